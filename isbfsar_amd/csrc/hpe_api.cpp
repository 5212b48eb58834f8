// C-ABI glue of the pose stage (include/isbfsar.h, isb_hpe_*): EfficientNetV2-L plan built from
// the weight blob, activation workspace, per-micro-batch launch sequence, stage-level test hooks.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <memory>
#include <utility>

#include "isb_common.h"
#include "kernels.h"

using namespace isb;

namespace {

struct ConvW {
    DevBuf w16, bias;
    int cout = 0, cin = 0, k = 0;
    bool f16 = false;             // w16 holds fp16 (the layer runs on fp16 operands) instead of bf16
};

struct BlockW {
    bool fused = false, residual = false;
    // fp16 stages (isb_hpe.f16_from): f16_in = the block's input (residual stream) is fp16, so its expand convolution and depthwise
    // taps are; f16 = everything after the expand tensor is (depthwise output, projection, block output). The block that
    // enters the fp16 stages has f16 && !f16_in: bf16 expand, depthwise bf16 -> fp16, fp16 projection.
    bool f16_in = false, f16 = false;
    int cin = 0, cout = 0, cexp = 0, stride = 1, cse = 0, in_hw = 0, out_hw = 0;
    ConvW expand, project;
    DevBuf dw_w16, dw_b, se_w1, se_b1, se_w2, se_b2;   // dw_w16: the depthwise taps in the stage's 16-bit type
    DevBuf mbf16_w1p;                 // stride-1 MBConv blocks with 192 / 224 inputs on 16 x 16 maps: the expand weights in fragment order (mbfront16_kernel)
    DevBuf fmb_w2p;                   // Fused-MBConv blocks: the projection weights in the register-E kernel's fragment order (launch_fmb_pack_w2)
    DevBuf mbf_w1p;                   // stride-1 MBConv blocks with 384 inputs on 8 x 8 maps: the expand weights in fragment order (mbfront8_kernel)
    DevBuf mb_w1p, mb_w2p, mb_se1p;   // stride-1 blocks of the 8 x 8 stages: the weights in mb8_chain_kernel's streaming layouts (conv_mb8.hip)
};

// public efficientnetv2-l table (mirrors isbfsar_amd/effnetv2.py::STAGES)
struct StageDef { bool fused; int repeats, expand, stride, cin, cout; bool se; };
constexpr int kMinSplit = 64;     // a batch this large is run as concurrent parts (>= 32 frames each)
constexpr int kMaxLanes = 4;
constexpr int kMaxJoints = 122;   // joints per pose after expansion (assets/32_to_122.npy); isb_hpe_set_joint_map checks n_out against it
constexpr int kRoiMinBatch = 2;   // isb_hpe_forward_host: ROI-only copies from this batch size on (one frame: the 9 KB round trip
                                  // for the homography costs what the smaller copy saves)
constexpr int kMaxMicroBatch = 1024;   // frames per micro-batch: keeps every activation tensor <= 2 GiB (32-bit byte offsets)
const StageDef kStages[] = {
    {true, 4, 1, 1, 32, 32, false},   {true, 7, 4, 2, 32, 64, false},   {true, 7, 4, 2, 64, 96, false},
    {false, 10, 4, 2, 96, 192, true}, {false, 19, 6, 1, 192, 224, true}, {false, 25, 6, 2, 224, 384, true},
    {false, 7, 6, 1, 384, 640, true},
};

// activation workspace of one micro-batch in flight. Two lanes on two streams run two halves of a batch
// concurrently: while one half sits in a bandwidth-bound kernel (depthwise conv, SE, tile write-back) the
// other half's GEMM tiles keep the matrix cores busy.
constexpr size_t kSplitKBytes = 8u << 20;

struct Lane {
    hipStream_t side = nullptr;   // lane 1 only: its own stream, forked from / joined into the caller's
    int ws_B = 0;
    DevBuf H, newK, R, crops, bufX, bufY, bufE, bufD, pooled, semid, gate, feat, logits;
    DevBuf part;                  // split-K partial tiles of the small-M projections (kSplitKBytes)
    void* X = nullptr;            // the residual stream's current input / output buffers while a forward pass is being enqueued
    void* Y = nullptr;
};

// Everything an engine READS during a forward pass and never writes: the folded / packed weights, the joint map, the zero line.
// An engine family (isb_hpe_create + isb_hpe_create_shared) holds one of these on the device -- 240 MB for EfficientNetV2-L --
// while every engine owns its streams, events and activation workspaces (the part that is written).
struct HpeModel {
    bool weights = false, jointmap = false;
    int n_out = 0;
    DevBuf stem_w, stem_wt, stem_b;
    std::vector<std::unique_ptr<BlockW>> blocks;
    ConvW headconv;
    DevBuf head_w, head_b;
    DevBuf expand, indices;
    bool has_indices = false;
    DevBuf zeros;
    int mb8_first = -1, mb8_count = 0;
    DevBuf mb8_desc;
    size_t device_bytes() const;
};

}  // namespace

struct isb_hpe {
    isb_hpe_cfg cfg{};
    hipStream_t own_stream = nullptr;
    // the stride-1 MBConv blocks of the two 8 x 8 stages (blocks mb8_first .. + mb8_count) as ONE launch (conv_mb8.hip: a workgroup
    // owns a sample for the whole chain; bit-identical to the five-launch path, tested). MEASURED SLOWER (round 4: 22.0 vs 16.7 ms per
    // 256-frame pose step; a 384 -> 2304 -> 384 block 545 k cycles against ~310 k for its five launches): every workgroup pulls ALL of
    // a block's weights -- 3.5 MB 16-bit + 1.8 MB f32 squeeze-excite -- through its own CU's L2 port for 64 rows, twice the L2 -> CU
    // bytes of the tiled GEMMs, and receives them at 5-15 B/clk (stamps: EXPERIMENTS.md round 4). Off by default; ISB_MB8=1 selects
    // it (this round's open experiment: the packed weights are only built then).
    bool mb8_on = false;
    // the FRONT half (expand + SiLU + depthwise + SiLU + pool) of the stride-1 MBConv blocks with 384 input channels on 8 x 8 maps in one
    // launch on stationary weights (conv_mb8.hip mbfront8_kernel; bit-identical to the two launches): batches >= mbf8_min_batch
    bool dwmm16 = true;           // small batches of the 16 x 16 blocks on the matrix-pipe depthwise kernel (the fused front's arithmetic); follows
                                  // ISB_MBF16 unless ISB_DWMM16 says otherwise (the bit-identity test's reference: ISB_MBF16=0 ISB_DWMM16=1)
    bool stamp16 = false;         // ISB_STAMP16=1: isb_debug_hpe_mb8_stamps arms the 16 x 16 front's clocks instead of the 8 x 8 front's
    bool mbf16_on = true;         // ISB_MBF16=0: expand GEMM + depthwise kernel on the 16 x 16 maps (the bit-identity test's reference)
    bool fmb_rege = true;         // ISB_FMB_REGE=0: Fused-MBConv blocks with the E tile in LDS (the round-2 form; A/B switch)
    bool dwmm_on = true;          // ISB_DWMM=0: (tests of the probe-only chain kernel) small batches of fused-front blocks on the v_dot2 depthwise kernel
    bool mbf8_on = true;          // ISB_MBF8=0: expand GEMM + depthwise kernel (the bit-identity test's reference)
    int mbf8_min_batch = 32;
    int mb8_min_batch = 48;
    DevBuf mb8_stamps;            // tuning probe (isb_debug_hpe_mb8_stamps)
    bool fuse_se = true;          // single-frame split-K projections compute their SE gate in the GEMM; ISB_FUSE_SE=0 disables (tests)
    // 16-bit storage type per stage: stages >= f16_from (index into kStages) and the 640 -> 1280 convolution keep activations AND
    // weights in IEEE fp16 instead of bf16 -- same MFMA rate, 3 more mantissa bits (per-stage error budget: DESIGN.md section 4,
    // oracle/error_budget.py). isb_hpe_cfg.precision: 0 (default) / 2 = fp16 everywhere, what the reference's TensorRT engines
    // run (7_create_engines.py:10; f16_from = 0, the stem stores fp16 too); 3 = bf16 with fp16 in the two 8x8 stages (round 3's
    // layout; f16_from = 5); 1 = bf16 everywhere (f16_from = 7)
    int f16_from = 0;
    double K[9] = {0};
    // weights: ONE device copy per engine family (isb_hpe_create_shared: the children hold the parent's model)
    std::shared_ptr<HpeModel> m;
    bool is_child = false;        // an engine made by isb_hpe_create_shared: it reads a model it may not replace (the model is kept alive by `m`)
    // test-time augmentation tables (hpe.py:88-93); n_aug = 0: off
    int n_aug = 0;
    DevBuf aug_rotflip, aug_scale;
    // workspace
    Lane lanes[kMaxLanes];
    int n_lanes = 2;              // ISB_HPE_LANES=1 disables the split, up to kMaxLanes
    hipEvent_t fork_ev = nullptr, join_ev[kMaxLanes] = {nullptr, nullptr, nullptr, nullptr};
    // the lanes' workspaces belong to one pass at a time: a pass enqueued on ANOTHER stream than the previous one (device-pointer
    // entry on the caller's stream after host batches on the handle's own stream, or the other way round) waits for it on the device
    hipEvent_t last_ev = nullptr;
    hipStream_t last_stream = nullptr;
    bool last_valid = false;
    // host-buffer entry point: persistent staging (grow-only) + a copy stream so that the H2D of chunk i + 1 travels
    // while chunk i computes
    hipStream_t copy_stream = nullptr;
    hipEvent_t h2d_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    DevBuf hs_frames, hs_bbox, hs_joints, hs_valid;
    int hs_B = 0;
    // ROI-only host input: the warp reads at most the crop square's pre-image, typically 150-480 KB of a 921 600-byte frame.
    // When the caller's frames are device-mapped pinned memory, isb_hpe_forward_host computes the crop homographies first (a
    // 9 KB round trip), bounds each frame's source rectangle on the host and ONE gather kernel pulls exactly those rectangles
    // over PCIe into a packed image (roi_gather_kernel). Measured and dropped: one hipMemcpy2DAsync per frame (2.3 ms EACH from
    // pinned memory: 608 ms per 256 frames) and one contiguous row-band copy per frame (24.2 ms against 22.9 for whole frames:
    // 256 copy calls cost more than the bytes they save). ISB_HPE_ROI: 1 = gather (default), 0 = whole frames.
    int roi_mode = 1;
    DevBuf hs_H, hs_newK, hs_R, hs_roi;
    // isb_hpe_submit_host / isb_hpe_wait_host: up to two host batches in flight. A slot owns its device staging, pinned result
    // buffers and events; the frames of batch k + 1 cross PCIe on the copy engine (copy_stream) while batch k computes, and the
    // lanes run batch after batch without a drain in between (what the device-pointer entry does for a resident caller)
    struct HostSlot {
        DevBuf frames, bbox, joints, valid;
        void* pin = nullptr;            // hipHostMalloc: [B * n_out * 12 bytes of joints | B bytes of valid]
        size_t pin_bytes = 0;
        std::vector<int32_t> boxes;     // the host copy the asynchronous upload reads
        hipEvent_t h2d = nullptr, done = nullptr;
        float* user_joints = nullptr;
        uint8_t* user_valid = nullptr;
        int B = 0, cap = 0, n_out = 0;  // n_out: joints per pose when the batch was submitted
        bool busy = false;
    };
    HostSlot slot[2];
    int sub_head = 0, sub_tail = 0;     // next slot to fill / oldest outstanding one
    const RoiDesc* roi_dev = nullptr;   // set for the duration of one isb_hpe_forward_host call
    const uint8_t* roi_src = nullptr;   // ... with the device address of the caller's mapped host frames: every lane gathers ITS
                                        // frames on its own stream, so lane 1's rectangles cross PCIe while lane 0 already computes
    // profiling of conv_igemm launches
    bool prof = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev_dw;      // the stand-alone depthwise launches (isb_hpe_profile_read_dw)
    double prof_dw_ms = 0.0;
    int64_t prof_dw_launches = 0;
    double prof_ms = 0.0;
    int64_t prof_launches = 0;
};

namespace {

inline uint16_t bf16_rne(float x) {             // round-to-nearest-even, as __bf16(x) on the device
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float bf16_to_float(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float x;
    memcpy(&x, &u, 4);
    return x;
}

int upload_conv(const std::map<std::string, BlobTensor>& m, const std::string& prefix, int cout, int k, int cin,
                ConvW& cw, hipStream_t st, bool f16 = false) {
    const BlobTensor *w, *sc, *sh;
    {
        auto it = m.find(prefix + ".w");
        ISB_REQUIRE(it != m.end(), ISB_ERR_WEIGHTS, "weight tensor '%s.w' missing", prefix.c_str());
        w = &it->second;
        ISB_REQUIRE((int)w->dims[0] == cout && (int)w->dims[1] == k && (int)w->dims[2] == k && (int)w->dims[3] == cin,
                    ISB_ERR_WEIGHTS, "'%s.w' has shape [%u,%u,%u,%u], expected [%d,%d,%d,%d]", prefix.c_str(), w->dims[0],
                    w->dims[1], w->dims[2], w->dims[3], cout, k, k, cin);
    }
    ISB_TRY(blob_get(m, (prefix + ".scale").c_str(), cout, 1, &sc));
    ISB_TRY(blob_get(m, (prefix + ".shift").c_str(), cout, 1, &sh));
    DevBuf tmp, dsc;
    ISB_TRY(upload(tmp, w->data, w->numel() * 4));
    ISB_TRY(upload(dsc, sc->data, (size_t)cout * 4));
    ISB_TRY(cw.w16.alloc(w->numel() * 2));
    ISB_TRY(launch_f32_to_bf16_rows(tmp.as<float>(), dsc.as<float>(), cw.w16.as<uint16_t>(), cout, (size_t)k * k * cin, st, f16 ? 1 : 0));
    ISB_HIP(hipStreamSynchronize(st));
    ISB_TRY(upload(cw.bias, sh->data, (size_t)cout * 4));
    cw.cout = cout; cw.cin = cin; cw.k = k; cw.f16 = f16;
    return ISB_OK;
}

// free functions (no handle): make the device that owns `dptr` current, like every handle-taking entry point does
int set_device_of(const void* dptr) {
    hipPointerAttribute_t at{};
    ISB_HIP(hipPointerGetAttributes(&at, dptr));
    ISB_REQUIRE(at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged, ISB_ERR_INVALID,
                "pointer %p is not device memory", dptr);
    ISB_HIP(hipSetDevice(at.device));
    return ISB_OK;
}

int ensure_ws(Lane& L, int Bm) {
    if (Bm <= L.ws_B) return ISB_OK;
    const size_t B = Bm;
    L.ws_B = 0;                   // failure-atomic: a failed allocation below leaves a lane that re-allocates on the next call
    ISB_TRY(L.H.alloc(B * 9 * 4));
    ISB_TRY(L.newK.alloc(B * 9 * 8));
    ISB_TRY(L.R.alloc(B * 9 * 8));
    ISB_TRY(L.crops.alloc(B * 256 * 256 * 3 * 4));
    ISB_TRY(L.bufX.alloc(B * 128 * 128 * 32 * 2));
    ISB_TRY(L.bufY.alloc(B * 128 * 128 * 32 * 2));
    ISB_TRY(L.bufE.alloc(B * 64 * 64 * 256 * 2));
    ISB_TRY(L.bufD.alloc(B * 32 * 32 * 384 * 2));
    ISB_TRY(L.pooled.alloc(B * 3840 * 4));
    ISB_TRY(L.semid.alloc(B * 160 * 32 * 4));       // SE fc1 partial sums: up to 32 channel slabs x 160 outputs
    ISB_TRY(L.gate.alloc(B * 3840 * 4));
    ISB_TRY(L.feat.alloc(B * 64 * 1280 * 4));
    ISB_TRY(L.logits.alloc(B * 64 * 288 * 4));
    if (!L.part.p) ISB_TRY(L.part.alloc(kSplitKBytes));
    L.ws_B = Bm;
    return ISB_OK;
}

int conv(isb_hpe* h, hipStream_t st, const ConvW& cw, const void* in, int B, int H, int W, int stride, bool act,
         const void* res, const float* gate, void* out, bool out_f32, Lane* lane = nullptr, const SeFcArgs* se = nullptr) {
    ConvArgs a{};
    a.f16 = cw.f16 ? 1 : 0;
    a.in = (const uint16_t*)in; a.w = cw.w16.as<uint16_t>(); a.bias = cw.bias.as<float>();
    a.res = (const uint16_t*)res; a.gate = gate; a.out = out;
    a.B = B; a.H = H; a.W = W; a.Cin = cw.cin; a.Cout = cw.cout; a.KH = cw.k; a.KW = cw.k; a.stride = stride;
    a.OH = H / stride; a.OW = W / stride;
    a.pad = (cw.k == 3 && stride == 1) ? 1 : 0;          // TF SAME: stride 2 on an even input pads bottom/right only
    a.M = B * a.OH * a.OW; a.K = cw.k * cw.k * cw.cin;
    a.act = act ? 1 : 0; a.out_f32 = out_f32 ? 1 : 0;
    a.zeros = h->m->zeros.as<uint16_t>();
    // ONE frame (the live loop) x a long K: the SE-gated projections of the last stages are 3-8 tiles of 64 x 128
    // each walking up to 120 k-tiles in series. Split K across workgroups so that the launch covers more of the chip.
    // Only for single-frame calls, and with a split count that depends on the layer alone: batches of two or more
    // frames keep one summation order whatever their size, so any sharding of a batch stays bit-identical.
    if (lane && B == 1 && cw.k == 1 && stride == 1 && !out_f32 && a.Cout >= 64 &&
        (!gate || (a.OH * a.OW) % 64 == 0)) {
        const int s = std::min(a.Cin / 32 / 6, 16);
        if (s > 1 && (size_t)s * a.M * a.Cout * 4 <= kSplitKBytes) { a.splits = s; a.part = lane->part.as<float>(); }
    }
    // se: the squeeze-excite FCs that produce `gate` have not run yet. A split-K launch whose k-ranges are at most
    // 256 channels computes each range's gate inside the GEMM (variant 149); otherwise the FC kernels run first.
    if (se) {
        const int nkt = a.Cin / 32;
        if (h->fuse_se && a.splits > 1 && se->nparts > 0 && ((nkt + a.splits - 1) / a.splits) * 32 <= 256 &&
            (a.OH * a.OW) % 64 == 0) {
            a.se_part = se->part; a.se_b1 = se->b1; a.se_w2t = se->w2t; a.se_b2 = se->b2;
            a.se_nparts = se->nparts; a.se_cse = se->cse;
            a.gate = nullptr;
            a.variant = 149;
        } else {
            ISB_TRY(launch_se_fcs(*se, st));
        }
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->prof) {
        ISB_HIP(hipEventCreate(&e0));
        ISB_HIP(hipEventCreate(&e1));
        ISB_HIP(hipEventRecord(e0, st));
    }
    ISB_TRY(launch_conv_igemm(a, st));
    if (h->prof) {
        ISB_HIP(hipEventRecord(e1, st));
        h->prof_ev.emplace_back(e0, e1);
        h->prof_launches += 1;
    }
    return ISB_OK;
}

int gemm(hipStream_t st, const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, int M,
         int N, int K, int act) {
    GemmF32Args g{};
    g.A = A; g.W = W; g.bias = bias; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw; g.ldc = ldc;
    g.add_period = 1; g.act = act;
    return launch_gemm_f32(g, st);
}

// crops f32 [B,256,256,3] (device) -> feat f32 [B*64,1280], logits f32 [B*64,288], in three pieces so that a caller with
// several lanes can ENQUEUE them round-robin (backbone_begin, backbone_blocks over block ranges, backbone_end): a lane's ~600
// launches take the host ~2 ms to submit, and submitted lane after lane the second lane's first kernel would wait that long
int backbone_begin(isb_hpe* h, Lane& L, hipStream_t st, const float* crops, int B) {
    StemArgs sa{};
    sa.in = crops; sa.w = h->m->stem_w.as<float>(); sa.wt = h->m->stem_wt.as<float>(); sa.bias = h->m->stem_b.as<float>(); sa.out = L.bufX.as<uint16_t>();
    sa.B = B; sa.H = 256; sa.W = 256; sa.out_f16 = h->f16_from <= 0 ? 1 : 0;
    ISB_TRY(launch_stem(sa, st));
    L.X = L.bufX.p;
    L.Y = L.bufY.p;
    return ISB_OK;
}

int backbone_blocks(isb_hpe* h, Lane& L, hipStream_t st, int B, size_t i0, size_t i1) {
    void*& X = L.X;
    void*& Y = L.Y;
    for (size_t bi = i0; bi < std::min(i1, h->m->blocks.size()); ++bi) {
        BlockW& b = *h->m->blocks[bi];
        if (h->mb8_on && h->m->mb8_count > 0 && B >= h->mb8_min_batch && (int)bi >= h->m->mb8_first && (int)bi < h->m->mb8_first + h->m->mb8_count) {
            if ((int)bi > h->m->mb8_first) continue;         // the chain ran when its first block came up
            Mb8Args a{};
            a.x = (const uint16_t*)X; a.out = (uint16_t*)Y; a.dscratch = L.bufE.p; a.dscratch_stride = (size_t)64 * 3840 * 2;
            a.blocks = h->m->mb8_desc.as<Mb8Block>(); a.nblocks = h->m->mb8_count; a.B = B; a.cin0 = b.cin; a.f16 = b.f16 ? 1 : 0;
            a.stamps = h->mb8_stamps.p ? h->mb8_stamps.as<uint64_t>() : nullptr;
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (h->prof) {
                ISB_HIP(hipEventCreate(&e0));
                ISB_HIP(hipEventCreate(&e1));
                ISB_HIP(hipEventRecord(e0, st));
            }
            ISB_TRY(launch_mb8_chain(a, st));
            if (h->prof) {
                ISB_HIP(hipEventRecord(e1, st));
                h->prof_ev.emplace_back(e0, e1);
                h->prof_launches += 1;
            }
            std::swap(X, Y);
            continue;
        }
        const void* res = b.residual ? X : nullptr;
        if (b.fused) {
            if (b.cexp == b.cin) {
                ISB_TRY(conv(h, st, b.expand, X, B, b.in_hw, b.in_hw, b.stride, true, res, nullptr, Y, false));
            } else if (b.cexp <= 256 && b.cout <= 128) {      // (384 expanded channels measured slower than two launches)
                // whole Fused-MBConv block in one launch: the expanded tensor never leaves the chip (bit-identical to
                // the two-launch path below; -26...32 % on the 64-channel stage, see launch_fused_mb)
                ConvArgs a{};
                a.in = (const uint16_t*)X; a.w = b.expand.w16.as<uint16_t>(); a.bias = b.expand.bias.as<float>();
                a.res = (const uint16_t*)res; a.out = Y;
                a.B = B; a.H = b.in_hw; a.W = b.in_hw; a.Cin = b.cin; a.Cout = b.cexp; a.KH = 3; a.KW = 3; a.stride = b.stride;
                a.OH = b.out_hw; a.OW = b.out_hw; a.pad = b.stride == 1 ? 1 : 0; a.M = B * b.out_hw * b.out_hw; a.K = 9 * b.cin;
                a.act = 1; a.f16 = b.f16 ? 1 : 0;
                a.w2 = b.project.w16.as<uint16_t>(); a.bias2 = b.project.bias.as<float>(); a.Cout2 = b.cout;
                a.w2p = h->fmb_rege ? b.fmb_w2p.p : nullptr;
                hipEvent_t e0 = nullptr, e1 = nullptr;
                if (h->prof) {
                    ISB_HIP(hipEventCreate(&e0));
                    ISB_HIP(hipEventCreate(&e1));
                    ISB_HIP(hipEventRecord(e0, st));
                }
                ISB_TRY(launch_fused_mb(a, st));
                if (h->prof) {
                    ISB_HIP(hipEventRecord(e1, st));
                    h->prof_ev.emplace_back(e0, e1);
                    h->prof_launches += 1;
                }
            } else {
                ISB_TRY(conv(h, st, b.expand, X, B, b.in_hw, b.in_hw, b.stride, true, nullptr, nullptr, L.bufE.p, false));
                ISB_TRY(conv(h, st, b.project, L.bufE.p, B, b.out_hw, b.out_hw, 1, false, res, nullptr, Y, false));
            }
        } else {
            int se_parts = 0;
            if (h->mbf8_on && b.mbf_w1p.p && B >= h->mbf8_min_batch) {
                MbFront8Args a{};
                a.x = (const uint16_t*)X; a.w1p = (const uint4*)b.mbf_w1p.p; a.b1 = b.expand.bias.as<float>();
                a.dww = b.dw_w16.as<uint16_t>(); a.dwb = b.dw_b.as<float>(); a.d = L.bufD.as<uint16_t>(); a.pooled = L.pooled.as<float>();
                a.B = B; a.cin = b.cin; a.f16 = b.f16 ? 1 : 0;
                a.stamps = (h->mb8_stamps.p && !h->stamp16) ? h->mb8_stamps.as<uint64_t>() : nullptr;
                if (a.stamps) a.form = 1;      // (tools/exp_mbf8.py reads the first kernel's phase clocks)
                hipEvent_t e0 = nullptr, e1 = nullptr;
                if (h->prof) {
                    ISB_HIP(hipEventCreate(&e0));
                    ISB_HIP(hipEventCreate(&e1));
                    ISB_HIP(hipEventRecord(e0, st));
                }
                ISB_TRY(launch_mbfront8(a, st));
                if (h->prof) {
                    ISB_HIP(hipEventRecord(e1, st));
                    h->prof_ev.emplace_back(e0, e1);
                    h->prof_launches += 1;
                }
            } else if (h->mbf16_on && b.mbf16_w1p.p && B >= h->mbf8_min_batch) {
                MbFront16Args a{};
                a.x = (const uint16_t*)X; a.w1p = (const uint4*)b.mbf16_w1p.p; a.b1 = b.expand.bias.as<float>();
                a.dww = b.dw_w16.as<uint16_t>(); a.dwb = b.dw_b.as<float>(); a.d = L.bufD.as<uint16_t>(); a.pooled = L.pooled.as<float>();
                a.B = B; a.cin = b.cin; a.cexp = b.cexp; a.f16 = b.f16 ? 1 : 0;
                a.stamps = (h->mb8_stamps.p && h->stamp16 && b.cin == 224) ? h->mb8_stamps.as<uint64_t>() : nullptr;
                if (a.stamps) a.form = 1;      // (tools/exp_mbf16.py reads the first kernel's phase clocks; the role kernel's: isb_debug_mbfront)
                hipEvent_t e0 = nullptr, e1 = nullptr;
                if (h->prof) {
                    ISB_HIP(hipEventCreate(&e0));
                    ISB_HIP(hipEventCreate(&e1));
                    ISB_HIP(hipEventRecord(e0, st));
                }
                ISB_TRY(launch_mbfront16(a, st));
                if (h->prof) {
                    ISB_HIP(hipEventRecord(e1, st));
                    h->prof_ev.emplace_back(e0, e1);
                    h->prof_launches += 1;
                }
            } else {
                ISB_TRY(conv(h, st, b.expand, X, B, b.in_hw, b.in_hw, 1, true, nullptr, nullptr, L.bufE.p, false));
                DwArgs d{};
                d.in = L.bufE.as<uint16_t>(); d.w = b.dw_w16.as<uint16_t>(); d.bias = b.dw_b.as<float>(); d.out = L.bufD.as<uint16_t>();
                d.B = B; d.H = b.in_hw; d.W = b.in_hw; d.C = b.cexp; d.OH = b.out_hw; d.OW = b.out_hw; d.stride = b.stride;
                d.pad = b.stride == 1 ? 1 : 0;
                d.in_f16 = b.f16_in ? 1 : 0; d.out_f16 = b.f16 ? 1 : 0;
                d.pooled = L.pooled.as<float>();
                // a block that has a fused front (mbfront8_kernel, batches >= mbf8_min_batch) runs the stand-alone kernel with ITS
                // arithmetic below that size, so that a frame's bits do not depend on the batch it arrives in
                d.general = (h->dwmm_on && b.stride == 1 && b.f16_in == b.f16 &&
                             ((b.in_hw == 8 && b.cin == 384) || (h->dwmm16 && b.in_hw == 16 && (b.cin == 192 || b.cin == 224)))) ? 3 : 0;
                if (B == 1) {   // one frame: FC1 of the squeeze-excite rides in the depthwise launch (batches: measured slower)
                    d.se_w1 = b.se_w1.as<float>(); d.se_part = L.semid.as<float>(); d.cse = b.cse;
                    se_parts = dw_slabs(d);
                }
                hipEvent_t e0 = nullptr, e1 = nullptr;
                if (h->prof) {
                    ISB_HIP(hipEventCreate(&e0));
                    ISB_HIP(hipEventCreate(&e1));
                    ISB_HIP(hipEventRecord(e0, st));
                }
                ISB_TRY(launch_dwconv3x3(d, st));
                if (h->prof) {
                    ISB_HIP(hipEventRecord(e1, st));
                    h->prof_ev_dw.emplace_back(e0, e1);
                    h->prof_dw_launches += 1;
                }
            }
            SeFcArgs se{};
            se.nparts = se_parts;
            se.pooled = L.pooled.as<float>(); se.w1 = b.se_w1.as<float>(); se.b1 = b.se_b1.as<float>();
            se.w2t = b.se_w2.as<float>(); se.b2 = b.se_b2.as<float>(); se.part = L.semid.as<float>();
            se.gate = L.gate.as<float>(); se.B = B; se.C = b.cexp; se.cse = b.cse;
            ISB_TRY(conv(h, st, b.project, L.bufD.p, B, b.out_hw, b.out_hw, 1, false, res, L.gate.as<float>(), Y, false, &L, &se));
        }
        std::swap(X, Y);
    }
    return ISB_OK;
}

int backbone_end(isb_hpe* h, Lane& L, hipStream_t st, int B) {
    void* X = L.X;
    ISB_TRY(conv(h, st, h->m->headconv, X, B, 8, 8, 1, true, nullptr, nullptr, L.feat.p, true));
    if (B == 1) {
        // one frame: 64 rows x 288 outputs are 5 tiles walking 40 k-tiles each -> 8 K-splits + an in-order reduction
        constexpr int kHeadSplits = 8;
        GemmF32Args g{};
        g.A = L.feat.as<float>(); g.lda = 1280; g.W = h->m->head_w.as<float>(); g.ldw = 1280; g.C = L.part.as<float>(); g.ldc = 288;
        g.M = 64; g.N = 288; g.K = 1280; g.add_period = 1; g.act = GEMM_ACT_NONE;
        g.splits = kHeadSplits; g.split_stride = (size_t)64 * 288;
        ISB_TRY(launch_gemm_f32(g, st));
        return launch_reduce_parts(L.part.as<float>(), kHeadSplits, g.split_stride, h->m->head_b.as<float>(), GEMM_ACT_NONE,
                                   L.logits.as<float>(), 64, 288, st);
    }
    ISB_TRY(gemm(st, L.feat.as<float>(), 1280, h->m->head_w.as<float>(), 1280, h->m->head_b.as<float>(), L.logits.as<float>(), 288,
                 B * 64, 288, 1280, GEMM_ACT_NONE));
    return ISB_OK;
}

int run_backbone(isb_hpe* h, Lane& L, hipStream_t st, const float* crops, int B) {
    ISB_TRY(backbone_begin(h, L, st, crops, B));
    ISB_TRY(backbone_blocks(h, L, st, B, 0, h->m->blocks.size()));
    return backbone_end(h, L, st, B);
}

int run_post(isb_hpe* h, Lane& L, hipStream_t st, const float* logits, int B, float* joints, uint8_t* valid, double* dbg,
             const int32_t* bbox = nullptr) {
    PostArgs a{};
    a.bbox = bbox;
    a.logits = logits; a.newK = L.newK.as<double>(); a.R = L.R.as<double>(); a.expand = h->m->expand.as<float>();
    a.indices = h->m->has_indices ? h->m->indices.as<int32_t>() : nullptr;
    a.joints = joints; a.valid = valid; a.dbg = dbg; a.B = B; a.n_out = h->m->n_out;
    return launch_hpe_post(a, st);
}

int run_crop_params(isb_hpe* h, Lane& L, hipStream_t st, const int32_t* d_bbox, int B) {
    CropParamArgs a{};
    a.bbox = d_bbox;
    for (int i = 0; i < 9; ++i) a.K[i] = h->K[i];
    a.H = L.H.as<float>(); a.newK = L.newK.as<double>(); a.R = L.R.as<double>();
    a.n_aug = h->n_aug; a.aug_rotflip = h->aug_rotflip.as<double>(); a.aug_scale = h->aug_scale.as<double>();
    a.B = B * std::max(h->n_aug, 1);        // B boxes -> B x n_aug parameter sets
    return launch_crop_params(a, st);
}

int run_warp(isb_hpe* h, Lane& L, hipStream_t st, const uint8_t* d_frames, int B, const RoiDesc* roi = nullptr) {
    WarpArgs a{};
    a.roi = roi;
    a.frames = d_frames; a.H = L.H.as<float>(); a.crops = L.crops.as<float>();
    a.n_aug = std::max(h->n_aug, 1);
    a.B = B * a.n_aug;                      // B frames -> B x n_aug crops
    a.FH = h->cfg.height; a.FW = h->cfg.width;
    return launch_warp(a, st);
}

// finished depthwise-launch event pairs -> prof_dw_ms (isb_hpe_profile_read_dw reports and resets it). Called by every reader and by
// isb_hpe_profile(0), so a caller of the family-only API does not accumulate events without bound.
int drain_prof_dw(isb_hpe* h) {
    for (auto& e : h->prof_ev_dw) {
        ISB_HIP(hipEventSynchronize(e.second));
        float ms = 0.f;
        ISB_HIP(hipEventElapsedTime(&ms, e.first, e.second));
        h->prof_dw_ms += ms;
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    h->prof_ev_dw.clear();
    return ISB_OK;
}

int create_streams(isb_hpe* h) {
    ISB_HIP(hipEventCreateWithFlags(&h->fork_ev, hipEventDisableTiming));
    ISB_HIP(hipEventCreateWithFlags(&h->last_ev, hipEventDisableTiming));
    ISB_HIP(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    for (auto& e : h->h2d_ev) ISB_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& sl : h->slot) {
        ISB_HIP(hipEventCreateWithFlags(&sl.h2d, hipEventDisableTiming));
        ISB_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    }
    for (int l = 1; l < kMaxLanes; ++l) {
        ISB_HIP(hipStreamCreateWithFlags(&h->lanes[l].side, hipStreamNonBlocking));
        ISB_HIP(hipEventCreateWithFlags(&h->join_ev[l], hipEventDisableTiming));
    }
    return ISB_OK;
}

size_t HpeModel::device_bytes() const {
    size_t n = stem_w.bytes + stem_wt.bytes + stem_b.bytes + headconv.w16.bytes + headconv.bias.bytes + head_w.bytes + head_b.bytes +
               expand.bytes + indices.bytes + zeros.bytes + mb8_desc.bytes;
    for (const auto& b : blocks)
        for (const DevBuf* d : {&b->expand.w16, &b->expand.bias, &b->project.w16, &b->project.bias, &b->dw_w16, &b->dw_b, &b->se_w1, &b->se_b1,
                                &b->se_w2, &b->se_b2, &b->mbf16_w1p, &b->fmb_w2p, &b->mbf_w1p, &b->mb_w1p, &b->mb_w2p, &b->mb_se1p})
            n += d->bytes;
    return n;
}

}  // namespace

extern "C" int isb_hpe_create(const isb_hpe_cfg* cfg, isb_hpe** out) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(cfg && out, ISB_ERR_INVALID, "isb_hpe_create: null argument");
    ISB_REQUIRE(cfg->width >= 16 && cfg->height >= 16 && cfg->width <= 8192 && cfg->height <= 8192, ISB_ERR_INVALID,
                "frame size %dx%d unsupported", cfg->width, cfg->height);
    ISB_REQUIRE(cfg->fx > 0 && cfg->fy > 0, ISB_ERR_INVALID, "focal lengths must be positive");
    ISB_REQUIRE(cfg->precision >= 0 && cfg->precision <= 3, ISB_ERR_INVALID,
                "isb_hpe_cfg.precision %d: 0 / 2 (fp16 everywhere), 1 (bf16 everywhere) or 3 (bf16, fp16 in the two 8x8 stages)", cfg->precision);
    int ndev = 0;
    ISB_HIP(hipGetDeviceCount(&ndev));
    ISB_REQUIRE(cfg->device >= 0 && cfg->device < ndev, ISB_ERR_INVALID, "device %d not in [0,%d)", cfg->device, ndev);
    ISB_HIP(hipSetDevice(cfg->device));
    std::unique_ptr<isb_hpe> h(new (std::nothrow) isb_hpe());
    ISB_REQUIRE(h, ISB_ERR_NOMEM, "out of host memory");
    h->cfg = *cfg;
    if (h->cfg.max_batch <= 0) h->cfg.max_batch = 64;
    // per-lane micro-batch limit (isbfsar.h): the convolution kernels address a tensor with 32-bit byte offsets and the
    // largest activation is 2 MiB per frame (64x64x256 bf16) -> 1024 frames = 2 GiB. Larger batches are micro-batched.
    h->cfg.max_batch = std::min(h->cfg.max_batch, kMaxMicroBatch);
    // K as float32 values (hpe.py:28-33)
    h->K[0] = (double)cfg->fx; h->K[2] = (double)cfg->ppx; h->K[4] = (double)cfg->fy; h->K[5] = (double)cfg->ppy; h->K[8] = 1.0;
    ISB_HIP(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->m = std::make_shared<HpeModel>();
    ISB_TRY(h->m->zeros.alloc(256));
    ISB_HIP(hipMemset(h->m->zeros.p, 0, 256));
    if (const char* e = getenv("ISB_FUSE_SE")) h->fuse_se = atoi(e) != 0;
#ifdef ISB_BUILD_PROBES
    if (const char* e = getenv("ISB_MB8")) h->mb8_on = atoi(e) != 0;       // (the per-sample chain kernel exists in probe builds only)
#endif
    if (const char* e = getenv("ISB_DWMM")) h->dwmm_on = atoi(e) != 0;
    if (const char* e = getenv("ISB_FMB_REGE")) h->fmb_rege = atoi(e) != 0;
    if (const char* e = getenv("ISB_MBF16")) h->mbf16_on = atoi(e) != 0;
    if (const char* e = getenv("ISB_STAMP16")) h->stamp16 = atoi(e) != 0;
    h->dwmm16 = h->mbf16_on;
    if (const char* e = getenv("ISB_DWMM16")) h->dwmm16 = atoi(e) != 0;
    if (const char* e = getenv("ISB_MBF8")) h->mbf8_on = atoi(e) != 0;
    if (!isb::mbf8_verified()) h->mbf8_on = false;      // fail closed: the build could not confirm the kernel's counted wait (wsreg_guard.cpp)
    if (const char* e = getenv("ISB_MB8_MIN_BATCH")) h->mb8_min_batch = std::max(1, atoi(e));
    h->f16_from = cfg->precision == 1 ? 7 : (cfg->precision == 3 ? 5 : 0);
    if (const char* e = getenv("ISB_HPE_ROI")) h->roi_mode = atoi(e) != 0 ? 1 : 0;
    if (const char* e = getenv("ISB_HPE_LANES")) h->n_lanes = std::max(1, std::min(kMaxLanes, atoi(e)));
    ISB_TRY(create_streams(h.get()));
    *out = h.release();
    return ISB_OK;
    });
}

// One more engine on the SAME device weights: the child shares the parent's model -- folded / packed weights, joint
// map -- and owns what a pass writes: its streams, events and activation workspaces. Keeping K batches in flight (one engine per batch,
// each on its own stream) then costs one copy of the weights + K workspaces instead of K copies of both.
extern "C" int isb_hpe_create_shared(isb_hpe* parent, isb_hpe** out) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(parent && out, ISB_ERR_INVALID, "isb_hpe_create_shared: null argument");
    ISB_HIP(hipSetDevice(parent->cfg.device));
    std::unique_ptr<isb_hpe> h(new (std::nothrow) isb_hpe());
    ISB_REQUIRE(h, ISB_ERR_NOMEM, "out of host memory");
    h->cfg = parent->cfg;
    memcpy(h->K, parent->K, sizeof(h->K));
    h->m = parent->m;                         // (the model outlives whichever of the two is destroyed first)
    h->is_child = true;
    // the plan switches select which packed weight images exist: a child runs the parent's plan
    h->mb8_on = parent->mb8_on; h->dwmm16 = parent->dwmm16; h->stamp16 = parent->stamp16; h->mbf16_on = parent->mbf16_on;
    h->fmb_rege = parent->fmb_rege; h->dwmm_on = parent->dwmm_on; h->mbf8_on = parent->mbf8_on;
    h->mbf8_min_batch = parent->mbf8_min_batch; h->mb8_min_batch = parent->mb8_min_batch; h->fuse_se = parent->fuse_se;
    h->f16_from = parent->f16_from; h->roi_mode = parent->roi_mode; h->n_lanes = parent->n_lanes;
    ISB_HIP(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    ISB_TRY(create_streams(h.get()));
    *out = h.release();
    return ISB_OK;
    });
}

extern "C" int isb_hpe_memory(isb_hpe* h, uint64_t* model_bytes, uint64_t* workspace_bytes, int32_t* engines_on_model) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(h, ISB_ERR_INVALID, "null handle");
    if (model_bytes) *model_bytes = h->m->device_bytes();
    if (workspace_bytes) {
        size_t w = 0;
        for (const Lane& L : h->lanes)
            for (const DevBuf* b : {&L.H, &L.newK, &L.R, &L.crops, &L.bufX, &L.bufY, &L.bufE, &L.bufD, &L.pooled, &L.semid, &L.gate, &L.feat,
                                    &L.logits, &L.part})
                w += b->bytes;
        for (const DevBuf* b : {&h->hs_frames, &h->hs_bbox, &h->hs_joints, &h->hs_valid, &h->hs_H, &h->hs_newK, &h->hs_R, &h->hs_roi,
                                &h->aug_rotflip, &h->aug_scale, &h->mb8_stamps})
            w += b->bytes;
        for (const auto& sl : h->slot) w += sl.frames.bytes + sl.bbox.bytes + sl.joints.bytes + sl.valid.bytes;
        *workspace_bytes = w;
    }
    if (engines_on_model) *engines_on_model = (int32_t)h->m.use_count();
    return ISB_OK;
    });
}

extern "C" void isb_hpe_destroy(isb_hpe* h) {
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    (void)hipDeviceSynchronize();
    for (auto* list : {&h->prof_ev, &h->prof_ev_dw})
        for (auto& e : *list) {
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    for (int l = 1; l < kMaxLanes; ++l) {
        if (h->lanes[l].side) (void)hipStreamDestroy(h->lanes[l].side);
        if (h->join_ev[l]) (void)hipEventDestroy(h->join_ev[l]);
    }
    if (h->fork_ev) (void)hipEventDestroy(h->fork_ev);
    if (h->last_ev) (void)hipEventDestroy(h->last_ev);
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    for (auto& sl : h->slot) {
        if (sl.h2d) (void)hipEventDestroy(sl.h2d);
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.pin) (void)hipHostFree(sl.pin);
    }
    for (auto& e : h->h2d_ev)
        if (e) (void)hipEventDestroy(e);
    delete h;
}

extern "C" int isb_hpe_load_weights(isb_hpe* h, const void* blob, size_t nbytes) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(h, ISB_ERR_INVALID, "null handle");
    ISB_REQUIRE(!h->slot[0].busy && !h->slot[1].busy, ISB_ERR_STATE, "submitted host batches are outstanding: isb_hpe_wait_host first");
    ISB_REQUIRE(!h->is_child, ISB_ERR_STATE, "isb_hpe_load_weights on an engine made by isb_hpe_create_shared: load into the parent (the family reads one model)");
    ISB_HIP(hipSetDevice(h->cfg.device));
    ISB_HIP(hipDeviceSynchronize());           // engines that share this model may have passes in flight on their own streams
    hipStream_t st = h->own_stream;
    std::map<std::string, BlobTensor> m;
    ISB_TRY(parse_blob(blob, nbytes, m));
    h->m->weights = false;
    h->m->blocks.clear();
    // stem: f32 [32][27] with the BN scale folded
    {
        auto it = m.find("bbone.stem.w");
        ISB_REQUIRE(it != m.end() && it->second.numel() == 32 * 27, ISB_ERR_WEIGHTS, "bbone.stem.w missing or mis-shaped");
        const BlobTensor *sc, *sh;
        ISB_TRY(blob_get(m, "bbone.stem.scale", 32, 1, &sc));
        ISB_TRY(blob_get(m, "bbone.stem.shift", 32, 1, &sh));
        std::vector<float> w(32 * 27);
        for (int o = 0; o < 32; ++o)
            for (int k = 0; k < 27; ++k) w[o * 27 + k] = it->second.data[o * 27 + k] * sc->data[o];
        ISB_TRY(upload(h->m->stem_w, w.data(), w.size() * 4));
        std::vector<float> wt(27 * 32);                    // pair-major copy [16][27][2]: stem_kernel reads channel pairs as adjacent scalars
        for (int o = 0; o < 32; ++o)
            for (int k = 0; k < 27; ++k) wt[((o >> 1) * 27 + k) * 2 + (o & 1)] = w[o * 27 + k];
        ISB_TRY(upload(h->m->stem_wt, wt.data(), wt.size() * 4));
        ISB_TRY(upload(h->m->stem_b, sh->data, 32 * 4));
    }
    int idx = 0, hw = 128, stage = 0;
    bool stream_f16 = false;        // type of the tensor the next block reads
    for (const StageDef& s : kStages) {
        const int si = stage++;
        for (int r = 0; r < s.repeats; ++r, ++idx) {
            std::unique_ptr<BlockW> b(new BlockW());
            b->f16_in = stream_f16;
            b->f16 = si >= h->f16_from;
            stream_f16 = b->f16;
            b->fused = s.fused;
            b->cin = r == 0 ? s.cin : s.cout;
            b->cout = s.cout;
            b->cexp = b->cin * s.expand;
            b->stride = r == 0 ? s.stride : 1;
            b->cse = s.se ? std::max(1, (int)(b->cin * 0.25)) : 0;
            b->residual = b->stride == 1 && b->cin == b->cout;
            b->in_hw = hw;
            b->out_hw = hw / b->stride;
            hw = b->out_hw;
            const std::string p = "bbone.b" + std::to_string(idx);
            if (b->fused) {
                if (b->cexp == b->cin) {
                    ISB_TRY(upload_conv(m, p + ".expand", b->cout, 3, b->cin, b->expand, st, b->f16));
                } else {
                    ISB_TRY(upload_conv(m, p + ".expand", b->cexp, 3, b->cin, b->expand, st, b->f16));
                    ISB_TRY(upload_conv(m, p + ".project", b->cout, 1, b->cexp, b->project, st, b->f16));
                    if (b->cexp <= 256 && b->cout <= 64) {
                        ISB_TRY(b->fmb_w2p.alloc(fmb_w2p_bytes(b->cout, b->cexp)));
                        ISB_TRY(launch_fmb_pack_w2(b->project.w16.as<uint16_t>(), b->fmb_w2p.p, b->cout, b->cexp, st));
                    }
                }
            } else {
                ISB_TRY(upload_conv(m, p + ".expand", b->cexp, 1, b->cin, b->expand, st, b->f16_in));
                ISB_TRY(upload_conv(m, p + ".project", b->cout, 1, b->cexp, b->project, st, b->f16));
                auto it = m.find(p + ".dw.w");
                ISB_REQUIRE(it != m.end() && it->second.numel() == (size_t)b->cexp * 9, ISB_ERR_WEIGHTS,
                            "%s.dw.w missing or mis-shaped", p.c_str());
                const BlobTensor *sc, *sh, *w1, *b1, *w2, *b2;
                ISB_TRY(blob_get(m, (p + ".dw.scale").c_str(), b->cexp, 1, &sc));
                ISB_TRY(blob_get(m, (p + ".dw.shift").c_str(), b->cexp, 1, &sh));
                std::vector<uint16_t> wt16((size_t)9 * b->cexp);  // tap-major, scale folded, rounded to bf16 (fp16 in the fp16 stages) like every conv weight
                for (int c = 0; c < b->cexp; ++c)
                    for (int t = 0; t < 9; ++t) {
                        const float wf = it->second.data[(size_t)c * 9 + t] * sc->data[c];
                        if (b->f16_in) {
                            const _Float16 hh = (_Float16)wf;            // round to nearest even, as the device's conversion
                            uint16_t hb;
                            memcpy(&hb, &hh, 2);
                            wt16[(size_t)t * b->cexp + c] = hb;
                        } else {
                            const uint16_t hb = bf16_rne(wf);
                            wt16[(size_t)t * b->cexp + c] = hb;
                        }
                    }
                if (h->mbf16_on && b->stride == 1 && b->in_hw == 16 && (b->cin == 192 || b->cin == 224) && b->f16_in == b->f16) {
                    ISB_TRY(b->mbf16_w1p.alloc((size_t)b->cexp * b->cin * 2));
                    ISB_TRY(launch_mb8_pack_frag(b->expand.w16.as<uint16_t>(), b->mbf16_w1p.p, b->cexp, b->cin, 1, st));
                    ISB_HIP(hipStreamSynchronize(st));
                }
                if (h->mbf8_on && b->stride == 1 && b->in_hw == 8 && b->cin == 384 && b->f16_in == b->f16) {
                    ISB_TRY(b->mbf_w1p.alloc((size_t)b->cexp * b->cin * 2));
                    ISB_TRY(launch_mb8_pack_frag(b->expand.w16.as<uint16_t>(), b->mbf_w1p.p, b->cexp, b->cin, 1, st));
                    ISB_HIP(hipStreamSynchronize(st));
                }
                ISB_TRY(upload(b->dw_w16, wt16.data(), wt16.size() * 2));
                ISB_TRY(upload(b->dw_b, sh->data, (size_t)b->cexp * 4));
                ISB_TRY(blob_get(m, (p + ".se.w1").c_str(), b->cse, b->cexp, &w1));
                ISB_TRY(blob_get(m, (p + ".se.b1").c_str(), b->cse, 1, &b1));
                ISB_TRY(blob_get(m, (p + ".se.w2").c_str(), b->cexp, b->cse, &w2));
                ISB_TRY(blob_get(m, (p + ".se.b2").c_str(), b->cexp, 1, &b2));
                ISB_TRY(upload(b->se_w1, w1->data, w1->numel() * 4));
                ISB_TRY(upload(b->se_b1, b1->data, b1->numel() * 4));
                {   // stored transposed [cse][C]: the gate kernel reads it with the channel on the lane
                    std::vector<float> w2t((size_t)b->cse * b->cexp);
                    for (int c = 0; c < b->cexp; ++c)
                        for (int j = 0; j < b->cse; ++j) w2t[(size_t)j * b->cexp + c] = w2->data[(size_t)c * b->cse + j];
                    ISB_TRY(upload(b->se_w2, w2t.data(), w2t.size() * 4));
                }
                ISB_TRY(upload(b->se_b2, b2->data, b2->numel() * 4));
            }
            h->m->blocks.push_back(std::move(b));
        }
    }
    ISB_REQUIRE(hw == 8, ISB_ERR_WEIGHTS, "internal: backbone plan ends at %dx%d", hw, hw);
    // the chain of stride-1 MBConv blocks on 8 x 8 maps (the tail of the network): weights in the fused kernel's streaming layouts
    {
        h->m->mb8_first = -1; h->m->mb8_count = 0;
        std::vector<Mb8Block> desc;
        for (size_t bi = 0; bi < h->m->blocks.size(); ++bi) {
            BlockW& b = *h->m->blocks[bi];
            const bool fits = h->mb8_on && !b.fused && b.stride == 1 && b.in_hw == 8 && b.f16_in == b.f16 && (b.cin == 384 || b.cin == 640) &&
                              (b.cout == 384 || b.cout == 640) && b.cexp == 6 * b.cin && b.cse == b.cin / 4 && !(b.cin == 640 && b.cout == 384);
            if (!fits) {
                ISB_REQUIRE(desc.empty(), ISB_ERR_WEIGHTS, "internal: the 8 x 8 chain is not the tail of the block list (block %zu)", bi);
                continue;
            }
            if (desc.empty()) h->m->mb8_first = (int)bi;
            ISB_TRY(b.mb_w1p.alloc((size_t)b.cexp * b.cin * 2));
            ISB_TRY(b.mb_w2p.alloc((size_t)b.cout * b.cexp * 2));
            ISB_TRY(b.mb_se1p.alloc((size_t)b.cse * b.cexp * 4));
            ISB_TRY(launch_mb8_pack_frag(b.expand.w16.as<uint16_t>(), b.mb_w1p.p, b.cexp, b.cin, 1, st));
            ISB_TRY(launch_mb8_pack_frag(b.project.w16.as<uint16_t>(), b.mb_w2p.p, b.cout, b.cexp, mb8_proj_group(b.cout), st));
            ISB_TRY(launch_mb8_pack_se1(b.se_w1.as<float>(), b.mb_se1p.as<float>(), b.cse, b.cexp, st));
            Mb8Block d{};
            d.w1p = (const uint4*)b.mb_w1p.p; d.b1 = b.expand.bias.as<float>(); d.dww = b.dw_w16.as<uint16_t>(); d.dwb = b.dw_b.as<float>();
            d.se_w1p = b.mb_se1p.as<float>(); d.se_b1 = b.se_b1.as<float>(); d.se_w2t = b.se_w2.as<float>(); d.se_b2 = b.se_b2.as<float>();
            d.w2p = (const uint4*)b.mb_w2p.p; d.b2 = b.project.bias.as<float>();
            d.cin = b.cin; d.cout = b.cout; d.residual = b.residual ? 1 : 0;
            desc.push_back(d);
        }
        ISB_HIP(hipStreamSynchronize(st));
        h->m->mb8_count = (int)desc.size();
        if (!desc.empty()) ISB_TRY(upload(h->m->mb8_desc, desc.data(), desc.size() * sizeof(Mb8Block)));
    }
    ISB_TRY(upload_conv(m, "bbone.head", 1280, 1, 640, h->m->headconv, st, stream_f16));
    const BlobTensor *hwt, *hb;
    ISB_TRY(blob_get(m, "head.weight", 288, 1280, &hwt));
    ISB_TRY(blob_get(m, "head.bias", 288, 1, &hb));
    ISB_TRY(upload(h->m->head_w, hwt->data, hwt->numel() * 4));
    ISB_TRY(upload(h->m->head_b, hb->data, 288 * 4));
    h->m->weights = true;
    return ISB_OK;
    });
}

// tuning probe: enable != 0 arms the s_memtime stamps of the fused 8 x 8 chain (the next forward passes write them), enable == 0
// copies them out ([32 workgroups][2 blocks][32 marks] uint64) and disarms
extern "C" int isb_debug_hpe_mb8_stamps(isb_hpe* h, int32_t enable, uint64_t* host_out) {
    return isb::guard([&]() -> int {
        ISB_REQUIRE(h, ISB_ERR_INVALID, "null handle");
        ISB_HIP(hipSetDevice(h->cfg.device));
        ISB_HIP(hipDeviceSynchronize());
        const size_t bytes = (size_t)32 * 2 * 32 * 8;
        if (enable) {
            ISB_TRY(h->mb8_stamps.alloc(bytes));
            ISB_HIP(hipMemset(h->mb8_stamps.p, 0, bytes));
        } else {
            ISB_REQUIRE(host_out && h->mb8_stamps.p, ISB_ERR_INVALID, "stamps were not armed");
            ISB_HIP(hipMemcpy(host_out, h->mb8_stamps.p, bytes, hipMemcpyDeviceToHost));
            h->mb8_stamps = DevBuf();
        }
        return ISB_OK;
    });
}

extern "C" int isb_hpe_set_joint_map(isb_hpe* h, const float* expand, const int32_t* indices, int32_t n_out) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(h && expand, ISB_ERR_INVALID, "null argument");
    ISB_REQUIRE(n_out >= 1 && n_out <= 122, ISB_ERR_INVALID, "n_out %d outside [1,122]", n_out);
    ISB_REQUIRE(indices || n_out == 122, ISB_ERR_INVALID, "without indices n_out must be 122");
    ISB_REQUIRE(!h->slot[0].busy && !h->slot[1].busy, ISB_ERR_STATE, "submitted host batches are outstanding: isb_hpe_wait_host first");
    ISB_REQUIRE(!h->is_child, ISB_ERR_STATE, "isb_hpe_set_joint_map on an engine made by isb_hpe_create_shared: set it on the parent (the family reads one model)");
    if (indices)
        for (int i = 0; i < n_out; ++i)
            ISB_REQUIRE(indices[i] >= 0 && indices[i] < 122, ISB_ERR_INVALID, "joint index %d outside [0,122)", indices[i]);
    ISB_HIP(hipSetDevice(h->cfg.device));
    ISB_HIP(hipDeviceSynchronize());           // (see isb_hpe_load_weights)
    ISB_TRY(upload(h->m->expand, expand, 32 * 122 * 4));
    h->m->has_indices = indices != nullptr;
    if (indices) ISB_TRY(upload(h->m->indices, indices, (size_t)n_out * 4));
    h->m->n_out = n_out;
    h->m->jointmap = true;
    return ISB_OK;
    });
}

extern "C" int isb_hpe_set_lanes(isb_hpe* h, int32_t n_lanes) {
    return isb::guard([&]() -> int {
        ISB_REQUIRE(h && n_lanes >= 1 && n_lanes <= kMaxLanes, ISB_ERR_INVALID, "n_lanes must be 1..%d", kMaxLanes);
        ISB_HIP(hipSetDevice(h->cfg.device));
        ISB_HIP(hipDeviceSynchronize());            // no forward pass of this engine is in flight when its lane count changes
        h->n_lanes = n_lanes;
        return ISB_OK;
    });
}

extern "C" int isb_hpe_set_augmentations(isb_hpe* h, int32_t n_aug, const double* rotflip, const double* scales) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(h, ISB_ERR_INVALID, "null handle");
    ISB_REQUIRE(n_aug >= 0 && n_aug <= 64, ISB_ERR_INVALID, "n_aug %d outside [0,64]", n_aug);
    ISB_REQUIRE(!h->slot[0].busy && !h->slot[1].busy, ISB_ERR_STATE, "submitted host batches are outstanding: isb_hpe_wait_host first");
    ISB_REQUIRE(n_aug == 0 || (rotflip && scales), ISB_ERR_INVALID, "augmentation tables missing");
    ISB_HIP(hipSetDevice(h->cfg.device));
    if (n_aug > 0) {
        ISB_TRY(upload(h->aug_rotflip, rotflip, (size_t)n_aug * 72));
        ISB_TRY(upload(h->aug_scale, scales, (size_t)n_aug * 8));
    }
    h->n_aug = n_aug;
    return ISB_OK;
    });
}

extern "C" int isb_hpe_forward(isb_hpe* h, const uint8_t* d_frames, const int32_t* d_bbox, int32_t B, float* d_joints,
                               uint8_t* d_valid, void* stream) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(h && d_frames && d_bbox && d_joints && d_valid, ISB_ERR_INVALID, "null argument");
    ISB_REQUIRE(B >= 1, ISB_ERR_INVALID, "batch %d < 1", B);
    ISB_REQUIRE(h->n_aug == 0, ISB_ERR_STATE,
                "test-time augmentation is on: the reference defines it up to the augmented crops only (hpe.py:88-100; its "
                "decode reshapes to one sample, hpe.py:108) -- use the crop_params / warp / backbone stages");
    ISB_REQUIRE(h->m->weights && h->m->jointmap, ISB_ERR_STATE, "isb_hpe_forward needs weights and a joint map");
    ISB_HIP(hipSetDevice(h->cfg.device));
    hipStream_t st = (hipStream_t)stream;   // NULL = the HIP null stream (torch's default stream)
    const int Bm_max = std::min<int>(B, h->cfg.max_batch);
    const size_t fsz = (size_t)h->cfg.height * h->cfg.width * 3;
    // a lane's pass in phases: 0 = crop parameters, (ROI gather,) warp, stem; 1 .. n = kPhaseBlocks blocks each; last = 640 -> 1280
    // convolution, pose head, decode + reconstruction
    constexpr size_t kPhaseBlocks = 6;
    const int n_phases = 2 + (int)((h->m->blocks.size() + kPhaseBlocks - 1) / kPhaseBlocks);
    auto run_phase = [&](Lane& L, hipStream_t s, int b0, int Bm, int ph, int lane_no) -> int {
        if (ph == 0) {
            ISB_TRY(ensure_ws(L, Bm));
            ISB_TRY(run_crop_params(h, L, s, d_bbox + (size_t)b0 * 4, Bm));
            if (h->roi_dev) {                        // packed ROI image: offsets live in the descriptors
                // (each lane gathers its own frames on its own stream. Measured and equal: one gather for the whole batch up front,
                // 19.5 ms per 256 frames; the lanes' gathers one after the other on the link, 19.5-19.7. What is exposed is the
                // transfer itself: 52 MB at the 20 GB/s a kernel reads mapped host memory, during which at most one lane computes.)
                (void)lane_no;
                ISB_TRY(launch_roi_gather(h->roi_src + (size_t)b0 * fsz, h->roi_dev + b0, const_cast<uint8_t*>(d_frames), Bm, h->cfg.height, h->cfg.width, s));
                ISB_TRY(run_warp(h, L, s, d_frames, Bm, h->roi_dev + b0));
            } else {
                ISB_TRY(run_warp(h, L, s, d_frames + (size_t)b0 * fsz, Bm));
            }
            return backbone_begin(h, L, s, L.crops.as<float>(), Bm);
        }
        if (ph < n_phases - 1) return backbone_blocks(h, L, s, Bm, (size_t)(ph - 1) * kPhaseBlocks, (size_t)ph * kPhaseBlocks);
        ISB_TRY(backbone_end(h, L, s, Bm));
        return run_post(h, L, s, L.logits.as<float>(), Bm, d_joints + (size_t)b0 * h->m->n_out * 3, d_valid + b0, nullptr,
                        d_bbox + (size_t)b0 * 4);
    };
    // (not while the stream is being captured into a graph: a replayed step is ordered by its own launches)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    const bool track = cap == hipStreamCaptureStatusNone;
    if (track && h->last_valid && h->last_stream != st) ISB_HIP(hipStreamWaitEvent(st, h->last_ev, 0));
    for (int b0 = 0; b0 < B; b0 += Bm_max) {
        const int Bm = std::min(Bm_max, B - b0);
        // every sample is independent, so the split changes no result. Small batches (latency regime) and the
        // per-launch profiling pass stay on one lane.
        const int nl = std::min(h->n_lanes, Bm / 32);
        if (nl >= 2 && Bm >= kMinSplit && !h->prof) {
            const int part = (Bm + nl - 1) / nl;
            ISB_HIP(hipEventRecord(h->fork_ev, st));
            for (int l = 1; l < nl; ++l)
                if (l * part < Bm) ISB_HIP(hipStreamWaitEvent(h->lanes[l].side, h->fork_ev, 0));
            // the lanes are fed round-robin, a phase at a time: each stream has work within ~0.2 ms of host time
            for (int ph = 0; ph < n_phases; ++ph)
                for (int l = 0; l < nl; ++l) {
                    const int lo = l * part, n = std::min(part, Bm - lo);
                    if (n <= 0) break;
                    ISB_TRY(run_phase(h->lanes[l], l == 0 ? st : h->lanes[l].side, b0 + lo, n, ph, l));
                }
            for (int l = 1; l < nl; ++l)
                if (l * part < Bm) {
                    ISB_HIP(hipEventRecord(h->join_ev[l], h->lanes[l].side));
                    ISB_HIP(hipStreamWaitEvent(st, h->join_ev[l], 0));
                }
        } else {
            for (int ph = 0; ph < n_phases; ++ph) ISB_TRY(run_phase(h->lanes[0], st, b0, Bm, ph, 0));
        }
    }
    if (track) {
        ISB_HIP(hipEventRecord(h->last_ev, st));
        h->last_stream = st;
        h->last_valid = true;
    }
    return ISB_OK;
    });
}

// the frames as the device sees them, if the caller's buffer is mapped pinned memory (hipHostMalloc / hipHostRegister;
// torch.Tensor.pin_memory()): only then can a kernel pull rectangles out of it. Null: pageable memory, whole frames are copied.
static const uint8_t* mapped_frames(isb_hpe* h, const uint8_t* frames, int B) {
    if (!(h->roi_mode > 0 && B >= kRoiMinBatch && h->cfg.width % 16 == 0)) return nullptr;
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, frames) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer)
        return static_cast<const uint8_t*>(at.devicePointer);
    (void)hipGetLastError();                     // pageable memory: not an error
    return nullptr;
}

// per frame the source rectangle its crop can reach (device bbox d_bbox -> crop homographies -> a 9-KB round trip -> host bounds);
// `off` = bytes of the packed image. Synchronises `st`.
static int host_rois(isb_hpe* h, const int32_t* db, int B, hipStream_t st, std::vector<RoiDesc>& roi, uint64_t& off) {
    const int FW = h->cfg.width, FH = h->cfg.height;
    // 1. the crop homographies of the whole batch (the kernel isb_hpe_forward runs again per lane: same inputs, same bits)
    if (h->hs_H.bytes < (size_t)B * 36) {
        ISB_TRY(h->hs_H.alloc((size_t)B * 36));
        ISB_TRY(h->hs_newK.alloc((size_t)B * 72));
        ISB_TRY(h->hs_R.alloc((size_t)B * 72));
    }
    if (h->hs_roi.bytes < (size_t)B * sizeof(RoiDesc)) ISB_TRY(h->hs_roi.alloc((size_t)B * sizeof(RoiDesc)));
    CropParamArgs ca{};
    ca.bbox = db;
    for (int i = 0; i < 9; ++i) ca.K[i] = h->K[i];
    ca.H = h->hs_H.as<float>(); ca.newK = h->hs_newK.as<double>(); ca.R = h->hs_R.as<double>(); ca.B = B;
    ISB_TRY(launch_crop_params(ca, st));
    std::vector<float> Hh((size_t)B * 9);
    ISB_HIP(hipMemcpyAsync(Hh.data(), h->hs_H.p, (size_t)B * 36, hipMemcpyDeviceToHost, st));
    ISB_HIP(hipStreamSynchronize(st));
    // 2. per frame: the bounding rectangle of the crop square's image under H. The map is projective with a positive
    // denominator over the square, so the square's image is the convex hull of its corners' images; + 2 px for the
    // float32 evaluation and the truncation in the kernel. Degenerate H -> the whole frame.
    roi.assign(B, RoiDesc{});
    off = 0;
    for (int b = 0; b < B; ++b) {
        const float* Hb = Hh.data() + (size_t)b * 9;
        double xlo = 1e30, xhi = -1e30, ylo = 1e30, yhi = -1e30;
        bool ok = std::isfinite(Hb[8]) && Hb[8] != 0.f;
        for (int c = 0; c < 4 && ok; ++c) {
            const double x = (c & 1) ? 255.0 : 0.0, y = (c & 2) ? 255.0 : 0.0, h8 = Hb[8];
            const double k = Hb[6] / h8 * x + Hb[7] / h8 * y + 1.0;
            if (!(k > 1e-6)) { ok = false; break; }
            const double xs = (Hb[0] / h8 * x + Hb[1] / h8 * y + Hb[2] / h8) / k;
            const double ys = (Hb[3] / h8 * x + Hb[4] / h8 * y + Hb[5] / h8) / k;
            if (!std::isfinite(xs) || !std::isfinite(ys)) { ok = false; break; }
            xlo = std::min(xlo, xs); xhi = std::max(xhi, xs); ylo = std::min(ylo, ys); yhi = std::max(yhi, ys);
        }
        int x0 = 0, x1 = FW - 1, y0 = 0, y1 = FH - 1;
        if (ok) {
            x0 = (int)std::max(0.0, std::floor(std::min(xlo, 1e9)) - 2.0);
            y0 = (int)std::max(0.0, std::floor(std::min(ylo, 1e9)) - 2.0);
            x1 = (int)std::min((double)FW - 1.0, std::ceil(std::max(xhi, -1e9)) + 2.0);
            y1 = (int)std::min((double)FH - 1.0, std::ceil(std::max(yhi, -1e9)) + 2.0);
            x0 &= ~15;                                              // 16-pixel (48-byte) alignment: whole 16-byte pieces
            x1 = std::min(FW - 1, x1 | 15);
        }
        RoiDesc& r = roi[b];
        if (x1 < x0 || y1 < y0) { r.x0 = 0; r.y0 = 0; r.w = 0; r.h = 0; r.off = off; continue; }   // the crop misses the frame
        r.x0 = x0; r.y0 = y0; r.w = x1 - x0 + 1; r.h = y1 - y0 + 1; r.off = off;
        off += ((uint64_t)r.w * r.h * 3 + 15) & ~15ull;
    }
    return ISB_OK;
}

extern "C" int isb_hpe_forward_host(isb_hpe* h, const uint8_t* frames, const int32_t* bbox, int32_t B, float* joints,
                                    uint8_t* valid) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(h && frames && bbox && joints && valid, ISB_ERR_INVALID, "null argument");
    ISB_REQUIRE(B >= 1, ISB_ERR_INVALID, "batch %d < 1", B);
    ISB_REQUIRE(h->n_aug == 0, ISB_ERR_STATE, "test-time augmentation is on: the reference defines it up to the augmented crops only (hpe.py:88-100)");
    ISB_REQUIRE(h->m->weights && h->m->jointmap, ISB_ERR_STATE, "isb_hpe_forward_host needs weights and a joint map");
    ISB_HIP(hipSetDevice(h->cfg.device));
    hipStream_t st = h->own_stream;
    const size_t fsz = (size_t)h->cfg.height * h->cfg.width * 3;
    if (B > h->hs_B) {                          // staging buffers live in the handle (no hipMalloc per call), grow-only
        h->hs_B = 0;
        ISB_TRY(h->hs_frames.alloc(fsz * B));
        ISB_TRY(h->hs_bbox.alloc((size_t)B * 16));
        ISB_TRY(h->hs_joints.alloc((size_t)B * kMaxJoints * 12));
        ISB_TRY(h->hs_valid.alloc((size_t)B));
        h->hs_B = B;
    }
    uint8_t* df = h->hs_frames.as<uint8_t>();
    int32_t* db = h->hs_bbox.as<int32_t>();
    float* dj = h->hs_joints.as<float>();
    uint8_t* dv = h->hs_valid.as<uint8_t>();
    ISB_HIP(hipMemcpyAsync(db, bbox, (size_t)B * 16, hipMemcpyHostToDevice, st));
    const uint8_t* frames_mapped = mapped_frames(h, frames, B);
    if (frames_mapped) {
        std::vector<RoiDesc> roi;
        uint64_t off = 0;
        ISB_TRY(host_rois(h, db, B, st, roi, off));
        ISB_REQUIRE(off <= h->hs_frames.bytes, ISB_ERR_INVALID, "internal: packed ROI image larger than the frame staging buffer");
        // 3. one gather kernel per lane pulls the rectangles out of the mapped host frames (aligned 16-byte reads over PCIe),
        // launched by isb_hpe_forward in front of each lane's warp
        ISB_HIP(hipMemcpyAsync(h->hs_roi.p, roi.data(), (size_t)B * sizeof(RoiDesc), hipMemcpyHostToDevice, st));
        h->roi_dev = h->hs_roi.as<RoiDesc>();
        h->roi_src = frames_mapped;
        const int rc = isb_hpe_forward(h, df, db, B, dj, dv, st);
        h->roi_dev = nullptr;
        h->roi_src = nullptr;
        if (rc != ISB_OK) return rc;
        // `roi` and `Hh` must outlive the asynchronous copies that read them: the synchronise below covers it
    } else {
    // up to four chunks of >= 256 frames: the frames of chunk i + 1 cross PCIe (copy stream) while chunk i computes. Smaller
    // batches go in one piece: measured at 256 frames, two chunks of 128 lose in launch efficiency (64-frame lanes) exactly what
    // the overlapped copy gains (23.6 ms either way, 18.0 with resident frames)
    const int n_chunks = std::max(1, std::min(4, B / 256));
    const int per = (B + n_chunks - 1) / n_chunks;
    for (int c = 0; c < n_chunks; ++c) {
        const int b0 = c * per, n = std::min(per, B - b0);
        if (n <= 0) break;
        ISB_HIP(hipMemcpyAsync(df + (size_t)b0 * fsz, frames + (size_t)b0 * fsz, fsz * n, hipMemcpyHostToDevice, h->copy_stream));
        ISB_HIP(hipEventRecord(h->h2d_ev[c], h->copy_stream));
    }
    for (int c = 0; c < n_chunks; ++c) {
        const int b0 = c * per, n = std::min(per, B - b0);
        if (n <= 0) break;
        ISB_HIP(hipStreamWaitEvent(st, h->h2d_ev[c], 0));
        ISB_TRY(isb_hpe_forward(h, df + (size_t)b0 * fsz, db + (size_t)b0 * 4, n, dj + (size_t)b0 * h->m->n_out * 3, dv + b0, st));
    }
    }
    ISB_HIP(hipMemcpyAsync(joints, dj, (size_t)B * h->m->n_out * 12, hipMemcpyDeviceToHost, st));
    ISB_HIP(hipMemcpyAsync(valid, dv, (size_t)B, hipMemcpyDeviceToHost, st));
    ISB_HIP(hipStreamSynchronize(st));
    return ISB_OK;
    });
}

static int complete_oldest(isb_hpe* h) {
    isb_hpe::HostSlot& sl = h->slot[h->sub_tail];
    ISB_HIP(hipEventSynchronize(sl.done));
    const size_t jb = (size_t)sl.B * sl.n_out * 12;
    memcpy(sl.user_joints, sl.pin, jb);
    memcpy(sl.user_valid, static_cast<const uint8_t*>(sl.pin) + jb, (size_t)sl.B);
    sl.busy = false;
    h->sub_tail ^= 1;
    return ISB_OK;
}

extern "C" int isb_hpe_submit_host(isb_hpe* h, const uint8_t* frames, const int32_t* bbox, int32_t B, float* joints, uint8_t* valid) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(h && frames && bbox && joints && valid, ISB_ERR_INVALID, "null argument");
    ISB_REQUIRE(B >= 1, ISB_ERR_INVALID, "batch %d < 1", B);
    ISB_REQUIRE(h->n_aug == 0, ISB_ERR_STATE, "test-time augmentation is on: the reference defines it up to the augmented crops only (hpe.py:88-100)");
    ISB_REQUIRE(h->m->weights && h->m->jointmap, ISB_ERR_STATE, "isb_hpe_submit_host needs weights and a joint map");
    ISB_HIP(hipSetDevice(h->cfg.device));
    isb_hpe::HostSlot& sl = h->slot[h->sub_head];
    if (sl.busy) ISB_TRY(complete_oldest(h));    // two in flight already: the oldest (this slot) is finished first
    const size_t fsz = (size_t)h->cfg.height * h->cfg.width * 3;
    if (B > sl.cap) {
        sl.cap = 0;
        ISB_TRY(sl.frames.alloc(fsz * B));
        ISB_TRY(sl.bbox.alloc((size_t)B * 16));
        ISB_TRY(sl.joints.alloc((size_t)B * kMaxJoints * 12));
        ISB_TRY(sl.valid.alloc((size_t)B));
        if (sl.pin) (void)hipHostFree(sl.pin);
        sl.pin = nullptr;
        sl.pin_bytes = (size_t)B * (kMaxJoints * 12 + 1);
        ISB_HIP(hipHostMalloc(&sl.pin, sl.pin_bytes, hipHostMallocDefault));
        sl.cap = B;
    }
    hipStream_t st = h->own_stream, cs = h->copy_stream;
    sl.boxes.assign(bbox, bbox + (size_t)B * 4);
    // whole frames on the copy engine (no CU: a rectangle gather's waves would sit on their CUs for the length of the transfer and
    // the previous batch's one-workgroup-per-CU kernels would find CUs taken -- measured, 21.3 against 19.2 ms per 256 frames)
    ISB_HIP(hipMemcpyAsync(sl.bbox.p, sl.boxes.data(), (size_t)B * 16, hipMemcpyHostToDevice, cs));
    ISB_HIP(hipMemcpyAsync(sl.frames.p, frames, fsz * B, hipMemcpyHostToDevice, cs));
    ISB_HIP(hipEventRecord(sl.h2d, cs));
    ISB_HIP(hipStreamWaitEvent(st, sl.h2d, 0));
    ISB_TRY(isb_hpe_forward(h, sl.frames.as<uint8_t>(), sl.bbox.as<int32_t>(), B, sl.joints.as<float>(), sl.valid.as<uint8_t>(), st));
    const size_t jb = (size_t)B * h->m->n_out * 12;
    ISB_HIP(hipMemcpyAsync(sl.pin, sl.joints.p, jb, hipMemcpyDeviceToHost, st));
    ISB_HIP(hipMemcpyAsync(static_cast<uint8_t*>(sl.pin) + jb, sl.valid.p, (size_t)B, hipMemcpyDeviceToHost, st));
    ISB_HIP(hipEventRecord(sl.done, st));
    sl.user_joints = joints; sl.user_valid = valid; sl.B = B; sl.n_out = h->m->n_out; sl.busy = true;
    h->sub_head ^= 1;
    return ISB_OK;
    });
}

extern "C" int isb_hpe_wait_host(isb_hpe* h) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(h, ISB_ERR_INVALID, "null argument");
    ISB_REQUIRE(h->slot[h->sub_tail].busy, ISB_ERR_STATE, "isb_hpe_wait_host: no submission outstanding");
    ISB_HIP(hipSetDevice(h->cfg.device));
    return complete_oldest(h);
    });
}

// ---------------------------------------------------------------- stage-level hooks (tests)
extern "C" int isb_hpe_crop_params_host(isb_hpe* h, const int32_t* bbox, int32_t B, float* H, double* newK, double* R) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(h && bbox && H && newK && R && B >= 1, ISB_ERR_INVALID, "bad argument");
    ISB_HIP(hipSetDevice(h->cfg.device));
    hipStream_t st = h->own_stream;
    Lane& L = h->lanes[0];
    const int n = B * std::max(h->n_aug, 1);           // parameter sets: [B][n_aug]
    ISB_REQUIRE(n <= h->cfg.max_batch, ISB_ERR_INVALID, "B x n_aug = %d exceeds max_batch %d", n, h->cfg.max_batch);
    ISB_TRY(ensure_ws(L, n));
    DevBuf db;
    ISB_TRY(upload(db, bbox, (size_t)B * 16));
    ISB_TRY(run_crop_params(h, L, st, db.as<int32_t>(), B));
    ISB_HIP(hipStreamSynchronize(st));
    ISB_HIP(hipMemcpy(H, L.H.p, (size_t)n * 36, hipMemcpyDeviceToHost));
    ISB_HIP(hipMemcpy(newK, L.newK.p, (size_t)n * 72, hipMemcpyDeviceToHost));
    ISB_HIP(hipMemcpy(R, L.R.p, (size_t)n * 72, hipMemcpyDeviceToHost));
    return ISB_OK;
    });
}

extern "C" int isb_hpe_warp_host(isb_hpe* h, const uint8_t* frames, const int32_t* bbox, int32_t B, float* crops) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(h && frames && bbox && crops && B >= 1, ISB_ERR_INVALID, "bad argument");
    ISB_HIP(hipSetDevice(h->cfg.device));
    hipStream_t st = h->own_stream;
    Lane& L = h->lanes[0];
    const int n = B * std::max(h->n_aug, 1);           // crops: [B][n_aug]
    ISB_REQUIRE(n <= h->cfg.max_batch, ISB_ERR_INVALID, "B x n_aug = %d exceeds max_batch %d", n, h->cfg.max_batch);
    ISB_TRY(ensure_ws(L, n));
    const size_t fsz = (size_t)h->cfg.height * h->cfg.width * 3;
    DevBuf df, db;
    ISB_TRY(upload(df, frames, fsz * B));
    ISB_TRY(upload(db, bbox, (size_t)B * 16));
    ISB_TRY(run_crop_params(h, L, st, db.as<int32_t>(), B));
    ISB_TRY(run_warp(h, L, st, df.as<uint8_t>(), B));
    ISB_HIP(hipStreamSynchronize(st));
    ISB_HIP(hipMemcpy(crops, L.crops.p, (size_t)n * 256 * 256 * 3 * 4, hipMemcpyDeviceToHost));
    return ISB_OK;
    });
}

extern "C" int isb_hpe_backbone_host(isb_hpe* h, const float* crops, int32_t B, float* features, float* logits) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(h && crops && B >= 1, ISB_ERR_INVALID, "bad argument");
    ISB_REQUIRE(h->m->weights, ISB_ERR_STATE, "isb_hpe_backbone_host before isb_hpe_load_weights");
    ISB_HIP(hipSetDevice(h->cfg.device));
    hipStream_t st = h->own_stream;
    Lane& L = h->lanes[0];
    ISB_TRY(ensure_ws(L, std::min<int>(B, h->cfg.max_batch)));
    ISB_REQUIRE(B <= L.ws_B, ISB_ERR_INVALID, "B %d exceeds max_batch %d", B, h->cfg.max_batch);
    ISB_HIP(hipMemcpy(L.crops.p, crops, (size_t)B * 256 * 256 * 3 * 4, hipMemcpyHostToDevice));
    ISB_TRY(run_backbone(h, L, st, L.crops.as<float>(), B));
    ISB_HIP(hipStreamSynchronize(st));
    if (features) ISB_HIP(hipMemcpy(features, L.feat.p, (size_t)B * 64 * 1280 * 4, hipMemcpyDeviceToHost));
    if (logits) ISB_HIP(hipMemcpy(logits, L.logits.p, (size_t)B * 64 * 288 * 4, hipMemcpyDeviceToHost));
    return ISB_OK;
    });
}

extern "C" int isb_hpe_post_host(isb_hpe* h, const float* logits, const int32_t* bbox, int32_t B, float* joints,
                                 uint8_t* valid, double* pred) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(h && logits && bbox && joints && valid && B >= 1, ISB_ERR_INVALID, "bad argument");
    ISB_REQUIRE(h->m->jointmap, ISB_ERR_STATE, "isb_hpe_post_host before isb_hpe_set_joint_map");
    ISB_REQUIRE(h->n_aug == 0, ISB_ERR_STATE, "test-time augmentation is on: the reference's decode takes one sample (hpe.py:108)");
    ISB_HIP(hipSetDevice(h->cfg.device));
    hipStream_t st = h->own_stream;
    Lane& L = h->lanes[0];
    ISB_TRY(ensure_ws(L, std::min<int>(B, h->cfg.max_batch)));
    ISB_REQUIRE(B <= L.ws_B, ISB_ERR_INVALID, "B %d exceeds max_batch %d", B, h->cfg.max_batch);
    DevBuf db, dl, dj, dv, dd;
    ISB_TRY(upload(db, bbox, (size_t)B * 16));
    ISB_TRY(upload(dl, logits, (size_t)B * 64 * 288 * 4));
    ISB_TRY(dj.alloc((size_t)B * h->m->n_out * 12));
    ISB_TRY(dv.alloc((size_t)B));
    if (pred) ISB_TRY(dd.alloc((size_t)B * 32 * 5 * 8));
    ISB_TRY(run_crop_params(h, L, st, db.as<int32_t>(), B));
    ISB_TRY(run_post(h, L, st, dl.as<float>(), B, dj.as<float>(), dv.as<uint8_t>(), pred ? dd.as<double>() : nullptr));
    ISB_HIP(hipStreamSynchronize(st));
    ISB_HIP(hipMemcpy(joints, dj.p, (size_t)B * h->m->n_out * 12, hipMemcpyDeviceToHost));
    ISB_HIP(hipMemcpy(valid, dv.p, (size_t)B, hipMemcpyDeviceToHost));
    if (pred) ISB_HIP(hipMemcpy(pred, dd.p, (size_t)B * 32 * 5 * 8, hipMemcpyDeviceToHost));
    return ISB_OK;
    });
}

extern "C" int isb_hpe_profile(isb_hpe* h, int32_t enable) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(h, ISB_ERR_INVALID, "null handle");
    h->prof = enable != 0;
    if (!h->prof) {          // a caller that never reads the depthwise list (isb_hpe_profile_read_dw) must not keep its events alive
        ISB_HIP(hipSetDevice(h->cfg.device));
        ISB_TRY(drain_prof_dw(h));
    }
    return ISB_OK;
    });
}

extern "C" int isb_hpe_profile_read(isb_hpe* h, double* ms_total, int64_t* launches) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(h && ms_total && launches, ISB_ERR_INVALID, "null argument");
    ISB_HIP(hipSetDevice(h->cfg.device));
    for (auto& e : h->prof_ev) {
        ISB_HIP(hipEventSynchronize(e.second));
        float ms = 0.f;
        ISB_HIP(hipEventElapsedTime(&ms, e.first, e.second));
        h->prof_ms += ms;
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    h->prof_ev.clear();
    ISB_TRY(drain_prof_dw(h));
    *ms_total = h->prof_ms;
    *launches = h->prof_launches;
    h->prof_ms = 0.0;
    h->prof_launches = 0;
    return ISB_OK;
    });
}

extern "C" int isb_hpe_profile_read_dw(isb_hpe* h, double* ms_total, int64_t* launches) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(h && ms_total && launches, ISB_ERR_INVALID, "null argument");
    ISB_HIP(hipSetDevice(h->cfg.device));
    ISB_TRY(drain_prof_dw(h));
    *ms_total = h->prof_dw_ms;
    *launches = h->prof_dw_launches;
    h->prof_dw_ms = 0.0;
    h->prof_dw_launches = 0;
    return ISB_OK;
    });
}

extern "C" int isb_pose_windows(const float* d_joints, int32_t n_cam, int32_t n_frames, int32_t J, int32_t L,
                                float* d_windows, void* stream) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(d_joints && d_windows, ISB_ERR_INVALID, "null argument");
    ISB_REQUIRE(n_cam >= 1 && J >= 1 && L >= 1 && n_frames >= L, ISB_ERR_INVALID,
                "bad shape n_cam=%d n_frames=%d J=%d L=%d", n_cam, n_frames, J, L);
    ISB_TRY(set_device_of(d_joints));         // no handle here: launch on the device that owns the buffers
    return launch_pose_windows(d_joints, n_cam, n_frames, J, L, d_windows, (hipStream_t)stream);
    });
}

extern "C" int isb_pose_distance(const float* d_joints, int32_t n, int32_t J, float* d_distance, void* stream) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(d_joints && d_distance, ISB_ERR_INVALID, "null argument");
    ISB_REQUIRE(n >= 1 && J >= 1, ISB_ERR_INVALID, "bad shape n=%d J=%d", n, J);
    ISB_TRY(set_device_of(d_joints));
    return launch_pose_distance(d_joints, n, J, d_distance, (hipStream_t)stream);
    });
}

// test / tuning hook: one conv_igemm layer on host tensors, timed with HIP events
extern "C" int isb_debug_conv(int32_t device, const uint16_t* x, const float* w, const float* scale, const float* shift,
                              const uint16_t* res, const float* gate, int32_t B, int32_t H, int32_t W, int32_t Cin,
                              int32_t Cout, int32_t k, int32_t stride, int32_t act, int32_t variant, int32_t iters,
                              uint16_t* out, float* ms_per_iter) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(x && w && scale && shift && out && ms_per_iter, ISB_ERR_INVALID, "null argument");
    const int sym2 = (stride & 0x100) ? 1 : 0;  // stride 2 with PyTorch's symmetric pad 1 (the detector / ResNet trunk) instead of TF-SAME
    stride &= 0xff;
    ISB_REQUIRE((k == 1 || k == 3) && (stride == 1 || stride == 2) && iters >= 1, ISB_ERR_INVALID, "bad conv parameters");
    ISB_HIP(hipSetDevice(device));
    const int f16 = (act & 0x100) ? 1 : 0;      // x / res / out hold fp16 bits and the weights are rounded to fp16
    act &= 0xff;
    const int OH = H / stride, OW = W / stride;
    const size_t nin = (size_t)B * H * W * Cin, nout = (size_t)B * OH * OW * Cout, nw = (size_t)Cout * k * k * Cin;
    DevBuf dx, dwf, dsc, dsh, dw16, dres, dgate, dout, dzero;
    ISB_TRY(dzero.alloc(256));
    ISB_HIP(hipMemset(dzero.p, 0, 256));
    ISB_TRY(upload(dx, x, nin * 2));
    ISB_TRY(upload(dwf, w, nw * 4));
    ISB_TRY(upload(dsc, scale, (size_t)Cout * 4));
    ISB_TRY(upload(dsh, shift, (size_t)Cout * 4));
    ISB_TRY(dw16.alloc(nw * 2));
    ISB_TRY(dout.alloc(nout * 2));
    if (res) ISB_TRY(upload(dres, res, nout * 2));
    if (gate) ISB_TRY(upload(dgate, gate, (size_t)B * Cin * 4));
    ISB_TRY(launch_f32_to_bf16_rows(dwf.as<float>(), dsc.as<float>(), dw16.as<uint16_t>(), Cout, (size_t)k * k * Cin, nullptr, f16));
    ConvArgs a{};
    a.f16 = f16;
    a.in = dx.as<uint16_t>(); a.w = dw16.as<uint16_t>(); a.bias = dsh.as<float>();
    a.res = res ? dres.as<uint16_t>() : nullptr; a.gate = gate ? dgate.as<float>() : nullptr; a.out = dout.p;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KH = k; a.KW = k; a.stride = stride; a.OH = OH; a.OW = OW;
    a.pad = (k == 3 && (stride == 1 || sym2)) ? 1 : 0; a.M = B * OH * OW; a.K = k * k * Cin; a.act = act; a.out_f32 = 0; a.zeros = dzero.as<uint16_t>();
    a.variant = variant % 1000;
    DevBuf dpart;
    if (variant >= 900000) {                    // kernels with in-kernel time stamps (tuning probes): 900181, 900131, 900143, ...
        ISB_TRY(dpart.alloc(128 * 1024));
        ISB_HIP(hipMemset(dpart.p, 0, 128 * 1024));
        a.part = dpart.as<float>();
        a.probe = 2;
        a.variant = variant - 900000;
    } else if (variant >= 2000) {               // variant = 1000 * splits + tile variant: split-K
        a.splits = variant / 1000;
        ISB_TRY(dpart.alloc((size_t)a.splits * a.M * Cout * 4));
        a.part = dpart.as<float>();
    }
    ISB_TRY(launch_conv_igemm(a, nullptr));       // warm-up + result
    ISB_HIP(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    ISB_HIP(hipEventCreate(&e0));
    ISB_HIP(hipEventCreate(&e1));
    ISB_HIP(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) ISB_TRY(launch_conv_igemm(a, nullptr));
    ISB_HIP(hipEventRecord(e1, nullptr));
    ISB_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    ISB_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_per_iter = ms / iters;
    ISB_HIP(hipMemcpy(out, dout.p, nout * 2, hipMemcpyDeviceToHost));
    if (variant >= 900000 && variant != 900181 && variant != 900182 && variant != 900183 && variant != 900184 && variant != 900185 && variant != 900186 && variant != 900187 && variant != 900188) {   // the tile GEMMs: per workgroup {prologue, DMA wait, barrier wait, k loop, epilogue, k-steps}
        std::vector<uint64_t> st(64 * 8);
        ISB_HIP(hipMemcpy(st.data(), dpart.p, 64 * 64, hipMemcpyDeviceToHost));
        double sum[5] = {0, 0, 0, 0, 0};
        int n = 0;
        for (int g = 0; g < 64; ++g) {
            const uint64_t* o = st.data() + (size_t)g * 8;
            if (!o[5]) continue;
            for (int i = 0; i < 5; ++i) sum[i] += (double)o[i];
            ++n;
            if (g < 4)
                fprintf(stderr, "wg %2d: prologue %6llu | k loop %7llu (%llu k-steps: DMA wait %6llu, barrier %6llu) | epilogue %6llu\n", g,
                        (unsigned long long)o[0], (unsigned long long)o[3], (unsigned long long)o[5], (unsigned long long)o[1],
                        (unsigned long long)o[2], (unsigned long long)o[4]);
        }
        if (n)
            fprintf(stderr, "mean of %d workgroups (cycles, wave 0): prologue %.0f | k loop %.0f of which DMA wait %.0f, barrier wait %.0f | epilogue %.0f\n",
                    n, sum[0] / n, sum[3] / n, sum[1] / n, sum[2] / n, sum[4] / n);
    }
    if (variant >= 900183 && variant <= 900188) {                        // wave 0: cycles per phase of tiles 1..4
        std::vector<uint64_t> st(64 * 128);
        ISB_HIP(hipMemcpy(st.data(), dpart.p, 64 * 1024, hipMemcpyDeviceToHost));
        {   // every workgroup's entry / exit on the constant 100 MHz counter
            std::vector<uint64_t> rt(2 * 1024);
            ISB_HIP(hipMemcpy(rt.data(), (const char*)dpart.p + 64 * 1024, 16 * 1024, hipMemcpyDeviceToHost));
            uint64_t e_min = ~0ull, e_max = 0, x_min = ~0ull, x_max = 0;
            int n = 0;
            for (int g = 0; g < 1024; ++g) {
                if (!rt[2 * g] || !rt[2 * g + 1]) continue;
                e_min = std::min(e_min, rt[2 * g]); e_max = std::max(e_max, rt[2 * g]);
                x_min = std::min(x_min, rt[2 * g + 1]); x_max = std::max(x_max, rt[2 * g + 1]);
                ++n;
            }
            if (n) {
                const int nwg = (a.variant == 184 || a.variant == 187) ? 512 : 256, nsl = cdiv(Cout, 128), Q = std::max(1, nwg / nsl);
                double xs[8] = {0}, ss[64] = {0}; int xn[8] = {0}, sn[64] = {0};
                for (int g = 0; g < nwg; ++g) {
                    if (!rt[2 * g] || !rt[2 * g + 1]) continue;
                    const double d = (double)(rt[2 * g + 1] - rt[2 * g]) * 0.01;
                    const int idx = (g & 7) * (nwg >> 3) + (g >> 3), sl = idx % nsl;
                    xs[g & 7] += d; ++xn[g & 7];
                    if (idx / nsl < Q && sl < 64) { ss[sl] += d; ++sn[sl]; }
                }
                {
                    std::vector<std::pair<double, int>> ds;
                    for (int g = 0; g < nwg; ++g)
                        if (rt[2 * g] && rt[2 * g + 1]) ds.push_back({(double)(rt[2 * g + 1] - rt[2 * g]) * 0.01, g});
                    std::sort(ds.begin(), ds.end());
                    fprintf(stderr, "workgroup duration deciles (us):");
                    for (int d = 0; d <= 10; ++d) fprintf(stderr, " %.1f", ds[std::min(ds.size() - 1, ds.size() * d / 10)].first);
                    fprintf(stderr, "\nslowest:");
                    for (size_t k = ds.size() - 8; k < ds.size(); ++k) {
                        const int g = ds[k].second, idx = (g & 7) * (nwg >> 3) + (g >> 3);
                        fprintf(stderr, " [g %d xcd %d slice %d q %d: %.1f]", g, g & 7, idx % nsl, idx / nsl, ds[k].first);
                    }
                    fprintf(stderr, "\nfastest:");
                    for (size_t k = 0; k < 8; ++k) {
                        const int g = ds[k].second, idx = (g & 7) * (nwg >> 3) + (g >> 3);
                        fprintf(stderr, " [g %d xcd %d slice %d q %d: %.1f]", g, g & 7, idx % nsl, idx / nsl, ds[k].first);
                    }
                    fprintf(stderr, "\n");
                }
                fprintf(stderr, "mean workgroup duration by XCD (us):");
                for (int x = 0; x < 8; ++x) fprintf(stderr, " %.1f", xn[x] ? xs[x] / xn[x] : 0.0);
                fprintf(stderr, "\nmean workgroup duration by channel slice (us):");
                for (int x = 0; x < nsl && x < 64; ++x) fprintf(stderr, " %.1f", sn[x] ? ss[x] / sn[x] : 0.0);
                fprintf(stderr, "\n");
            }
            if (n)
                fprintf(stderr, "%d workgroups: entries spread over %.2f us, first exit at %.2f us, last exit at %.2f us after the first entry\n", n,
                        (double)(e_max - e_min) * 0.01, (double)(x_min - e_min) * 0.01, (double)(x_max - e_min) * 0.01);
        }
        uint64_t t_first = ~0ull;
        for (int g = 0; g < 64; ++g)
            if (st[(size_t)g * 128 + 120]) t_first = std::min(t_first, st[(size_t)g * 128 + 120]);
        for (int g : {0, 1, 2, 8, 16, 33, 63}) {
            const uint64_t* w = st.data() + (size_t)g * 128 + 120;
            if (!w[0]) continue;
            fprintf(stderr, "wg %2d: entry at +%6lld | prologue %6lld | %2lld tiles in %7lld | last epilogue %6lld | entry to exit %7lld cycles = %.1f us (%.2f GHz)\n", g,
                    (long long)(w[0] - t_first), (long long)(w[1] - w[0]), (long long)w[4], (long long)(w[2] - w[1]), (long long)(w[3] - w[2]),
                    (long long)(w[3] - w[0]), (double)(w[6] - w[5]) * 0.01, (double)(w[3] - w[0]) / ((double)(w[6] - w[5]) * 10.0));
        }
        for (int g : {0, 1, 8, 33})
            for (int tl = 0; tl < 4; ++tl) {
                const uint64_t* s0 = st.data() + (size_t)g * 128 + tl * 8;
                if (!s0[0]) continue;
                fprintf(stderr, "wg %2d tile %d: barrier %5lld  tile pass %5lld  accumulator copy %5lld  total %5lld\n", g, tl + 1,
                        (long long)(s0[1] - s0[0]), (long long)(s0[2] - s0[1]), (long long)(s0[3] - s0[2]), (long long)(s0[3] - s0[0]));
            }
    }
    if (variant == 900181 || variant == 900182) {   // print the stamps of the last launch: cycles per phase, per tile
        std::vector<uint64_t> st(64 * 128);
        ISB_HIP(hipMemcpy(st.data(), dpart.p, 64 * 1024, hipMemcpyDeviceToHost));
        for (int g : {0, 1, 8, 33}) {
            for (int tl = 0; tl < 4; ++tl)
                for (int half = 0; half < 2; ++half) {
                    const uint64_t* s0 = st.data() + (size_t)g * 128 + (tl * 2 + half) * 8;
                    if (!s0[0]) continue;
                    fprintf(stderr, "wg %2d tile %d wave %c: barrier %5lld  first phase %5lld  second phase %5lld  total %5lld\n", g,
                            tl + 1, half ? 'B' : 'A', (long long)(s0[1] - s0[0]), (long long)(s0[2] - s0[1]), (long long)(s0[3] - s0[2]),
                            (long long)(s0[3] - s0[0]));
                }
        }
    }
    return ISB_OK;
    });
}

extern "C" int isb_hpe_select_person(isb_hpe* h, const float* d_boxes, const float* d_confs, int32_t B, float conf_thresh,
                                     int32_t* d_bbox, uint8_t* d_found, void* stream) {
    return isb::guard([&]() -> int {
        ISB_REQUIRE(h && d_boxes && d_confs && d_bbox && B >= 1, ISB_ERR_INVALID, "bad argument");
        ISB_HIP(hipSetDevice(h->cfg.device));
        return launch_select_person(d_boxes, d_confs, B, 4032, 80, conf_thresh, h->cfg.width, h->cfg.height, d_bbox, d_found,
                                    (hipStream_t)stream);
    });
}

extern "C" int isb_hpe_select_person_host(isb_hpe* h, const float* boxes, const float* confs, int32_t B, float conf_thresh,
                                          int32_t* bbox, uint8_t* found) {
    return isb::guard([&]() -> int {
        ISB_REQUIRE(h && boxes && confs && bbox && found && B >= 1, ISB_ERR_INVALID, "bad argument");
        ISB_HIP(hipSetDevice(h->cfg.device));
        DevBuf db, dc, dbb, df;
        ISB_TRY(upload(db, boxes, (size_t)B * 4032 * 4 * 4));
        ISB_TRY(upload(dc, confs, (size_t)B * 4032 * 80 * 4));
        ISB_TRY(dbb.alloc((size_t)B * 16));
        ISB_TRY(df.alloc((size_t)B));
        ISB_TRY(launch_select_person(db.as<float>(), dc.as<float>(), B, 4032, 80, conf_thresh, h->cfg.width, h->cfg.height,
                                     dbb.as<int32_t>(), df.as<uint8_t>(), h->own_stream));
        ISB_HIP(hipStreamSynchronize(h->own_stream));
        ISB_HIP(hipMemcpy(bbox, dbb.p, (size_t)B * 16, hipMemcpyDeviceToHost));
        ISB_HIP(hipMemcpy(found, df.p, (size_t)B, hipMemcpyDeviceToHost));
        return ISB_OK;
    });
}

// test / tuning hook: a whole Fused-MBConv block in one launch on host tensors
extern "C" int isb_debug_fused_mb(int32_t device, const uint16_t* x, const float* w1, const float* scale1, const float* shift1,
                                  const float* w2, const float* scale2, const float* shift2, const uint16_t* res, int32_t B,
                                  int32_t H, int32_t Cin, int32_t Cexp, int32_t Cout2, int32_t stride, int32_t iters,
                                  uint16_t* out, float* ms_per_iter) {
    return isb::guard([&]() -> int {
        ISB_REQUIRE(x && w1 && scale1 && shift1 && w2 && scale2 && shift2 && out && ms_per_iter, ISB_ERR_INVALID, "null argument");
        const int f16 = (stride & 0x100) ? 1 : 0;      // x / res / out hold fp16 bits and the weights are rounded to fp16
        const int stamps = (stride & 0x200) ? 1 : 0;   // tuning probe: phase clocks of workgroups 256 .. 319 to stderr
        const int lds_e = (stride & 0x400) ? 1 : 0;    // the E tile through LDS (round 2's form) instead of the projection from the accumulators
        stride &= 0xff;
        ISB_REQUIRE((stride == 1 || stride == 2) && iters >= 1 && B >= 1, ISB_ERR_INVALID, "bad parameters");
        ISB_HIP(hipSetDevice(device));
        const int OH = H / stride;
        const size_t nin = (size_t)B * H * H * Cin, nout = (size_t)B * OH * OH * Cout2;
        const size_t nw1 = (size_t)Cexp * 9 * Cin, nw2 = (size_t)Cout2 * Cexp;
        DevBuf dx, dw1f, ds1, db1, dw1, dw2f, ds2, db2, dw2, dres, dout;
        ISB_TRY(upload(dx, x, nin * 2));
        ISB_TRY(upload(dw1f, w1, nw1 * 4));
        ISB_TRY(upload(ds1, scale1, (size_t)Cexp * 4));
        ISB_TRY(upload(db1, shift1, (size_t)Cexp * 4));
        ISB_TRY(upload(dw2f, w2, nw2 * 4));
        ISB_TRY(upload(ds2, scale2, (size_t)Cout2 * 4));
        ISB_TRY(upload(db2, shift2, (size_t)Cout2 * 4));
        ISB_TRY(dw1.alloc(nw1 * 2));
        ISB_TRY(dw2.alloc(nw2 * 2));
        ISB_TRY(dout.alloc(nout * 2));
        if (res) ISB_TRY(upload(dres, res, nout * 2));
        ISB_TRY(launch_f32_to_bf16_rows(dw1f.as<float>(), ds1.as<float>(), dw1.as<uint16_t>(), Cexp, (size_t)9 * Cin, nullptr, f16));
        ISB_TRY(launch_f32_to_bf16_rows(dw2f.as<float>(), ds2.as<float>(), dw2.as<uint16_t>(), Cout2, (size_t)Cexp, nullptr, f16));
        ConvArgs a{};
        a.f16 = f16;
        a.in = dx.as<uint16_t>(); a.w = dw1.as<uint16_t>(); a.bias = db1.as<float>();
        a.res = res ? dres.as<uint16_t>() : nullptr; a.out = dout.p;
        a.B = B; a.H = H; a.W = H; a.Cin = Cin; a.Cout = Cexp; a.KH = 3; a.KW = 3; a.stride = stride; a.OH = OH; a.OW = OH;
        a.pad = stride == 1 ? 1 : 0; a.M = B * OH * OH; a.K = 9 * Cin; a.act = 1;
        a.w2 = dw2.as<uint16_t>(); a.bias2 = db2.as<float>(); a.Cout2 = Cout2;
        DevBuf dw2p;
        if (!lds_e && Cout2 <= 64 && Cexp <= 256) {
            ISB_TRY(dw2p.alloc(fmb_w2p_bytes(Cout2, Cexp)));
            ISB_TRY(launch_fmb_pack_w2(dw2.as<uint16_t>(), dw2p.p, Cout2, Cexp, nullptr));
            a.w2p = dw2p.p;
        }
        DevBuf dstamps;
        if (stamps) {
            ISB_TRY(dstamps.alloc(64 * 64));
            ISB_HIP(hipMemset(dstamps.p, 0, 64 * 64));
            a.part = dstamps.as<float>();
            a.probe = 2;
        }
        ISB_TRY(launch_fused_mb(a, nullptr));
        ISB_HIP(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        ISB_HIP(hipEventCreate(&e0));
        ISB_HIP(hipEventCreate(&e1));
        ISB_HIP(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; ++i) ISB_TRY(launch_fused_mb(a, nullptr));
        ISB_HIP(hipEventRecord(e1, nullptr));
        ISB_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        ISB_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *ms_per_iter = ms / iters;
        ISB_HIP(hipMemcpy(out, dout.p, nout * 2, hipMemcpyDeviceToHost));
        if (stamps) {
            std::vector<uint64_t> st(64 * 8);
            ISB_HIP(hipMemcpy(st.data(), dstamps.p, 64 * 64, hipMemcpyDeviceToHost));
            double sum[7] = {0, 0, 0, 0, 0, 0, 0};
            int n = 0;
            for (int g = 0; g < 64; ++g) {
                if (!st[(size_t)g * 8 + 7]) continue;
                for (int i = 0; i < 7; ++i) sum[i] += (double)st[(size_t)g * 8 + i];
                ++n;
            }
            if (n)
                fprintf(stderr, "fused_mb, mean of %d workgroups (cycles, wave 0): prologue %.0f | k loop %.0f | E epilogue %.0f | its publish %.0f | "
                                "GEMM 2 %.0f | out epilogue %.0f | workgroup %.0f\n",
                        n, sum[0] / n, sum[1] / n, sum[2] / n, sum[3] / n, sum[4] / n, sum[5] / n, sum[6] / n);
        }
        return ISB_OK;
    });
}

// test / tuning hook: the FRONT half of a stride-1 MBConv block on 16 x 16 or 8 x 8 maps (1x1 expand + BN + SiLU -> depthwise 3x3 + BN +
// SiLU -> D + squeeze-excite pool) on host tensors, three ways that must give the same bits: form 0 = the two launches (expand GEMM, then
// dwconv3x3_mm_kernel), 1 = mbfront16_kernel / mbfront8_kernel, 2 = mbfront16r_kernel / mbfront8r_kernel (conv_mb16.hip, conv_mb8.hip)
// The squeeze-excite FCs of a batch on their own (tests): pooled [B,C] f32, w1 [cse,C], b1 [cse], w2t [cse,C], b2 [C] -> gate [B,C]
// through launch_se_fcs (se_fc1_part_kernel + se_fc2_kernel).
extern "C" int isb_debug_se_fcs(int32_t device, const float* pooled, const float* w1, const float* b1, const float* w2t, const float* b2,
                                int32_t B, int32_t C, int32_t cse, int32_t iters, float* gate, float* ms_per_iter) {
    return isb::guard([&]() -> int {
        ISB_REQUIRE(pooled && w1 && b1 && w2t && b2 && gate && ms_per_iter, ISB_ERR_INVALID, "null argument");
        ISB_REQUIRE(B >= 1 && iters >= 1 && C >= 4 && cse >= 1, ISB_ERR_INVALID, "bad parameters");
        ISB_HIP(hipSetDevice(device));
        DevBuf dp, dw1, db1, dw2, db2, dpart, dgate;
        ISB_TRY(upload(dp, pooled, (size_t)B * C * 4));
        ISB_TRY(upload(dw1, w1, (size_t)cse * C * 4));
        ISB_TRY(upload(db1, b1, (size_t)cse * 4));
        ISB_TRY(upload(dw2, w2t, (size_t)cse * C * 4));
        ISB_TRY(upload(db2, b2, (size_t)C * 4));
        ISB_TRY(dpart.alloc((size_t)B * 160 * 32 * 4));
        ISB_TRY(dgate.alloc((size_t)B * C * 4));
        ISB_HIP(hipMemset(dgate.p, 0xff, (size_t)B * C * 4));
        ISB_HIP(hipMemset(dpart.p, 0xff, (size_t)B * 160 * 32 * 4));
        SeFcArgs se{};
        se.pooled = dp.as<float>(); se.w1 = dw1.as<float>(); se.b1 = db1.as<float>(); se.w2t = dw2.as<float>(); se.b2 = db2.as<float>();
        se.part = dpart.as<float>(); se.gate = dgate.as<float>(); se.B = B; se.C = C; se.cse = cse;
        ISB_TRY(launch_se_fcs(se, nullptr));
        ISB_HIP(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        ISB_HIP(hipEventCreate(&e0));
        ISB_HIP(hipEventCreate(&e1));
        ISB_HIP(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; ++i) ISB_TRY(launch_se_fcs(se, nullptr));
        ISB_HIP(hipEventRecord(e1, nullptr));
        ISB_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        ISB_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *ms_per_iter = ms / iters;
        ISB_HIP(hipMemcpy(gate, dgate.p, (size_t)B * C * 4, hipMemcpyDeviceToHost));
        return ISB_OK;
    });
}

extern "C" int isb_debug_mbfront(int32_t device, int32_t hw, const uint16_t* x, const float* w1, const float* scale1, const float* shift1,
                                 const float* dww, const float* dwscale, const float* dwshift, int32_t B, int32_t cin, int32_t cexp,
                                 int32_t f16, int32_t form, int32_t iters, uint16_t* d_out, float* pooled, float* ms_per_iter) {
    return isb::guard([&]() -> int {
        ISB_REQUIRE(x && w1 && scale1 && shift1 && dww && dwscale && dwshift && d_out && pooled && ms_per_iter, ISB_ERR_INVALID, "null argument");
        const bool stamps = (form & 0x100) != 0;           // tuning probe (form 2, cin 224, fp16): the tick loops' clocks to stderr
        form &= 0xff;
        ISB_REQUIRE(B >= 1 && iters >= 1 && form >= 0 && form <= 2 &&
                        ((hw == 16 && (cin == 192 || cin == 224) && cexp % 32 == 0 && cexp >= 128) || (hw == 8 && cin == 384 && cexp == 2304)),
                    ISB_ERR_INVALID, "bad parameters (16 x 16 maps: cin 192 / 224, cexp a multiple of 32; 8 x 8 maps: 384 -> 2304; form 0..2)");
        const int npx = hw * hw;
        ISB_HIP(hipSetDevice(device));
        const size_t nin = (size_t)B * npx * cin, nout = (size_t)B * npx * cexp;
        std::vector<uint16_t> wt16((size_t)9 * cexp);
        for (int c = 0; c < cexp; ++c)
            for (int t = 0; t < 9; ++t) {
                const float wf = dww[(size_t)c * 9 + t] * dwscale[c];
                if (f16) {
                    const _Float16 hh = (_Float16)wf;
                    memcpy(&wt16[(size_t)t * cexp + c], &hh, 2);
                } else {
                    wt16[(size_t)t * cexp + c] = bf16_rne(wf);
                }
            }
        DevBuf dx, dw1f, ds1, db1, dw1, dw1p, ddw, ddb, dE, dD, dpool, dzero;
        ISB_TRY(upload(dx, x, nin * 2));
        ISB_TRY(upload(dw1f, w1, (size_t)cexp * cin * 4));
        ISB_TRY(upload(ds1, scale1, (size_t)cexp * 4));
        ISB_TRY(upload(db1, shift1, (size_t)cexp * 4));
        ISB_TRY(upload(ddw, wt16.data(), wt16.size() * 2));
        ISB_TRY(upload(ddb, dwshift, (size_t)cexp * 4));
        ISB_TRY(dw1.alloc((size_t)cexp * cin * 2));
        ISB_TRY(dw1p.alloc((size_t)cexp * cin * 2));
        ISB_TRY(dD.alloc(nout * 2));
        ISB_TRY(dpool.alloc((size_t)B * cexp * 4));
        ISB_TRY(dzero.alloc(256));
        ISB_HIP(hipMemset(dzero.p, 0, 256));
        ISB_HIP(hipMemset(dD.p, 0xff, nout * 2));
        ISB_HIP(hipMemset(dpool.p, 0xff, (size_t)B * cexp * 4));
        ISB_TRY(launch_f32_to_bf16_rows(dw1f.as<float>(), ds1.as<float>(), dw1.as<uint16_t>(), cexp, (size_t)cin, nullptr, f16));
        ISB_TRY(launch_mb8_pack_frag(dw1.as<uint16_t>(), dw1p.p, cexp, cin, 1, nullptr));
        if (form == 0) ISB_TRY(dE.alloc(nout * 2));
        DevBuf dstamps;
        if (stamps) {
            ISB_REQUIRE(form == 2, ISB_ERR_INVALID, "stamps: form 2");
            ISB_TRY(dstamps.alloc(32 * 12 * 8 * 8));
            ISB_HIP(hipMemset(dstamps.p, 0, 32 * 12 * 8 * 8));
        }
        auto run = [&]() -> int {
            if (form == 0) {
                ConvArgs a{};
                a.f16 = f16;
                a.in = dx.as<uint16_t>(); a.w = dw1.as<uint16_t>(); a.bias = db1.as<float>(); a.out = dE.p;
                a.B = B; a.H = hw; a.W = hw; a.Cin = cin; a.Cout = cexp; a.KH = 1; a.KW = 1; a.stride = 1; a.OH = hw; a.OW = hw;
                a.pad = 0; a.M = B * npx; a.K = cin; a.act = 1; a.zeros = dzero.as<uint16_t>();
                ISB_TRY(launch_conv_igemm(a, nullptr));
                DwArgs d{};
                d.in = dE.as<uint16_t>(); d.w = ddw.as<uint16_t>(); d.bias = ddb.as<float>(); d.out = dD.as<uint16_t>();
                d.pooled = dpool.as<float>(); d.B = B; d.H = hw; d.W = hw; d.C = cexp; d.OH = hw; d.OW = hw; d.stride = 1; d.pad = 1;
                d.in_f16 = f16; d.out_f16 = f16; d.general = 3;
                return launch_dwconv3x3(d, nullptr);
            }
            if (hw == 8) {
                MbFront8Args a{};
                a.x = dx.as<uint16_t>(); a.w1p = (const uint4*)dw1p.p; a.b1 = db1.as<float>(); a.dww = ddw.as<uint16_t>(); a.dwb = ddb.as<float>();
                a.d = dD.as<uint16_t>(); a.pooled = dpool.as<float>(); a.B = B; a.cin = cin; a.f16 = f16; a.form = form;
                a.stamps = stamps ? dstamps.as<uint64_t>() : nullptr;
                return launch_mbfront8(a, nullptr);
            }
            MbFront16Args a{};
            a.x = dx.as<uint16_t>(); a.w1p = (const uint4*)dw1p.p; a.b1 = db1.as<float>(); a.dww = ddw.as<uint16_t>(); a.dwb = ddb.as<float>();
            a.d = dD.as<uint16_t>(); a.pooled = dpool.as<float>(); a.B = B; a.cin = cin; a.cexp = cexp; a.f16 = f16; a.form = form;
            a.stamps = stamps ? dstamps.as<uint64_t>() : nullptr;
            return launch_mbfront16(a, nullptr);
        };
        ISB_TRY(run());
        ISB_HIP(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        ISB_HIP(hipEventCreate(&e0));
        ISB_HIP(hipEventCreate(&e1));
        ISB_HIP(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; ++i) ISB_TRY(run());
        ISB_HIP(hipEventRecord(e1, nullptr));
        ISB_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        ISB_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *ms_per_iter = ms / iters;
        ISB_HIP(hipMemcpy(d_out, dD.p, nout * 2, hipMemcpyDeviceToHost));
        ISB_HIP(hipMemcpy(pooled, dpool.p, (size_t)B * cexp * 4, hipMemcpyDeviceToHost));
        if (stamps && hw == 8) {
            std::vector<uint64_t> st(32 * 12 * 8);
            ISB_HIP(hipMemcpy(st.data(), dstamps.p, st.size() * 8, hipMemcpyDeviceToHost));
            const char* names[2][5] = {{"MFMAs of row block 0", "its epilogue", "MFMAs of row block 1", "its epilogue", "barrier"},
                                       {"input requests", "fragment reads + MFMAs", "SiLU + D + readback + stores", "pool", "barrier"}};
            for (int role = 0; role < 2; ++role) {
                double sum[6] = {0, 0, 0, 0, 0, 0};
                int n = 0;
                for (int g = 0; g < 32; ++g)
                    for (int w = role ? 4 : 0; w < (role ? 12 : 4); ++w) {
                        const uint64_t* o = &st[((size_t)g * 12 + w) * 8];
                        if (!o[0]) continue;
                        for (int k = 0; k < 6; ++k) sum[k] += (double)o[k];
                        ++n;
                    }
                if (!n) continue;
                fprintf(stderr, "mbfront8r %s waves (%d), cycles per tick:", role ? "consumer" : "producer", n);
                double tot = 0;
                for (int k = 0; k < 5; ++k) { fprintf(stderr, " %s %.0f |", names[role][k], sum[1 + k] / sum[0]); tot += sum[1 + k] / sum[0]; }
                fprintf(stderr, " tick %.0f (%.1f ticks per wave)\n", tot, sum[0] / n);
            }
        } else if (stamps) {
            std::vector<uint64_t> st(32 * 8 * 4);
            ISB_HIP(hipMemcpy(st.data(), dstamps.p, st.size() * 8, hipMemcpyDeviceToHost));
            for (int role = 0; role < 2; ++role) {
                double loop = 0, wait = 0, ticks = 0, real = 0;
                int n = 0;
                for (int g = 0; g < 32; ++g)
                    for (int w = role * 4; w < role * 4 + 4; ++w) {
                        const uint64_t* o = &st[((size_t)g * 8 + w) * 4];
                        if (!o[2]) continue;
                        loop += (double)o[0]; wait += (double)o[1]; ticks += (double)o[2]; real += (double)o[3];
                        ++n;
                    }
                if (n)
                    fprintf(stderr, "mbfront16r %s waves (%d): %.0f ticks, %.0f cycles per tick, of those %.0f at the barrier (%.0f %%); clock %.0f MHz\n",
                            role ? "consumer" : "producer", n, ticks / n, loop / ticks, wait / ticks, 100.0 * wait / loop, loop / real * 100.0);
            }
        }
        return ISB_OK;
    });
}

// test / tuning hook: depthwise 3x3 + SiLU + SE mean on host tensors
// test / tuning hook: one launch_gemm_f32 on host tensors (every A-operand option of GemmF32Args), timed with HIP events
extern "C" int isb_debug_gemm_f32(int32_t device, const float* A, const float* W, const float* bias, const float* a_bias,
                                  const float* a_add, int32_t M, int32_t N, int32_t K, int32_t a_parts, int32_t a_act,
                                  int32_t add_period, int32_t act, int32_t splits, int32_t a_offset, int32_t iters, float* C,
                                  float* ms_per_iter) {
    return isb::guard([&]() -> int {
        ISB_REQUIRE(A && W && C && ms_per_iter, ISB_ERR_INVALID, "null argument");
        ISB_REQUIRE(M >= 1 && N >= 1 && K >= 1 && iters >= 1 && a_parts >= 1 && splits >= 1 && a_offset >= 0 && a_offset < 4 &&
                        (!a_add || add_period >= 1),
                    ISB_ERR_INVALID, "bad gemm parameters");
        ISB_HIP(hipSetDevice(device));
        const size_t na = (size_t)a_parts * M * K;
        DevBuf dA, dW, dB, dAb, dAdd, dC, dParts;
        // a_offset floats of slack in front of A: exercises the less aligned staging paths on an aligned tensor
        ISB_TRY(dA.alloc((na + 4) * 4));
        ISB_HIP(hipMemcpy(dA.as<float>() + a_offset, A, na * 4, hipMemcpyHostToDevice));
        ISB_TRY(upload(dW, W, (size_t)N * K * 4));
        if (bias) ISB_TRY(upload(dB, bias, (size_t)N * 4));
        if (a_bias) ISB_TRY(upload(dAb, a_bias, (size_t)K * 4));
        if (a_add) ISB_TRY(upload(dAdd, a_add, (size_t)add_period * K * 4));
        ISB_TRY(dC.alloc((size_t)M * N * 4));
        GemmF32Args g{};
        g.A = dA.as<float>() + a_offset; g.W = dW.as<float>(); g.bias = bias ? dB.as<float>() : nullptr;
        g.Aadd = a_add ? dAdd.as<float>() : nullptr; g.C = dC.as<float>();
        g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = N; g.ldadd = K; g.add_period = a_add ? add_period : 1;
        g.act = act; g.a_bias = a_bias ? dAb.as<float>() : nullptr; g.a_act = a_act; g.a_parts = a_parts;
        g.a_part_stride = (size_t)M * K;
        if (splits > 1) {
            ISB_TRY(dParts.alloc((size_t)splits * M * N * 4));
            g.splits = splits; g.split_stride = (size_t)M * N; g.C = dParts.as<float>();
        }
        auto run = [&]() -> int {
            ISB_TRY(launch_gemm_f32(g, nullptr));
            if (splits > 1) ISB_TRY(launch_reduce_parts(dParts.as<float>(), splits, (size_t)M * N, g.bias, act, dC.as<float>(), M, N, nullptr));
            return ISB_OK;
        };
        ISB_TRY(run());
        ISB_HIP(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        ISB_HIP(hipEventCreate(&e0));
        ISB_HIP(hipEventCreate(&e1));
        ISB_HIP(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; ++i) ISB_TRY(run());
        ISB_HIP(hipEventRecord(e1, nullptr));
        ISB_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        ISB_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *ms_per_iter = ms / iters;
        ISB_HIP(hipMemcpy(C, dC.p, (size_t)M * N * 4, hipMemcpyDeviceToHost));
        return ISB_OK;
    });
}

// se_w1 [cse, C] (optional): the squeeze-excite FC1 rides in the launch, se_part [dw_slabs][B][cse] receives the slabs' partial sums
static int debug_dwconv_impl(int32_t device, const uint16_t* x, const float* w, const float* scale, const float* shift,
                             int32_t B, int32_t H, int32_t C, int32_t stride, int32_t iters, uint16_t* out, float* pooled,
                             float* ms_per_iter, const float* se_w1, int32_t cse, float* se_part, int32_t* n_parts) {
    return isb::guard([&]() -> int {
        ISB_REQUIRE(x && w && scale && shift && out && pooled && ms_per_iter, ISB_ERR_INVALID, "null argument");
        const int in_f16 = (stride & 0x100) ? 1 : 0, out_f16 = (stride & 0x200) ? 1 : 0;   // fp16 input + taps / fp16 output
        const int general = ((stride & 0x400) ? 1 : 0) | ((stride & 0x800) ? 2 : 0);        // DwArgs.general: 1 = the general kernel, 2 = the LDS-map kernel, 3 = taps on the matrix pipe
        stride &= 0xff;
        ISB_REQUIRE((stride == 1 || stride == 2) && iters >= 1 && B >= 1, ISB_ERR_INVALID, "bad depthwise parameters");
        ISB_HIP(hipSetDevice(device));
        const int OH = H / stride;
        const size_t nin = (size_t)B * H * H * C, nout = (size_t)B * OH * OH * C;
        std::vector<uint16_t> wt16((size_t)9 * C);
        for (int c = 0; c < C; ++c)
            for (int t = 0; t < 9; ++t) {
                const float wf = w[(size_t)c * 9 + t] * scale[c];
                if (in_f16) {
                    const _Float16 hh = (_Float16)wf;
                    memcpy(&wt16[(size_t)t * C + c], &hh, 2);
                } else {
                    wt16[(size_t)t * C + c] = bf16_rne(wf);
                }
            }
        DevBuf dx, dw, db, dout, dpool;
        ISB_TRY(upload(dx, x, nin * 2));
        ISB_TRY(upload(dw, wt16.data(), wt16.size() * 2));
        ISB_TRY(upload(db, shift, (size_t)C * 4));
        ISB_TRY(dout.alloc(nout * 2));
        ISB_TRY(dpool.alloc((size_t)B * C * 4));
        DwArgs d{};
        d.in = dx.as<uint16_t>(); d.w = dw.as<uint16_t>(); d.bias = db.as<float>(); d.out = dout.as<uint16_t>();
        d.pooled = dpool.as<float>(); d.B = B; d.H = H; d.W = H; d.C = C; d.OH = OH; d.OW = OH; d.stride = stride;
        d.pad = stride == 1 ? 1 : 0;
        d.in_f16 = in_f16; d.out_f16 = out_f16; d.general = general;
        DevBuf dw1, dpart;
        if (se_w1) {
            ISB_REQUIRE(se_part && n_parts && cse >= 1 && cse <= 160, ISB_ERR_INVALID, "FC1 fold: se_part, n_parts and 1 <= cse <= 160");
            ISB_TRY(upload(dw1, se_w1, (size_t)cse * C * 4));
            ISB_TRY(dpart.alloc((size_t)dw_slabs(d) * B * cse * 4));
            d.se_w1 = dw1.as<float>(); d.se_part = dpart.as<float>(); d.cse = cse;
        }
        ISB_TRY(launch_dwconv3x3(d, nullptr));
        ISB_HIP(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        ISB_HIP(hipEventCreate(&e0));
        ISB_HIP(hipEventCreate(&e1));
        ISB_HIP(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; ++i) ISB_TRY(launch_dwconv3x3(d, nullptr));
        ISB_HIP(hipEventRecord(e1, nullptr));
        ISB_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        ISB_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *ms_per_iter = ms / iters;
        ISB_HIP(hipMemcpy(out, dout.p, nout * 2, hipMemcpyDeviceToHost));
        ISB_HIP(hipMemcpy(pooled, dpool.p, (size_t)B * C * 4, hipMemcpyDeviceToHost));
        if (se_w1) {
            *n_parts = dw_slabs(d);
            ISB_HIP(hipMemcpy(se_part, dpart.p, (size_t)dw_slabs(d) * B * cse * 4, hipMemcpyDeviceToHost));
        }
        return ISB_OK;
    });
}

extern "C" int isb_debug_dwconv(int32_t device, const uint16_t* x, const float* w, const float* scale, const float* shift,
                                int32_t B, int32_t H, int32_t C, int32_t stride, int32_t iters, uint16_t* out, float* pooled,
                                float* ms_per_iter) {
    return debug_dwconv_impl(device, x, w, scale, shift, B, H, C, stride, iters, out, pooled, ms_per_iter, nullptr, 0, nullptr, nullptr);
}

extern "C" int isb_debug_dwconv_fc1(int32_t device, const uint16_t* x, const float* w, const float* scale, const float* shift,
                                    int32_t B, int32_t H, int32_t C, int32_t stride, int32_t iters, uint16_t* out, float* pooled,
                                    float* ms_per_iter, const float* se_w1, int32_t cse, float* se_part, int32_t* n_parts) {
    ISB_REQUIRE(se_w1, ISB_ERR_INVALID, "null argument");
    return debug_dwconv_impl(device, x, w, scale, shift, B, H, C, stride, iters, out, pooled, ms_per_iter, se_w1, cse, se_part, n_parts);
}

