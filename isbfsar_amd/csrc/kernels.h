// Kernel argument blocks + host launchers shared between the .hip translation units and the
// C-ABI glue (ar_api.cpp / hpe_api.cpp).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace isb {

// ---------------------------------------------------------------- gemm_f32.hip
enum { GEMM_ACT_NONE = 0, GEMM_ACT_RELU = 1, GEMM_ACT_SILU = 2, GEMM_ACT_SIGMOID = 3 };
struct GemmF32Args {
    const float* A;      // [M,K], row stride lda
    const float* W;      // [N,K], row stride ldw (torch Linear weight)
    const float* bias;   // [N] or null
    const float* Aadd;   // optional additive table [add_period, K] (positional encoding), stride ldadd
    float* C;            // [M,N], row stride ldc
    int M, N, K;
    int lda, ldw, ldc, ldadd;
    int add_period;
    int act;
    // split-K: grid.z = splits, split z covers a contiguous range of k-tiles and writes its raw
    // partial sums (no bias / act) to C + z * split_stride; the consumer adds the parts in order
    int splits;            // 0/1 = off
    size_t split_stride;
    // A-operand transform on load: v = a_act( sum_{s < a_parts} A[s * a_part_stride + m*lda + k] + a_bias[k] )
    const float* a_bias;   // [K] or null
    int a_act;             // GEMM_ACT_*
    int a_parts;           // 0/1 = plain A
    size_t a_part_stride;
};
int launch_gemm_f32(const GemmF32Args& a, hipStream_t st);
// out[m,n] = act(bias[n] + sum_s parts[s*stride + m*N + n]) -- fixed-order reduction of split-K partials
int launch_reduce_parts(const float* parts, int n_parts, size_t stride, const float* bias, int act, float* out, int M, int N, hipStream_t st);

// ---------------------------------------------------------------- ar_kernels.hip
// Fragment-ordered bf16 operand images (see ar_kernels.hip header):
//   KF  [n_items][NT][8 ks][64 lanes][8]      K of every tuple, A/B operand of S^T = Kc * Kq^T
//   VtF [n_items][NT][4 dt][2 s][64 lanes][8] V^T of every support tuple, A operand of P^T
struct ArTupleArgs {
    const float* proj;      // [n_items*L, 512] = [Ak | Bk | Av | Bv] per frame (bias-free)
    const float* bk;        // [128] k_linear.bias
    const float* bv;        // [128] v_linear.bias
    const float* gamma;     // [128] norm_k.weight
    const float* beta;      // [128] norm_k.bias
    const int16_t* tup;     // [Tp][2] frame indices of tuple t (padded rows = -1)
    uint16_t* KF;           // out, hi part
    uint16_t* KF_lo;        // out, lo part (bf16x3) or null
    uint16_t* VtF;          // out (support only) or null
    uint16_t* VtF_lo;       // out or null
    uint16_t* KF16;         // out or null: the same K image in IEEE fp16 (ISB_AR_PREC_F16: operands of the all-classes pass)
    uint16_t* VtF16;        // out or null: the V^T image in fp16
    float* VqF;             // out (query only) or null: f32 V of every tuple in ar_proto's epilogue order,
                            // [item][it][piece = 4 dt + q][lane = 32 h + r][4] = V[32 it + r][32 dt + 8 q + 4 h ..+4]
    float kscale;           // folded into K before rounding (query: log2(e)/sqrt(128); support: 1)
    int n_items, L, T, NT;
};
int launch_ar_tuples(const ArTupleArgs& a, hipStream_t st);

struct ArStatsArgs {
    const uint16_t* KqF;    // [B][NT][8][64][8]
    const uint16_t* KqF_lo;
    const uint16_t* KcF;    // [n][NT][8][64][8]
    const uint16_t* KcF_lo;
    float* lse2;            // out [B][n][Tp]: -log2 sum_i exp2(s'[i,j])  (negated: the C operand of ar_proto's S^T chain)
    int B, n, T, NT;
    int x3;
    int f16;                // KqF / KcF hold fp16 fragments (ArTupleArgs.KF16): v_mfma_f32_32x32x16_f16; not with x3 / chosen
    int online;             // 1: running-max variant (bound too loose to exclude underflow)
    int wt;                 // grid decode: windows per L2 block (set by the launcher, <= STATS_WT)
    const int32_t* chosen;  // null: every class -> lse2 [B][n][Tp]; else ONLY class chosen[b] of window b -> lse2 [B][Tp]
};
int launch_ar_stats(const ArStatsArgs& a, hipStream_t st);

struct ArProtoArgs {
    const uint16_t* KqF;
    const uint16_t* KqF_lo;
    const uint16_t* KcF;
    const uint16_t* KcF_lo;
    const uint16_t* VtF;
    const uint16_t* VtF_lo;
    const float* lse2;      // [B][n][Tp], negated (ArStatsArgs.lse2)
    const float* proj;      // [B*L,512] query projections (Av at +256, Bv at +384)
    const float* VqF;       // query V in epilogue order (ArTupleArgs.VqF)
    const float* bv;        // [128]
    const int16_t* tup;     // [Tp][2]
    const int32_t* chosen;  // null: all classes -> part; else only class chosen[b] -> diff
    float* part;            // out [B][n][NT] partial sum of squares
    float* diff;            // out [B][T][128] (chosen mode)
    int B, n, L, T, NT;
    int x3;
    int f16;                // KqF / KcF / VtF hold fp16 fragments and A^T is formed in fp16; not with x3 / chosen
    int wt;                 // grid decode: window groups per L2 block (set by the launcher, <= PROTO_WT)
    int lse_per_window;     // chosen mode: lse2 is [B][Tp] (ArStatsArgs.chosen), not [B][n][Tp]
    uint64_t* stamps;       // tuning probe (all-classes pass): per wave of the first 64 workgroups [prologue, tile loop, epilogue] cycles + start
};
int launch_ar_proto(const ArProtoArgs& a, hipStream_t st);

struct ArFinalArgs {
    const float* part;      // [B][n][NT]
    float* logits;          // [B][n] (row stride n)
    int32_t* chosen;        // [B]
    int B, n, T, NT;
};
int launch_ar_finalize(const ArFinalArgs& a, hipStream_t st);

struct ArDiscTailArgs {
    const float* h1;        // [n_parts][B][256] split-K partial sums of fc1 (bias/ReLU applied here)
    const float* b1;        // [256] fc1 bias
    int n_parts;
    size_t part_stride;
    const float* w2;        // [64][256]
    const float* b2;        // [64]
    const float* w3;        // [64]
    const float* b3;        // [1]
    float* is_true;         // [B]
    int B;
};
int launch_ar_disc_tail(const ArDiscTailArgs& a, hipStream_t st);

// ---------------------------------------------------------------- conv_dispatch.hip + conv_*.hip (the convolution family)
struct ConvArgs {
    const uint16_t* in;     // bf16 [B,H,W,Cin]
    const uint16_t* w;      // bf16 [Cout, KH*KW*Cin] (BN scale folded)
    const float* bias;      // [Cout] folded-BN shift
    const uint16_t* res;    // bf16 [M,Cout] residual or null
    const float* gate;      // f32 [B,Cin] squeeze-excite gate applied to the A operand, or null
    void* out;              // bf16 (or f32 when out_f32) [M,Cout]
    int B, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad;
    int M, K;
    int act;                // 0 none, 1 SiLU, 2 Mish, 3 LeakyReLU(0.1), 4 ReLU
    int act_after_res;      // 1: out = act(conv + bias + res) (a ResNet bottleneck's tail) instead of act(conv + bias) + res; shared epilogue only
    int out_f32;
    // row stride of `out` in elements, 0 = Cout: the 16-bit output lands in a channel slice of a wider tensor (the detector's
    // concatenations are written in place). Shared / wave-local epilogues only (launch_conv_igemm refuses it elsewhere).
    int out_ld;
    // in / w / res / out (unless out_f32) hold IEEE fp16 instead of bf16: the two 8x8 stages and the 640 -> 1280 convolution of
    // the pose backbone under isb_hpe_cfg.precision 0. Implemented by the gemm1x1 variants the 8x8 stages
    // select (131, 132, 138; gated 141, 144, 146, 147, 149; weights-stationary 185 / 186) and the split-K reduction.
    int f16;
    int variant;            // tile variant, 0 = chosen per layer shape (conv_dispatch.hip)
    const uint16_t* zeros;  // >= 16 bytes of zeros (source of padding taps for the LDS-DMA kernels)
    // workgroup -> tile mapping (set by launch_conv_igemm): 0 = (blockIdx.x, blockIdx.y) = (M tile, N tile);
    // 1/2 = 1-D grid decoded per XCD (workgroup id % 8 = XCD): all N tiles of an M tile run back to back on
    // ONE XCD, so its L2 fetches the A rows once (1: M tiles interleaved over XCDs, 2: contiguous M ranges)
    int grid_mode, grid_m, grid_n;
    int grid_bias_off;      // gated gemm1x1 kernels: LDS byte offset of the tile's bias row (set by the launcher)
    // split-K (gemm1x1 kernels on single-frame launches: a few rows x a long K would run on a handful of CUs):
    // blockIdx.z = split; each split writes its f32 partial tile to part[split][M][Cout]; launch_splitk_reduce
    // adds the partials in split order + bias + residual and rounds once
    int splits;
    float* part;
    // split-K gated projection of ONE frame with the second squeeze-excite FC folded in (variant 149): each workgroup
    // computes the gate of its own k-range (at most 256 channels) from the FC1 partials, exactly as se_fc2_kernel does
    const float* se_part;   // [se_nparts][B][se_cse] FC1 partial sums (SeFcArgs.part / DwArgs.se_part)
    const float* se_b1;     // [se_cse]
    const float* se_w2t;    // [se_cse][Cin]
    const float* se_b2;     // [Cin]
    int se_nparts, se_cse;
    // fused Fused-MBConv block (launch_fused_mb): 3x3 expand (w, bias, act) -> bf16 -> 1x1 project (w2, bias2) + residual
    // tuning probes of gemm1x1_wsreg_kernel (isb_debug_conv variant 9181 / ISB_WS_PROBE): bit 0 = no output stores,
    // bit 1 = s_memtime stamps of the first workgroups' wave 0 into `part` (1 KiB per workgroup)
    int probe;
    int exp;                // open experiments (isb::exp_flags() = ISB_EXP, set by the launchers); no bit is in use (EXPERIMENTS.md round 5: bit 8 was
                            // s_setprio around the k loops)
    const uint16_t* w2;     // bf16 [Cout2][Cout] (BN scale folded)
    const void* w2p;        // the same weights in the register-E kernel's fragment order (launch_fmb_pack_w2; fmb_w2p_bytes) or null:
                            // projections to <= 64 channels then run from the accumulators (fused_mb_kernel<.., REGE>)
    const float* bias2;     // [Cout2]
    int Cout2;
};
int launch_conv_igemm(const ConvArgs& a, hipStream_t st);
// conv_ws.hip: the weights-stationary expand GEMMs (tile variants 181 - 188); aa = a with the grid fields the launcher fills
int launch_conv_ws(const ConvArgs& a, ConvArgs& aa, int v, hipStream_t st);
int launch_fused_mb(const ConvArgs& a, hipStream_t st);
// projection weights [Cout2][Cexp] (16-bit) -> [tile c][Cexp / 32 tiles j][2 k16 steps s][64 lanes][8]: lane (m = lane & 31, h = lane >> 5) holds
// W[32 c + m][32 j + 16 s + 8 (t / 4) + 4 h + t % 4], t = 0..7 (rows past Cout2 are zero)
size_t fmb_w2p_bytes(int Cout2, int Cexp);
int launch_fmb_pack_w2(const uint16_t* w2, void* dst, int Cout2, int Cexp, hipStream_t st);
int launch_splitk_reduce(const ConvArgs& a, hipStream_t st);

// ---------------------------------------------------------------- conv_mb8.hip: the stride-1 MBConv blocks of the 8 x 8 stages as one launch
struct Mb8Block {           // one block of the chain (device-resident array); every pointer device memory
    const uint4* w1p;       // expand weights [Cexp][Cin] 16-bit (BN scale folded), fragment-packed (launch_mb8_pack_frag, G = 1)
    const float* b1;        // [Cexp] folded-BN shift of the expand convolution
    const uint16_t* dww;    // depthwise taps [9][Cexp] 16-bit, tap-major (BlockW.dw_w16)
    const float* dwb;       // [Cexp]
    const float* se_w1p;    // squeeze-excite FC1 weights, packed (launch_mb8_pack_se1)
    const float* se_b1;     // [cse]
    const float* se_w2t;    // [cse][Cexp] (se.w2 transposed, as SeFcArgs.w2t)
    const float* se_b2;     // [Cexp]
    const uint4* w2p;       // projection weights [Cout][Cexp] 16-bit, fragment-packed (G = mb8_proj_group(Cout))
    const float* b2;        // [Cout]
    int cin, cout;          // 384 -> 384, 384 -> 640 or 640 -> 640 (Cexp = 6 Cin, cse = Cin / 4)
    int residual;           // out += in (Cin == Cout)
    int pad_;
};
struct Mb8Args {
    const uint16_t* x;      // [B][64][cin0] 16-bit: input of the first block (NHWC, 8 x 8 map)
    uint16_t* out;          // [B][64][cout of the last block]; store_all: block i's output at out + i * out_block_stride
    void* dscratch;         // per sample dscratch_stride bytes (>= 64 x 3840 x 2): the depthwise outputs as projection A tiles
    size_t dscratch_stride;
    const Mb8Block* blocks;
    int nblocks, B, cin0, f16;
    int store_all;
    size_t out_block_stride;
    uint64_t* stamps;       // tuning probe or null: [32 workgroups][2 blocks][32] s_memtime marks of the phases (tools/exp_mb8.py)
};
int launch_mb8_chain(const Mb8Args& a, hipStream_t st);
// the front half of a stride-1 MBConv block on 8 x 8 maps in one launch: expand 1x1 + SiLU -> depthwise 3x3 + SiLU -> D, pooled
struct MbFront8Args {
    const uint16_t* x;      // [B][64][cin] 16-bit block input (NHWC)
    const uint4* w1p;       // expand weights, fragment-packed (launch_mb8_pack_frag, G = 1)
    const float* b1;        // [cexp]
    const uint16_t* dww;    // depthwise taps [9][cexp] 16-bit
    const float* dwb;       // [cexp]
    uint16_t* d;            // out [B][64][cexp] 16-bit: the depthwise output the gated projection reads
    float* pooled;          // out [B][cexp] f32 spatial means
    int B, cin, f16;
    int form;               // 0 = the library's choice; 1 = mbfront8_kernel; 2 = mbfront8r_kernel (producer / consumer waves, round 6). Same bits.
    int exp;                // open experiments (isb::exp_flags(), set by the launcher)
    uint64_t* stamps;       // tuning probe or null: [32 workgroups][4 waves][16] = loop cycles, waiting at the loop top, bodies, iterations,
                            // then the bodies' phases: expand MFMAs, second barrier, E epilogue, depthwise + stores + pool, and the last
                            // one's parts: its MFMAs, SiLU + pack + sums, D through LDS + stores, pooled means
};
int launch_mbfront8(const MbFront8Args& a, hipStream_t st);
int mbf8_verified();        // wsreg_guard.cpp: 1 only if the build confirmed mbfront8_kernel's counted wait in the disassembly
// conv_mb16.hip: the same front half on 16 x 16 maps (192 / 224 inputs; expanded channels in slices of 192), in bands of two image rows
struct MbFront16Args {
    const uint16_t* x;      // [B][256][cin] 16-bit block input (NHWC)
    const uint4* w1p;       // expand weights, fragment-packed (launch_mb8_pack_frag, G = 1)
    const float* b1;        // [cexp]
    const uint16_t* dww;    // depthwise taps [9][cexp] 16-bit
    const float* dwb;       // [cexp]
    uint16_t* d;            // out [B][256][cexp] 16-bit
    float* pooled;          // out [B][cexp] f32 spatial means
    int B, cin, cexp, f16;
    int form;               // 0 = the library's choice; 1 = mbfront16_kernel (every wave does everything, three waves per SIMD); 2 = mbfront16r_kernel
                            // (producer / consumer waves, four per SIMD; round 6). Same bits either way.
    int exp;                // open experiments (isb::exp_flags(), set by the launcher): bit 16 = the stamps are a census of every workgroup's start / end
    uint64_t* stamps;       // tuning probe or null: [32 workgroups][6 waves][10] = loop cycles, band steps, then per-phase sums: tile wait + barrier,
                            // expand MFMAs, second barrier, E epilogue, depthwise MFMAs, SiLU + D rows, pooled means
};
int launch_mbfront16(const MbFront16Args& a, hipStream_t st);
int launch_mb8_pack_frag(const uint16_t* w, void* dst, int N, int K, int G, hipStream_t st);
int launch_mb8_pack_se1(const float* w1, float* dst, int cse, int C, hipStream_t st);
int mb8_proj_group(int cout);

struct DwArgs {
    const uint16_t* in;     // bf16 [B,H,W,C]
    const uint16_t* w;      // bf16 [9][C] tap-major, BN scale folded (rounded once at load, like every conv weight)
    const float* bias;      // [C]
    uint16_t* out;          // bf16 [B,OH,OW,C]
    float* pooled;          // f32 [B,C] spatial mean of `out` (squeeze-excite input) or null
    int B, H, W, C, OH, OW, stride, pad;
    // single-frame calls: the workgroup that owns a channel slab of a sample also owns its pooled means, so it writes
    // the slab's share of the first squeeze-excite FC: se_part[slab][b][j] = sum_{c in slab} pooled[b][c] * se_w1[j][c]
    const float* se_w1;     // [cse,C] or null
    float* se_part;         // [dw_slabs(a)][B][cse]
    int cse;
    int general;            // kernel choice on stride-1 8 x 8 / 16 x 16 maps: 0 / 2 = the LDS-map kernel with v_dot2 taps, 1 = the general kernel
                            // (tests compare the two bit for bit), 3 = taps on the matrix pipe (dwconv3x3_mm_kernel: the arithmetic of the
                            // fused 8 x 8 front, within f32 rounding of the others)
    int in_f16, out_f16;    // `in` + `w` / `out` hold fp16 instead of bf16 (ConvArgs.f16); in bf16 -> out fp16 is the block that
                            // enters the fp16 stages
};
int launch_dwconv3x3(const DwArgs& a, hipStream_t st);
int dw_slabs(const DwArgs& a);      // channel slabs (= grid.x) of the launch

struct SeFcArgs {
    const float* pooled;    // [B,C]
    const float* w1;        // [cse,C]
    const float* b1;        // [cse]
    const float* w2t;       // [cse,C]  (se.w2 transposed)
    const float* b2;        // [C]
    float* part;            // scratch [ceil(C/256), B, cse]: fc1 partial sums per 256-channel chunk
    float* gate;            // out [B,C]
    int B, C, cse;
    int nparts;             // 0: run fc1 over 256-channel chunks; > 0: part already holds this many slabs (DwArgs.se_part)
};
constexpr int SE_MAX_PARTS = 32;
int launch_se_fcs(const SeFcArgs& a, hipStream_t st);

struct StemArgs {
    const float* in;        // f32 [B,H,W,3]
    const float* w;         // f32 [32][27], BN scale folded (the detector's stem)
    const float* wt;        // f32 [16][27][2]: the same weights, wt[co / 2][k][co & 1] (the pose backbone's stem: a channel pair's taps are 54 consecutive scalars)
    const float* bias;      // [32]
    uint16_t* out;          // bf16 (fp16 when out_f16) [B,H/2,W/2,32]
    int B, H, W;
    int out_f16;
};
int launch_stem(const StemArgs& a, hipStream_t st);
// out[r][c] = T(in[r][c] * row_scale[r]), T = bf16 or (f16 != 0) fp16, round to nearest even
int launch_f32_to_bf16_rows(const float* in, const float* row_scale, uint16_t* out, size_t rows, size_t cols, hipStream_t st, int f16 = 0);

// ---------------------------------------------------------------- hpe_kernels.hip
struct CropParamArgs {
    const int32_t* bbox;    // [B,4] x1,x2,y1,y2
    double K[9];            // camera intrinsics (float32 values, hpe.py:28-33)
    float* H;               // out [B,9]  f32 homography, hpe.py:96-97
    double* newK;           // out [B,9]
    double* R;              // out [B,9]
    int B;                  // outputs = boxes x max(n_aug, 1); output i uses box i / n_aug and augmentation i % n_aug
    // test-time augmentation (hpe.py:88-93): new_K[k][:2,:2] *= scale[k]; homo_inv[k] = rotflip[k] @ homo_inv
    int n_aug;              // 0 = off
    const double* aug_rotflip;   // [n_aug,9]
    const double* aug_scale;     // [n_aug]
};
int launch_crop_params(const CropParamArgs& a, hipStream_t st);

// ROI-only frames (isb_hpe_forward_host): only the source rectangle the crop's homography can reach was copied to the
// device. Frame b's rectangle [x0, x0 + w) x [y0, y0 + h) sits packed (row pitch w * 3 bytes) at byte offset `off` of
// WarpArgs.frames. Pixels inside the frame but outside the rectangle cannot be requested by construction (the rectangle
// bounds the image of the crop square + 2 px); the kernel still treats them as out of range, never as an address.
struct RoiDesc {
    int32_t x0, y0, w, h;
    uint64_t off;
};

struct WarpArgs {
    const uint8_t* frames;  // [B,FH,FW,3]
    const float* H;         // [B,9]
    float* crops;           // out [B,256,256,3] f32 in [0,1]
    int B, FH, FW;
    int n_aug;              // > 1: crop i is cut from frame i / n_aug (test-time augmentation)
    const RoiDesc* roi;     // [B] or null: `frames` is the packed ROI image instead of whole frames
};
int launch_warp(const WarpArgs& a, hipStream_t st);
// ROI gather: frames in device-mapped (pinned) HOST memory -> the packed ROI image in HBM. One launch for the whole batch; every
// rectangle is 16-pixel aligned in x (48-byte = three 16-byte pieces), so the reads over PCIe are aligned 16-byte lane loads of
// contiguous row segments. frames_mapped: device address of the host frames [B,FH,FW,3]; roi [B] (device); dst: packed image.
int launch_roi_gather(const uint8_t* frames_mapped, const RoiDesc* roi, uint8_t* dst, int B, int FH, int FW, hipStream_t st);

struct PostArgs {
    const float* logits;    // [B,8,8,288] pose-head output
    const double* newK;     // [B,9]
    const double* R;        // [B,9]
    const float* expand;    // [32,122] joint expansion (assets/32_to_122.npy)
    const int32_t* indices; // [n_out] selected joints or null (-> all 122)
    float* joints;          // out [B,n_out,3]
    uint8_t* valid;         // out [B]
    double* dbg;            // optional [B,32,5] pred2d(2) pred3d(3) or null
    const int32_t* bbox;    // optional [B,4]: x1 < 0 marks 'no person' -> valid = 0
    int B, n_out;
};
int launch_hpe_post(const PostArgs& a, hipStream_t st);
int launch_select_person(const float* boxes, const float* confs, int B, int n_anchor, int n_cls, float thresh, int width, int height,
                         int32_t* bbox, uint8_t* found, hipStream_t st);
int launch_pose_windows(const float* joints, int n_cam, int n_frames, int J, int L, float* windows, hipStream_t st);
int launch_pose_distance(const float* joints, int n, int J, float* distance, hipStream_t st);

// ---------------------------------------------------------------- rgb_kernels.hip (ResNet-50 trunk of the hybrid AR branch)
struct RgbStemArgs {
    const float* in;        // f32 images: [N,3,H,W] (nchw = 1, what main.py:91 hands over) or [N,H,W,3]
    const float* w;         // f32 [64][7][7][3], BN scale folded
    const float* bias;      // [64] folded-BN shift
    uint16_t* out;          // bf16 [N,H/2,W/2,64], ReLU applied
    int N, H, W, nchw;
};
int launch_rgb_stem(const RgbStemArgs& a, hipStream_t st);
int launch_maxpool3x3s2(const uint16_t* in, uint16_t* out, int N, int H, int W, int C, hipStream_t st);   // padding 1
int launch_avgpool(const uint16_t* in, float* out, int N, int HW, int C, hipStream_t st);                 // -> f32 [N,C]

// ---------------------------------------------------------------- det_kernels.hip (YOLOv4 person detector)
int launch_det_preprocess(const uint8_t* frames, int B, int FH, int FW, float* out, hipStream_t st);
int launch_det_stem(const StemArgs& a, hipStream_t st);      // conv 3x3 s1 p1, 3 -> 32, Mish (StemArgs.H/W = map size)
int launch_concat(const uint16_t* a, const uint16_t* b, uint16_t* out, int B, int H, int W, int Ca, int Cb, int up_b, hipStream_t st);
int launch_spp(const uint16_t* in, uint16_t* out, int B, int H, int W, int C, hipStream_t st);
int launch_yolo_decode(const float* map, int B, int H, int W, int ldm, const float* anchors_wh, float sxy, float* boxes, float* confs,
                       int n_boxes, int box_off, hipStream_t st);

}  // namespace isb
