// C-ABI glue of the YOLOv4 person detector (include/isbfsar.h, isb_det_*): the layer plan of the public architecture,
// weight upload (BatchNorm folded, bf16), per-batch launch sequence. The network is what the reference runs as
// `yolo.engine` (modules/hpe/hpe.py:42,59-60; export contract modules/hpe/setup/1_extract_yolo_onnx.py:21-25,44-60).
#include <algorithm>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "isb_common.h"
#include "kernels.h"

using namespace isb;

namespace {

constexpr int kActLinear = 0, kActMish = 2, kActLeaky = 3;
constexpr int kNBoxes = 4032, kNCls = 80, kNo = 255, kNoPad = 256;
constexpr int kAnchors[18] = {12, 16, 19, 36, 40, 28, 36, 75, 76, 55, 72, 146, 142, 110, 192, 243, 459, 401};
constexpr int kMaxBatch = 256;     // activations of one frame reach 4 MiB (256 x 256 x 32 bf16): 256 frames = 1 GiB, every tensor < 2 GiB

struct DetConv {
    std::string name;
    int cin, cout, k, stride, act;
    bool bn;
    // device weights: bf16 [cout_pad][k*k*cin] with the BN scale folded (f32 [32][27] for the first conv), f32 bias
    DevBuf w, bias;
    int cout_pad = 0;
};

enum OpKind { OP_STEM, OP_CONV, OP_CONCAT, OP_SPP, OP_DECODE };
struct Op {
    OpKind kind;
    int in0 = -1, in1 = -1, out = -1, conv = -1, res = -1, up = 0, scale = 0;
};
struct Tensor {
    int hw = 0, c = 0;
    bool f32 = false;
    // alias >= 0: this tensor is only ever read by ONE concatenation, so its producer writes it straight into channels
    // [c_off, c_off + c) of that concatenation's output (ConvArgs.out_ld) and it owns no buffer
    int alias = -1, c_off = 0;
    DevBuf buf;
};

}  // namespace

struct isb_det {
    isb_det_cfg cfg{};
    hipStream_t own_stream = nullptr;
    bool weights = false;
    std::vector<DetConv> convs;
    std::vector<Op> ops;
    std::vector<Tensor> tens;
    int t_image = -1;
    int ws_B = 0;
    DevBuf zeros;
    // host-buffer entry points: grow-only staging (frames in, boxes / confidences out) so that the per-frame live loop
    // (HumanPoseEstimator calls isb_det_forward_host once per frame) makes no hipMalloc / hipFree -- hipFree synchronises
    DevBuf hs_frames, hs_boxes, hs_confs;
    int hs_B = 0;
};

namespace {

// ---- the plan: the forward() of the public model, module by module ----
struct Builder {
    isb_det* d;
    int tensor(int hw, int c, bool f32 = false) {
        Tensor t;
        t.hw = hw; t.c = c; t.f32 = f32;
        d->tens.push_back(std::move(t));
        return (int)d->tens.size() - 1;
    }
    int conv(const std::string& name, int in, int cout, int k, int stride = 1, int act = kActMish, int res = -1, bool bn = true) {
        const Tensor& ti = d->tens[in];
        DetConv c;
        c.name = name; c.cin = ti.c; c.cout = cout; c.k = k; c.stride = stride; c.act = act; c.bn = bn;
        d->convs.push_back(std::move(c));
        Op o;
        o.kind = d->convs.size() == 1 ? OP_STEM : OP_CONV;
        o.in0 = in; o.conv = (int)d->convs.size() - 1; o.res = res;
        o.out = tensor(ti.hw / stride, bn ? cout : kNoPad, !bn);
        d->ops.push_back(o);
        return o.out;
    }
    int cat(int a, int b, int up_b = 0) {
        Op o;
        o.kind = OP_CONCAT; o.in0 = a; o.in1 = b; o.up = up_b;
        o.out = tensor(d->tens[a].hw, d->tens[a].c + d->tens[b].c);
        d->ops.push_back(o);
        return o.out;
    }
    int spp(int in) {
        Op o;
        o.kind = OP_SPP; o.in0 = in;
        o.out = tensor(d->tens[in].hw, 4 * d->tens[in].c);
        d->ops.push_back(o);
        return o.out;
    }
    void decode(int in, int scale) {
        Op o;
        o.kind = OP_DECODE; o.in0 = in; o.scale = scale;
        d->ops.push_back(o);
    }
    int csp(const std::string& p, int x, int ch, int n) {           // DownSample2..5
        const int x1 = conv(p + ".conv1", x, 2 * ch, 3, 2);
        const int x2 = conv(p + ".conv2", x1, ch, 1);
        int x3 = conv(p + ".conv3", x1, ch, 1);
        for (int i = 0; i < n; ++i) {                               // ResBlock: x = x + conv3x3(conv1x1(x))
            const int h = conv(p + ".resblock.module_list." + std::to_string(i) + ".0", x3, ch, 1);
            x3 = conv(p + ".resblock.module_list." + std::to_string(i) + ".1", h, ch, 3, 1, kActMish, x3);
        }
        const int x4 = conv(p + ".conv4", x3, ch, 1);
        return conv(p + ".conv5", cat(x4, x2), 2 * ch, 1);
    }
    void build() {
        d->t_image = tensor(256, 3, true);
        // DownSample1
        const int x1 = conv("down1.conv1", d->t_image, 32, 3);
        const int x2 = conv("down1.conv2", x1, 64, 3, 2);
        const int x3 = conv("down1.conv3", x2, 64, 1);
        const int x4 = conv("down1.conv4", x2, 64, 1);
        const int x5 = conv("down1.conv5", x4, 32, 1);
        const int x6 = conv("down1.conv6", x5, 64, 3, 1, kActMish, x4);          // + shortcut
        const int x7 = conv("down1.conv7", x6, 64, 1);
        const int d1 = conv("down1.conv8", cat(x7, x3), 64, 1);
        const int d2 = csp("down2", d1, 64, 2);
        const int d3 = csp("down3", d2, 128, 8);
        const int d4 = csp("down4", d3, 256, 8);
        const int d5 = csp("down5", d4, 512, 4);
        // neck: SPP + PANet
        const int lk = kActLeaky;
        int n = conv("neek.conv1", d5, 512, 1, 1, lk);
        n = conv("neek.conv2", n, 1024, 3, 1, lk);
        n = conv("neek.conv3", n, 512, 1, 1, lk);
        n = conv("neek.conv4", spp(n), 512, 1, 1, lk);
        n = conv("neek.conv5", n, 1024, 3, 1, lk);
        const int n6 = conv("neek.conv6", n, 512, 1, 1, lk);
        const int n7 = conv("neek.conv7", n6, 256, 1, 1, lk);
        n = cat(conv("neek.conv8", d4, 256, 1, 1, lk), n7, 1);                   // [lateral, upsampled]
        n = conv("neek.conv9", n, 256, 1, 1, lk);
        n = conv("neek.conv10", n, 512, 3, 1, lk);
        n = conv("neek.conv11", n, 256, 1, 1, lk);
        n = conv("neek.conv12", n, 512, 3, 1, lk);
        const int n13 = conv("neek.conv13", n, 256, 1, 1, lk);
        const int n14 = conv("neek.conv14", n13, 128, 1, 1, lk);
        n = cat(conv("neek.conv15", d3, 128, 1, 1, lk), n14, 1);
        n = conv("neek.conv16", n, 128, 1, 1, lk);
        n = conv("neek.conv17", n, 256, 3, 1, lk);
        n = conv("neek.conv18", n, 128, 1, 1, lk);
        n = conv("neek.conv19", n, 256, 3, 1, lk);
        const int n20 = conv("neek.conv20", n, 128, 1, 1, lk);
        // heads
        int h = conv("head.conv1", n20, 256, 3, 1, lk);
        decode(conv("head.conv2", h, kNo, 1, 1, kActLinear, -1, false), 0);
        h = cat(conv("head.conv3", n20, 256, 3, 2, lk), n13);
        h = conv("head.conv4", h, 256, 1, 1, lk);
        h = conv("head.conv5", h, 512, 3, 1, lk);
        h = conv("head.conv6", h, 256, 1, 1, lk);
        h = conv("head.conv7", h, 512, 3, 1, lk);
        const int h8 = conv("head.conv8", h, 256, 1, 1, lk);
        h = conv("head.conv9", h8, 512, 3, 1, lk);
        decode(conv("head.conv10", h, kNo, 1, 1, kActLinear, -1, false), 1);
        h = cat(conv("head.conv11", h8, 512, 3, 2, lk), n6);
        h = conv("head.conv12", h, 512, 1, 1, lk);
        h = conv("head.conv13", h, 1024, 3, 1, lk);
        h = conv("head.conv14", h, 512, 1, 1, lk);
        h = conv("head.conv15", h, 1024, 3, 1, lk);
        h = conv("head.conv16", h, 512, 1, 1, lk);
        h = conv("head.conv17", h, 1024, 3, 1, lk);
        decode(conv("head.conv18", h, kNo, 1, 1, kActLinear, -1, false), 2);
    }
};

// concatenations in place: an input that a convolution produces and nothing but the concatenation reads (both halves of the five
// CSP joins, the lateral / down-sampling halves of the four PANet joins) is written by that convolution into its slice of the
// joined tensor; what is left for concat_kernel is the halves that have other readers or are up-sampled on the way.
void alias_concat_inputs(isb_det* d) {
    const char* e = getenv("ISB_DET_ALIAS");               // A/B switch, read when the handle is created
    if (e && atoi(e) == 0) return;
    std::vector<int> readers(d->tens.size(), 0), producer(d->tens.size(), -1);
    for (size_t i = 0; i < d->ops.size(); ++i) {
        const Op& o = d->ops[i];
        for (int t : {o.in0, o.in1, o.res})
            if (t >= 0) readers[t] += 1;
        if (o.kind == OP_CONV && o.out >= 0) producer[o.out] = (int)i;
    }
    for (const Op& o : d->ops) {
        if (o.kind != OP_CONCAT) continue;
        const int side[2] = {o.in0, o.in1};
        for (int s = 0; s < 2; ++s) {
            const int t = side[s];
            if ((s == 1 && o.up) || producer[t] < 0 || readers[t] != 1 || d->tens[t].f32 || d->tens[t].c % 8 != 0) continue;
            d->tens[t].alias = o.out;
            d->tens[t].c_off = s == 0 ? 0 : d->tens[o.in0].c;
        }
    }
}

int ensure_ws(isb_det* d, int B) {
    if (B <= d->ws_B) return ISB_OK;
    d->ws_B = 0;
    for (Tensor& t : d->tens)
        if (t.alias < 0) ISB_TRY(t.buf.alloc((size_t)B * t.hw * t.hw * t.c * (t.f32 ? 4 : 2)));
    d->ws_B = B;
    return ISB_OK;
}

int run(isb_det* d, hipStream_t st, const uint8_t* d_frames, int B, float* d_boxes, float* d_confs) {
    ISB_TRY(launch_det_preprocess(d_frames, B, d->cfg.height, d->cfg.width, d->tens[d->t_image].buf.as<float>(), st));
    for (const Op& o : d->ops) {
        switch (o.kind) {
            case OP_STEM: {
                const DetConv& c = d->convs[o.conv];
                StemArgs a{};
                a.in = d->tens[o.in0].buf.as<float>(); a.w = c.w.as<float>(); a.wt = c.w.as<float>() + 32 * 27; a.bias = c.bias.as<float>();
                a.out = d->tens[o.out].buf.as<uint16_t>(); a.B = B; a.H = 256; a.W = 256;
                ISB_TRY(launch_det_stem(a, st));
                break;
            }
            case OP_CONV: {
                const DetConv& c = d->convs[o.conv];
                const Tensor &ti = d->tens[o.in0], &to = d->tens[o.out];
                ConvArgs a{};
                a.in = ti.buf.as<uint16_t>(); a.w = c.w.as<uint16_t>(); a.bias = c.bias.as<float>();
                a.res = o.res >= 0 ? d->tens[o.res].buf.as<uint16_t>() : nullptr;
                a.out = to.buf.p;
                if (to.alias >= 0) {                                           // written into its slice of the concatenation
                    const Tensor& tc = d->tens[to.alias];
                    a.out = tc.buf.as<uint16_t>() + to.c_off;
                    a.out_ld = tc.c;
                }
                a.B = B; a.H = ti.hw; a.W = ti.hw; a.Cin = c.cin; a.Cout = c.cout_pad; a.KH = c.k; a.KW = c.k; a.stride = c.stride;
                a.OH = to.hw; a.OW = to.hw; a.pad = (c.k - 1) / 2;             // PyTorch padding: symmetric, also at stride 2
                a.M = B * to.hw * to.hw; a.K = c.k * c.k * c.cin;
                a.act = c.act; a.out_f32 = to.f32 ? 1 : 0;
                a.zeros = d->zeros.as<uint16_t>();
                ISB_TRY(launch_conv_igemm(a, st));
                break;
            }
            case OP_CONCAT: {
                const Tensor &ta = d->tens[o.in0], &tb = d->tens[o.in1], &to = d->tens[o.out];
                ISB_TRY(launch_concat(ta.alias >= 0 ? nullptr : ta.buf.as<uint16_t>(), tb.alias >= 0 ? nullptr : tb.buf.as<uint16_t>(),
                                      to.buf.as<uint16_t>(), B, to.hw, to.hw, ta.c, tb.c, o.up, st));
                break;
            }
            case OP_SPP: {
                const Tensor &ti = d->tens[o.in0], &to = d->tens[o.out];
                ISB_TRY(launch_spp(ti.buf.as<uint16_t>(), to.buf.as<uint16_t>(), B, ti.hw, ti.hw, ti.c, st));
                break;
            }
            case OP_DECODE: {
                const Tensor& ti = d->tens[o.in0];
                static const int stride[3] = {8, 16, 32};
                static const float sxy[3] = {1.2f, 1.1f, 1.05f};
                static const int off[3] = {0, 3 * 32 * 32, 3 * 32 * 32 + 3 * 16 * 16};
                float wh[6];
                for (int a = 0; a < 3; ++a) {
                    wh[2 * a] = (float)kAnchors[2 * (3 * o.scale + a)] / (float)stride[o.scale];
                    wh[2 * a + 1] = (float)kAnchors[2 * (3 * o.scale + a) + 1] / (float)stride[o.scale];
                }
                ISB_TRY(launch_yolo_decode(ti.buf.as<float>(), B, ti.hw, ti.hw, kNoPad, wh, sxy[o.scale], d_boxes, d_confs, kNBoxes, off[o.scale], st));
                break;
            }
        }
    }
    return ISB_OK;
}

}  // namespace

extern "C" void isb_det_destroy(isb_det* d);

namespace {
int ensure_host_staging(isb_det* d, int B) {
    if (B <= d->hs_B) return ISB_OK;
    const size_t fsz = (size_t)d->cfg.height * d->cfg.width * 3;
    d->hs_B = 0;                                  // failure-atomic: re-allocated on the next call
    ISB_TRY(d->hs_frames.alloc(fsz * B));
    ISB_TRY(d->hs_boxes.alloc((size_t)B * kNBoxes * 16));
    ISB_TRY(d->hs_confs.alloc((size_t)B * kNBoxes * kNCls * 4));
    d->hs_B = B;
    return ISB_OK;
}
}  // namespace

extern "C" int isb_det_create(const isb_det_cfg* cfg, isb_det** out) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(cfg && out, ISB_ERR_INVALID, "isb_det_create: null argument");
    ISB_REQUIRE(cfg->width >= 256 && cfg->height >= 256 && cfg->width <= 8192 && cfg->height <= 8192, ISB_ERR_INVALID,
                "frame size %dx%d unsupported (the detector shrinks frames to 256 x 256)", cfg->width, cfg->height);
    int ndev = 0;
    ISB_HIP(hipGetDeviceCount(&ndev));
    ISB_REQUIRE(cfg->device >= 0 && cfg->device < ndev, ISB_ERR_INVALID, "device %d not in [0,%d)", cfg->device, ndev);
    ISB_HIP(hipSetDevice(cfg->device));
    // a failure after the stream exists must destroy it too: release through isb_det_destroy, not a plain delete
    std::unique_ptr<isb_det, void (*)(isb_det*)> d(new (std::nothrow) isb_det(), isb_det_destroy);
    ISB_REQUIRE(d, ISB_ERR_NOMEM, "out of host memory");
    d->cfg = *cfg;
    if (d->cfg.max_batch <= 0) d->cfg.max_batch = 16;
    d->cfg.max_batch = std::min(d->cfg.max_batch, kMaxBatch);
    ISB_HIP(hipStreamCreateWithFlags(&d->own_stream, hipStreamNonBlocking));
    ISB_TRY(d->zeros.alloc(256));
    ISB_HIP(hipMemset(d->zeros.p, 0, 256));
    Builder b{d.get()};
    b.build();
    alias_concat_inputs(d.get());
    *out = d.release();
    return ISB_OK;
    });
}

extern "C" void isb_det_destroy(isb_det* d) {
    if (!d) return;
    (void)hipSetDevice(d->cfg.device);
    (void)hipDeviceSynchronize();
    if (d->own_stream) (void)hipStreamDestroy(d->own_stream);
    delete d;
}

extern "C" int isb_det_n_convs(void) {
    static const int n = [] {
        isb_det tmp;
        Builder b{&tmp};
        b.build();
        return (int)tmp.convs.size();
    }();
    return n;
}

extern "C" int isb_det_describe(int32_t idx, char* name, int32_t name_cap, int32_t* dims) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(name && dims && name_cap > 0, ISB_ERR_INVALID, "null argument");
    isb_det tmp;
    Builder b{&tmp};
    b.build();
    ISB_REQUIRE(idx >= 0 && idx < (int)tmp.convs.size(), ISB_ERR_INVALID, "layer %d outside [0,%zu)", idx, tmp.convs.size());
    const DetConv& c = tmp.convs[idx];
    snprintf(name, (size_t)name_cap, "%s", c.name.c_str());
    dims[0] = c.cin; dims[1] = c.cout; dims[2] = c.k; dims[3] = c.stride; dims[4] = c.act; dims[5] = c.bn ? 1 : 0;
    return ISB_OK;
    });
}

extern "C" int isb_det_load_weights(isb_det* d, const void* blob, size_t nbytes) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(d, ISB_ERR_INVALID, "null handle");
    ISB_HIP(hipSetDevice(d->cfg.device));
    hipStream_t st = d->own_stream;
    std::map<std::string, BlobTensor> m;
    ISB_TRY(parse_blob(blob, nbytes, m));
    d->weights = false;
    for (size_t i = 0; i < d->convs.size(); ++i) {
        DetConv& c = d->convs[i];
        const std::string p = "yolo." + c.name;
        auto it = m.find(p + ".w");
        ISB_REQUIRE(it != m.end(), ISB_ERR_WEIGHTS, "weight tensor '%s.w' missing", p.c_str());
        const BlobTensor& w = it->second;
        ISB_REQUIRE((int)w.dims[0] == c.cout && (int)w.dims[1] == c.k && (int)w.dims[2] == c.k && (int)w.dims[3] == c.cin, ISB_ERR_WEIGHTS,
                    "'%s.w' has shape [%u,%u,%u,%u], expected [%d,%d,%d,%d]", p.c_str(), w.dims[0], w.dims[1], w.dims[2], w.dims[3], c.cout,
                    c.k, c.k, c.cin);
        const BlobTensor *sc, *sh;
        ISB_TRY(blob_get(m, (p + ".scale").c_str(), c.cout, 1, &sc));
        ISB_TRY(blob_get(m, (p + ".shift").c_str(), c.cout, 1, &sh));
        const size_t kk = (size_t)c.k * c.k * c.cin;
        if (i == 0) {                       // first conv: f32 [32][27], scale folded
            ISB_REQUIRE(c.cin == 3 && c.cout == 32 && c.k == 3, ISB_ERR_WEIGHTS, "internal: first layer is not the 3 -> 32 stem");
            // [32][27] row-major, then the same weights pair-major [16][27][2] (det_stem_kernel: a channel pair's taps are 54 consecutive
            // scalars, one v_pk_fma_f32 per pair and tap)
            std::vector<float> wf(2 * 32 * 27);
            for (int o = 0; o < 32; ++o)
                for (int k = 0; k < 27; ++k) {
                    wf[o * 27 + k] = w.data[o * 27 + k] * sc->data[o];
                    wf[32 * 27 + ((o >> 1) * 27 + k) * 2 + (o & 1)] = wf[o * 27 + k];
                }
            ISB_TRY(upload(c.w, wf.data(), wf.size() * 4));
            ISB_TRY(upload(c.bias, sh->data, 32 * 4));
            c.cout_pad = 32;
            continue;
        }
        c.cout_pad = c.bn ? c.cout : kNoPad;                            // 255 detection channels are padded to 256 zero rows
        ISB_REQUIRE(c.cin % 32 == 0 && c.cout_pad % 32 == 0, ISB_ERR_WEIGHTS, "internal: layer %s has Cin %d / Cout %d", c.name.c_str(), c.cin, c.cout_pad);
        std::vector<float> wf((size_t)c.cout_pad * kk, 0.f), scp(c.cout_pad, 1.f), shp(c.cout_pad, 0.f);
        memcpy(wf.data(), w.data, (size_t)c.cout * kk * 4);
        memcpy(scp.data(), sc->data, (size_t)c.cout * 4);
        memcpy(shp.data(), sh->data, (size_t)c.cout * 4);
        DevBuf tmp, dsc;
        ISB_TRY(upload(tmp, wf.data(), wf.size() * 4));
        ISB_TRY(upload(dsc, scp.data(), scp.size() * 4));
        ISB_TRY(c.w.alloc(wf.size() * 2));
        ISB_TRY(launch_f32_to_bf16_rows(tmp.as<float>(), dsc.as<float>(), c.w.as<uint16_t>(), c.cout_pad, kk, st));
        ISB_HIP(hipStreamSynchronize(st));
        ISB_TRY(upload(c.bias, shp.data(), shp.size() * 4));
    }
    d->weights = true;
    return ISB_OK;
    });
}

extern "C" int isb_det_forward(isb_det* d, const uint8_t* d_frames, int32_t B, float* d_boxes, float* d_confs, void* stream) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(d && d_frames && d_boxes && d_confs, ISB_ERR_INVALID, "null argument");
    ISB_REQUIRE(B >= 1, ISB_ERR_INVALID, "batch %d < 1", B);
    ISB_REQUIRE(d->weights, ISB_ERR_STATE, "isb_det_forward before isb_det_load_weights");
    ISB_HIP(hipSetDevice(d->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    const int Bm = std::min<int>(B, d->cfg.max_batch);
    ISB_TRY(ensure_ws(d, Bm));
    const size_t fsz = (size_t)d->cfg.height * d->cfg.width * 3;
    for (int b0 = 0; b0 < B; b0 += Bm) {
        const int n = std::min(Bm, B - b0);
        ISB_TRY(run(d, st, d_frames + (size_t)b0 * fsz, n, d_boxes + (size_t)b0 * kNBoxes * 4, d_confs + (size_t)b0 * kNBoxes * kNCls));
    }
    return ISB_OK;
    });
}

extern "C" int isb_det_forward_host(isb_det* d, const uint8_t* frames, int32_t B, float* boxes, float* confs) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(d && frames && boxes && confs, ISB_ERR_INVALID, "null argument");
    ISB_REQUIRE(B >= 1, ISB_ERR_INVALID, "batch %d < 1", B);
    ISB_REQUIRE(d->weights, ISB_ERR_STATE, "isb_det_forward_host before isb_det_load_weights");
    ISB_HIP(hipSetDevice(d->cfg.device));
    hipStream_t st = d->own_stream;
    const size_t fsz = (size_t)d->cfg.height * d->cfg.width * 3;
    ISB_TRY(ensure_host_staging(d, B));
    DevBuf &df = d->hs_frames, &db = d->hs_boxes, &dc = d->hs_confs;
    ISB_HIP(hipMemcpyAsync(df.p, frames, fsz * B, hipMemcpyHostToDevice, st));
    ISB_TRY(isb_det_forward(d, df.as<uint8_t>(), B, db.as<float>(), dc.as<float>(), st));
    ISB_HIP(hipMemcpyAsync(boxes, db.p, (size_t)B * kNBoxes * 16, hipMemcpyDeviceToHost, st));
    ISB_HIP(hipMemcpyAsync(confs, dc.p, (size_t)B * kNBoxes * kNCls * 4, hipMemcpyDeviceToHost, st));
    ISB_HIP(hipStreamSynchronize(st));
    return ISB_OK;
    });
}

// stage hooks (tests): the preprocessed image f32 [B,256,256,3] RGB, and the three raw detection maps f32 [B,H,W,256]
extern "C" int isb_det_debug_host(isb_det* d, const uint8_t* frames, int32_t B, float* image, float* map8, float* map16, float* map32) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(d && frames && B >= 1 && B <= d->cfg.max_batch, ISB_ERR_INVALID, "bad argument (B must be <= max_batch)");
    ISB_REQUIRE(d->weights, ISB_ERR_STATE, "isb_det_debug_host before isb_det_load_weights");
    ISB_HIP(hipSetDevice(d->cfg.device));
    hipStream_t st = d->own_stream;
    const size_t fsz = (size_t)d->cfg.height * d->cfg.width * 3;
    ISB_TRY(ensure_host_staging(d, B));
    DevBuf &df = d->hs_frames, &db = d->hs_boxes, &dc = d->hs_confs;
    ISB_HIP(hipMemcpyAsync(df.p, frames, fsz * B, hipMemcpyHostToDevice, st));
    ISB_TRY(isb_det_forward(d, df.as<uint8_t>(), B, db.as<float>(), dc.as<float>(), st));
    ISB_HIP(hipStreamSynchronize(st));
    if (image) ISB_HIP(hipMemcpy(image, d->tens[d->t_image].buf.p, (size_t)B * 256 * 256 * 3 * 4, hipMemcpyDeviceToHost));
    float* outs[3] = {map8, map16, map32};
    for (const Op& o : d->ops)
        if (o.kind == OP_DECODE && outs[o.scale]) {
            const Tensor& t = d->tens[o.in0];
            ISB_HIP(hipMemcpy(outs[o.scale], t.buf.p, (size_t)B * t.hw * t.hw * kNoPad * 4, hipMemcpyDeviceToHost));
        }
    return ISB_OK;
    });
}
