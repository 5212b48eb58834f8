// C-ABI glue of the ResNet-50 trunk (include/isbfsar.h, isb_rgb_*): what the reference builds as
// nn.Sequential(*list(resnet50(pretrained=True).children())[:-1]) for its RGB / hybrid input types
// (modules/ar/utils/model.py:270-277) and runs per frame on 224 x 224 person crops (main.py:85-92): images -> [N, 2048] trunk
// features, the input of PostResNet (model.py:207-216, inside isb_ar_infer_hybrid). Public torchvision architecture
// (Bottleneck [3,4,6,3], stride on the 3x3 convolution), BatchNorm folded at load time, bf16 storage / f32 accumulate on the
// conv_igemm kernel family with a ReLU epilogue; the block table mirrors isbfsar_amd/resnet50.py.
#include <algorithm>
#include <memory>
#include <string>
#include <vector>

#include "isb_common.h"
#include "kernels.h"

using namespace isb;

namespace {

constexpr int kActReLU = 4;
constexpr int kImg = 224;
constexpr int kMaxBatch = 512;           // the largest activation is 1.6 MB per image (112 x 112 x 64 bf16): < 2 GiB per tensor

struct RConv {
    DevBuf w, bias;
    int cin = 0, cout = 0, k = 0;
};
struct RBlock {
    int cin, planes, stride, in_hw, out_hw;
    bool down;
    RConv c1, c2, c3, cd;
};
const int kLayers[4][3] = {{64, 3, 1}, {128, 4, 2}, {256, 6, 2}, {512, 3, 2}};     // planes, blocks, stride of the first block

}  // namespace

struct isb_rgb {
    isb_rgb_cfg cfg{};
    hipStream_t own_stream = nullptr;
    bool weights = false;
    DevBuf stem_w, stem_b, zeros;
    std::vector<RBlock> blocks;
    int ws_N = 0;
    DevBuf bufS, bufX, bufY, bufT1, bufT2, bufD;       // stem out, block in / out, the two intermediates, the down-sampled skip
    DevBuf hs_img, hs_out;                             // host entry point staging (grow-only)
    int hs_N = 0;
};

namespace {

int upload_conv(const std::map<std::string, BlobTensor>& m, const std::string& prefix, int cout, int k, int cin, RConv& c, hipStream_t st) {
    auto it = m.find(prefix + ".w");
    ISB_REQUIRE(it != m.end(), ISB_ERR_WEIGHTS, "weight tensor '%s.w' missing", prefix.c_str());
    const BlobTensor& w = it->second;
    ISB_REQUIRE((int)w.dims[0] == cout && (int)w.dims[1] == k && (int)w.dims[2] == k && (int)w.dims[3] == cin, ISB_ERR_WEIGHTS,
                "'%s.w' has shape [%u,%u,%u,%u], expected [%d,%d,%d,%d]", prefix.c_str(), w.dims[0], w.dims[1], w.dims[2], w.dims[3], cout, k, k, cin);
    const BlobTensor *sc, *sh;
    ISB_TRY(blob_get(m, (prefix + ".scale").c_str(), cout, 1, &sc));
    ISB_TRY(blob_get(m, (prefix + ".shift").c_str(), cout, 1, &sh));
    DevBuf tmp, dsc;
    ISB_TRY(upload(tmp, w.data, w.numel() * 4));
    ISB_TRY(upload(dsc, sc->data, (size_t)cout * 4));
    ISB_TRY(c.w.alloc(w.numel() * 2));
    ISB_TRY(launch_f32_to_bf16_rows(tmp.as<float>(), dsc.as<float>(), c.w.as<uint16_t>(), cout, (size_t)k * k * cin, st));
    ISB_HIP(hipStreamSynchronize(st));
    ISB_TRY(upload(c.bias, sh->data, (size_t)cout * 4));
    c.cin = cin; c.cout = cout; c.k = k;
    return ISB_OK;
}

int ensure_ws(isb_rgb* r, int N) {
    if (N <= r->ws_N) return ISB_OK;
    r->ws_N = 0;
    const size_t act = (size_t)N * 112 * 112 * 64 * 2;            // = 56 x 56 x 256: the largest tensors of the network
    for (DevBuf* b : {&r->bufS, &r->bufX, &r->bufY, &r->bufT1, &r->bufT2, &r->bufD}) ISB_TRY(b->alloc(act));
    r->ws_N = N;
    return ISB_OK;
}

int conv(isb_rgb* r, hipStream_t st, const RConv& c, const void* in, int N, int hw, int stride, const void* res, bool act_after_res, void* out) {
    ConvArgs a{};
    a.in = (const uint16_t*)in; a.w = c.w.as<uint16_t>(); a.bias = c.bias.as<float>(); a.res = (const uint16_t*)res; a.out = out;
    a.B = N; a.H = hw; a.W = hw; a.Cin = c.cin; a.Cout = c.cout; a.KH = c.k; a.KW = c.k; a.stride = stride;
    a.OH = hw / stride; a.OW = hw / stride; a.pad = (c.k - 1) / 2;                 // PyTorch padding: symmetric, also at stride 2
    a.M = N * a.OH * a.OW; a.K = c.k * c.k * c.cin;
    a.act = kActReLU; a.act_after_res = act_after_res ? 1 : 0;
    a.zeros = r->zeros.as<uint16_t>();
    return launch_conv_igemm(a, st);
}

int run(isb_rgb* r, hipStream_t st, const float* d_images, int N, int nchw, float* d_trunk) {
    RgbStemArgs sa{};
    sa.in = d_images; sa.w = r->stem_w.as<float>(); sa.bias = r->stem_b.as<float>(); sa.out = r->bufS.as<uint16_t>();
    sa.N = N; sa.H = kImg; sa.W = kImg; sa.nchw = nchw;
    ISB_TRY(launch_rgb_stem(sa, st));
    ISB_TRY(launch_maxpool3x3s2(r->bufS.as<uint16_t>(), r->bufX.as<uint16_t>(), N, 112, 112, 64, st));
    void* X = r->bufX.p;
    void* Y = r->bufY.p;
    for (RBlock& b : r->blocks) {
        // Bottleneck.forward (torchvision): relu(bn1(conv1 x)) -> relu(bn2(conv2 .)) [stride here] -> bn3(conv3 .) + identity -> relu
        ISB_TRY(conv(r, st, b.c1, X, N, b.in_hw, 1, nullptr, false, r->bufT1.p));
        ISB_TRY(conv(r, st, b.c2, r->bufT1.p, N, b.in_hw, b.stride, nullptr, false, r->bufT2.p));
        const void* skip = X;
        if (b.down) {                              // downsample: conv1x1 (stride) + bn, NO activation
            ConvArgs a{};
            a.in = (const uint16_t*)X; a.w = b.cd.w.as<uint16_t>(); a.bias = b.cd.bias.as<float>(); a.out = r->bufD.p;
            a.B = N; a.H = b.in_hw; a.W = b.in_hw; a.Cin = b.cd.cin; a.Cout = b.cd.cout; a.KH = 1; a.KW = 1; a.stride = b.stride;
            a.OH = b.out_hw; a.OW = b.out_hw; a.pad = 0; a.M = N * b.out_hw * b.out_hw; a.K = b.cd.cin; a.act = 0;
            a.zeros = r->zeros.as<uint16_t>();
            ISB_TRY(launch_conv_igemm(a, st));
            skip = r->bufD.p;
        }
        ISB_TRY(conv(r, st, b.c3, r->bufT2.p, N, b.out_hw, 1, skip, true, Y));
        std::swap(X, Y);
    }
    return launch_avgpool((const uint16_t*)X, d_trunk, N, 49, 2048, st);
}

}  // namespace

extern "C" void isb_rgb_destroy(isb_rgb* r) {
    if (!r) return;
    (void)hipSetDevice(r->cfg.device);
    (void)hipDeviceSynchronize();
    if (r->own_stream) (void)hipStreamDestroy(r->own_stream);
    delete r;
}

extern "C" int isb_rgb_create(const isb_rgb_cfg* cfg, isb_rgb** out) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(cfg && out, ISB_ERR_INVALID, "isb_rgb_create: null argument");
    int ndev = 0;
    ISB_HIP(hipGetDeviceCount(&ndev));
    ISB_REQUIRE(cfg->device >= 0 && cfg->device < ndev, ISB_ERR_INVALID, "device %d not in [0,%d)", cfg->device, ndev);
    ISB_HIP(hipSetDevice(cfg->device));
    std::unique_ptr<isb_rgb, void (*)(isb_rgb*)> r(new (std::nothrow) isb_rgb(), isb_rgb_destroy);
    ISB_REQUIRE(r, ISB_ERR_NOMEM, "out of host memory");
    r->cfg = *cfg;
    if (r->cfg.max_batch <= 0) r->cfg.max_batch = 64;
    r->cfg.max_batch = std::min(r->cfg.max_batch, kMaxBatch);
    ISB_HIP(hipStreamCreateWithFlags(&r->own_stream, hipStreamNonBlocking));
    ISB_TRY(r->zeros.alloc(256));
    ISB_HIP(hipMemset(r->zeros.p, 0, 256));
    int hw = kImg / 4, cin = 64;
    for (int li = 0; li < 4; ++li)
        for (int i = 0; i < kLayers[li][1]; ++i) {
            RBlock b{};
            b.planes = kLayers[li][0];
            b.stride = i == 0 ? kLayers[li][2] : 1;
            b.cin = cin; b.in_hw = hw; b.out_hw = hw / b.stride;
            b.down = i == 0 && (b.stride != 1 || cin != 4 * b.planes);
            hw = b.out_hw; cin = 4 * b.planes;
            r->blocks.push_back(std::move(b));
        }
    *out = r.release();
    return ISB_OK;
    });
}

extern "C" int isb_rgb_load_weights(isb_rgb* r, const void* blob, size_t nbytes) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(r, ISB_ERR_INVALID, "null handle");
    ISB_HIP(hipSetDevice(r->cfg.device));
    hipStream_t st = r->own_stream;
    std::map<std::string, BlobTensor> m;
    ISB_TRY(parse_blob(blob, nbytes, m));
    r->weights = false;
    {   // conv1: f32 [64][147] with the BN scale folded
        auto it = m.find("rgb.conv1.w");
        ISB_REQUIRE(it != m.end() && it->second.numel() == 64 * 147, ISB_ERR_WEIGHTS, "rgb.conv1.w missing or mis-shaped");
        const BlobTensor *sc, *sh;
        ISB_TRY(blob_get(m, "rgb.conv1.scale", 64, 1, &sc));
        ISB_TRY(blob_get(m, "rgb.conv1.shift", 64, 1, &sh));
        std::vector<float> w(64 * 147);
        for (int o = 0; o < 64; ++o)
            for (int k = 0; k < 147; ++k) w[o * 147 + k] = it->second.data[o * 147 + k] * sc->data[o];
        ISB_TRY(upload(r->stem_w, w.data(), w.size() * 4));
        ISB_TRY(upload(r->stem_b, sh->data, 64 * 4));
    }
    int li = 1, idx = 0, left = kLayers[0][1];
    for (RBlock& b : r->blocks) {
        const std::string p = "rgb.layer" + std::to_string(li) + "." + std::to_string(idx);
        ISB_TRY(upload_conv(m, p + ".conv1", b.planes, 1, b.cin, b.c1, st));
        ISB_TRY(upload_conv(m, p + ".conv2", b.planes, 3, b.planes, b.c2, st));
        ISB_TRY(upload_conv(m, p + ".conv3", 4 * b.planes, 1, b.planes, b.c3, st));
        if (b.down) ISB_TRY(upload_conv(m, p + ".downsample", 4 * b.planes, 1, b.cin, b.cd, st));
        ++idx;
        if (--left == 0 && li < 4) { left = kLayers[li][1]; ++li; idx = 0; }
    }
    r->weights = true;
    return ISB_OK;
    });
}

extern "C" int isb_rgb_forward(isb_rgb* r, const float* d_images, int32_t N, int32_t nchw, float* d_trunk, void* stream) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(r && d_images && d_trunk, ISB_ERR_INVALID, "null argument");
    ISB_REQUIRE(N >= 1, ISB_ERR_INVALID, "batch %d < 1", N);
    ISB_REQUIRE(r->weights, ISB_ERR_STATE, "isb_rgb_forward before isb_rgb_load_weights");
    ISB_HIP(hipSetDevice(r->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    const int Nm = std::min<int>(N, r->cfg.max_batch);
    ISB_TRY(ensure_ws(r, Nm));
    const size_t isz = (size_t)3 * kImg * kImg;
    for (int n0 = 0; n0 < N; n0 += Nm) {
        const int n = std::min(Nm, N - n0);
        ISB_TRY(run(r, st, d_images + (size_t)n0 * isz, n, nchw ? 1 : 0, d_trunk + (size_t)n0 * 2048));
    }
    return ISB_OK;
    });
}

extern "C" int isb_rgb_forward_host(isb_rgb* r, const float* images, int32_t N, int32_t nchw, float* trunk) {
    return isb::guard([&]() -> int {
    ISB_REQUIRE(r && images && trunk, ISB_ERR_INVALID, "null argument");
    ISB_REQUIRE(N >= 1, ISB_ERR_INVALID, "batch %d < 1", N);
    ISB_REQUIRE(r->weights, ISB_ERR_STATE, "isb_rgb_forward_host before isb_rgb_load_weights");
    ISB_HIP(hipSetDevice(r->cfg.device));
    hipStream_t st = r->own_stream;
    const size_t isz = (size_t)3 * kImg * kImg * 4;
    if (N > r->hs_N) {
        r->hs_N = 0;
        ISB_TRY(r->hs_img.alloc(isz * N));
        ISB_TRY(r->hs_out.alloc((size_t)N * 2048 * 4));
        r->hs_N = N;
    }
    ISB_HIP(hipMemcpyAsync(r->hs_img.p, images, isz * N, hipMemcpyHostToDevice, st));
    ISB_TRY(isb_rgb_forward(r, r->hs_img.as<float>(), N, nchw, r->hs_out.as<float>(), st));
    ISB_HIP(hipMemcpyAsync(trunk, r->hs_out.p, (size_t)N * 2048 * 4, hipMemcpyDeviceToHost, st));
    ISB_HIP(hipStreamSynchronize(st));
    return ISB_OK;
    });
}
