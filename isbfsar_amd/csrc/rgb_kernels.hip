// The three kernels of the ResNet-50 trunk (RGB / hybrid action-recognition branch, reference modules/ar/utils/model.py:270-277:
// nn.Sequential(*list(resnet50(pretrained=True).children())[:-1])) that are not convolutions of the conv_igemm family:
//   rgb_stem     conv1 7x7 / 2 (3 -> 64) + folded BatchNorm + ReLU on the f32 image (NCHW as main.py:91 hands it over, or NHWC)
//   maxpool3x3s2 3x3 / 2 max-pool, padding 1
//   avgpool      global average pool of the last map -> f32 [N, 2048] ("trunk features")
// The 48 bottleneck convolutions + 4 down-sampling convolutions run on launch_conv_igemm (ReLU epilogue, residual added before it).
#include "conv_common.h"

namespace isb {

// grid (tiles of 16 x 16 output pixels, N); block 256: thread = (2 x 2 block of output pixels, group of 16 output channels).
// LDS: the 37 x 37 x 3 input patch of the tile (zeros outside the image) + the 147 x 64 weights, tap-major. A tap's four
// weight vectors feed four pixels (8 LDS reads per 64 FMAs; one pixel per thread read 5 per 16 and the launch was bound by
// them: 2.06 ms per 256 images). Every output still sums its taps in (ky, kx, c) order: the same bits.
__global__ __launch_bounds__(256) void rgb_stem_kernel(RgbStemArgs p) {
    __shared__ float patch[3][37][38];
    __shared__ __attribute__((aligned(16))) float wt[147][64];
    const int tid = threadIdx.x;
    const int OH = p.H / 2, OW = p.W / 2;
    const int tiles_x = (OW + 15) / 16;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int n = blockIdx.y;
    for (int i = tid; i < 147 * 64; i += 256) {                 // p.w [64][7][7][3] -> wt[(ky * 7 + kx) * 3 + c][o]
        const int o = i / 147, t = i - o * 147;
        wt[t][o] = p.w[i];
    }
    const int y0 = ty * 32 - 3, x0 = tx * 32 - 3;               // input origin of the patch (stride 2, pad 3)
    for (int i = tid; i < 3 * 37 * 37; i += 256) {
        const int c = i / 1369, r = (i - c * 1369) / 37, q = i - c * 1369 - r * 37;
        const int y = y0 + r, x = x0 + q;
        float v = 0.f;
        if (y >= 0 && y < p.H && x >= 0 && x < p.W)
            v = p.nchw ? p.in[(((size_t)n * 3 + c) * p.H + y) * p.W + x] : p.in[(((size_t)n * p.H + y) * p.W + x) * 3 + c];
        patch[c][r][q] = v;
    }
    __syncthreads();
    const int quad = tid >> 2, g = tid & 3;
    const int qy = quad >> 3, qx = quad & 7;                    // outputs (2 qy + dy, 2 qx + dx) of the tile
    float acc[4][16];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[o][e] = 0.f;
    for (int ky = 0; ky < 7; ++ky)
        for (int kx = 0; kx < 7; ++kx)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float4* wr = reinterpret_cast<const float4*>(&wt[(ky * 7 + kx) * 3 + c][g * 16]);
                const float4 w0 = wr[0], w1 = wr[1], w2 = wr[2], w3 = wr[3];
                const float wv[16] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z, w3.w};
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const float v = patch[c][(2 * qy + (o >> 1)) * 2 + ky][(2 * qx + (o & 1)) * 2 + kx];
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[o][e] = fmaf(v, wv[e], acc[o][e]);
                }
            }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const int oy = ty * 16 + 2 * qy + (o >> 1), ox = tx * 16 + 2 * qx + (o & 1);
        if (oy >= OH || ox >= OW) continue;
        uint32_t pk[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float a = fmaxf(acc[o][2 * e] + p.bias[g * 16 + 2 * e], 0.f), b = fmaxf(acc[o][2 * e + 1] + p.bias[g * 16 + 2 * e + 1], 0.f);
            pk[e] = (uint32_t)f2bf_(a) | ((uint32_t)f2bf_(b) << 16);
        }
        uint4* d = reinterpret_cast<uint4*>(p.out + (((size_t)n * OH + oy) * OW + ox) * 64 + g * 16);
        d[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        d[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
    }
}

int launch_rgb_stem(const RgbStemArgs& a, hipStream_t st) {
    if (a.H % 2 || a.W % 2 || a.N < 1) {
        set_error("rgb_stem: image %dx%d unsupported", a.W, a.H);
        return ISB_ERR_INVALID;
    }
    const int OH = a.H / 2, OW = a.W / 2;
    hipLaunchKernelGGL(rgb_stem_kernel, dim3(cdiv(OH, 16) * cdiv(OW, 16), a.N), dim3(256), 0, st, a);
    ISB_LAUNCHED("rgb_stem", st);
    return ISB_OK;
}

// thread = (output pixel, 8 channels); padding 1 contributes nothing (PyTorch pads max-pool with -inf)
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const uint16_t* in, uint16_t* out, int N, int H, int W, int C) {
    const int OH = H / 2, OW = W / 2, cg = C / 8;
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (size_t)N * OH * OW * cg) return;
    const int c8 = (int)(id % cg);
    const size_t pix = id / cg;
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), n = (int)(pix / ((size_t)OW * OH));
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -3.0e38f;
    for (int ky = 0; ky < 3; ++ky) {
        const int y = 2 * oy - 1 + ky;
        if (y < 0 || y >= H) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int x = 2 * ox - 1 + kx;
            if (x < 0 || x >= W) continue;
            const uint4 v = *reinterpret_cast<const uint4*>(in + (((size_t)n * H + y) * W + x) * C + c8 * 8);
            const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                m[2 * e] = fmaxf(m[2 * e], bf2f_((uint16_t)(u[e] & 0xffff)));
                m[2 * e + 1] = fmaxf(m[2 * e + 1], bf2f_((uint16_t)(u[e] >> 16)));
            }
        }
    }
    uint32_t pk[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) pk[e] = (uint32_t)f2bf_(m[2 * e]) | ((uint32_t)f2bf_(m[2 * e + 1]) << 16);
    *reinterpret_cast<uint4*>(out + (((size_t)n * OH + oy) * OW + ox) * C + c8 * 8) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
}

int launch_maxpool3x3s2(const uint16_t* in, uint16_t* out, int N, int H, int W, int C, hipStream_t st) {
    if (H % 2 || W % 2 || C % 8) {
        set_error("maxpool3x3s2: shape %dx%dx%d unsupported", H, W, C);
        return ISB_ERR_INVALID;
    }
    const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, st, in, out, N, H, W, C);
    ISB_LAUNCHED("maxpool3x3s2", st);
    return ISB_OK;
}

// thread = (image, 8 channels): mean over the HW pixels in pixel order, f32
__global__ __launch_bounds__(256) void avgpool_kernel(const uint16_t* in, float* out, int N, int HW, int C) {
    const int cg = C / 8;
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (size_t)N * cg) return;
    const int c8 = (int)(id % cg), n = (int)(id / cg);
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    for (int px = 0; px < HW; ++px) {
        const uint4 v = *reinterpret_cast<const uint4*>(in + ((size_t)n * HW + px) * C + c8 * 8);
        const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[2 * e] += bf2f_((uint16_t)(u[e] & 0xffff)); s[2 * e + 1] += bf2f_((uint16_t)(u[e] >> 16)); }
    }
    const float inv = 1.0f / (float)HW;
    float4* d = reinterpret_cast<float4*>(out + (size_t)n * C + c8 * 8);
    d[0] = make_float4(s[0] * inv, s[1] * inv, s[2] * inv, s[3] * inv);
    d[1] = make_float4(s[4] * inv, s[5] * inv, s[6] * inv, s[7] * inv);
}

int launch_avgpool(const uint16_t* in, float* out, int N, int HW, int C, hipStream_t st) {
    if (C % 8) {
        set_error("avgpool: C=%d unsupported", C);
        return ISB_ERR_INVALID;
    }
    hipLaunchKernelGGL(avgpool_kernel, dim3((unsigned)cdivz((size_t)N * (C / 8), 256)), dim3(256), 0, st, in, out, N, HW, C);
    ISB_LAUNCHED("avgpool", st);
    return ISB_OK;
}

}  // namespace isb
