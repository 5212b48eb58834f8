// EfficientNetV2-L building blocks on gfx950 (bf16 storage, f32 accumulate):
//   conv_igemm  : 3x3 / 1x1 convolution as an implicit GEMM on v_mfma_f32_32x32x16_bf16
//                 (M = B*OH*OW output pixels, N = Cout, K = KH*KW*Cin), NHWC activations,
//                 folded-BN bias + SiLU + residual epilogue, optional SE gate on the A operand
//   dwconv3x3   : depthwise 3x3 (+bias, SiLU), HBM-bound
//   se_pool     : global average pool feeding the squeeze-excite FCs (f32 GEMM)
//   stem        : conv3x3/s2 3->32 in f32 on the f32 crop
// The backbone is what the reference runs as `bbone1.engine` (utils/params.py:29, hpe.py:103);
// layer semantics follow the public efficientnetv2-l definition (isbfsar_amd/effnetv2.py).
//
// conv_igemm tiling: a k-tile is 32 input channels of ONE filter tap (Cin % 32 == 0), so a row of
// the A tile is one contiguous 64-B run of the NHWC input (or zeros for padding).  A and B tiles
// sit in LDS as [row][64 B] with the 16-B chunk index XOR-swizzled by (row>>2)&3: with that the
// ds_read_b128 fragment reads of the 32x32x16 MFMA (lane -> row lane&31, chunk 2*ks + lane>>5)
// are bank-conflict free (4-way without it).  Two LDS buffers, next tile's global loads in
// flight during the MFMAs, one barrier per k-tile.
#include "isb_common.h"
#include "kernels.h"

namespace isb {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float bf2f_(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t f2bf_(float x) { return __builtin_bit_cast(uint16_t, (__bf16)x); }
__device__ __forceinline__ float silu_(float x) { return x / (1.0f + __expf(-x)); }

constexpr int CK = 32;            // k-tile (bf16 elements) = 64 B per row
constexpr int ROWB = 64;          // bytes per LDS row

__device__ __forceinline__ int swz(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <int TM, int TN, int WGM, int WGN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs p) {
    constexpr int BM = 32 * TM * WGM;
    constexpr int BN = 32 * TN * WGN;
    constexpr int A_PASS = BM / 64;                       // 16-B chunks per thread for the A tile
    constexpr int B_CHUNKS = BN * 4;
    constexpr int B_PASS = (B_CHUNKS + 255) / 256;
    constexpr int BUF = (BM + BN) * ROWB;
    static_assert(WGM * WGN == 4, "4 waves");
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int chunk = tid & 3;

    // ---- per-thread A rows (fixed over the k loop): pixel coordinates
    int a_b[A_PASS], a_iy[A_PASS], a_ix[A_PASS];
    bool a_ok[A_PASS];
    const int ohw = p.OH * p.OW;
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
        const int m = m0 + (tid >> 2) + 64 * i;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
        const int b = mm / ohw, rem = mm - b * ohw;
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        a_b[i] = b;
        a_iy[i] = oy * p.stride - p.pad;
        a_ix[i] = ox * p.stride - p.pad;
    }
    uint4 ra[A_PASS], rb[B_PASS];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);

    auto gload = [&](int kt) {
        const int k0 = kt * CK;
        const int tap = k0 / p.Cin;
        const int c0 = k0 - tap * p.Cin;
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) {
            const int iy = a_iy[i] + ky, ix = a_ix[i] + kx;
            uint4 v = zero4;
            if (a_ok[i] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
                v = *reinterpret_cast<const uint4*>(p.in + ((size_t)(a_b[i] * p.H + iy) * p.W + ix) * p.Cin + c0 + chunk * 8);
                if (p.gate) {      // squeeze-excite gate of the producing depthwise conv (1x1 convs only)
                    const float* g = p.gate + (size_t)a_b[i] * p.Cin + c0 + chunk * 8;
                    const float4 g0 = *reinterpret_cast<const float4*>(g), g1 = *reinterpret_cast<const float4*>(g + 4);
                    uint32_t w[4] = {v.x, v.y, v.z, v.w};
                    const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lo = bf2f_((uint16_t)(w[e] & 0xffff)) * gg[2 * e];
                        const float hi = bf2f_((uint16_t)(w[e] >> 16)) * gg[2 * e + 1];
                        w[e] = (uint32_t)f2bf_(lo) | ((uint32_t)f2bf_(hi) << 16);
                    }
                    v = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_PASS; ++i) {
            const int cidx = tid + 256 * i;
            uint4 v = zero4;
            if (cidx < B_CHUNKS) {
                const int n = n0 + (cidx >> 2);
                if (n < p.Cout) v = *reinterpret_cast<const uint4*>(p.w + (size_t)n * p.K + k0 + (cidx & 3) * 8);
            }
            rb[i] = v;
        }
    };
    auto lstore = [&](int buf) {
        unsigned char* As = lds + buf * BUF;
        unsigned char* Bs = As + BM * ROWB;
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) *reinterpret_cast<uint4*>(As + swz((tid >> 2) + 64 * i, chunk)) = ra[i];
#pragma unroll
        for (int i = 0; i < B_PASS; ++i) {
            const int cidx = tid + 256 * i;
            if (cidx < B_CHUNKS) *reinterpret_cast<uint4*>(Bs + swz(cidx >> 2, cidx & 3)) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nkt = p.K / CK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) gload(kt + 1);
        const unsigned char* As = lds + cur * BUF;
        const unsigned char* Bs = As + BM * ROWB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(As + swz((wm * TM + i) * 32 + r, 2 * ks + h)));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bfr[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Bs + swz((wn * TN + j) * 32 + r, 2 * ks + h)));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nkt) lstore(cur ^ 1);
        __syncthreads();
    }

    // epilogue: D col = lane&31 -> n, row = (e&3) + 8*(e>>2) + 4*(lane>>5) -> m
    uint16_t* out16 = reinterpret_cast<uint16_t*>(p.out);
    float* out32 = reinterpret_cast<float*>(p.out);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + r;
        if (n >= p.Cout) continue;
        const float bias = p.bias[n];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + (wm * TM + i) * 32 + 4 * h;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = mb + (e & 3) + 8 * (e >> 2);
                if (m < p.M) {
                    float v = acc[i][j][e] + bias;
                    if (p.act) v = silu_(v);
                    const size_t o = (size_t)m * p.Cout + n;
                    if (p.res) v += bf2f_(p.res[o]);
                    if (p.out_f32) out32[o] = v;
                    else out16[o] = f2bf_(v);
                }
            }
        }
    }
}

int launch_conv_igemm(const ConvArgs& a, hipStream_t st) {
    if (a.Cin % 32 != 0 || a.Cout % 32 != 0 || a.K != a.KH * a.KW * a.Cin || a.M <= 0) {
        set_error("conv_igemm: unsupported shape Cin=%d Cout=%d K=%d M=%d", a.Cin, a.Cout, a.K, a.M);
        return ISB_ERR_INVALID;
    }
    if (a.gate && (a.KH != 1 || a.stride != 1)) {
        set_error("conv_igemm: SE gate only on 1x1 convolutions");
        return ISB_ERR_INVALID;
    }
    if (a.Cout % 128 == 0) {
        dim3 g(cdiv(a.M, 128), a.Cout / 128);
        hipLaunchKernelGGL((conv_igemm_kernel<2, 2, 2, 2>), g, dim3(256), 0, st, a);
    } else if (a.Cout % 96 == 0) {
        dim3 g(cdiv(a.M, 128), a.Cout / 96);
        hipLaunchKernelGGL((conv_igemm_kernel<1, 3, 4, 1>), g, dim3(256), 0, st, a);
    } else if (a.Cout % 64 == 0) {
        dim3 g(cdiv(a.M, 128), a.Cout / 64);
        hipLaunchKernelGGL((conv_igemm_kernel<2, 1, 2, 2>), g, dim3(256), 0, st, a);
    } else if (a.Cout == 224) {
        dim3 g(cdiv(a.M, 128), 1);
        hipLaunchKernelGGL((conv_igemm_kernel<1, 7, 4, 1>), g, dim3(256), 0, st, a);
    } else {
        dim3 g(cdiv(a.M, 256), a.Cout / 32);
        hipLaunchKernelGGL((conv_igemm_kernel<2, 1, 4, 1>), g, dim3(256), 0, st, a);
    }
    ISB_LAUNCHED("conv_igemm", st);
    return ISB_OK;
}

// =====================================================================================
// depthwise 3x3 (+ folded-BN bias + SiLU). thread = (output pixel, 8 channels)
// weights tap-major f32 [9][C] (BN scale folded in)
// =====================================================================================
__global__ __launch_bounds__(256) void dwconv3x3_kernel(DwArgs p) {
    const int cg = p.C >> 3;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)p.B * p.OH * p.OW * cg;
    if (idx >= total) return;
    const int c8 = (int)(idx % cg);
    size_t pix = idx / cg;
    const int ox = (int)(pix % p.OW); pix /= p.OW;
    const int oy = (int)(pix % p.OH);
    const int b = (int)(pix / p.OH);
    const int c = c8 * 8;
    float acc[8];
    {
        const float4 s0 = *reinterpret_cast<const float4*>(p.bias + c), s1 = *reinterpret_cast<const float4*>(p.bias + c + 4);
        acc[0] = s0.x; acc[1] = s0.y; acc[2] = s0.z; acc[3] = s0.w; acc[4] = s1.x; acc[5] = s1.y; acc[6] = s1.z; acc[7] = s1.w;
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * p.stride - p.pad + ky;
        if ((unsigned)iy >= (unsigned)p.H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * p.stride - p.pad + kx;
            if ((unsigned)ix >= (unsigned)p.W) continue;
            const uint4 v = *reinterpret_cast<const uint4*>(p.in + ((size_t)(b * p.H + iy) * p.W + ix) * p.C + c);
            const float* w = p.w + (size_t)(ky * 3 + kx) * p.C + c;
            const float4 w0 = *reinterpret_cast<const float4*>(w), w1 = *reinterpret_cast<const float4*>(w + 4);
            acc[0] = fmaf(bf2f_((uint16_t)(v.x & 0xffff)), w0.x, acc[0]);
            acc[1] = fmaf(bf2f_((uint16_t)(v.x >> 16)), w0.y, acc[1]);
            acc[2] = fmaf(bf2f_((uint16_t)(v.y & 0xffff)), w0.z, acc[2]);
            acc[3] = fmaf(bf2f_((uint16_t)(v.y >> 16)), w0.w, acc[3]);
            acc[4] = fmaf(bf2f_((uint16_t)(v.z & 0xffff)), w1.x, acc[4]);
            acc[5] = fmaf(bf2f_((uint16_t)(v.z >> 16)), w1.y, acc[5]);
            acc[6] = fmaf(bf2f_((uint16_t)(v.w & 0xffff)), w1.z, acc[6]);
            acc[7] = fmaf(bf2f_((uint16_t)(v.w >> 16)), w1.w, acc[7]);
        }
    }
    uint32_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
        o[e] = (uint32_t)f2bf_(silu_(acc[2 * e])) | ((uint32_t)f2bf_(silu_(acc[2 * e + 1])) << 16);
    *reinterpret_cast<uint4*>(p.out + (((size_t)(b * p.OH + oy) * p.OW + ox) * p.C + c)) = make_uint4(o[0], o[1], o[2], o[3]);
}

int launch_dwconv3x3(const DwArgs& a, hipStream_t st) {
    if (a.C % 8 != 0) {
        set_error("dwconv3x3: C=%d not a multiple of 8", a.C);
        return ISB_ERR_INVALID;
    }
    const size_t total = (size_t)a.B * a.OH * a.OW * (a.C / 8);
    hipLaunchKernelGGL(dwconv3x3_kernel, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, st, a);
    ISB_LAUNCHED("dwconv3x3", st);
    return ISB_OK;
}

// =====================================================================================
// SE squeeze: mean over HW per (b, c). grid (C/64, B), block 256 = 8 channel-chunks x 32 pixel lanes
// =====================================================================================
__global__ __launch_bounds__(256) void se_pool_kernel(PoolArgs p) {
    __shared__ float red[32][65];
    const int b = blockIdx.y;
    const int cchunk = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int c = blockIdx.x * 64 + cchunk * 8;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c < p.C) {
        for (int px = pl; px < p.HW; px += 32) {
            const uint4 v = *reinterpret_cast<const uint4*>(p.in + ((size_t)b * p.HW + px) * p.C + c);
            s[0] += bf2f_((uint16_t)(v.x & 0xffff)); s[1] += bf2f_((uint16_t)(v.x >> 16));
            s[2] += bf2f_((uint16_t)(v.y & 0xffff)); s[3] += bf2f_((uint16_t)(v.y >> 16));
            s[4] += bf2f_((uint16_t)(v.z & 0xffff)); s[5] += bf2f_((uint16_t)(v.z >> 16));
            s[6] += bf2f_((uint16_t)(v.w & 0xffff)); s[7] += bf2f_((uint16_t)(v.w >> 16));
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[pl][cchunk * 8 + e] = s[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        const int cc = blockIdx.x * 64 + threadIdx.x;
        if (cc < p.C) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 32; ++q) t += red[q][threadIdx.x];
            p.out[(size_t)b * p.C + cc] = t / (float)p.HW;
        }
    }
}

int launch_se_pool(const PoolArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(se_pool_kernel, dim3(cdiv(a.C, 64), a.B), dim3(256), 0, st, a);
    ISB_LAUNCHED("se_pool", st);
    return ISB_OK;
}

// =====================================================================================
// stem: conv3x3 stride 2 (TF SAME on an even input: pad bottom/right), 3 -> 32, bias, SiLU.
// f32 crop [B,256,256,3] -> bf16 [B,128,128,32]. thread = one output pixel, all 32 channels;
// weights [32][3][3][3] f32 (scale folded) are wave-uniform -> scalar loads.
// =====================================================================================
__global__ __launch_bounds__(256) void stem_kernel(StemArgs p) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int OH = p.H / 2, OW = p.W / 2;
    if (idx >= (size_t)p.B * OH * OW) return;
    const int ox = (int)(idx % OW), oy = (int)((idx / OW) % OH), b = (int)(idx / ((size_t)OW * OH));
    float x[27];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = 2 * oy + ky, ix = 2 * ox + kx;
            const bool ok = iy < p.H && ix < p.W;
            const float* src = p.in + ((size_t)(b * p.H + (ok ? iy : 0)) * p.W + (ok ? ix : 0)) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) x[(ky * 3 + kx) * 3 + c] = ok ? src[c] : 0.f;
        }
    uint32_t o[16];
#pragma unroll
    for (int co = 0; co < 32; co += 2) {
        float a0 = p.bias[co], a1 = p.bias[co + 1];
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            a0 = fmaf(x[k], p.w[co * 27 + k], a0);
            a1 = fmaf(x[k], p.w[(co + 1) * 27 + k], a1);
        }
        o[co >> 1] = (uint32_t)f2bf_(silu_(a0)) | ((uint32_t)f2bf_(silu_(a1)) << 16);
    }
    uint4* dst = reinterpret_cast<uint4*>(p.out + idx * 32);
    dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
    dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
    dst[2] = make_uint4(o[8], o[9], o[10], o[11]);
    dst[3] = make_uint4(o[12], o[13], o[14], o[15]);
}

int launch_stem(const StemArgs& a, hipStream_t st) {
    const size_t total = (size_t)a.B * (a.H / 2) * (a.W / 2);
    hipLaunchKernelGGL(stem_kernel, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, st, a);
    ISB_LAUNCHED("stem", st);
    return ISB_OK;
}

// f32 -> bf16 (weights at load time), with an optional per-row scale (folded BN)
__global__ void f32_to_bf16_rows_kernel(const float* in, const float* row_scale, uint16_t* out, size_t rows, size_t cols) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * cols) return;
    const float s = row_scale ? row_scale[i / cols] : 1.f;
    out[i] = f2bf_(in[i] * s);
}

int launch_f32_to_bf16_rows(const float* in, const float* row_scale, uint16_t* out, size_t rows, size_t cols, hipStream_t st) {
    hipLaunchKernelGGL(f32_to_bf16_rows_kernel, dim3((unsigned)cdivz(rows * cols, 256)), dim3(256), 0, st, in, row_scale, out, rows, cols);
    ISB_LAUNCHED("f32_to_bf16_rows", st);
    return ISB_OK;
}

}  // namespace isb
