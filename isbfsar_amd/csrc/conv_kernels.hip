// EfficientNetV2-L building blocks on gfx950 (bf16 storage, f32 accumulate):
//   conv_igemm  : 3x3 / 1x1 convolution as an implicit GEMM on v_mfma_f32_32x32x16_bf16
//                 (M = B*OH*OW output pixels, N = Cout, K = KH*KW*Cin), NHWC activations,
//                 folded-BN bias + SiLU + residual epilogue, optional SE gate on the A operand
//   dwconv3x3   : depthwise 3x3 (+bias, SiLU) fused with the squeeze-excite average pool, HBM-bound
//   se_fc1/fc2  : the two squeeze-excite FCs in f32 on the vector ALU
//   stem        : conv3x3/s2 3->32 in f32 on the f32 crop
// The backbone is what the reference runs as `bbone1.engine` (utils/params.py:29, hpe.py:103);
// layer semantics follow the public efficientnetv2-l definition (isbfsar_amd/effnetv2.py).
//
// conv_igemm tiling: a k-tile is 32 input channels of ONE filter tap (Cin % 32 == 0), so a row of
// the A tile is one contiguous 64-B run of the NHWC input (or zeros for padding).  A and B tiles
// sit in LDS as [row][64 B] with the 16-B chunk index XOR-swizzled by (row>>2)&3: with that the
// ds_read_b128 fragment reads of the 32x32x16 MFMA (lane -> row lane&31, chunk 2*ks + lane>>5)
// are bank-conflict free (4-way without it).  Two LDS buffers, next tile's global loads in
// flight during the MFMAs, one barrier per k-tile.  The MFMA computes the TRANSPOSED output tile
// (weights as the A operand) so each lane ends up with 4 consecutive channels of one pixel: bias,
// SiLU and the residual are applied in registers, the bf16 tile is staged through LDS (row stride
// BN*2+16 B: conflict-free 8-byte writes) and leaves as full 16-byte pieces.
#include <algorithm>
#include <type_traits>

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

__device__ __forceinline__ float silu_fast(float x) {
    // x * sigmoid(x) with v_exp_f32 / v_rcp_f32 (about 1 ulp each; the result is rounded to bf16)
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

// epilogue shared by the register-staged and the LDS-DMA kernels
template <int TM, int TN, int WGM, int WGN>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x16 (&acc)[TM][TN], unsigned char* lds, int m0, int n0,
                                              int wm, int wn, int r, int h, int tid) {
    constexpr int NT = 64 * WGM * WGN;
    constexpr int BM = 32 * TM * WGM;
    constexpr int BN = 32 * TN * WGN;
    constexpr int CROW = BN * 2 + 16;
    // ---- epilogue. acc[i][j][e]: channel n = n0 + (wn*TN+j)*32 + 8*(e>>2) + 4*h + (e&3), pixel m = m0 + (wm*TM+i)*32 + r
    if (p.out_f32) {          // f32 output (last 1x1 conv feeding the f32 pose head): direct stores
        float* out32 = reinterpret_cast<float*>(p.out);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + (wm * TM + i) * 32 + r;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n0 + (wn * TN + j) * 32 + 8 * q + 4 * h;
                    if (n >= p.Cout) continue;
                    const float4 bs = *reinterpret_cast<const float4*>(p.bias + n);
                    float4 v = make_float4(acc[i][j][4 * q] + bs.x, acc[i][j][4 * q + 1] + bs.y, acc[i][j][4 * q + 2] + bs.z,
                                           acc[i][j][4 * q + 3] + bs.w);
                    if (p.act) { v.x = silu_fast(v.x); v.y = silu_fast(v.y); v.z = silu_fast(v.z); v.w = silu_fast(v.w); }
                    *reinterpret_cast<float4*>(out32 + (size_t)m * p.Cout + n) = v;
                }
        }
        return;
    }
    // bf16 output: bias + SiLU + residual in registers (one rounding), stage the tile in LDS, then
    // write full 16-byte pieces, 256 B contiguous per pixel row. Tiles too wide to stage (> 64 KiB)
    // store their 8-byte packed pieces straight from registers (the L2 merges the partial lines).
    constexpr bool STAGE = BM * CROW <= 65536;
    uint16_t* out16 = reinterpret_cast<uint16_t*>(p.out);
    unsigned char* Cs = lds;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ml = (wm * TM + i) * 32 + r;
        const int m = m0 + ml;
        const bool mok = m < p.M;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = (wn * TN + j) * 32 + 8 * q + 4 * h;
                const int n = n0 + nl;
                float v0 = acc[i][j][4 * q], v1 = acc[i][j][4 * q + 1], v2 = acc[i][j][4 * q + 2], v3 = acc[i][j][4 * q + 3];
                if (n < p.Cout) {
                    const float4 bs = *reinterpret_cast<const float4*>(p.bias + n);
                    v0 += bs.x; v1 += bs.y; v2 += bs.z; v3 += bs.w;
                    if (p.act) { v0 = silu_fast(v0); v1 = silu_fast(v1); v2 = silu_fast(v2); v3 = silu_fast(v3); }
                    if (p.res && mok) {
                        const uint2 rr = *reinterpret_cast<const uint2*>(p.res + (size_t)m * p.Cout + n);
                        v0 += bf2f_((uint16_t)(rr.x & 0xffff)); v1 += bf2f_((uint16_t)(rr.x >> 16));
                        v2 += bf2f_((uint16_t)(rr.y & 0xffff)); v3 += bf2f_((uint16_t)(rr.y >> 16));
                    }
                }
                uint2 pk;
                pk.x = (uint32_t)f2bf_(v0) | ((uint32_t)f2bf_(v1) << 16);
                pk.y = (uint32_t)f2bf_(v2) | ((uint32_t)f2bf_(v3) << 16);
                if constexpr (STAGE) *reinterpret_cast<uint2*>(Cs + ml * CROW + nl * 2) = pk;
                else if (mok && n < p.Cout) *reinterpret_cast<uint2*>(out16 + (size_t)m * p.Cout + n) = pk;
            }
    }
    if constexpr (!STAGE) return;
    __syncthreads();
    constexpr int CPR = BN / 8;                            // 16-byte pieces per tile row
#pragma unroll 4
    for (int id = tid; id < BM * CPR; id += NT) {
        const int row = id / CPR, cc = id - row * CPR;
        const int m = m0 + row, n = n0 + cc * 8;
        if (m < p.M && n < p.Cout)
            *reinterpret_cast<uint4*>(out16 + (size_t)m * p.Cout + n) = *reinterpret_cast<const uint4*>(Cs + row * CROW + cc * 16);
    }
}

template <int TM, int TN, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_igemm_kernel(ConvArgs p) {
    constexpr int NT = 64 * WGM * WGN;                    // threads: 4 or 8 waves
    constexpr int BM = 32 * TM * WGM;
    constexpr int BN = 32 * TN * WGN;
    constexpr int A_ROWS = NT / 4;                        // tile rows staged per pass (4 x 16-B chunks per row)
    constexpr int A_PASS = (BM + A_ROWS - 1) / A_ROWS;
    constexpr int B_CHUNKS = BN * 4;
    constexpr int B_PASS = (B_CHUNKS + NT - 1) / NT;
    constexpr int BUF = (BM + BN) * ROWB;
    constexpr int CROW = BN * 2 + 16;                     // staged C tile row (bf16) + pad: conflict-free b64 writes
    constexpr int LDS_BYTES = (2 * BUF > BM * CROW || BM * CROW > 65536) ? 2 * BUF : BM * CROW;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int chunk = tid & 3;

    // ---- per-thread A rows (fixed over the k loop): pixel coordinates, 32-bit element offsets
    int a_off[A_PASS], a_iy[A_PASS], a_ix[A_PASS], a_gate[A_PASS];
    const int ohw = p.OH * p.OW;
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
        const int m = m0 + (tid >> 2) + A_ROWS * i;
        const bool ok = m < p.M && (tid >> 2) + A_ROWS * i < BM;
        const int mm = ok ? m : 0;
        const int b = mm / ohw, rem = mm - b * ohw;
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        a_iy[i] = ok ? oy * p.stride - p.pad : -100000;   // rows past M never pass the bounds test
        a_ix[i] = ox * p.stride - p.pad;
        a_off[i] = ok ? ((b * p.H + a_iy[i]) * p.W + a_ix[i]) * p.Cin + chunk * 8 : 0;
        a_gate[i] = b * p.Cin + chunk * 8;
    }
    uint4 rA[1][A_PASS], rB[1][B_PASS];
    float4 gA[A_PASS][2];
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) gA[i][0] = gA[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    using St0 = std::integral_constant<int, 0>;

    auto gload = [&](int kt, auto stage) {
        constexpr int SG = decltype(stage)::value;
        const int k0 = kt * CK;
        const int tap = k0 / p.Cin;
        const int c0 = k0 - tap * p.Cin;
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
        const int tap_off = (ky * p.W + kx) * p.Cin + c0;
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) {
            const int iy = a_iy[i] + ky, ix = a_ix[i] + kx;
            uint4 v = zero4;
            if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
                v = *reinterpret_cast<const uint4*>(p.in + (a_off[i] + tap_off));
                if (p.gate) {      // squeeze-excite gate (1x1 convs only): fetched now, applied when the tile is
                                   // written to LDS, so neither load is waited for before the MFMAs of this step
                    const float* g = p.gate + (a_gate[i] + c0);
                    gA[i][0] = *reinterpret_cast<const float4*>(g);
                    gA[i][1] = *reinterpret_cast<const float4*>(g + 4);
                }
            }
            rA[SG][i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_PASS; ++i) {
            const int cidx = tid + NT * i;
            uint4 v = zero4;
            if (cidx < B_CHUNKS) {
                const int n = n0 + (cidx >> 2);
                if (n < p.Cout) v = *reinterpret_cast<const uint4*>(p.w + (size_t)n * p.K + k0 + (cidx & 3) * 8);
            }
            rB[SG][i] = v;
        }
    };
    auto lstore = [&](int buf, auto stage) {
        constexpr int SG = decltype(stage)::value;
        unsigned char* As = lds + buf * BUF;
        unsigned char* Bs = As + BM * ROWB;
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) {
            uint4 v = rA[SG][i];
            if (p.gate) {
                uint32_t w[4] = {v.x, v.y, v.z, v.w};
                const float gg[8] = {gA[i][0].x, gA[i][0].y, gA[i][0].z, gA[i][0].w, gA[i][1].x, gA[i][1].y, gA[i][1].z, gA[i][1].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = bf2f_((uint16_t)(w[e] & 0xffff)) * gg[2 * e];
                    const float hi = bf2f_((uint16_t)(w[e] >> 16)) * gg[2 * e + 1];
                    w[e] = (uint32_t)f2bf_(lo) | ((uint32_t)f2bf_(hi) << 16);
                }
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            if ((tid >> 2) + A_ROWS * i < BM) *reinterpret_cast<uint4*>(As + swz((tid >> 2) + A_ROWS * i, chunk)) = v;
        }
#pragma unroll
        for (int i = 0; i < B_PASS; ++i) {
            const int cidx = tid + NT * i;
            if (cidx < B_CHUNKS) *reinterpret_cast<uint4*>(Bs + swz(cidx >> 2, cidx & 3)) = rB[SG][i];
        }
    };

    // accumulators hold the TRANSPOSED tile: D = W_tile (rows n) x X_tile^T (cols m), so a lane owns one
    // pixel (m = lane&31) and 4 consecutive channels per register quad -> 8-byte packed bf16 pieces
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nkt = p.K / CK;
    auto compute = [&](int cur) {
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
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
    };
    gload(0, St0{});
    lstore(0, St0{});
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) gload(kt + 1, St0{});            // in flight during this tile's MFMAs
        compute(cur);
        if (kt + 1 < nkt) lstore(cur ^ 1, St0{});
        __syncthreads();
    }

    conv_epilogue<TM, TN, WGM, WGN>(p, acc, lds, m0, n0, wm, wn, r, h, tid);
}

// -------------------------------------------------------------------------------------------
// Fused MBConv front half: 1x1 expand (this GEMM) -> folded BN + SiLU -> depthwise 3x3 (stride 1)
// -> folded BN + SiLU -> D, plus the squeeze-excite average pool -- without the expanded tensor E ever
// leaving the chip. The M tile is exactly ONE sample (BM == H*W pixels in raster order), so after the
// bias/SiLU'd bf16 E tile has been staged in LDS as [pixel][channel] the 3x3 neighbourhood of every
// output is in LDS. thread = (8-channel chunk, pixel quad) as in dwconv3x3_pool_kernel.
// -------------------------------------------------------------------------------------------
template <int TM, int TN, int WGM, int WGN>
__device__ __forceinline__ void conv_epilogue_dw(const ConvArgs& p, f32x16 (&acc)[TM][TN], unsigned char* lds, int m0, int n0,
                                                 int wm, int wn, int r, int h, int tid) {
    constexpr int NT = 64 * WGM * WGN;
    constexpr int BM = 32 * TM * WGM;
    constexpr int BN = 32 * TN * WGN;
    constexpr int CROW = BN * 2 + 16;
    constexpr int CH = BN / 8;                             // 8-channel chunks in the slab
    constexpr int NQ = BM / 4;                             // pixel quads of the sample
    static_assert(CH * NQ == NT, "one (chunk, quad) item per thread");
    unsigned char* Cs = lds;
    float* red = reinterpret_cast<float*>(lds + BM * CROW);   // [NQ][BN] partial pool sums
    // stage E = silu(acc + bias) as bf16 [pixel][channel]
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ml = (wm * TM + i) * 32 + r;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = (wn * TN + j) * 32 + 8 * q + 4 * h;
                const float4 bs = *reinterpret_cast<const float4*>(p.bias + n0 + nl);
                const float v0 = silu_fast(acc[i][j][4 * q] + bs.x), v1 = silu_fast(acc[i][j][4 * q + 1] + bs.y);
                const float v2 = silu_fast(acc[i][j][4 * q + 2] + bs.z), v3 = silu_fast(acc[i][j][4 * q + 3] + bs.w);
                uint2 pk;
                pk.x = (uint32_t)f2bf_(v0) | ((uint32_t)f2bf_(v1) << 16);
                pk.y = (uint32_t)f2bf_(v2) | ((uint32_t)f2bf_(v3) << 16);
                *reinterpret_cast<uint2*>(Cs + ml * CROW + nl * 2) = pk;
            }
    }
    __syncthreads();
    // depthwise 3x3, stride 1, pad 1 over the staged sample
    const int W = p.OW, Hh = p.OH;
    const int cl = tid % CH, pq = tid / CH;
    const int c = n0 + cl * 8;                             // global expanded channel
    const int qpr = W >> 2;
    const int oy = pq / qpr, ox0 = (pq - oy * qpr) * 4;
    const int b = m0 / BM;
    float w[9][8], a4[4][8], psum[8];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float4 w0 = *reinterpret_cast<const float4*>(p.dw_w + (size_t)t * p.Cout + c), w1 = *reinterpret_cast<const float4*>(p.dw_w + (size_t)t * p.Cout + c + 4);
        w[t][0] = w0.x; w[t][1] = w0.y; w[t][2] = w0.z; w[t][3] = w0.w; w[t][4] = w1.x; w[t][5] = w1.y; w[t][6] = w1.z; w[t][7] = w1.w;
    }
    {
        const float4 s0 = *reinterpret_cast<const float4*>(p.dw_bias + c), s1 = *reinterpret_cast<const float4*>(p.dw_bias + c + 4);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            a4[o][0] = s0.x; a4[o][1] = s0.y; a4[o][2] = s0.z; a4[o][3] = s0.w; a4[o][4] = s1.x; a4[o][5] = s1.y; a4[o][6] = s1.z; a4[o][7] = s1.w;
        }
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy - 1 + ky;
        if ((unsigned)iy >= (unsigned)Hh) continue;
#pragma unroll
        for (int col = 0; col < 6; ++col) {
            const int ix = ox0 - 1 + col;
            if ((unsigned)ix >= (unsigned)W) continue;
            const uint4 v = *reinterpret_cast<const uint4*>(Cs + (iy * W + ix) * CROW + cl * 16);
            float x[8];
            x[0] = bf2f_((uint16_t)(v.x & 0xffff)); x[1] = bf2f_((uint16_t)(v.x >> 16));
            x[2] = bf2f_((uint16_t)(v.y & 0xffff)); x[3] = bf2f_((uint16_t)(v.y >> 16));
            x[4] = bf2f_((uint16_t)(v.z & 0xffff)); x[5] = bf2f_((uint16_t)(v.z >> 16));
            x[6] = bf2f_((uint16_t)(v.w & 0xffff)); x[7] = bf2f_((uint16_t)(v.w >> 16));
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const int kx = col - o;
                if (kx >= 0 && kx < 3) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) a4[o][e] = fmaf(x[e], w[ky * 3 + kx][e], a4[o][e]);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) psum[e] = 0.f;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        uint32_t pk[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint16_t lo = f2bf_(silu_fast(a4[o][2 * e])), hi = f2bf_(silu_fast(a4[o][2 * e + 1]));
            pk[e] = (uint32_t)lo | ((uint32_t)hi << 16);
            psum[2 * e] += bf2f_(lo);
            psum[2 * e + 1] += bf2f_(hi);
        }
        *reinterpret_cast<uint4*>(p.dw_out + (((size_t)(b * Hh + oy) * W + ox0 + o) * p.Cout + c)) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[pq * BN + cl * 8 + e] = psum[e];
    __syncthreads();
    if (tid < BN) {
        float t = 0.f;
        for (int s2 = 0; s2 < NQ; ++s2) t += red[s2 * BN + tid];
        p.pooled[(size_t)b * p.Cout + n0 + tid] = t / (float)BM;
    }
}

// -------------------------------------------------------------------------------------------
// LDS-DMA variant (convolutions without an SE gate): tiles go global -> LDS directly
// (global_load_lds_dwordx4: 1 KiB per wave-instruction, destination = wave-uniform base + lane*16,
// so the XOR swizzle is applied to the per-lane SOURCE address). No staging registers, no
// ds_write instructions; padding taps and rows past M read a 16-byte zero line. The DMA of tile
// k+1 is in flight while tile k feeds the MFMAs; __syncthreads() waits for it (vmcnt) before the swap.
// -------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int TM, int TN, int WGM, int WGN, int NB, bool DW = false>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_igemm_dma_kernel(ConvArgs p) {
    constexpr int NW = WGM * WGN;
    constexpr int BM = 32 * TM * WGM;
    constexpr int BN = 32 * TN * WGN;
    constexpr int A_INST = BM / 16, B_INST = BN / 16;    // 1-KiB pieces (16 tile rows) per tile
    constexpr int A_PW = (A_INST + NW - 1) / NW, B_PW = (B_INST + NW - 1) / NW;
    constexpr int BUF = (BM + BN) * ROWB;
    constexpr int CROW = BN * 2 + 16;
    constexpr int LDS_PLAIN = (NB * BUF > BM * CROW || BM * CROW > 65536) ? NB * BUF : BM * CROW;
    constexpr int LDS_DW = BM * CROW + (BM / 4) * BN * 4;   // staged E tile + partial pool sums
    constexpr int LDS_BYTES = DW ? (LDS_DW > NB * BUF ? LDS_DW : NB * BUF) : LDS_PLAIN;
    static_assert(NB == 2 || (A_INST % NW == 0 && B_INST % NW == 0), "ring mode needs the same DMA count in every wave");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

    // lane -> (tile row, logical chunk) of the 1-KiB piece it fills: LDS position lane*16 holds
    // physical chunk lane&3 of row lane>>2; the source chunk is the inverse swizzle of that
    int a_off[A_PW], a_iy[A_PW], a_ix[A_PW], b_off[B_PW];
    const int ohw = p.OH * p.OW;
#pragma unroll
    for (int s = 0; s < A_PW; ++s) {
        const int row = 16 * (wave + NW * s) + (lane >> 2);
        const int logical = (lane & 3) ^ ((row >> 2) & 3);
        const int m = m0 + row;
        const bool ok = m < p.M && row < BM;
        const int mm = ok ? m : 0;
        const int b = mm / ohw, rem = mm - b * ohw;
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        a_iy[s] = ok ? oy * p.stride - p.pad : -100000;
        a_ix[s] = ox * p.stride - p.pad;
        a_off[s] = ok ? ((b * p.H + a_iy[s]) * p.W + a_ix[s]) * p.Cin + logical * 8 : 0;
    }
#pragma unroll
    for (int s = 0; s < B_PW; ++s) {
        const int row = 16 * (wave + NW * s) + (lane >> 2);
        const int logical = (lane & 3) ^ ((row >> 2) & 3);
        const int n = n0 + row;
        b_off[s] = (n < p.Cout && row < BN) ? n * p.K + logical * 8 : -1;
    }

    auto dma = [&](int kt, int buf) {
        const int k0 = kt * CK;
        const int tap = k0 / p.Cin;
        const int c0 = k0 - tap * p.Cin;
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
        const int tap_off = (ky * p.W + kx) * p.Cin + c0;
        unsigned char* base = lds + buf * BUF;
#pragma unroll
        for (int s = 0; s < A_PW; ++s) {
            const int q = wave + NW * s;
            if (q < A_INST) {
                const int iy = a_iy[s] + ky, ix = a_ix[s] + kx;
                const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const uint16_t* src = ok ? p.in + (a_off[s] + tap_off) : p.zeros;
                __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(base + q * 1024), 16, 0, 0);
            }
        }
#pragma unroll
        for (int s = 0; s < B_PW; ++s) {
            const int q = wave + NW * s;
            if (q < B_INST) {
                const uint16_t* src = b_off[s] >= 0 ? p.w + (b_off[s] + k0) : p.zeros;
                __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(base + BM * ROWB + q * 1024), 16, 0, 0);
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto compute = [&](int cur) {
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
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
    };

    const int nkt = p.K / CK;
    if constexpr (NB == 2) {
        dma(0, 0);
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nkt) dma(kt + 1, cur ^ 1);
            compute(cur);
            __syncthreads();
        }
    } else if constexpr (NB == 3) {
        // three buffers, two tiles in flight: tile kt+2 is requested before tile kt is consumed and is
        // waited for one iteration later with a COUNTED vmcnt (the newest tile stays in flight across
        // the barrier). The barrier after the MFMAs also orders the next DMA (into the buffer just
        // consumed) behind every wave's reads.
        constexpr int PER = A_PW + B_PW;
        dma(0, 0);
        if (nkt > 1) {
            dma(1, 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt < nkt; ++kt) {
            if (kt + 2 < nkt) dma(kt + 2, (kt + 2) % 3);
            compute(kt % 3);
            if (kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    } else {
        // ring of NB buffers, D = NB-2 tiles in flight beyond the one being consumed. One barrier per
        // tile, placed BEFORE the MFMAs; the DMA issued in iteration kt overwrites the buffer of tile
        // kt-2, which every wave finished before it arrived at barrier kt-1 (hence NB = D + 2).
        // Waits are counted (never vmcnt(0) in steady state) and the barrier is the raw s_barrier:
        // __syncthreads() would drain the DMA queue.
        constexpr int D = NB - 2;
        constexpr int PER = A_PW + B_PW;                    // DMA instructions per wave per tile
#pragma unroll
        for (int t = 0; t < D; ++t)
            if (t < nkt) dma(t, t);
        for (int kt = 0; kt < nkt; ++kt) {
            if (kt + D < nkt) dma(kt + D, (kt + D) % NB);
            const int ahead = min(D, nkt - 1 - kt);         // tiles issued after tile kt
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            compute(kt % NB);
        }
        __syncthreads();                                    // all MFMA reads done before the tile staging reuses LDS
    }
    if constexpr (DW) conv_epilogue_dw<TM, TN, WGM, WGN>(p, acc, lds, m0, n0, wm, wn, r, h, tid);
    else conv_epilogue<TM, TN, WGM, WGN>(p, acc, lds, m0, n0, wm, wn, r, h, tid);
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
    // tile variants: 0 = pick by Cout
    int v = a.variant;
    if (v == 0) {
        // measured on MI355X (tools/conv_sweep.py, profiles/): without an SE gate the LDS-DMA kernels win,
        // 8-wave 256-row tiles for wide outputs; gated projections stay on the register-staged kernel
        // High occupancy wins on this chip (many waves of 32-row x 64..96-column sub-tiles); projections
        // that carry an SE gate stay on the register-staged kernel and prefer full-width tiles (the big
        // A operand is then read once).
        if (a.M <= 2048 && a.Cout >= 64) {
            // latency regime (a single frame): a 128/256-row tile would leave most CUs idle; 64-row tiles
            // and 64-wide columns give M/64 * Cout/64 workgroups
            v = (!a.gate && a.zeros) ? 64 : 75;
        } else if (!a.gate && a.zeros) {
            if (a.Cout == 32) v = 59;                 // 256 x  32, 8 waves
            else if (a.Cout % 192 == 0) v = 54;       // 128 x 192, 8 waves of 32 x 96
            else if (a.Cout % 128 == 0) v = 55;       // 128 x 128, 8 waves of 32 x 64
            else if (a.Cout == 64) v = 57;            // 256 x  64
            else if (a.Cout == 224) v = 14;
            else v = 55;
        } else {
            if (a.Cout % 320 == 0) v = 43;            // 128 x 320, 8 waves
            else if (a.Cout % 192 == 0) v = 44;       // 128 x 192, 8 waves
            else if (a.Cout == 224) v = 4;            // 128 x 224
            else if (a.Cout % 128 == 0) v = 1;
            else if (a.Cout % 96 == 0) v = 2;
            else if (a.Cout % 64 == 0) v = 3;
            else v = 5;
        }
    }
    const bool is_dma = (v >= 11 && v <= 39) || (v >= 51 && v <= 69);
    if (is_dma && (a.gate || !a.zeros)) {
        set_error("conv_igemm: the LDS-DMA variants take no SE gate and need the zero line");
        return ISB_ERR_INVALID;
    }
#define ISB_CONV_LAUNCH(TM, TN, WGM, WGN)                                                              \
    do {                                                                                               \
        dim3 g(cdiv(a.M, 32 * TM * WGM), cdiv(a.Cout, 32 * TN * WGN));                                 \
        hipLaunchKernelGGL((conv_igemm_kernel<TM, TN, WGM, WGN>), g, dim3(64 * WGM * WGN), 0, st, a); \
    } while (0)
    switch (v) {
        case 1: ISB_CONV_LAUNCH(2, 2, 2, 2); break;      // 128 x 128, 4 waves
        case 2: ISB_CONV_LAUNCH(1, 3, 4, 1); break;      // 128 x  96
        case 3: ISB_CONV_LAUNCH(2, 1, 2, 2); break;      // 128 x  64
        case 4: ISB_CONV_LAUNCH(1, 7, 4, 1); break;      // 128 x 224
        case 5: ISB_CONV_LAUNCH(2, 1, 4, 1); break;      // 256 x  32
        case 6: ISB_CONV_LAUNCH(2, 2, 4, 2); break;      // 256 x 128, 8 waves
        case 7: ISB_CONV_LAUNCH(4, 2, 2, 4); break;      // 256 x 256, 8 waves
        case 8: ISB_CONV_LAUNCH(2, 3, 4, 2); break;      // 256 x 192, 8 waves
        case 9: ISB_CONV_LAUNCH(2, 1, 4, 2); break;      // 256 x  64, 8 waves
        case 41: ISB_CONV_LAUNCH(1, 6, 4, 1); break;     // 128 x 192: full-width tiles read the A operand once
        case 42: ISB_CONV_LAUNCH(1, 6, 4, 2); break;     // 128 x 384, 8 waves
        case 43: ISB_CONV_LAUNCH(1, 5, 4, 2); break;     // 128 x 320, 8 waves
        case 44: ISB_CONV_LAUNCH(1, 3, 4, 2); break;     // 128 x 192, 8 waves
        case 45: ISB_CONV_LAUNCH(1, 2, 4, 2); break;     // 128 x 128, 8 waves of 32 x 64
        case 47: ISB_CONV_LAUNCH(1, 4, 4, 2); break;     // 128 x 256, 8 waves
        case 48: ISB_CONV_LAUNCH(1, 7, 4, 2); break;     // 128 x 448, 8 waves
        case 71: ISB_CONV_LAUNCH(1, 3, 8, 2); break;     // 256 x 192, 16 waves
        case 75: ISB_CONV_LAUNCH(1, 1, 2, 2); break;     //  64 x  64 (small M)
        case 76: ISB_CONV_LAUNCH(1, 2, 2, 2); break;     //  64 x 128 (small M)
        case 72: ISB_CONV_LAUNCH(1, 3, 4, 4); break;     // 128 x 384, 16 waves
        case 73: ISB_CONV_LAUNCH(1, 5, 4, 4); break;     // 128 x 640, 16 waves
        case 74: ISB_CONV_LAUNCH(1, 2, 8, 2); break;     // 256 x 128, 16 waves
#define ISB_CONV_LAUNCH_DMA(TM, TN, WGM, WGN)                                                              \
    do {                                                                                                   \
        dim3 g(cdiv(a.M, 32 * TM * WGM), cdiv(a.Cout, 32 * TN * WGN));                                     \
        hipLaunchKernelGGL((conv_igemm_dma_kernel<TM, TN, WGM, WGN, 2>), g, dim3(64 * WGM * WGN), 0, st, a); \
    } while (0)
        case 11: ISB_CONV_LAUNCH_DMA(2, 2, 2, 2); break;
        case 12: ISB_CONV_LAUNCH_DMA(1, 3, 4, 1); break;
        case 13: ISB_CONV_LAUNCH_DMA(2, 1, 2, 2); break;
        case 14: ISB_CONV_LAUNCH_DMA(1, 7, 4, 1); break;
        case 15: ISB_CONV_LAUNCH_DMA(2, 1, 4, 1); break;
        case 16: ISB_CONV_LAUNCH_DMA(2, 2, 4, 2); break;
        case 17: ISB_CONV_LAUNCH_DMA(4, 2, 2, 4); break;
        case 18: ISB_CONV_LAUNCH_DMA(2, 3, 4, 2); break;
        case 19: ISB_CONV_LAUNCH_DMA(2, 1, 4, 2); break;
        case 51: ISB_CONV_LAUNCH_DMA(1, 6, 4, 1); break;
        case 52: ISB_CONV_LAUNCH_DMA(1, 6, 4, 2); break;
        case 53: ISB_CONV_LAUNCH_DMA(1, 5, 4, 2); break;
        case 54: ISB_CONV_LAUNCH_DMA(1, 3, 4, 2); break;
        case 55: ISB_CONV_LAUNCH_DMA(1, 2, 4, 2); break;
        case 56: ISB_CONV_LAUNCH_DMA(1, 4, 4, 2); break;
        case 57: ISB_CONV_LAUNCH_DMA(1, 2, 8, 1); break;    // 256 x 64
        case 58: ISB_CONV_LAUNCH_DMA(1, 3, 8, 1); break;    // 256 x 96
        case 59: ISB_CONV_LAUNCH_DMA(1, 1, 8, 1); break;    // 256 x 32
        case 60: ISB_CONV_LAUNCH_DMA(1, 2, 8, 2); break;    // 256 x 128, 16 waves of 32 x 64
        case 64: ISB_CONV_LAUNCH_DMA(1, 1, 2, 2); break;    //  64 x  64: small-M launches (single frames) need many workgroups
        case 65: ISB_CONV_LAUNCH_DMA(1, 2, 2, 2); break;    //  64 x 128
        case 61: ISB_CONV_LAUNCH_DMA(1, 3, 8, 2); break;    // 256 x 192, 16 waves of 32 x 96
        case 62: ISB_CONV_LAUNCH_DMA(1, 2, 4, 4); break;    // 128 x 256, 16 waves
        case 63: ISB_CONV_LAUNCH_DMA(1, 3, 4, 4); break;    // 128 x 384, 16 waves
#undef ISB_CONV_LAUNCH_DMA
#define ISB_CONV_LAUNCH_RING(TM, TN, WGM, WGN)                                                                \
    do {                                                                                                      \
        dim3 g(cdiv(a.M, 32 * TM * WGM), cdiv(a.Cout, 32 * TN * WGN));                                        \
        hipLaunchKernelGGL((conv_igemm_dma_kernel<TM, TN, WGM, WGN, 4>), g, dim3(64 * WGM * WGN), 0, st, a); \
    } while (0)
#define ISB_CONV_LAUNCH_3B(TM, TN, WGM, WGN)                                                                  \
    do {                                                                                                      \
        dim3 g(cdiv(a.M, 32 * TM * WGM), cdiv(a.Cout, 32 * TN * WGN));                                        \
        hipLaunchKernelGGL((conv_igemm_dma_kernel<TM, TN, WGM, WGN, 3>), g, dim3(64 * WGM * WGN), 0, st, a); \
    } while (0)
        case 31: ISB_CONV_LAUNCH_3B(2, 2, 2, 2); break;     // 128 x 128, 4 waves, 3 buffers / 2 tiles in flight
        case 33: ISB_CONV_LAUNCH_3B(2, 1, 2, 2); break;     // 128 x  64
        case 36: ISB_CONV_LAUNCH_3B(2, 2, 4, 2); break;     // 256 x 128, 8 waves
        case 37: ISB_CONV_LAUNCH_3B(4, 2, 2, 4); break;     // 256 x 256, 8 waves
#undef ISB_CONV_LAUNCH_3B
        case 21: ISB_CONV_LAUNCH_RING(2, 2, 2, 2); break;   // 128 x 128, 4 waves, 4-buffer ring
        case 23: ISB_CONV_LAUNCH_RING(2, 1, 2, 2); break;   // 128 x  64
        case 26: ISB_CONV_LAUNCH_RING(2, 2, 4, 2); break;   // 256 x 128, 8 waves
        case 27: ISB_CONV_LAUNCH_RING(4, 2, 2, 4); break;   // 256 x 256, 8 waves
        case 28: ISB_CONV_LAUNCH_RING(4, 1, 2, 4); break;   // 256 x 128 as 2x4 waves of 128x32
#undef ISB_CONV_LAUNCH_RING
        default:
            set_error("conv_igemm: unknown tile variant %d", v);
            return ISB_ERR_INVALID;
    }
#undef ISB_CONV_LAUNCH
    ISB_LAUNCHED("conv_igemm", st);
    return ISB_OK;
}

int launch_conv_expand_dw(const ConvArgs& a, hipStream_t st) {
    const int hw = a.OH * a.OW;
    if (a.KH != 1 || a.stride != 1 || a.gate || a.res || !a.zeros || !a.dw_w || !a.dw_bias || !a.dw_out || !a.pooled || a.OH != a.OW ||
        a.Cin % 32 != 0) {
        set_error("conv_expand_dw: needs an un-gated stride-1 1x1 expand conv with depthwise weights");
        return ISB_ERR_INVALID;
    }
    if (hw == 256 && a.Cout % 64 == 0) {           // one 16x16 sample per tile: 256 x 64, 8 waves
        dim3 g(a.B, a.Cout / 64);
        hipLaunchKernelGGL((conv_igemm_dma_kernel<1, 2, 8, 1, 2, true>), g, dim3(512), 0, st, a);
    } else if (hw == 64 && a.Cout % 128 == 0) {    // one 8x8 sample per tile: 64 x 128, 4 waves
        dim3 g(a.B, a.Cout / 128);
        hipLaunchKernelGGL((conv_igemm_dma_kernel<1, 2, 2, 2, 2, true>), g, dim3(256), 0, st, a);
    } else {
        set_error("conv_expand_dw: unsupported shape hw=%d Cout=%d", hw, a.Cout);
        return ISB_ERR_INVALID;
    }
    ISB_LAUNCHED("conv_expand_dw", st);
    return ISB_OK;
}

// =====================================================================================
// depthwise 3x3 (+ folded-BN bias + SiLU) fused with the squeeze-excite average pool.
// WG = one sample x a slab of CH 8-channel chunks x all output pixels; thread = (chunk, pixel-quad):
// 4 horizontally adjacent outputs share their input columns ((4-1)*S+3 columns x 3 rows of 16-B loads
// instead of 36). The per-(sample, channel) mean is reduced inside the WG in a fixed order (no atomics:
// results do not depend on scheduling) and written straight to pooled[b][c].
// weights tap-major f32 [9][C] (BN scale folded)
// =====================================================================================
template <int S>
__global__ __launch_bounds__(256) void dwconv3x3_pool_kernel(DwArgs p) {
    __shared__ float red[32][129];
    constexpr int NCOL = 3 * S + 3;
    const int nq = (p.OH * p.OW) >> 2;                    // pixel quads per sample
    const int PQ = nq >= 32 ? 32 : nq;                    // quad slots in the WG
    const int CH = 256 / PQ;                              // chunks per WG (8 or 16)
    const int cl = threadIdx.x % CH, pq = threadIdx.x / CH;
    const int b = blockIdx.y;
    const int c = (blockIdx.x * CH + cl) * 8;
    const bool cok = c < p.C;
    float w[9][8], bias[8], psum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) psum[e] = 0.f;
    if (cok) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float4 w0 = *reinterpret_cast<const float4*>(p.w + (size_t)t * p.C + c), w1 = *reinterpret_cast<const float4*>(p.w + (size_t)t * p.C + c + 4);
            w[t][0] = w0.x; w[t][1] = w0.y; w[t][2] = w0.z; w[t][3] = w0.w; w[t][4] = w1.x; w[t][5] = w1.y; w[t][6] = w1.z; w[t][7] = w1.w;
        }
        const float4 s0 = *reinterpret_cast<const float4*>(p.bias + c), s1 = *reinterpret_cast<const float4*>(p.bias + c + 4);
        bias[0] = s0.x; bias[1] = s0.y; bias[2] = s0.z; bias[3] = s0.w; bias[4] = s1.x; bias[5] = s1.y; bias[6] = s1.z; bias[7] = s1.w;
        const int qpr = p.OW >> 2;                        // quads per output row
        for (int q = pq; q < nq; q += PQ) {
            const int oy = q / qpr, ox0 = (q - oy * qpr) * 4;
            float acc[4][8];
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[o][e] = bias[e];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = oy * S - p.pad + ky;
                const bool yok = (unsigned)iy < (unsigned)p.H;
                const uint16_t* rowp = p.in + ((size_t)(b * p.H + (yok ? iy : 0)) * p.W) * p.C + c;
                uint4 v[NCOL];
#pragma unroll
                for (int col = 0; col < NCOL; ++col) {     // branch-free: every load of the row is in flight at once
                    const int ix = ox0 * S - p.pad + col;
                    const bool ok = yok && (unsigned)ix < (unsigned)p.W;
                    const uint4 t = *reinterpret_cast<const uint4*>(rowp + (size_t)(ok ? ix : 0) * p.C);
                    v[col] = ok ? t : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int col = 0; col < NCOL; ++col) {
                    float x[8];
                    x[0] = bf2f_((uint16_t)(v[col].x & 0xffff)); x[1] = bf2f_((uint16_t)(v[col].x >> 16));
                    x[2] = bf2f_((uint16_t)(v[col].y & 0xffff)); x[3] = bf2f_((uint16_t)(v[col].y >> 16));
                    x[4] = bf2f_((uint16_t)(v[col].z & 0xffff)); x[5] = bf2f_((uint16_t)(v[col].z >> 16));
                    x[6] = bf2f_((uint16_t)(v[col].w & 0xffff)); x[7] = bf2f_((uint16_t)(v[col].w >> 16));
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        const int kx = col - o * S;       // tap of output o that reads this column
                        if (kx >= 0 && kx < 3) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) acc[o][e] = fmaf(x[e], w[ky * 3 + kx][e], acc[o][e]);
                        }
                    }
                }
            }
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                uint32_t pk[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint16_t lo = f2bf_(silu_fast(acc[o][2 * e])), hi = f2bf_(silu_fast(acc[o][2 * e + 1]));
                    pk[e] = (uint32_t)lo | ((uint32_t)hi << 16);
                    psum[2 * e] += bf2f_(lo);              // the pool sees the stored (rounded) activations
                    psum[2 * e + 1] += bf2f_(hi);
                }
                *reinterpret_cast<uint4*>(p.out + (((size_t)(b * p.OH + oy) * p.OW + ox0 + o) * p.C + c)) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            }
        }
    }
    if (p.pooled) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[pq][cl * 8 + e] = psum[e];
        __syncthreads();
        if ((int)threadIdx.x < CH * 8) {
            const int cc = blockIdx.x * CH * 8 + threadIdx.x;
            if (cc < p.C) {
                float t = 0.f;
                for (int s2 = 0; s2 < PQ; ++s2) t += red[s2][threadIdx.x];
                p.pooled[(size_t)b * p.C + cc] = t / (float)(p.OH * p.OW);
            }
        }
    }
}

int launch_dwconv3x3(const DwArgs& a, hipStream_t st) {
    if (a.C % 8 != 0 || a.OW % 4 != 0 || ((a.OH * a.OW) >> 2) < 8 || 256 % std::min(32, (a.OH * a.OW) >> 2) != 0) {
        set_error("dwconv3x3: unsupported shape C=%d OH=%d OW=%d", a.C, a.OH, a.OW);
        return ISB_ERR_INVALID;
    }
    const int nq = (a.OH * a.OW) >> 2;
    const int CH = 256 / std::min(32, nq);
    dim3 grid(cdiv(a.C / 8, CH), a.B);
    if (a.stride == 1) hipLaunchKernelGGL(dwconv3x3_pool_kernel<1>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(dwconv3x3_pool_kernel<2>, grid, dim3(256), 0, st, a);
    ISB_LAUNCHED("dwconv3x3_pool", st);
    return ISB_OK;
}

// =====================================================================================
// squeeze-excite FCs in f32 on the vector ALU (tiny GEMMs: latency, not FLOPs, is what matters)
//   se_fc1: mid[b][j]  = silu(b1[j] + sum_c pooled[b][c] * W1[j][c])    wave = (4 j, 8 samples)
//   se_fc2: gate[b][c] = sigmoid(b2[c] + sum_j mid[b][j] * W2T[j][c])   thread = (c, 4 samples)
// fixed summation order -> independent of scheduling and of how the batch is sharded
// =====================================================================================
__global__ __launch_bounds__(256) void se_fc1_kernel(SeFcArgs p) {
    // wave = 4 outputs j x 8 samples: every weight / activation vector fetched from L2 feeds 8 / 4 FMAs
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j0 = (blockIdx.x * 4 + wave) * 4;
    const int b0 = blockIdx.y * 8;
    if (j0 >= p.cse) return;
    float acc[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int s = 0; s < 8; ++s) acc[q][s] = 0.f;
    for (int c = lane * 4; c < p.C; c += 256) {
        float4 wv[4], xv[8];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            wv[q] = *reinterpret_cast<const float4*>(p.w1 + (size_t)min(j0 + q, p.cse - 1) * p.C + c);
#pragma unroll
        for (int s = 0; s < 8; ++s)
            xv[s] = *reinterpret_cast<const float4*>(p.pooled + (size_t)min(b0 + s, p.B - 1) * p.C + c);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int s = 0; s < 8; ++s)
                acc[q][s] = fmaf(xv[s].x, wv[q].x, fmaf(xv[s].y, wv[q].y, fmaf(xv[s].z, wv[q].z, fmaf(xv[s].w, wv[q].w, acc[q][s]))));
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float v = acc[q][s];
#pragma unroll
            for (int sh = 32; sh >= 1; sh >>= 1) v += __shfl_xor(v, sh, 64);
            if (lane == 0 && j0 + q < p.cse && b0 + s < p.B) {
                v += p.b1[j0 + q];
                p.mid[(size_t)(b0 + s) * p.cse + j0 + q] = v / (1.0f + expf(-v));
            }
        }
}

__global__ __launch_bounds__(256) void se_fc2_kernel(SeFcArgs p) {
    __shared__ float mids[4][160];
    const int b0 = blockIdx.y * 4;
    for (int i = threadIdx.x; i < 4 * p.cse; i += 256) {
        const int s = i / p.cse, j = i - s * p.cse;
        mids[s][j] = (b0 + s < p.B) ? p.mid[(size_t)(b0 + s) * p.cse + j] : 0.f;
    }
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= p.C) return;
    float acc[4];
    const float bias = p.b2[c];
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[s] = bias;
#pragma unroll 8
    for (int j = 0; j < p.cse; ++j) {
        const float wv = p.w2t[(size_t)j * p.C + c];
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[s] = fmaf(mids[s][j], wv, acc[s]);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
        if (b0 + s < p.B) p.gate[(size_t)(b0 + s) * p.C + c] = 1.0f / (1.0f + expf(-acc[s]));
}

int launch_se_fcs(const SeFcArgs& a, hipStream_t st) {
    if (a.cse > 160 || a.C % 4 != 0) {
        set_error("se_fcs: unsupported shape cse=%d C=%d", a.cse, a.C);
        return ISB_ERR_INVALID;
    }
    hipLaunchKernelGGL(se_fc1_kernel, dim3(cdiv(a.cse, 16), cdiv(a.B, 8)), dim3(256), 0, st, a);
    hipLaunchKernelGGL(se_fc2_kernel, dim3(cdiv(a.C, 256), cdiv(a.B, 4)), dim3(256), 0, st, a);
    ISB_LAUNCHED("se_fcs", st);
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
