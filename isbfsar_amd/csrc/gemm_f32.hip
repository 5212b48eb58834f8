// Exact-f32 Linear layer on the gfx950 matrix cores:
//     C[m,n] = act( sum_k (A[m,k] + Aadd[m % add_period, k]) * W[n,k] + bias[n] )
// W is a torch.nn.Linear weight ([N,K] row major), so both operands are K-contiguous.
// v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain (bitwise f32), so this path carries the
// layers whose inputs must not be rounded to bf16: the skeleton MLP (reference
// modules/ar/utils/model.py:164-180), the factorised tuple projections (model.py:75-78), the
// Discriminator (model.py:194-204) and the MetrABS head (modules/hpe/setup/4_create_heads_onnx.py:10-15).
//
// Tile: WG = 4 waves arranged WGM x WGN, each wave owns TM x TN MFMA tiles of 32x32; k-tiles of 32.
// Staging: thread = (row, 16-byte chunk of 4 consecutive k): one dwordx4 (VEC 4), two dwordx2 (VEC 2: rows that are only
// 8-byte aligned, K = 366) or four dword loads per chunk, transforms applied per component, one ds_write_b128 per chunk.
// LDS rows are 36 floats apart: 36 * row mod 64 runs through all multiples of 4, so the 16 lanes that a ds_read_b128
// serves per cycle (rows of one lane group) touch 64 different banks. The two lane halves take k 0..15 and 16..31 of a k-tile
// (four ds_read_b128 per 32-row operand block instead of sixteen ds_read_b32): the MFMA chain of an output element adds the
// products of k = (0, 16), (1, 17), ... (15, 31) of each k-tile in that order -- a fixed order that does not depend on the
// batch or the tiling.
#include "isb_common.h"
#include "kernels.h"

namespace isb {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GEMM_BK = 32;
constexpr int GEMM_LDK = 36;

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == GEMM_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == GEMM_ACT_SILU) return v / (1.0f + expf(-v));
    if (act == GEMM_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}

// 4 consecutive floats at p[0..3] of which the first `nv` exist (0..4); VEC = guaranteed alignment in floats
template <int VEC>
__device__ __forceinline__ float4 load4(const float* p, int nv) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nv >= 4) {
        if constexpr (VEC == 4) return *reinterpret_cast<const float4*>(p);
        if constexpr (VEC == 2) {
            const float2 a = *reinterpret_cast<const float2*>(p), b = *reinterpret_cast<const float2*>(p + 2);
            return make_float4(a.x, a.y, b.x, b.y);
        }
        return make_float4(p[0], p[1], p[2], p[3]);
    }
    if (nv > 0) v.x = p[0];
    if (nv > 1) v.y = p[1];
    if (nv > 2) v.z = p[2];
    return v;
}

template <int TM, int TN, int WGM, int WGN, int VEC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmF32Args p) {
    constexpr int BM = 32 * TM * WGM;
    constexpr int BN = 32 * TN * WGN;
    static_assert(WGM * WGN == 4, "4 waves per workgroup");
    __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * GEMM_LDK];
    float* As = lds;
    float* Bs = lds + BM * GEMM_LDK;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    // staging map: thread -> (row = tid / 8 + 32 * pass, chunk = tid % 8): 128-B contiguous row segments per 8 threads
    constexpr int A_PASSES = BM / 32;
    constexpr int B_PASSES = BN / 32;
    const int sc = tid & 7;
    const int sr = tid >> 3;
    float4 ra[A_PASSES];
    float4 rb[B_PASSES];
    const bool plain_a = p.a_parts <= 1 && !p.a_bias && p.a_act == GEMM_ACT_NONE && !p.Aadd;

    auto gload = [&](int kt) {
        const int k = kt * GEMM_BK + sc * 4;
        const int nv = min(max(p.K - k, 0), 4);
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            const int m = m0 + sr + 32 * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (nv > 0 && m < p.M) {
                v = load4<VEC>(p.A + (size_t)m * p.lda + k, nv);
                if (!plain_a) {
                    for (int s = 1; s < p.a_parts; ++s) {
                        const float4 u = load4<VEC>(p.A + (size_t)s * p.a_part_stride + (size_t)m * p.lda + k, nv);
                        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
                    }
                    if (p.a_bias) {
                        const float4 u = load4<1>(p.a_bias + k, nv);
                        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
                    }
                    v.x = apply_act(v.x, p.a_act); v.y = apply_act(v.y, p.a_act); v.z = apply_act(v.z, p.a_act); v.w = apply_act(v.w, p.a_act);
                    if (p.Aadd) {
                        const float4 u = load4<1>(p.Aadd + (size_t)(m % p.add_period) * p.ldadd + k, nv);
                        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
                    }
                    // components past K stay out of the sum whatever the transform made of them
                    if (nv < 4) v.w = 0.f;
                    if (nv < 3) v.z = 0.f;
                    if (nv < 2) v.y = 0.f;
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) {
            const int n = n0 + sr + 32 * i;
            rb[i] = (nv > 0 && n < p.N) ? load4<VEC>(p.W + (size_t)n * p.ldw + k, nv) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nkt_all = (p.K + GEMM_BK - 1) / GEMM_BK;
    int kt_begin = 0, nkt = nkt_all;
    const bool split = p.splits > 1;
    if (split) {
        const int per = (nkt_all + p.splits - 1) / p.splits;
        kt_begin = blockIdx.z * per;
        nkt = min(nkt_all, kt_begin + per);
    }
    float* Cout = split ? p.C + (size_t)blockIdx.z * p.split_stride : p.C;
    if (kt_begin < nkt) gload(kt_begin);
    const int lr = lane & 31;
    const int lh = lane >> 5;
    for (int kt = kt_begin; kt < nkt; ++kt) {
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) *reinterpret_cast<float4*>(As + (sr + 32 * i) * GEMM_LDK + sc * 4) = ra[i];
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) *reinterpret_cast<float4*>(Bs + (sr + 32 * i) * GEMM_LDK + sc * 4) = rb[i];
        __syncthreads();
        if (kt + 1 < nkt) gload(kt + 1);
        float a[TM][16], b[TN][16];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(As + ((wm * TM + i) * 32 + lr) * GEMM_LDK + 16 * lh + 4 * q);
                a[i][4 * q] = v.x; a[i][4 * q + 1] = v.y; a[i][4 * q + 2] = v.z; a[i][4 * q + 3] = v.w;
            }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(Bs + ((wn * TN + j) * 32 + lr) * GEMM_LDK + 16 * lh + 4 * q);
                b[j][4 * q] = v.x; b[j][4 * q + 1] = v.y; b[j][4 * q + 2] = v.z; b[j][4 * q + 3] = v.w;
            }
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
        __syncthreads();
    }

    // epilogue: D tile col = lane&31 (n), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (m)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + lr;
        if (n >= p.N) continue;
        const float bv = (p.bias && !split) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + (wm * TM + i) * 32 + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                if (m < p.M) {
                    float v = acc[i][j][r] + bv;
                    if (!split) v = apply_act(v, p.act);
                    Cout[(size_t)m * p.ldc + n] = v;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void reduce_parts_kernel(const float* parts, int n_parts, size_t stride, const float* bias, int act,
                                                           float* out, int M, int N) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)M * N) return;
    float v = bias ? bias[i % N] : 0.f;
    for (int s = 0; s < n_parts; ++s) v += parts[(size_t)s * stride + i];
    out[i] = apply_act(v, act);
}

int launch_reduce_parts(const float* parts, int n_parts, size_t stride, const float* bias, int act, float* out, int M, int N,
                        hipStream_t st) {
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)cdivz((size_t)M * N, 256)), dim3(256), 0, st, parts, n_parts, stride, bias,
                       act, out, M, N);
    ISB_LAUNCHED("reduce_parts", st);
    return ISB_OK;
}

int launch_gemm_f32(const GemmF32Args& a, hipStream_t st) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) {
        set_error("gemm_f32: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
        return ISB_ERR_INVALID;
    }
    const int z = a.splits > 1 ? a.splits : 1;
    // alignment of every row start and of the k origin of a chunk (k is a multiple of 4), in floats
    auto al = [](const void* ptr, size_t ld) {
        const uintptr_t u = (uintptr_t)ptr;
        return (u % 16 == 0 && ld % 4 == 0) ? 4 : (u % 8 == 0 && ld % 2 == 0) ? 2 : 1;
    };
    int vec = std::min(al(a.A, (size_t)a.lda), al(a.W, (size_t)a.ldw));
    if (a.a_parts > 1) vec = std::min(vec, al(a.A + a.a_part_stride, (size_t)a.lda));
#define ISB_GEMM_GO(TM, TN, WGM, WGN)                                                                    \
    do {                                                                                                 \
        dim3 grid(cdiv(a.M, 32 * TM * WGM), cdiv(a.N, 32 * TN * WGN), z);                                \
        if (vec == 4) hipLaunchKernelGGL((gemm_f32_kernel<TM, TN, WGM, WGN, 4>), grid, dim3(256), 0, st, a);      \
        else if (vec == 2) hipLaunchKernelGGL((gemm_f32_kernel<TM, TN, WGM, WGN, 2>), grid, dim3(256), 0, st, a); \
        else hipLaunchKernelGGL((gemm_f32_kernel<TM, TN, WGM, WGN, 1>), grid, dim3(256), 0, st, a);               \
    } while (0)
    if (a.N <= 32) ISB_GEMM_GO(2, 1, 4, 1);
    else if (a.N <= 64) ISB_GEMM_GO(2, 1, 2, 2);
    else if ((long)cdiv(a.M, 128) * cdiv(a.N, 128) * z < 256) ISB_GEMM_GO(1, 1, 2, 2);   // fewer 128 x 128 tiles than CUs (AR layers at a few thousand rows): 64 x 64 tiles
    else if (a.N % 128 != 0 && a.N % 96 == 0) ISB_GEMM_GO(1, 3, 4, 1);                   // 128 x 96 tiles: no padded columns (the pose head's 288 outputs)
    else ISB_GEMM_GO(2, 2, 2, 2);
#undef ISB_GEMM_GO
    ISB_LAUNCHED("gemm_f32", st);
    return ISB_OK;
}

}  // namespace isb
