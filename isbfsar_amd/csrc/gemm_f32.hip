// Exact-f32 Linear layer on the gfx950 matrix cores:
//     C[m,n] = act( sum_k (A[m,k] + Aadd[m % add_period, k]) * W[n,k] + bias[n] )
// W is a torch.nn.Linear weight ([N,K] row major), so both operands are K-contiguous.
// v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain (bitwise f32), so this path carries the
// layers whose inputs must not be rounded to bf16: the skeleton MLP (reference
// modules/ar/utils/model.py:164-180), the factorised tuple projections (model.py:75-78), the
// Discriminator (model.py:194-204) and the MetrABS head (modules/hpe/setup/4_create_heads_onnx.py:10-15).
//
// Tile: WG = 4 waves arranged WGM x WGN, each wave owns TM x TN MFMA tiles of 32x32.
// LDS rows are padded to BK+1 floats: lane l reads row (l&31), k = 2*kk + (l>>5); with an odd
// row stride the 32 lanes of a half-wave hit 32 distinct banks (ds_read_b32 banks = dword % 32).
#include "isb_common.h"
#include "kernels.h"

namespace isb {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GEMM_BK = 32;
constexpr int GEMM_LDK = GEMM_BK + 1;

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == GEMM_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == GEMM_ACT_SILU) return v / (1.0f + expf(-v));
    if (act == GEMM_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}

template <int TM, int TN, int WGM, int WGN>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmF32Args p) {
    constexpr int BM = 32 * TM * WGM;
    constexpr int BN = 32 * TN * WGN;
    static_assert(WGM * WGN == 4, "4 waves per workgroup");
    __shared__ float lds[(BM + BN) * GEMM_LDK];
    float* As = lds;
    float* Bs = lds + BM * GEMM_LDK;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    // staging map: thread -> (row = tid/32 + 8*pass, k = tid%32): 128-B contiguous row segments
    constexpr int A_PASSES = BM / 8;
    constexpr int B_PASSES = BN / 8;
    const int sk = tid & 31;
    const int sr = tid >> 5;
    float ra[A_PASSES];
    float rb[B_PASSES];

    auto gload = [&](int kt) {
        const int k = kt * GEMM_BK + sk;
        const bool kin = k < p.K;
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            const int m = m0 + sr + 8 * i;
            float v = 0.f;
            if (kin && m < p.M) {
                v = p.A[(size_t)m * p.lda + k];
                for (int s = 1; s < p.a_parts; ++s) v += p.A[(size_t)s * p.a_part_stride + (size_t)m * p.lda + k];
                if (p.a_bias) v += p.a_bias[k];
                v = apply_act(v, p.a_act);
                if (p.Aadd) v += p.Aadd[(size_t)(m % p.add_period) * p.ldadd + k];
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) {
            const int n = n0 + sr + 8 * i;
            rb[i] = (kin && n < p.N) ? p.W[(size_t)n * p.ldw + k] : 0.f;
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
        for (int i = 0; i < A_PASSES; ++i) As[(sr + 8 * i) * GEMM_LDK + sk] = ra[i];
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) Bs[(sr + 8 * i) * GEMM_LDK + sk] = rb[i];
        __syncthreads();
        if (kt + 1 < nkt) gload(kt + 1);
#pragma unroll
        for (int kk = 0; kk < GEMM_BK / 2; ++kk) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = As[((wm * TM + i) * 32 + lr) * GEMM_LDK + 2 * kk + lh];
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[j] = Bs[((wn * TN + j) * 32 + lr) * GEMM_LDK + 2 * kk + lh];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
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
    if (a.N <= 32) {
        dim3 grid(cdiv(a.M, 256), cdiv(a.N, 32), z);
        hipLaunchKernelGGL((gemm_f32_kernel<2, 1, 4, 1>), grid, dim3(256), 0, st, a);
    } else if (a.N <= 64) {
        dim3 grid(cdiv(a.M, 128), cdiv(a.N, 64), z);
        hipLaunchKernelGGL((gemm_f32_kernel<2, 1, 2, 2>), grid, dim3(256), 0, st, a);
    } else if ((long)cdiv(a.M, 128) * cdiv(a.N, 128) * z < 256) {
        // fewer 128 x 128 tiles than CUs (AR layers at a few thousand rows): 64 x 64 tiles
        dim3 grid(cdiv(a.M, 64), cdiv(a.N, 64), z);
        hipLaunchKernelGGL((gemm_f32_kernel<1, 1, 2, 2>), grid, dim3(256), 0, st, a);
    } else {
        dim3 grid(cdiv(a.M, 128), cdiv(a.N, 128), z);
        hipLaunchKernelGGL((gemm_f32_kernel<2, 2, 2, 2>), grid, dim3(256), 0, st, a);
    }
    ISB_LAUNCHED("gemm_f32", st);
    return ISB_OK;
}

}  // namespace isb
