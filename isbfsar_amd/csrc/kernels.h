// Kernel argument blocks + host launchers shared between the .hip translation units and the
// C-ABI glue (ar_api.cpp / hpe_api.cpp).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace isb {

// ---------------------------------------------------------------- gemm_f32.hip
enum { GEMM_ACT_NONE = 0, GEMM_ACT_RELU = 1 };
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
};
int launch_gemm_f32(const GemmF32Args& a, hipStream_t st);

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
    float* ub;              // out (support only): [n_items][Tp] = |kc_j| * qnorm_bound, or null
    float kscale;           // folded into K before rounding (query: log2(e)/sqrt(128); support: 1)
    float qnorm_bound;      // upper bound of |kq'| (support side only)
    int n_items, L, T, NT;
};
int launch_ar_tuples(const ArTupleArgs& a, hipStream_t st);

struct ArStatsArgs {
    const uint16_t* KqF;    // [B][NT][8][64][8]
    const uint16_t* KqF_lo;
    const uint16_t* KcF;    // [n][NT][8][64][8]
    const uint16_t* KcF_lo;
    const float* ub;        // [n][Tp]
    float* lse2;            // out [B][n][Tp]: log2 sum_i exp2(s'[i,j])
    int B, n, T, NT;
    int x3;
    int online;             // 1: running-max variant (bound too loose to exclude underflow)
};
int launch_ar_stats(const ArStatsArgs& a, hipStream_t st);

struct ArProtoArgs {
    const uint16_t* KqF;
    const uint16_t* KqF_lo;
    const uint16_t* KcF;
    const uint16_t* KcF_lo;
    const uint16_t* VtF;
    const uint16_t* VtF_lo;
    const float* lse2;      // [B][n][Tp]
    const float* proj;      // [B*L,512] query projections (Av at +256, Bv at +384)
    const float* bv;        // [128]
    const int16_t* tup;     // [Tp][2]
    const int32_t* chosen;  // null: all classes -> part; else only class chosen[b] -> diff
    float* part;            // out [B][n][NT] partial sum of squares
    float* diff;            // out [B][T][128] (chosen mode)
    int B, n, L, T, NT;
    int x3;
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
    const float* h1;        // [B][256] relu(fc1)
    const float* w2;        // [64][256]
    const float* b2;        // [64]
    const float* w3;        // [64]
    const float* b3;        // [1]
    float* is_true;         // [B]
    int B;
};
int launch_ar_disc_tail(const ArDiscTailArgs& a, hipStream_t st);

}  // namespace isb
