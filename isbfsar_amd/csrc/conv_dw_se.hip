// The bandwidth- and latency-bound layers of the pose backbone beside the tile GEMMs:
//   dwconv3x3   : depthwise 3x3 (+bias, SiLU) fused with the squeeze-excite average pool, HBM-bound
//   se_fc1/fc2  : the two squeeze-excite FCs in f32 on the vector ALU
//   stem        : conv3x3/s2 3->32 in f32 on the f32 crop
//   splitk_reduce, f32_to_bf16_rows
// The backbone is what the reference runs as `bbone1.engine` (utils/params.py:29, hpe.py:103);
// layer semantics follow the public efficientnetv2-l definition (isbfsar_amd/effnetv2.py).
#include "conv_common.h"
#include "dw_mm.h"

namespace isb {

// split-K reduction: out[m][n] = bf16( sum_s part[s][m][n] (in split order) + bias[n] (+ res[m][n]) )
__global__ __launch_bounds__(256) void splitk_reduce_kernel(ConvArgs p) {
    if (p.f16) fp16_ovfl_on();
    const size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x;       // one thread = 4 consecutive channels
    const size_t total4 = (size_t)p.M * p.Cout / 4;
    if (i4 >= total4) return;
    const size_t e = i4 * 4;
    const int n = (int)(e % p.Cout);
    float4 v = *reinterpret_cast<const float4*>(p.part + e);
    for (int s = 1; s < p.splits; ++s) {
        const float4 u = *reinterpret_cast<const float4*>(p.part + (size_t)s * p.M * p.Cout + e);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    const float4 bs = *reinterpret_cast<const float4*>(p.bias + n);
    v.x += bs.x; v.y += bs.y; v.z += bs.z; v.w += bs.w;
    if (p.act) { v.x = silu_fast(v.x); v.y = silu_fast(v.y); v.z = silu_fast(v.z); v.w = silu_fast(v.w); }
    if (p.res) {
        const uint2 rr = *reinterpret_cast<const uint2*>(p.res + e);
        if (p.f16) { v.x += T16<true>::lo(rr.x); v.y += T16<true>::hi(rr.x); v.z += T16<true>::lo(rr.y); v.w += T16<true>::hi(rr.y); }
        else { v.x += T16<false>::lo(rr.x); v.y += T16<false>::hi(rr.x); v.z += T16<false>::lo(rr.y); v.w += T16<false>::hi(rr.y); }
    }
    uint2 pk;
    if (p.f16) { pk.x = T16<true>::pack2(v.x, v.y); pk.y = T16<true>::pack2(v.z, v.w); }
    else {
        pk.x = (uint32_t)f2bf_(v.x) | ((uint32_t)f2bf_(v.y) << 16);
        pk.y = (uint32_t)f2bf_(v.z) | ((uint32_t)f2bf_(v.w) << 16);
    }
    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.out) + e) = pk;
}

int launch_splitk_reduce(const ConvArgs& a, hipStream_t st) {
    if (a.splits < 2 || !a.part || a.Cout % 4 != 0 || a.out_f32) {
        set_error("splitk_reduce: needs splits >= 2, a partial buffer and bf16 output");
        return ISB_ERR_INVALID;
    }
    const size_t total4 = (size_t)a.M * a.Cout / 4;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdivz(total4, 256)), dim3(256), 0, st, a);
    ISB_LAUNCHED("splitk_reduce", st);
    return ISB_OK;
}

// =====================================================================================
// depthwise 3x3 (+ folded-BN bias + SiLU) fused with the squeeze-excite average pool.
// WG = one sample x a slab of CH 8-channel chunks x all output pixels; thread = (chunk, pixel-quad):
// 4 horizontally adjacent outputs share their input columns ((4-1)*S+3 columns x 3 rows of 16-B loads
// instead of 36). The per-(sample, channel) mean is reduced inside the WG in a fixed order (no atomics:
// results do not depend on scheduling) and written straight to pooled[b][c].
// weights tap-major bf16 [9][C] (BN scale folded); the 9-tap sums run on v_dot2c_f32_bf16 with the other half
// of the weight pair zeroed = exact f32 FMAs fed by the packed activations
// =====================================================================================
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// acc += x.lo * w.lo + x.hi * w.hi on bf16 pairs (v_dot2c_f32_bf16). With one half of w zero this is the exact
// f32 FMA of ONE channel straight from the packed activations: no bf16 -> f32 unpacking, no operand shuffles.
__device__ __forceinline__ float dot2_bf16(uint32_t x, uint32_t w, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, x), __builtin_bit_cast(bf16x2_t, w), acc, false);
}

// sum over the 16 lanes of a DPP row (row_ror 8, 4, 2, 1), the total in every lane
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}

// IN_F16 / OUT_F16: input + taps / output in fp16 instead of bf16 (DwArgs.in_f16 / out_f16)
template <int S, bool FC1 = false, bool IN_F16 = false, bool OUT_F16 = false>
__global__ __launch_bounds__(256) void dwconv3x3_pool_kernel(DwArgs p) {
    T16<OUT_F16>::enter();
    __shared__ float red[32][129];
    __shared__ __attribute__((aligned(16))) float pmean[FC1 ? 128 : 4];
    constexpr int NCOL = 3 * S + 3;
    const int nq = (p.OH * p.OW) >> 2;                    // pixel quads per sample
    const int PQ = nq >= 32 ? 32 : nq;                    // quad slots in the WG
    const int CH = 256 / PQ;                              // chunks per WG (8 or 16)
    const int cl = threadIdx.x % CH, pq = threadIdx.x / CH;
    const int b = blockIdx.y;
    const int c = (blockIdx.x * CH + cl) * 8;
    const bool cok = c < p.C;
    // FC1: the slab's squeeze-excite weights do not depend on anything computed here -- request them first, they
    // arrive while the taps run (one frame = one workgroup per CU: the 80 registers cost no occupancy)
    const int sub = threadIdx.x & 31, grp = threadIdx.x >> 5;
    bool fc_ok = false;
    float4 wv[FC1 ? 20 : 1];
    if constexpr (FC1) {
        const int c0 = blockIdx.x * CH * 8, nch = min(CH * 8, p.C - c0);
        fc_ok = sub * 4 < nch;
#pragma unroll
        for (int q = 0; q < 20; ++q) {
            const int j = grp + 8 * q;
            wv[q] = (fc_ok && j < p.cse) ? *reinterpret_cast<const float4*>(p.se_w1 + (size_t)j * p.C + c0 + sub * 4)
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    uint32_t wlo[9][4], whi[9][4];                        // tap weights of the even / odd channel of each pair
    float bias[8], psum[8];
    // bf16 (1, 0) and (0, 1) in registers: as a literal 0x3f800000 becomes the INLINE constant 1.0, which a packed
    // bf16 operand reads as (1, 0) -- the pool would sum the wrong channel
    uint32_t one_lo, one_hi;
    if constexpr (OUT_F16) asm volatile("v_mov_b32 %0, 0x3c00\n\tv_lshlrev_b32 %1, 16, %0" : "=v"(one_lo), "=v"(one_hi));
    else asm volatile("v_mov_b32 %0, 0x3f80\n\tv_lshlrev_b32 %1, 16, %0" : "=v"(one_lo), "=v"(one_hi));
#pragma unroll
    for (int e = 0; e < 8; ++e) psum[e] = 0.f;
    if (cok) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const uint4 wv = *reinterpret_cast<const uint4*>(p.w + (size_t)t * p.C + c);
            const uint32_t wp[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { wlo[t][e] = wp[e] & 0xffffu; whi[t][e] = wp[e] & 0xffff0000u; }
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
                    const uint32_t x[4] = {v[col].x, v[col].y, v[col].z, v[col].w};
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        const int kx = col - o * S;       // tap of output o that reads this column
                        if (kx >= 0 && kx < 3) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                acc[o][2 * e] = T16<IN_F16>::dot2(x[e], wlo[ky * 3 + kx][e], acc[o][2 * e]);
                                acc[o][2 * e + 1] = T16<IN_F16>::dot2(x[e], whi[ky * 3 + kx][e], acc[o][2 * e + 1]);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                uint32_t pk[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint16_t lo = T16<OUT_F16>::from_f32(silu_fast(acc[o][2 * e])), hi = T16<OUT_F16>::from_f32(silu_fast(acc[o][2 * e + 1]));
                    pk[e] = (uint32_t)lo | ((uint32_t)hi << 16);
                    // the pool sees the stored (rounded) activations: x * 1.0 + psum, one instruction per channel
                    psum[2 * e] = T16<OUT_F16>::dot2(pk[e], one_lo, psum[2 * e]);
                    psum[2 * e + 1] = T16<OUT_F16>::dot2(pk[e], one_hi, psum[2 * e + 1]);
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
            float mean = 0.f;
            if (cc < p.C) {
                float t = 0.f;
                for (int s2 = 0; s2 < PQ; ++s2) t += red[s2][threadIdx.x];
                mean = t / (float)(p.OH * p.OW);
                p.pooled[(size_t)b * p.C + cc] = mean;
            }
            if constexpr (FC1) pmean[threadIdx.x] = mean;
        }
        if constexpr (FC1) {
            // this slab's share of squeeze-excite FC1: 32 lanes per output row j hold the row's slab (requested at
            // kernel start), multiply with the pooled means and meet in a butterfly. 8 groups x 20 rows: cse <= 160.
            __syncthreads();
            const float4 pv = fc_ok ? *reinterpret_cast<const float4*>(&pmean[sub * 4]) : make_float4(0.f, 0.f, 0.f, 0.f);
            float racc[20], rup[20];
#pragma unroll
            for (int q = 0; q < 20; ++q)      // 16-lane sums on the vector ALU (DPP row rotations), no LDS round trips
                racc[q] = row16_sum(fmaf(pv.w, wv[q].w, fmaf(pv.z, wv[q].z, fmaf(pv.y, wv[q].y, pv.x * wv[q].x))));
#pragma unroll
            for (int q = 0; q < 20; ++q) rup[q] = __shfl_xor(racc[q], 16, 64);     // the other half-row: 20 permutes in flight
#pragma unroll
            for (int q = 0; q < 20; ++q) {
                const int j = grp + 8 * q;
                if (sub == 0 && j < p.cse) p.se_part[((size_t)blockIdx.x * p.B + b) * p.cse + j] = racc[q] + rup[q];
            }
        }
    }
}

// Whole-sample maps of 8 x 8 and 16 x 16 pixels at stride 1 (stages 3-6: 54 of the 61 depthwise launches). On these the kernel
// above moved its bytes at 3.5 TB/s (5.0-5.5 on the two stride-2 launches): a workgroup's working set is 16-32 KiB (one sample x
// 128 / 64 channels), but its threads request it 4.5 times over through the texture path (six 16-byte columns x three rows per
// four outputs), and only a few hundred bytes per thread are ever in flight. Here the slab goes to LDS ONCE, as fully coalesced
// 256- / 128-byte pixel rows (every thread has its 4 / 8 16-byte loads in flight at the start), inside a ring of zero pixels --
// TF-SAME padding becomes data -- and the taps read LDS (conflict-free: the 16 lanes of a ds_read_b128 group cover whole
// pixels). Same thread <-> (pixel quad, channel chunk) map, same tap order, same accumulators and pool order as
// dwconv3x3_pool_kernel<1>: bit-identical (tested). 8 x 8: 3.5 -> 4.7 TB/s.
template <bool F16, int HW, bool FC1 = false>
__global__ __launch_bounds__(256) void dwconv3x3_map_kernel(DwArgs p) {
    T16<F16>::enter();
    constexpr int NQ = HW * HW / 4;                        // pixel quads per sample
    constexpr int PQ = NQ >= 32 ? 32 : NQ;                 // quad slots in the workgroup (16 on 8 x 8 maps)
    constexpr int CH = 256 / PQ;                           // 8-channel chunks per workgroup: 16 (128 channels) / 8 (64 channels)
    constexpr int TW = HW + 2;                             // tile width with the zero ring
    constexpr int NLD = HW * HW * CH / 256;                // 16-byte loads per thread: 4 / 8
    constexpr int QPR = HW / 4;                            // quads per row
    __shared__ __attribute__((aligned(16))) uint4 tile[TW * TW * CH];     // [y + 1][x + 1][chunk]
    __shared__ float red[PQ][CH * 8 + 1];
    __shared__ __attribute__((aligned(16))) float pmean[FC1 ? 128 : 4];
    const int tid = threadIdx.x;
    const int cl = tid % CH, pq = tid / CH;
    const int b = blockIdx.y, c0 = blockIdx.x * (CH * 8);
    const int c = c0 + cl * 8;
    const bool cok = c < p.C;
    // the slab: pixel px = idx / CH, chunk idx % CH -- CH consecutive threads fetch one pixel's contiguous bytes
    uint4 ld[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int idx = tid + 256 * k, px = idx / CH, ch = idx % CH;
        ld[k] = (c0 + ch * 8 < p.C) ? *reinterpret_cast<const uint4*>(p.in + ((size_t)b * (HW * HW) + px) * p.C + c0 + ch * 8) : make_uint4(0, 0, 0, 0);
    }
    for (int i = tid; i < (4 * HW + 4) * CH; i += 256) {   // the ring of zero pixels: top row, bottom row, left / right columns
        const int q = i / CH, ch = i % CH;
        const int y = q < TW ? 0 : (q < 2 * TW ? TW - 1 : 1 + ((q - 2 * TW) >> 1)), x = q < TW ? q : (q < 2 * TW ? q - TW : ((q - 2 * TW) & 1) * (TW - 1));
        tile[(y * TW + x) * CH + ch] = make_uint4(0, 0, 0, 0);
    }
    uint32_t wlo[9][4], whi[9][4];
    float bias[8], psum[8];
    uint32_t one_lo, one_hi;
    if constexpr (F16) asm volatile("v_mov_b32 %0, 0x3c00\n\tv_lshlrev_b32 %1, 16, %0" : "=v"(one_lo), "=v"(one_hi));
    else asm volatile("v_mov_b32 %0, 0x3f80\n\tv_lshlrev_b32 %1, 16, %0" : "=v"(one_lo), "=v"(one_hi));
#pragma unroll
    for (int e = 0; e < 8; ++e) psum[e] = 0.f;
    if (cok) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const uint4 wv = *reinterpret_cast<const uint4*>(p.w + (size_t)t * p.C + c);
            const uint32_t wp[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { wlo[t][e] = wp[e] & 0xffffu; whi[t][e] = wp[e] & 0xffff0000u; }
        }
        const float4 s0 = *reinterpret_cast<const float4*>(p.bias + c), s1 = *reinterpret_cast<const float4*>(p.bias + c + 4);
        bias[0] = s0.x; bias[1] = s0.y; bias[2] = s0.z; bias[3] = s0.w; bias[4] = s1.x; bias[5] = s1.y; bias[6] = s1.z; bias[7] = s1.w;
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int idx = tid + 256 * k, px = idx / CH, ch = idx % CH;
        tile[((px / HW + 1) * TW + (px % HW) + 1) * CH + ch] = ld[k];
    }
    __syncthreads();
    if (cok) {
        for (int q = pq; q < NQ; q += PQ) {
            const int oy = q / QPR, ox0 = (q - oy * QPR) * 4;
            float acc[4][8];
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[o][e] = bias[e];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                uint4 v[6];
#pragma unroll
                for (int col = 0; col < 6; ++col) v[col] = tile[((oy + ky) * TW + ox0 + col) * CH + cl];
#pragma unroll
                for (int col = 0; col < 6; ++col) {
                    const uint32_t x[4] = {v[col].x, v[col].y, v[col].z, v[col].w};
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        const int kx = col - o;
                        if (kx >= 0 && kx < 3) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                acc[o][2 * e] = T16<F16>::dot2(x[e], wlo[ky * 3 + kx][e], acc[o][2 * e]);
                                acc[o][2 * e + 1] = T16<F16>::dot2(x[e], whi[ky * 3 + kx][e], acc[o][2 * e + 1]);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                uint32_t pk[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint16_t lo = T16<F16>::from_f32(silu_fast(acc[o][2 * e])), hi = T16<F16>::from_f32(silu_fast(acc[o][2 * e + 1]));
                    pk[e] = (uint32_t)lo | ((uint32_t)hi << 16);
                    psum[2 * e] = T16<F16>::dot2(pk[e], one_lo, psum[2 * e]);
                    psum[2 * e + 1] = T16<F16>::dot2(pk[e], one_hi, psum[2 * e + 1]);
                }
                *reinterpret_cast<uint4*>(p.out + (((size_t)b * (HW * HW) + oy * HW + ox0 + o) * p.C + c)) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            }
        }
    }
    if (p.pooled) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[pq][cl * 8 + e] = psum[e];
        // FC1: this slab's rows of the squeeze-excite weights, requested before the pool sums meet (same thread <-> (row, channel)
        // map, same sums as dwconv3x3_pool_kernel's: bit-identical partials)
        const int sub = tid & 31, grp = tid >> 5;
        bool fc_ok = false;
        float4 wv[FC1 ? 20 : 1];
        if constexpr (FC1) {
            fc_ok = sub * 4 < min(CH * 8, p.C - c0);
#pragma unroll
            for (int q = 0; q < 20; ++q) {
                const int j = grp + 8 * q;
                wv[q] = (fc_ok && j < p.cse) ? *reinterpret_cast<const float4*>(p.se_w1 + (size_t)j * p.C + c0 + sub * 4)
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        __syncthreads();
        if (tid < CH * 8) {
            const int cc = c0 + tid;
            float mean = 0.f;
            if (cc < p.C) {
                float t = 0.f;
                for (int s2 = 0; s2 < PQ; ++s2) t += red[s2][tid];
                mean = t / (float)(HW * HW);
                p.pooled[(size_t)b * p.C + cc] = mean;
            }
            if constexpr (FC1) pmean[tid] = mean;
        }
        if constexpr (FC1) {
            __syncthreads();
            const float4 pv = fc_ok ? *reinterpret_cast<const float4*>(&pmean[sub * 4]) : make_float4(0.f, 0.f, 0.f, 0.f);
            float racc[20], rup[20];
#pragma unroll
            for (int q = 0; q < 20; ++q)
                racc[q] = row16_sum(fmaf(pv.w, wv[q].w, fmaf(pv.z, wv[q].z, fmaf(pv.y, wv[q].y, pv.x * wv[q].x))));
#pragma unroll
            for (int q = 0; q < 20; ++q) rup[q] = __shfl_xor(racc[q], 16, 64);
#pragma unroll
            for (int q = 0; q < 20; ++q) {
                const int j = grp + 8 * q;
                if (sub == 0 && j < p.cse) p.se_part[((size_t)blockIdx.x * p.B + b) * p.cse + j] = racc[q] + rup[q];
            }
        }
    }
}


// Stride-1 depthwise 3x3 on whole 8 x 8 / 16 x 16 maps with the TAPS ON THE MATRIX PIPE (dw_mm.h; round 5; DwArgs.general == 3): the
// stand-alone partner of the fused 8 x 8 front (same arithmetic, same bits), selected for blocks that have one when the batch is too
// small for it. Staging as in dwconv3x3_map_kernel (the slab goes to LDS once, inside a ring of zero pixels), but a
// wave owns CPW channels of the slab for ALL pixels: per 8-channel group and 32-pixel tile, three 16-byte fragment reads + three
// v_mfma_f32_16x16x32 replace 288 v_dot2 (per lane: 4 outputs instead of 32 per pass, 7 vector instructions per output instead of
// 17). The 16-bit results are staged in LDS and leave as full pixel rows; pooled means: per lane over its tiles in order, then the
// 32 (pixel pair, pixel) lanes in dw_mm.h's butterfly.
// LDS images: in  [TW x TW pixels][RB bytes], 16-byte chunk slot = chunk ^ f(y, x) with f chosen so that the sixteen lanes a
//                 ds_read_b128 serves together (pixels 2n + j of two rows) fall into sixteen different slots;
//             out [HW x HW pixels][RB bytes], chunk slot = chunk ^ (pixel >> 1).
// (launch bounds: at least two waves per SIMD = at most 256 registers, with which the compiler keeps the MFMA results in ordinary
// vector registers -- with 512 allowed it accumulates in AGPRs and reads every result back with a v_accvgpr_read)
template <bool F16, int HW>
__global__ __launch_bounds__(256, 2) void dwconv3x3_mm_kernel(DwArgs p) {
    T16<F16>::enter();
    constexpr int TW = HW + 2;
    constexpr int CPW = HW == 16 ? 16 : 32;                // channels per wave
    constexpr int GPW = CPW / 8;                           // 8-channel groups per wave
    constexpr int CWG = 4 * CPW;                           // channels per workgroup: 64 / 128 (dwconv3x3_map_kernel's slabs)
    constexpr int RB = CWG * 2;                            // bytes per pixel in LDS
    constexpr int NCH = RB / 16;                           // 16-byte chunks per pixel: 8 / 16
    constexpr int NT = HW * HW / 32;                       // 32-pixel tiles per map: 8 / 2
    constexpr int PPR = HW / 2, RPT = 32 / HW;             // pixel pairs per row, rows per tile
    constexpr int NLD = HW * HW * NCH / 256;               // 16-byte loads per thread: 8 / 4
    constexpr int IN_BYTES = TW * TW * RB;                 // 41 472 / 25 600 (+ the output image: dwmm_lds_bytes, dynamic LDS)
    unsigned char* const tin = conv_lds_dyn;
    unsigned char* const tout = conv_lds_dyn + IN_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y, c0 = blockIdx.x * CWG;
    auto fsw = [](int y, int x) { return HW == 16 ? ((x >> 1) & 7) : (((x >> 1) & 3) | ((y & 3) << 2)); };

    uint4 ld[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int idx = tid + 256 * k, px = idx / NCH, ch = idx % NCH;
        ld[k] = *reinterpret_cast<const uint4*>(p.in + ((size_t)b * (HW * HW) + px) * p.C + c0 + ch * 8);
    }
    for (int i = tid; i < (4 * HW + 4) * NCH; i += 256) {  // the ring of zero pixels
        const int q = i / NCH, ch = i % NCH;
        const int y = q < TW ? 0 : (q < 2 * TW ? TW - 1 : 1 + ((q - 2 * TW) >> 1)), x = q < TW ? q : (q < 2 * TW ? q - TW : ((q - 2 * TW) & 1) * (TW - 1));
        *reinterpret_cast<uint4*>(tin + (y * TW + x) * RB + ch * 16) = make_uint4(0, 0, 0, 0);
    }
    // the wave's weight fragments (block-diagonal Toeplitz rows, dw_mm.h) and its lanes' biases
    const DwmmLane wl(lane);
    const int n = lane & 15, j = lane >> 4, s = j >> 1;
    uint4 afr[GPW][3];
    f32x4 bias[GPW];
#pragma unroll
    for (int g = 0; g < GPW; ++g) {
        const int cg = c0 + wave * CPW + g * 8;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int tap = ky * 3 + min(max(wl.d, 0), 2);
            afr[g][ky] = wl.place((uint32_t)p.w[(size_t)tap * p.C + cg + wl.c]);
        }
        const float4 bs = *reinterpret_cast<const float4*>(p.bias + cg + 4 * (j & 1));
        bias[g] = f32x4{bs.x, bs.y, bs.z, bs.w};
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int idx = tid + 256 * k, px = idx / NCH, ch = idx % NCH;
        const int y = px / HW + 1, x = px % HW + 1;
        *reinterpret_cast<uint4*>(tin + (y * TW + x) * RB + ((ch ^ fsw(y, x)) << 4)) = ld[k];
    }
    __syncthreads();

    // lane (n, j): pixel pair n of a tile = row ry, first pixel xp; B fragment = padded pixel (row + ky, xp + j)
    const int ry = n / PPR, xp = 2 * (n % PPR);
    int boff[GPW][3];
#pragma unroll
    for (int g = 0; g < GPW; ++g)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
            boff[g][ky] = ((ry + ky) * TW + xp + j) * RB + (((wave * GPW + g) ^ fsw(ry + ky, xp + j)) << 4);     // (rows of later tiles: + t RPT TW RB; f is periodic in them)
    static_assert(HW == 16 || RPT % 4 == 0, "f(y, x) must not change from tile to tile");
    int ooff[GPW];
    {
        const int px0 = ry * HW + xp + s;                   // the lane's output pixel in tile 0; tile t: + 32 t (h(pixel) is periodic in 32)
#pragma unroll
        for (int g = 0; g < GPW; ++g) ooff[g] = px0 * RB + ((((wave * GPW + g) ^ (px0 >> 1)) & (NCH - 1)) << 4) + (j & 1) * 8;
    }
    uint32_t one_lo, one_hi;
    if constexpr (F16) asm volatile("v_mov_b32 %0, 0x3c00\n\tv_lshlrev_b32 %1, 16, %0" : "=v"(one_lo), "=v"(one_hi));
    else asm volatile("v_mov_b32 %0, 0x3f80\n\tv_lshlrev_b32 %1, 16, %0" : "=v"(one_lo), "=v"(one_hi));
    float psum[GPW][4];
#pragma unroll
    for (int g = 0; g < GPW; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) psum[g][i] = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int g = 0; g < GPW; ++g) {
            f32x4 acc = bias[g];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const uint4 bf = *reinterpret_cast<const uint4*>(tin + boff[g][ky] + t * (RPT * TW * RB));
                acc = mfma16<F16>(afr[g][ky], bf, acc);
            }
            const uint32_t pk0 = T16<F16>::pack2(silu_fast(acc[0]), silu_fast(acc[1]));
            const uint32_t pk1 = T16<F16>::pack2(silu_fast(acc[2]), silu_fast(acc[3]));
            psum[g][0] = T16<F16>::dot2(pk0, one_lo, psum[g][0]);      // the pool sees the stored (rounded) activations
            psum[g][1] = T16<F16>::dot2(pk0, one_hi, psum[g][1]);
            psum[g][2] = T16<F16>::dot2(pk1, one_lo, psum[g][2]);
            psum[g][3] = T16<F16>::dot2(pk1, one_hi, psum[g][3]);
            *reinterpret_cast<uint2*>(tout + ooff[g] + t * (32 * RB)) = make_uint2(pk0, pk1);
        }
    }
    __syncthreads();                                        // every fragment read is done: the input image is free
    if (p.pooled) {
        // pooled means, one fixed order (the fused fronts walk the same one): a lane's sums over its tiles, then dw_mm.h's butterfly over
        // the 32 lanes that hold the same channels; lanes 0 and 16 (pixel pair 0, first pixel: the two channel halves) store
        const int xaddr = (lane ^ 32) << 2;
#pragma unroll
        for (int g = 0; g < GPW; ++g) {
            const float4 tot = dwmm_pool_sum4(psum[g], xaddr, 1.0f / (float)(HW * HW));
            if ((lane & 47) == 0) *reinterpret_cast<float4*>(p.pooled + (size_t)b * p.C + c0 + wave * CPW + g * 8 + 4 * (j & 1)) = tot;
        }
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int idx = tid + 256 * k, px = idx / NCH, ch = idx % NCH;
        *reinterpret_cast<uint4*>(p.out + ((size_t)b * (HW * HW) + px) * p.C + c0 + ch * 8) =
            *reinterpret_cast<const uint4*>(tout + px * RB + (((ch ^ (px >> 1)) & (NCH - 1)) << 4));
    }
}

static constexpr int dwmm_lds_bytes(int hw) { return (hw + 2) * (hw + 2) * (hw == 16 ? 128 : 256) + hw * hw * (hw == 16 ? 128 : 256); }

int dw_slabs(const DwArgs& a) {
    const int nq = (a.OH * a.OW) >> 2;
    return cdiv(a.C / 8, 256 / std::min(32, std::max(nq, 1)));
}

int launch_dwconv3x3(const DwArgs& a, hipStream_t st) {
    if (a.C % 8 != 0 || a.OW % 4 != 0 || ((a.OH * a.OW) >> 2) < 8 || 256 % std::min(32, (a.OH * a.OW) >> 2) != 0) {
        set_error("dwconv3x3: unsupported shape C=%d OH=%d OW=%d", a.C, a.OH, a.OW);
        return ISB_ERR_INVALID;
    }
    dim3 grid(dw_slabs(a), a.B);
    if (a.se_w1) {
        if (!a.pooled || !a.se_part || a.cse < 1 || a.cse > 160 || a.C % 8 != 0 || (int)grid.x > SE_MAX_PARTS) {
            set_error("dwconv3x3: the folded SE FC1 needs pooled + se_part, cse <= 160 and at most %d slabs (C=%d)", SE_MAX_PARTS, a.C);
            return ISB_ERR_INVALID;
        }
    }
    // fp16 forms: fp16 -> fp16 at either stride (the blocks of the fp16 stages), stride 2 bf16 -> fp16 (the block that enters them
    // under isb_hpe_cfg.precision 0)
    const int form = (a.in_f16 ? 1 : 0) | (a.out_f16 ? 2 : 0);
    if ((form == 2 && a.stride != 2) || form == 1) {
        set_error("dwconv3x3: fp16 forms are fp16 -> fp16 and stride 2 bf16 -> fp16 (in_f16=%d out_f16=%d stride=%d)", a.in_f16, a.out_f16, a.stride);
        return ISB_ERR_INVALID;
    }
    if (a.stride == 1 && a.H == a.W && (a.H == 8 || a.H == 16) && a.OH == a.H && a.OW == a.W && a.pad == 1 && (form == 0 || form == 3) &&
        a.general == 3 && !a.se_w1 && a.C % (a.H == 8 ? 128 : 64) == 0) {
        // the taps on the matrix pipe (same slabs, so grid.x = dw_slabs(a)): the arithmetic of the fused 8 x 8 front (mbfront8_kernel), for
        // the batches too small for it. As a stand-alone kernel it is SLOWER than the v_dot2 form below (66 vs 60 us on 16 x 16 maps, 58 vs
        // 52 on 8 x 8 at 256 frames: these launches move their bytes at 6 TB/s either way, and this form holds two workgroups per CU)
        static DevOnce attr_set;
        if (attr_set.need()) {
            ISB_HIP(hipFuncSetAttribute((const void*)dwconv3x3_mm_kernel<true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, dwmm_lds_bytes(8)));
            ISB_HIP(hipFuncSetAttribute((const void*)dwconv3x3_mm_kernel<false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, dwmm_lds_bytes(8)));
            ISB_HIP(hipFuncSetAttribute((const void*)dwconv3x3_mm_kernel<true, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, dwmm_lds_bytes(16)));
            ISB_HIP(hipFuncSetAttribute((const void*)dwconv3x3_mm_kernel<false, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, dwmm_lds_bytes(16)));
            attr_set.mark();
        }
        const int ldsb = dwmm_lds_bytes(a.H);
        if (a.H == 8) { if (form == 3) hipLaunchKernelGGL((dwconv3x3_mm_kernel<true, 8>), grid, dim3(256), ldsb, st, a); else hipLaunchKernelGGL((dwconv3x3_mm_kernel<false, 8>), grid, dim3(256), ldsb, st, a); }
        else { if (form == 3) hipLaunchKernelGGL((dwconv3x3_mm_kernel<true, 16>), grid, dim3(256), ldsb, st, a); else hipLaunchKernelGGL((dwconv3x3_mm_kernel<false, 16>), grid, dim3(256), ldsb, st, a); }
        ISB_LAUNCHED("dwconv3x3_mm", st);
        return ISB_OK;
    }
    if (a.stride == 1 && a.H == a.W && (a.H == 8 || a.H == 16) && a.OH == a.H && a.OW == a.W && a.pad == 1 && (form == 0 || form == 3) &&
        a.general != 1) {
        // grid.x = dw_slabs(a): slabs of 128 (8 x 8 maps) / 64 (16 x 16 maps) channels, as in the general kernel
#define ISB_DW_MAP(F16_, HW_)                                                                                            \
    do {                                                                                                                 \
        if (a.se_w1) hipLaunchKernelGGL((dwconv3x3_map_kernel<F16_, HW_, true>), grid, dim3(256), 0, st, a);             \
        else hipLaunchKernelGGL((dwconv3x3_map_kernel<F16_, HW_, false>), grid, dim3(256), 0, st, a);                    \
    } while (0)
        if (a.H == 8) { if (form == 3) ISB_DW_MAP(true, 8); else ISB_DW_MAP(false, 8); }
        else { if (form == 3) ISB_DW_MAP(true, 16); else ISB_DW_MAP(false, 16); }
#undef ISB_DW_MAP
        ISB_LAUNCHED("dwconv3x3_map", st);
        return ISB_OK;
    }
#define ISB_DW_POOL(S_, FC1_)                                                                                                \
    do {                                                                                                                     \
        if (form == 3) hipLaunchKernelGGL((dwconv3x3_pool_kernel<S_, FC1_, true, true>), grid, dim3(256), 0, st, a);         \
        else if (form == 2) hipLaunchKernelGGL((dwconv3x3_pool_kernel<2, FC1_, false, true>), grid, dim3(256), 0, st, a);    \
        else hipLaunchKernelGGL((dwconv3x3_pool_kernel<S_, FC1_>), grid, dim3(256), 0, st, a);                               \
    } while (0)
    if (a.se_w1) { if (a.stride == 1) ISB_DW_POOL(1, true); else ISB_DW_POOL(2, true); }
    else { if (a.stride == 1) ISB_DW_POOL(1, false); else ISB_DW_POOL(2, false); }
#undef ISB_DW_POOL
    ISB_LAUNCHED("dwconv3x3_pool", st);
    return ISB_OK;
}

// =====================================================================================
// squeeze-excite FCs in f32 on the vector ALU. Two tiny GEMMs (B x cse x C, 0.1 GFLOP) between a
// global pool and the gated projection: what costs is latency and parallelism, not FLOPs.
//   se_fc1_part: part[kc][b][j] = sum_{c in chunk kc} pooled[b][c] * W1[j][c]
//                WG = 16 samples x 16 outputs x one 256-channel chunk, operands staged in LDS once
//                (grid = cse/16 x B/16 x C/256 workgroups: every CU gets work, each makes ONE round trip)
//   se_fc2:      mid[b][j]  = silu(b1[j] + sum_kc part[kc][b][j])           (prologue, fixed order)
//                gate[b][c] = sigmoid(b2[c] + sum_j mid[b][j] * W2T[j][c])
//                WG = 8 samples x 256 channels; lane = 4 channels (16-B weight loads), the 4 waves split j,
//                partial sums meet in LDS and are added in wave order
// every sum runs in a fixed order that depends on neither scheduling nor the batch size
// =====================================================================================
constexpr int SE_CHUNK = 256;      // channels per fc1 workgroup
constexpr int SE_ROW = SE_CHUNK + 4;   // LDS row stride in floats (1040 B: consecutive rows land in consecutive 16-B slots)

__global__ __launch_bounds__(256) void se_fc1_part_kernel(SeFcArgs p) {
    __shared__ __attribute__((aligned(16))) float Ps[16][SE_ROW];
    __shared__ __attribute__((aligned(16))) float Ws[16][SE_ROW];
    const int t = threadIdx.x;
    const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 16, c0 = blockIdx.z * SE_CHUNK;
    {
        const int row = t >> 4, q = (t & 15) * 4;
        const int b = b0 + row, j = j0 + row;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int cl = pass * 64 + q, c = c0 + cl;
            float4 pv = make_float4(0.f, 0.f, 0.f, 0.f), wv = pv;
            if (c < p.C) {
                if (b < p.B) pv = *reinterpret_cast<const float4*>(p.pooled + (size_t)b * p.C + c);
                if (j < p.cse) wv = *reinterpret_cast<const float4*>(p.w1 + (size_t)j * p.C + c);
            }
            *reinterpret_cast<float4*>(&Ps[row][cl]) = pv;
            *reinterpret_cast<float4*>(&Ws[row][cl]) = wv;
        }
    }
    __syncthreads();
    const int s = t >> 4, jl = t & 15;
    float acc = 0.f;
#pragma unroll 8
    for (int c4 = 0; c4 < SE_CHUNK / 4; ++c4) {
        const float4 pv = *reinterpret_cast<const float4*>(&Ps[s][c4 * 4]);
        const float4 wv = *reinterpret_cast<const float4*>(&Ws[jl][c4 * 4]);
        acc = fmaf(pv.x, wv.x, acc);
        acc = fmaf(pv.y, wv.y, acc);
        acc = fmaf(pv.z, wv.z, acc);
        acc = fmaf(pv.w, wv.w, acc);
    }
    const int b = b0 + s, j = j0 + jl;
    if (b < p.B && j < p.cse) p.part[((size_t)blockIdx.z * p.B + b) * p.cse + j] = acc;
}

// PRE (launches of a few samples: one workgroup per CU, latency is everything): the wave's W2T rows are requested
// before the FC1 partials are even read, so the kernel makes one memory round trip instead of six
template <bool PRE>
__global__ __launch_bounds__(256) void se_fc2_kernel(SeFcArgs p) {
    __shared__ __attribute__((aligned(16))) float mids[160][8];          // [j][sample]
    __shared__ __attribute__((aligned(16))) float4 red[4][8][64];        // [wave][sample][lane]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int b0 = blockIdx.y * 8;
    float4 wpre[PRE ? 40 : 1];
    if constexpr (PRE) {
        const int c = blockIdx.x * 256 + lane * 4;
        const int jq = (p.cse + 3) >> 2, jb = wave * jq, je = min(p.cse, jb + jq);
#pragma unroll
        for (int q = 0; q < 40; ++q)
            wpre[q] = (c < p.C && jb + q < je) ? *reinterpret_cast<const float4*>(p.w2t + (size_t)(jb + q) * p.C + c)
                                              : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int nkc = p.nparts > 0 ? p.nparts : (p.C + SE_CHUNK - 1) / SE_CHUNK;
    for (int i = t; i < 8 * p.cse; i += 256) {
        const int sm = i / p.cse, j = i - sm * p.cse;
        float v = 0.f;
        if (b0 + sm < p.B) {
            float pv[SE_MAX_PARTS];            // all partials requested at once (one memory round trip), added in order
#pragma unroll
            for (int kc = 0; kc < SE_MAX_PARTS; ++kc) pv[kc] = kc < nkc ? p.part[((size_t)kc * p.B + b0 + sm) * p.cse + j] : 0.f;
            v = p.b1[j];
#pragma unroll
            for (int kc = 0; kc < SE_MAX_PARTS; ++kc)
                if (kc < nkc) v += pv[kc];
            v = v / (1.0f + expf(-v));
        }
        mids[j][sm] = v;
    }
    __syncthreads();
    const int c = blockIdx.x * 256 + lane * 4;
    const bool cok = c < p.C;
    const int jq = (p.cse + 3) >> 2;
    const int jb = wave * jq, je = min(p.cse, jb + jq);
    float4 acc[8];
#pragma unroll
    for (int sm = 0; sm < 8; ++sm) acc[sm] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int ns = min(8, p.B - b0);
    if constexpr (PRE) {
#pragma unroll
        for (int q = 0; q < 40; ++q) {
            if (jb + q >= je) break;
            const float4 wv = wpre[q];
            const float4 m0 = *reinterpret_cast<const float4*>(&mids[jb + q][0]);
            const float4 m1 = *reinterpret_cast<const float4*>(&mids[jb + q][4]);
            const float mm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
            for (int sm = 0; sm < 8; ++sm) {
                if (sm >= ns) break;           // a single frame pays for one sample, not eight
                acc[sm].x = fmaf(mm[sm], wv.x, acc[sm].x);
                acc[sm].y = fmaf(mm[sm], wv.y, acc[sm].y);
                acc[sm].z = fmaf(mm[sm], wv.z, acc[sm].z);
                acc[sm].w = fmaf(mm[sm], wv.w, acc[sm].w);
            }
        }
    } else if (cok) {
#pragma unroll 8
        for (int j = jb; j < je; ++j) {
            const float4 wv = *reinterpret_cast<const float4*>(p.w2t + (size_t)j * p.C + c);
            const float4 m0 = *reinterpret_cast<const float4*>(&mids[j][0]);
            const float4 m1 = *reinterpret_cast<const float4*>(&mids[j][4]);
            const float mm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
            for (int sm = 0; sm < 8; ++sm) {
                acc[sm].x = fmaf(mm[sm], wv.x, acc[sm].x);
                acc[sm].y = fmaf(mm[sm], wv.y, acc[sm].y);
                acc[sm].z = fmaf(mm[sm], wv.z, acc[sm].z);
                acc[sm].w = fmaf(mm[sm], wv.w, acc[sm].w);
            }
        }
    }
#pragma unroll
    for (int sm = 0; sm < 8; ++sm) red[wave][sm][lane] = acc[sm];
    __syncthreads();
    if (!cok) return;
    const float4 bias = *reinterpret_cast<const float4*>(p.b2 + c);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int sm = wave * 2 + q;
        if (b0 + sm >= p.B) continue;
        float4 v = red[0][sm][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float4 u = red[w][sm][lane];
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        v.x = 1.0f / (1.0f + expf(-(v.x + bias.x)));
        v.y = 1.0f / (1.0f + expf(-(v.y + bias.y)));
        v.z = 1.0f / (1.0f + expf(-(v.z + bias.z)));
        v.w = 1.0f / (1.0f + expf(-(v.w + bias.w)));
        *reinterpret_cast<float4*>(p.gate + (size_t)(b0 + sm) * p.C + c) = v;
    }
}

int launch_se_fcs(const SeFcArgs& a, hipStream_t st) {
    if (a.cse > 160 || a.C % 4 != 0 || a.C > 15 * SE_CHUNK || !a.part || a.nparts > SE_MAX_PARTS) {
        set_error("se_fcs: unsupported shape cse=%d C=%d parts=%d", a.cse, a.C, a.nparts);
        return ISB_ERR_INVALID;
    }
    if (a.nparts <= 0)
        hipLaunchKernelGGL(se_fc1_part_kernel, dim3(cdiv(a.cse, 16), cdiv(a.B, 16), cdiv(a.C, SE_CHUNK)), dim3(256), 0, st, a);
    if (a.B <= 8) hipLaunchKernelGGL(se_fc2_kernel<true>, dim3(cdiv(a.C, 256), 1), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(se_fc2_kernel<false>, dim3(cdiv(a.C, 256), cdiv(a.B, 8)), dim3(256), 0, st, a);
    ISB_LAUNCHED("se_fcs", st);
    return ISB_OK;
}

// =====================================================================================
// stem: conv3x3 stride 2 (TF SAME on an even input: pad bottom/right), 3 -> 32, bias, SiLU.
// f32 crop [B,256,256,3] -> bf16 [B,128,128,32]. thread = one output pixel, all 32 channels;
// weights [16 channel pairs][27 taps][2] f32 (scale folded) are wave-uniform -> scalar loads, 54 consecutive floats per pair.
// =====================================================================================
template <bool OUT_F16, int SILU>
__global__ __launch_bounds__(256) void stem_kernel(StemArgs p) {
    const int OH = p.H / 2, OW = p.W / 2;
    const unsigned pix = blockIdx.x * 256u + threadIdx.x;       // grid = (pixels of an image / 256, B): 32-bit index arithmetic
    if (pix - (threadIdx.x & 63) >= (unsigned)(OH * OW)) return;                 // (whole waves only: a wave's lanes store for one another)
    const unsigned pixc = min(pix, (unsigned)(OH * OW) - 1u);
    const int b = blockIdx.y, oy = (int)(pixc / (unsigned)OW), ox = (int)(pixc - (unsigned)oy * (unsigned)OW);
    const size_t idx = (size_t)b * OH * OW + pix;
    float x[27];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = 2 * oy + ky, ix = 2 * ox + kx;
            const bool ok = iy < p.H && ix < p.W;
            // unconditional loads from clamped addresses, the select on the VALUE (a select on the load is a branch + vmcnt(0) per tap)
            const float* src = p.in + ((size_t)(b * p.H + min(iy, p.H - 1)) * p.W + min(ix, p.W - 1)) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = src[c];
                x[(ky * 3 + kx) * 3 + c] = ok ? v : 0.f;
            }
        }
    // output channels in PAIRS: the pair-major weights wt[co / 2][k][co & 1] put (co, co + 1) into adjacent scalar registers, so one v_pk_fma_f32
    // (same IEEE fma per half, same k order: the bits of the scalar chain) does two of the 864 multiply-adds of a pixel. Round 5: the
    // kernel ran AT the vector ALU's issue rate (valu_active_share 1.05, 184 us per 256 frames; 1104 v_fma + a 15-instruction IEEE
    // division per SiLU); SiLU now is the x * rcp(1 + exp2(-x log2 e)) of every other layer.
    const f32x2_t* w2 = reinterpret_cast<const f32x2_t*>(p.wt);
    const f32x2_t* b2 = reinterpret_cast<const f32x2_t*>(p.bias);
    uint32_t o[16];
#pragma unroll
    for (int c2 = 0; c2 < 16; ++c2) {                  // one channel pair at a time: its 27 weight pairs fit the scalar registers
        f32x2_t acc = b2[c2];
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            const f32x2_t xk = {x[k], x[k]};
            acc = __builtin_elementwise_fma(xk, w2[c2 * 27 + k], acc);
        }
        // SILU 0: x * v_rcp(1 + v_exp(..)) (1 ulp each); 1: the reciprocal refined by one Newton step (the stem's output is what all 79 blocks
        // amplify: the hard weight profile's distance to fp32, priced in EXPERIMENTS.md round 6); 2: IEEE division (round 4's form)
        const float a0 = SILU == 2 ? silu_(acc.x) : SILU == 1 ? silu_nr(acc.x) : silu_fast(acc.x);
        const float a1 = SILU == 2 ? silu_(acc.y) : SILU == 1 ? silu_nr(acc.y) : silu_fast(acc.y);
        o[c2] = OUT_F16 ? ((uint32_t)f2h_(a0) | ((uint32_t)f2h_(a1) << 16)) : ((uint32_t)f2bf_(a0) | ((uint32_t)f2bf_(a1) << 16));
    }
    // a lane holds its pixel's 64 bytes; stored from here every instruction would write 16 bytes into each of 64 different rows. Through
    // LDS (wave-local, slots rotated by (pixel >> 2) & 3: conflict-free both ways) a store instruction writes 16 whole pixels = 1 KiB
    __shared__ __attribute__((aligned(16))) unsigned char tile[256 * 64];
    {
        const int t = threadIdx.x, sw = (t >> 2) & 3;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            *reinterpret_cast<uint4*>(tile + t * 64 + ((c ^ sw) << 4)) = make_uint4(o[4 * c], o[4 * c + 1], o[4 * c + 2], o[4 * c + 3]);
        const int lane = t & 63, w0 = t & ~63, cc = lane & 3;
        uint16_t* const base = p.out + (idx - lane) * 32;          // the wave's first pixel (a wave never straddles images: OH * OW % 64 == 0)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pp = (lane >> 2) + 16 * i;
            const uint4 v = *reinterpret_cast<const uint4*>(tile + (w0 + pp) * 64 + ((cc ^ ((pp >> 2) & 3)) << 4));
            if (pix - lane + pp < (unsigned)(OH * OW)) *reinterpret_cast<uint4*>(base + pp * 32 + cc * 8) = v;
        }
    }
}

int launch_stem(const StemArgs& a, hipStream_t st) {
    const dim3 grid((unsigned)cdiv((a.H / 2) * (a.W / 2), 256), (unsigned)a.B);
    static const int silu = [] { const char* e = getenv("ISB_STEM_SILU"); return e ? atoi(e) : 0; }();
    if (a.out_f16) {
        if (silu == 2) hipLaunchKernelGGL((stem_kernel<true, 2>), grid, dim3(256), 0, st, a);
        else if (silu == 1) hipLaunchKernelGGL((stem_kernel<true, 1>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((stem_kernel<true, 0>), grid, dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL((stem_kernel<false, 0>), grid, dim3(256), 0, st, a);
    }
    ISB_LAUNCHED("stem", st);
    return ISB_OK;
}

// f32 -> bf16 (weights at load time), with an optional per-row scale (folded BN)
__global__ void f32_to_bf16_rows_kernel(const float* in, const float* row_scale, uint16_t* out, size_t rows, size_t cols, int f16) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * cols) return;
    const float s = row_scale ? row_scale[i / cols] : 1.f;
    out[i] = f16 ? f2h_(in[i] * s) : f2bf_(in[i] * s);
}

int launch_f32_to_bf16_rows(const float* in, const float* row_scale, uint16_t* out, size_t rows, size_t cols, hipStream_t st, int f16) {
    hipLaunchKernelGGL(f32_to_bf16_rows_kernel, dim3((unsigned)cdivz(rows * cols, 256)), dim3(256), 0, st, in, row_scale, out, rows, cols, f16);
    ISB_LAUNCHED("f32_to_bf16_rows", st);
    return ISB_OK;
}

}  // namespace isb
