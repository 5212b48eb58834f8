// Depthwise 3x3 taps on the MATRIX pipe (round 5). Shared by dwconv3x3_mm_kernel (conv_dw_se.hip) and the fused front of the 8 x 8
// MBConv blocks (conv_mb8.hip).
//
// A depthwise convolution has no contraction over channels, so as a GEMM it is block-diagonal -- but a filter ROW is a Toeplitz
// band, and v_mfma_f32_16x16x32 has room for it: for one group of 8 channels and one filter row ky
//     D[m = (s, c)][n] += sum_{k = (j, c')} A[m][k] * B[k][n]
//       n : 16 pixel PAIRS (32 consecutive outputs: pair n = pixels 2n, 2n + 1 of the tile's rows)
//       m : s = which pixel of the pair (0 / 1) x c = channel of the group (0..7)
//       k : j = input column 0..3 relative to the pair's first tap x c' = channel
//       A[(s, c)][(j, c')] = w[c][ky][j - s] if c' == c and 0 <= j - s <= 2, else 0        (weights: block-diagonal Toeplitz)
//       B[(j, c')][n]      = in[row + ky][2n + j][c']                                       (activations, zero ring = padding)
// The B fragment of lane (n = lane & 15, j = lane >> 4) is ONE 16-byte read of 8 consecutive channels of pixel 2n + j -- the NHWC
// layout as it lies in LDS, no im2col, no transpose -- and three chained MFMAs (ky = 0, 1, 2; the C operand starts as the bias)
// give 8 channels x 32 pixels = 256 outputs in 48 matrix cycles, where the vector ALU spent 9 v_dot2 per output (144 issue cycles
// per 256 outputs, the largest item of the depthwise kernels' instruction count: tools/valu_floor.py). Products of two 16-bit
// values are exact in f32 and the accumulation is f32, as with v_dot2; the summation ORDER is the matrix core's (so results agree
// with the v_dot2 kernels to f32 rounding, not bit for bit).
// The lane's four results are channels 4 (j & 1) .. + 3 of pixel 2n + (j >> 1): D row m = 4 (lane >> 4) + i.
#pragma once
#include "conv_common.h"

namespace isb {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool F16>
__device__ __forceinline__ f32x4 mfma16(uint4 a, uint4 b, f32x4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Lane constants of the weight (A) fragment: row m = lane & 15 = (s, c), k group j = lane >> 4.
struct DwmmLane {
    int c;          // channel of the group this lane's A row belongs to (0..7)
    int d;          // tap column j - s this lane's k group holds for that row; outside 0..2: the fragment is zero
    __device__ __forceinline__ explicit DwmmLane(int lane) : c(lane & 7), d((lane >> 4) - ((lane >> 3) & 1)) {}
    __device__ __forceinline__ bool live() const { return d >= 0 && d <= 2; }
    // the fragment: 8 halves = channels c' = 0..7 of input column j; only c' == c is non-zero. w16 = the tap's 16 bits.
    __device__ __forceinline__ uint4 place(uint32_t w16) const {
        const uint32_t v = live() ? ((c & 1) ? (w16 << 16) : (w16 & 0xffffu)) : 0u;
        const int q = c >> 1;
        return make_uint4(q == 0 ? v : 0u, q == 1 ? v : 0u, q == 2 ? v : 0u, q == 3 ? v : 0u);
    }
};

// The pool's cross-lane sum (round 6): v summed over the 32 lanes that hold the same channels as this one -- lanes n + 16 (j & 1) + 32 s
// for the pixel pairs n = 0..15 and the two pixels s of a pair -- in ONE fixed butterfly: n ^ 1, n ^ 2 (quad permutes: every lane of a
// quad then holds the quad's sum), 7 - n within each 8 (pairs the quads), 15 - n within the 16, then lane ^ 32. Both lanes of a pair add
// the same two values (f32 addition commutes), so all 32 end with the same bits; five dependent adds per value, no LDS image, no barrier.
// EVERY kernel that pools a dw_mm result uses it (dwconv3x3_mm_kernel, both forms of both fused fronts): the order is part of the
// arithmetic they share. Before: each lane's sums went through a per-wave LDS scratch and one lane per channel walked the 32 slots in
// order -- 1 250-1 500 of the 6 500 cycles of mbfront8r's consumer tick. xaddr = (lane ^ 32) << 2.
__device__ __forceinline__ float dwmm_pool_sum(float v, int xaddr) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));   // row_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(xaddr, __builtin_bit_cast(int, v)));
    return v;
}

// the four channels 4 (j & 1) .. + 3 of group g, pooled: scale = 1 / pixels of the map (a power of two: the product is the quotient)
__device__ __forceinline__ float4 dwmm_pool_sum4(const float (&ps)[4], int xaddr, float scale) {
    return make_float4(dwmm_pool_sum(ps[0], xaddr) * scale, dwmm_pool_sum(ps[1], xaddr) * scale, dwmm_pool_sum(ps[2], xaddr) * scale,
                       dwmm_pool_sum(ps[3], xaddr) * scale);
}

__device__ __forceinline__ float4 sel4(bool c, float4 a, float4 b) { return make_float4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w); }

}  // namespace isb
