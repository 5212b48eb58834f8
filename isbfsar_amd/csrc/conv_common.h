// Device helpers shared by the convolution translation units (conv_*.hip): operand typedefs, the LDS row
// layout and its swizzle, activations, the LDS-DMA issue helper, the dynamic LDS symbol.
#pragma once
#include <algorithm>
#include <type_traits>
#include <utility>

#include "isb_common.h"
#include "kernels.h"

namespace isb {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float bf2f_(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t f2bf_(float x) { return __builtin_bit_cast(uint16_t, (__bf16)x); }
__device__ __forceinline__ float silu_(float x) { return x / (1.0f + __expf(-x)); }

// SE gate on eight bf16 values: bf16(f32(x) * g), two elements per instruction (shift / mask unpack, v_pk_mul_f32,
// v_cvt_pk_bf16_f32) -- 4 VALU instructions per dword
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t gate_bf16x2(uint32_t w, float g_lo, float g_hi) {
    f32x2_t x = {__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
    const f32x2_t g = {g_lo, g_hi};
    x = x * g;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bf16x2_t));
}
__device__ __forceinline__ uint4 gate_bf16x8(uint4 v, float4 g0, float4 g1) {
    return make_uint4(gate_bf16x2(v.x, g0.x, g0.y), gate_bf16x2(v.y, g0.z, g0.w), gate_bf16x2(v.z, g1.x, g1.y),
                      gate_bf16x2(v.w, g1.z, g1.w));
}

// ---- 16-bit storage type of a layer: bf16 (stages 0-4 of the pose backbone, the detector) or IEEE fp16 (the two 8x8 stages
// and the 640 -> 1280 convolution under isb_hpe_cfg.precision 0: 3 more mantissa bits at the same MFMA rate; per-stage error
// budget in DESIGN.md section 4). One traits struct so that a kernel takes `bool F16` and nothing else changes.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float h2f_(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
// round to nearest even, saturating at the largest finite fp16 (an inf would poison every later layer). Two ways:
//  * f2h_: an explicit clamp in front of the conversion (kernels that learn the storage type at run time: stem, weight conversion);
//  * the kernels templated on `bool F16` set MODE.FP16_OVFL once at their start (T16<F16>::enter()): with that bit every fp16
//    VALU result that would overflow is clamped to +-65504 by the conversion itself (true infinities pass; probed on gfx950,
//    tools/probes/fp16_ovfl.hip), so an fp16 pair costs ONE v_cvt_pk_f16_f32 like a bf16 pair -- the two v_med3 per pair of the
//    explicit clamp made fp16 storage 2.9 % slower than bf16 on the 256-frame pose step (epilogue-bound expand GEMMs).
__device__ __forceinline__ float f16_sat(float x) { return __builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f); }
__device__ __forceinline__ uint16_t f2h_(float x) { return __builtin_bit_cast(uint16_t, (_Float16)f16_sat(x)); }
__device__ __forceinline__ void fp16_ovfl_on() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1"); }

template <bool F16>
struct T16 {
    // first statement of every kernel that stores through this struct: fp16 conversions saturate from here on (see above)
    static __device__ __forceinline__ void enter() {
        if constexpr (F16) fp16_ovfl_on();
    }
    static __device__ __forceinline__ float to_f32(uint16_t h) {
        if constexpr (F16) return h2f_(h); else return bf2f_(h);
    }
    static __device__ __forceinline__ uint16_t from_f32(float x) {
        if constexpr (F16) return __builtin_bit_cast(uint16_t, (_Float16)x); else return f2bf_(x);      // (saturating: enter())
    }
    static __device__ __forceinline__ float lo(uint32_t w) { return to_f32((uint16_t)(w & 0xffffu)); }
    static __device__ __forceinline__ float hi(uint32_t w) { return to_f32((uint16_t)(w >> 16)); }
    static __device__ __forceinline__ uint32_t pack2(float a, float b) {
        if constexpr (F16) {
            const f32x2_t v = {a, b};
            return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));      // (saturating: enter())
        } else {
            const f32x2_t v = {a, b};
            return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
        }
    }
    // D = A (32 x 16) * B (16 x 32) + C on the matrix cores; operands as the raw 16 bytes of a fragment
    static __device__ __forceinline__ f32x16 mfma32(uint4 a, uint4 b, f32x16 c) {
        if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {     // fragments held as 8 x 16-bit vectors
        return mfma32(__builtin_bit_cast(uint4, a), __builtin_bit_cast(uint4, b), c);
    }
    // acc + x.lo * w.lo + x.hi * w.hi in f32 (v_dot2_f32_*: with one half of w zero, an f32 FMA straight from the pair)
    static __device__ __forceinline__ float dot2(uint32_t x, uint32_t w, float acc) {
        if constexpr (F16) return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, x), __builtin_bit_cast(f16x2_t, w), acc, false);
        else return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, x), __builtin_bit_cast(bf16x2_t, w), acc, false);
    }
    static constexpr uint32_t ONE = F16 ? 0x3c00u : 0x3f80u;        // 1.0 in the low half
    // SE gate on a pair: T(f32(x) * g)
    static __device__ __forceinline__ uint32_t gate2(uint32_t w, float g_lo, float g_hi) {
        if constexpr (F16) return pack2(lo(w) * g_lo, hi(w) * g_hi);
        else return gate_bf16x2(w, g_lo, g_hi);
    }
    static __device__ __forceinline__ uint4 gate8(uint4 v, float4 g0, float4 g1) {
        return make_uint4(gate2(v.x, g0.x, g0.y), gate2(v.y, g0.z, g0.w), gate2(v.z, g1.x, g1.y), gate2(v.w, g1.z, g1.w));
    }
};

constexpr int CK = 32;            // k-tile (bf16 elements) = 64 B per row
constexpr int ROWB = 64;          // bytes per LDS row

__device__ __forceinline__ int swz(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 2) & 3)) << 4); }

__device__ __forceinline__ float silu_fast(float x) {
    // x * sigmoid(x) with v_exp_f32 / v_rcp_f32 (about 1 ulp each; the result is rounded to bf16)
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

// two values at once, the arithmetic of silu_fast bit for bit (the same IEEE multiply / add per half): the three plain operations as packed
// f32 instructions (one issue slot per pair instead of two), the transcendentals per value
__device__ __forceinline__ f32x2_t silu_fast2(f32x2_t x) {
    const f32x2_t t = x * f32x2_t{-1.4426950408889634f, -1.4426950408889634f};
    const f32x2_t d = f32x2_t{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + f32x2_t{1.0f, 1.0f};
    return x * f32x2_t{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}

// the same with the reciprocal refined by one Newton step (r' = r + r (1 - d r): two v_fma more; <= 0.5 ulp of 1 / d instead of 1)
__device__ __forceinline__ float silu_nr(float x) {
    const float d = 1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x);
    const float r = __builtin_amdgcn_rcpf(d);
    return x * __builtin_fmaf(r, __builtin_fmaf(-d, r, 1.0f), r);
}

// activation codes (ConvArgs.act): 0 none, 1 SiLU (the pose backbone), 2 Mish, 3 LeakyReLU(0.1) (the YOLOv4 detector),
// 4 ReLU (the ResNet-50 of the RGB / hybrid action-recognition branch)
__device__ __forceinline__ float mish_fast(float x) {
    // x * tanh(softplus(x)) with n = e^x: tanh(ln(1 + n)) = n (n + 2) / (n (n + 2) + 2); one v_exp + one v_rcp.
    // n is clamped so that n (n + 2) stays finite (for x > 20 the factor is 1 to f32 precision anyway)
    const float n = __builtin_amdgcn_exp2f(1.4426950408889634f * fminf(x, 20.0f));
    const float w = n * (n + 2.0f);
    return x * w * __builtin_amdgcn_rcpf(w + 2.0f);
}
__device__ __forceinline__ float act_other(int act, float x) {     // act >= 2 (wave-uniform)
    return act == 2 ? mish_fast(x) : (act == 4 ? fmaxf(x, 0.f) : (x > 0.f ? x : 0.1f * x));
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

extern __shared__ __attribute__((aligned(16))) unsigned char conv_lds_dyn[];

// LDS-DMA: 16 bytes per lane from (wave-uniform base + 32-bit lane offset) to LDS address lds_addr + lane * 16
__device__ __forceinline__ void dma16_s(const void* sbase, uint32_t voff, uint32_t lds_addr) {
    const uint32_t la = __builtin_amdgcn_readfirstlane(lds_addr);
    const uint64_t b = (uint64_t)(uintptr_t)sbase;          // wave-uniform by construction; make the compiler see it
    const uint32_t b_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
    const uint32_t b_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);   // the builtin returns int: widen unsigned
    const uint64_t sb = ((uint64_t)b_hi << 32) | (uint64_t)b_lo;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sb), "s"(la) : "memory");
}

}  // namespace isb
