// Issue cost of the depthwise taps' candidate instructions on gfx950: shader cycles per wave-instruction, one and two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
// 16 independent accumulators per wave, 256 x 16 instructions between two s_memtime stamps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__(512) void rate_kernel(uint64_t* out, const uint32_t* in, float* sink) {
    uint32_t x = in[threadIdx.x & 63], w = in[64 + (threadIdx.x & 63)];
    float a[16];
    float pk[2] = {1.0f, 1.0f};
    asm volatile("" : "+v"(*reinterpret_cast<float2*>(&pk)));
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = (float)i;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 256; ++it) {
#define DOT2C(i) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(w));
#define DOT2CB(i) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(w));
#define FMAMIX(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "+v"(a[i]) : "v"(x), "v"(w));
#define FMA(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(w));
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#define RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define CVT(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
#define AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
#define PKMUL(i) if ((i) % 2 == 0) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<float2*>(&a[i])) : "v"(*reinterpret_cast<float2*>(&pk)));
#define PKFMA(i) if ((i) % 2 == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*reinterpret_cast<float2*>(&a[i])) : "v"(*reinterpret_cast<float2*>(&pk)));
#define PKADD(i) if ((i) % 2 == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*reinterpret_cast<float2*>(&a[i])) : "v"(*reinterpret_cast<float2*>(&pk)));
#define MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
        if constexpr (OP == 0) { REP16(DOT2C) }
        if constexpr (OP == 1) { REP16(FMAMIX) }
        if constexpr (OP == 2) { REP16(FMA) }
        if constexpr (OP == 3) { REP16(EXP) }
        if constexpr (OP == 4) { REP16(RCP) }
        if constexpr (OP == 5) { REP16(CVT) }
        if constexpr (OP == 6) { REP16(DOT2CB) }
        if constexpr (OP == 7) { REP16(AND) }
        if constexpr (OP == 8) { REP16(PKMUL) REP16(PKMUL) }
        if constexpr (OP == 9) { REP16(PKFMA) REP16(PKFMA) }
        if constexpr (OP == 10) { REP16(PKADD) REP16(PKADD) }
        if constexpr (OP == 11) { REP16(MUL) }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
static void run(const char* name, uint64_t* d_out, const uint32_t* d_in, float* d_sink) {
    for (int waves = 4; waves <= 8; waves += 4) {
        hipLaunchKernelGGL(rate_kernel<OP>, dim3(256), dim3(64 * waves), 0, 0, d_out, d_in, d_sink);
        hipDeviceSynchronize();
        std::vector<uint64_t> h(256 * 8);
        hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
        double sum = 0; int n = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) { sum += (double)h[b * 8 + w]; ++n; }
        printf("%-18s %d wave(s) per SIMD: %.2f cycles per wave-instruction of ONE wave -> %.2f per instruction issued on the SIMD\n", name, waves / 4,
               sum / n / 4096.0, sum / n / 4096.0 / (waves / 4));
    }
}

int main() {
    uint64_t* d_out; uint32_t* d_in; float* d_sink;
    hipMalloc(&d_out, 256 * 8 * 8); hipMalloc(&d_in, 128 * 4); hipMalloc(&d_sink, 4);
    std::vector<uint32_t> h(128, 0x3c003c00u);
    hipMemcpy(d_in, h.data(), 512, hipMemcpyHostToDevice);
    run<0>("v_dot2c_f32_f16", d_out, d_in, d_sink);
    run<6>("v_dot2c_f32_bf16", d_out, d_in, d_sink);
    run<1>("v_fma_mix_f32", d_out, d_in, d_sink);
    run<2>("v_fmac_f32", d_out, d_in, d_sink);
    run<3>("v_exp_f32", d_out, d_in, d_sink);
    run<4>("v_rcp_f32", d_out, d_in, d_sink);
    run<5>("v_cvt_pk_f16_f32", d_out, d_in, d_sink);
    run<7>("v_and_b32", d_out, d_in, d_sink);
    run<11>("v_mul_f32", d_out, d_in, d_sink);
    run<8>("v_pk_mul_f32", d_out, d_in, d_sink);
    run<10>("v_pk_add_f32", d_out, d_in, d_sink);
    run<9>("v_pk_fma_f32", d_out, d_in, d_sink);
    return 0;
}
