// Do the matrix pipe and the vector ALU of one SIMD overlap when DIFFERENT waves feed them? One workgroup of 8 waves per CU (two per
// SIMD): waves 0-3 run MFMA chains, waves 4-7 run plain / transcendental vector instructions; each group alone, then both together.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/overlap.hip -o /tmp/overlap && /tmp/overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// mode bit 0: the MFMA waves work; bit 1: the vector waves work; bit 2: vector waves issue transcendentals instead of FMAs;
// bit 3: MFMA chains are DEPENDENT (one accumulator) instead of four independent ones
__global__ __launch_bounds__(512) void overlap_kernel(uint64_t* out, float* sink, int mode, int iters) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool is_mfma = wave < 4;
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * lane); b[e] = (_Float16)(0.002f * e); }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = 1.0f + 0.01f * i + lane;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    if (is_mfma && (mode & 1)) {
        for (int it = 0; it < iters; ++it) {
            if (mode & 8) {
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[0], 0, 0, 0);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
            }
        }
    }
    if (!is_mfma && (mode & 2)) {
        for (int it = 0; it < iters; ++it) {
            if (mode & 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
            }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 1234.5f) sink[0] = s;
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    uint64_t* d_out; float* d_sink;
    hipMalloc(&d_out, 256 * 8 * 8); hipMalloc(&d_sink, 4);
    const int iters = 2000;
    auto run = [&](const char* name, int mode) {
        hipLaunchKernelGGL(overlap_kernel, dim3(256), dim3(512), 0, 0, d_out, d_sink, mode, iters);
        hipDeviceSynchronize();
        std::vector<uint64_t> h(256 * 8);
        hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
        double m = 0, v = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += (double)h[b * 8 + w];
        m /= 256 * 4; v /= 256 * 4;
        const double n_mfma = 16.0 * iters, n_valu = ((mode & 4) ? 64.0 : 128.0) * iters;
        printf("%-44s MFMA waves %8.0f cycles (%.1f per MFMA)   vector waves %8.0f cycles (%.2f per instruction)\n", name, m, (mode & 1) ? m / n_mfma : 0.0, v,
               (mode & 2) ? v / n_valu : 0.0);
    };
    run("MFMA alone (4 independent accumulators)", 1);
    run("MFMA alone (one dependent chain)", 1 | 8);
    run("v_fmac alone", 2);
    run("v_exp alone", 2 | 4);
    run("MFMA (independent) + v_fmac", 1 | 2);
    run("MFMA (independent) + v_exp", 1 | 2 | 4);
    run("MFMA (dependent) + v_fmac", 1 | 2 | 8);
    run("MFMA (dependent) + v_exp", 1 | 2 | 4 | 8);
    return 0;
}
