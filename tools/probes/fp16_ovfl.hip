// Does MODE.FP16_OVFL (bit 23) make v_cvt_pk_f16_f32 saturate at +-65504 on gfx950?   hipcc --offload-arch=gfx950 -O2 fp16_ovfl.hip -o /tmp/fp16_ovfl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, uint32_t* out, int mode) {
    if (mode) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
    const f32x2 v = {in[2 * threadIdx.x], in[2 * threadIdx.x + 1]};
    uint32_t r;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(v.x), "v"(v.y));
    out[threadIdx.x] = r;
    _Float16 h = (_Float16)in[2 * threadIdx.x];       // scalar conversion (v_cvt_f16_f32)
    out[64 + threadIdx.x] = __builtin_bit_cast(uint16_t, h);
}
int main() {
    float h[8] = {1e6f, -1e6f, 65504.f, 65520.f, 70000.f, 1.0f, __builtin_inff(), -__builtin_inff()};
    float* d; uint32_t* o; uint32_t r[128];
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof r); hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, d, o, mode);
        hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
        printf("mode %d: pk", mode);
        for (int i = 0; i < 4; ++i) printf(" %08x", r[i]);
        printf(" | scalar");
        for (int i = 0; i < 4; ++i) printf(" %04x", r[64 + i]);
        printf("\n");
    }
    return 0;
}
