// 1x1 stride-1 convolutions without an SE gate on gemm1x1_dma_kernel (k_gemm1x1.h): tile variants 131 - 140, 150.
#include "k_gemm1x1.h"

namespace isb {

int launch_tiles_gemm1x1(int v, const ConvArgs& a, ConvArgs& aa, hipStream_t st) {
    if (a.gate || a.KH != 1 || a.stride != 1 || a.pad != 0) {
        set_error("conv_igemm: variants 131-140 / 150 are un-gated 1x1 stride-1 GEMMs");
        return ISB_ERR_INVALID;
    }
    // ISB_G1H: the variant exists in both 16-bit operand types (ConvArgs.f16); ISB_G1: bf16 only
#define ISB_G1_BF16(TM, TN, WGM, WGN)                                                                                        \
    do {                                                                                                                     \
        if (a.probe & 2) ISB_G1_STAMPS(TM, TN, WGM, WGN);                                                                    \
        else hipLaunchKernelGGL((gemm1x1_dma_kernel<TM, TN, WGM, WGN>), g, dim3(64 * WGM * WGN), 0, st, aa);                 \
    } while (0)
#define ISB_G1(TM, TN, WGM, WGN)                                                                                             \
    do {                                                                                                                     \
        if (a.f16) { set_error("conv_igemm: tile variant %d has no fp16 form", v); return ISB_ERR_INVALID; }                 \
        const dim3 g = conv_grid(aa, 32 * TM * WGM, 32 * TN * WGN);                                                          \
        ISB_G1_BF16(TM, TN, WGM, WGN);                                                                                       \
    } while (0)
#define ISB_G1H(TM, TN, WGM, WGN)                                                                                            \
    do {                                                                                                                     \
        const dim3 g = conv_grid(aa, 32 * TM * WGM, 32 * TN * WGN);                                                          \
        if (a.f16) hipLaunchKernelGGL((gemm1x1_dma_kernel<TM, TN, WGM, WGN, 0, false, 2, true>), g, dim3(64 * WGM * WGN), 0, st, aa); \
        else ISB_G1_BF16(TM, TN, WGM, WGN);                                                                                  \
    } while (0)
#ifdef ISB_BUILD_PROBES
#define ISB_G1_STAMPS(TM, TN, WGM, WGN) hipLaunchKernelGGL((gemm1x1_dma_kernel<TM, TN, WGM, WGN, 0, true>), g, dim3(64 * WGM * WGN), 0, st, aa)
#else
#define ISB_G1_STAMPS(TM, TN, WGM, WGN) do { set_error("conv_igemm: s_memtime stamps need a -DISB_BUILD_PROBES build"); return ISB_ERR_INVALID; } while (0)
#endif
    switch (v) {
        case 131: ISB_G1H(1, 3, 4, 2); break;     // 128 x 192
        case 132: ISB_G1H(1, 2, 4, 2); break;     // 128 x 128
        case 135: ISB_G1(1, 2, 8, 1); break;     // 256 x  64
        case 138: ISB_G1H(1, 1, 2, 2); break;     //  64 x  64
        case 150: ISB_G1(1, 2, 2, 2); break;     //  64 x 128, 4 waves
#ifdef ISB_BUILD_PROBES
        case 133: ISB_G1(2, 3, 4, 2); break;     // 256 x 192 (8 waves of 64 x 96)
        case 134: ISB_G1(2, 2, 4, 2); break;     // 256 x 128
        case 136: ISB_G1(1, 3, 8, 1); break;     // 256 x  96
        case 137: ISB_G1(1, 7, 4, 1); break;     // 128 x 224
        case 139: ISB_G1(2, 4, 4, 2); break;     // 256 x 256 (8 waves of 64 x 128)
        case 140: ISB_G1(1, 3, 2, 2); break;     //  64 x 192, 4 waves (four workgroups per CU: more independent phases)
#endif
        default:
            set_error("conv_igemm: tile variant %d is not in this build (un-gated 1x1: 131, 132, 135, 138, 150)", v);
            return ISB_ERR_INVALID;
    }
#undef ISB_G1
#undef ISB_G1H
#undef ISB_G1_BF16
#undef ISB_G1_STAMPS
    return ISB_OK;
}

}  // namespace isb
