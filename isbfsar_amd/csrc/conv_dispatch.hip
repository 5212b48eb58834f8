// launch_conv_igemm: one convolution of the pose backbone / detector / ResNet trunk -> the tile kernel that runs it.
// The choice per layer shape was made by measurement on MI355X (tools/conv_sweep.py, tools/layer_breakdown.py,
// EXPERIMENTS.md); every kernel of the family keeps the same (tap, channel) summation order, so the choice never changes a
// result bit. The kernels live in conv_igemm.hip (general), conv_gemm1x1*.hip (1x1), conv_3x3.hip, conv_ws.hip
// (weights-stationary expand GEMMs); tile variants outside the selected set exist only in -DISB_BUILD_PROBES builds.
#include "conv_tiles.h"

namespace isb {

extern "C" int isb_wsreg_verified(void);      // wsreg_guard.cpp: 1 only if the build checked the staging registers in the disassembly

// tile variant for a layer (ConvArgs.variant == 0)
static int choose_variant(const ConvArgs& a) {
    //   * 1x1 stride-1 convolutions are plain GEMMs and run on the lean gemm1x1 kernels (131-150 without, 141-156 with an SE
    //     gate), 8 waves of 32 x 64..96 sub-tiles: high occupancy beats big tiles here;
    //   * 3x3 convolutions run on conv3x3_dma_kernel (161-171: raw-buffer A operand, hardware zero padding);
    //     the general LDS-DMA kernels remain for shapes outside its contract;
    //   * a single frame (M <= 2048) needs many small workgroups: 64-row tiles.
    const bool g1 = a.KH == 1 && a.stride == 1 && a.pad == 0 && a.zeros;
    const int ohw = a.OH * a.OW;
    // (stride 2: TF-SAME on an even input = pad 0, bottom / right overhang; or PyTorch's symmetric pad 1 -- the lean 3x3 kernel
    // addresses its taps from a shifted buffer base and a per-lane validity mask, whatever the padding)
    const bool c3 = !a.gate && a.KH == 3 && a.KW == 3 && ((a.stride == 1 && a.pad == 1) || (a.stride == 2 && (a.pad == 0 || a.pad == 1))) &&
                    (size_t)a.B * a.H * a.W * a.Cin * 2 + (size_t)(a.W + 1) * a.Cin * 2 < 0x7ffffff0ull;
    const bool ws_ok = isb_wsreg_verified() != 0;       // fail closed: without the build's verdict the tile kernels run
    if (a.M <= 2048 && a.Cout >= 64) {
        if (g1 && !a.gate) return 138;
        if (g1 && a.gate && ohw % 64 == 0) return 147;
        if (c3) return 169;                         //  64 x  64 on the lean 3x3 kernel (one frame's 32 x 32 maps: 96 workgroups)
        return (!a.gate && a.zeros) ? 64 : 75;
    }
    if (a.M <= 8192 && a.Cout >= 128 && a.Cout % 128 == 0 && !a.gate && !a.f16 && (c3 || g1) &&
        (long)cdiv(a.M, 128) * cdiv(a.Cout, 128) < 384) {
        // a few thousand rows (the detector's 8 x 8 / 16 x 16 maps at 64 frames, the ResNet-50's 7 x 7 and 14 x 14 maps): 128-row
        // tiles would leave most CUs idle -- 64 x 128 tiles of 4 waves double the workgroups
        return c3 ? 168 : 150;
    }
#ifdef ISB_BUILD_PROBES
    const bool ws_act = a.act <= 1;                     // the probe build keeps the no-activation instantiations of 184 / 186
#else
    const bool ws_act = a.act == 1;                     // the product build holds the SiLU forms only (launch_conv_ws refuses the rest): act 0 -> 131 / 132
#endif
    if (g1 && !a.gate && !a.res && !a.out_f32 && ws_act && a.splits <= 1 && ws_ok &&
        (a.Cin == 96 || a.Cin == 192 || a.Cin == 224 || a.Cin == 384) && a.Cout % 32 == 0 && a.M >= 64 * 128 &&
        (size_t)a.M * a.Cin * 2 < 0xffffffffull) {
        // weights-stationary persistent GEMM (short-K expand convolutions), two waves per SIMD: 2 workgroups x 4 waves, or
        // (K = 384) 1 x 8 -- eight waves pay from 256 tiles on
        return a.Cin == 384 ? ((a.M >= 16384 || !a.act) ? 186 : 185) : 184;
    }
    if (g1 && !a.gate) {
        if (a.Cout % 192 == 0) return 131;          // 128 x 192
        if (a.Cout == 64 && !a.f16) return 135;     // 256 x  64
        if (a.Cout == 32) return 59;
        return 132;                                 // 128 x 128
    }
    if (g1 && a.gate && (ohw % 128 == 0 || 128 % ohw == 0)) {
        // the 8 x 8 stages: 128-row tiles leave at most one workgroup per CU (a half-batch lane: half the CUs);
        // 64 x 192 tiles of 4 waves measured +14 % (384 outputs) and +41 % (640 outputs) at 8192 rows
        const long wgs128 = (long)cdiv(a.M, 128) * cdiv(a.Cout, a.Cout % 320 == 0 ? 320 : 192);
        if (ohw == 64 && a.M >= 4096 && a.Cin % 64 == 0 && a.Cin >= 256 && a.splits <= 1 && !a.out_f32 &&
            (a.Cout % 192 == 0 || a.Cout % 320 == 0) && (size_t)a.M * a.Cin * 2 < 0xffffffffull &&
            2 * a.Cin * 4 + (a.Cout % 192 == 0 || a.M < 16384 ? 6 * 320 : 4 * 448) * 64 + 2048 <= 160 * 1024) {
            // the 8 x 8 stages: 128-row tiles with loader waves, one workgroup per CU (gemm1x1_lw_kernel; bit-identical).
            // 2304 -> 384: 53.7 vs 60.3 us at 256 frames, 39.6 vs 46.7 at 128; 3840 -> 640: 113.7 vs 123.0 / 76.0 vs 84.1
            // (round 6 sweep, tools/exp_sweep_b1024.py: at 1024 frames -- M = 65 536 -- the A operand no longer stays in the L2 / Infinity
            // Cache between the N tiles and two plain 128 x 192 workgroups per CU beat the loader-wave form, 214 vs 222-232 us on 2304 -> 384)
            if (a.Cout % 192 == 0 && a.M >= 65536) return 141;
            return (a.Cout % 192 == 0 || a.M < 16384) ? 155 : 156;
        }
        if (ohw % 64 == 0 && a.Cout % 64 == 0 && ((a.Cout % 192 == 0 && wgs128 <= 256) || (a.Cout % 320 == 0 && wgs128 < 256))) return 146;
        if (a.Cout % 320 == 0) return 144;          // 128 x 320
        if (a.Cout == 224) return 143;              // 128 x 224
        if (a.Cout % 192 == 0) return 141;          // 128 x 192
        return 142;                                 // 128 x 128
    }
    if (c3) {
        if (a.Cout == 32 && a.Cin == 32 && a.stride == 1 && a.W == 128 && a.H % 2 == 0 && !a.out_f32 && a.act <= 1)
            return 171;                             // rows ring in LDS, +50 % over the implicit GEMM (bit-identical)
        if (a.Cout == 32) return 163;               // 256 x  32   (lean 3x3, buffer-addressed A operand)
        if (a.Cout % 192 == 0 && a.stride == 1 && a.Cin == 96 && a.W == 32 && a.H % 4 == 0)
            return 167;                             // 128 x 192, halo-tile A operand
        if (a.Cout % 192 == 0) return 161;          // 128 x 192
        // the detector's / the ResNet trunk's plain 3x3 layers (same k order in every tile shape: bit-identical). At 256 frames:
        // 64 outputs on 256 x 64 tiles instead of half-empty 128 x 128 ones (32 -> 64 @128: 417 vs 584 us, 64 -> 64 @64: 157 vs
        // 227); widening layers from 512 outputs on 256 x 128 (256 -> 512 @16: 193 vs 227, 512 -> 1024 @8: 174 vs 218)
        if (a.Cout == 64 && a.M >= 32768) return 165;                                             // 256 x  64
        if (a.Cout >= 512 && a.Cout % 128 == 0 && a.Cout > a.Cin && a.M >= 16384) return 164;     // 256 x 128
        return 162;                                 // 128 x 128
    }
    if (!a.gate && a.zeros) {                       // general LDS-DMA kernel
        if (a.Cout == 32) return 59;                // 256 x  32, 8 waves
        if (a.Cout % 192 == 0) return 54;           // 128 x 192, 8 waves of 32 x 96
        if (a.Cout % 128 == 0) return 55;           // 128 x 128, 8 waves of 32 x 64
        if (a.Cout == 64) return 57;                // 256 x  64
        if (a.Cout == 224) return 14;
        return 55;
    }
    // register-staged fallback (a gate on tiles that do not align with samples, or no zero line)
    if (a.Cout % 128 == 0) return 1;
    if (a.Cout % 64 == 0) return 3;
    return 5;
}

static int launch_conv_igemm_impl(const ConvArgs& a, hipStream_t st) {
    ConvArgs aa = a;
    aa.exp = exp_flags();
    aa.grid_mode = 1;               // 1-D grid decoded per XCD (conv_tile_origin): +4 % over the 2-D grid
    if (a.Cin % 32 != 0 || a.Cout % 32 != 0 || a.K != a.KH * a.KW * a.Cin || a.M <= 0) {
        set_error("conv_igemm: unsupported shape Cin=%d Cout=%d K=%d M=%d", a.Cin, a.Cout, a.K, a.M);
        return ISB_ERR_INVALID;
    }
    if (a.gate && (a.KH != 1 || a.stride != 1)) {
        set_error("conv_igemm: SE gate only on 1x1 convolutions");
        return ISB_ERR_INVALID;
    }
    const int v = a.variant ? a.variant : choose_variant(a);
    const bool is_g1 = (v >= 131 && v <= 140) || v == 150, is_g1g = (v >= 141 && v <= 149) || (v >= 152 && v <= 156) || (v >= 191 && v <= 197);
    const bool is_wsk = v == 157;
    const bool is_ws = v >= 181 && v <= 188, is_c3 = v >= 161 && v <= 171;
    if (a.act_after_res && (a.act < 2 || a.out_f32 || aa.splits > 1 || v == 171 || is_ws || is_wsk)) {
        set_error("conv_igemm: act_after_res takes act 2-4 on the kernels with the shared 16-bit epilogue (variant %d)", v);
        return ISB_ERR_INVALID;
    }
    if (a.out_ld && (a.out_ld < a.Cout || a.out_ld % 8 != 0 || a.out_f32 || aa.splits > 1 || v == 171 || v == 149 || is_ws || is_wsk)) {
        set_error("conv_igemm: out_ld (a channel slice of a wider 16-bit tensor) needs out_ld >= Cout, a multiple of 8, and a kernel with the shared epilogue (variant %d)", v);
        return ISB_ERR_INVALID;
    }
    if (a.f16 && !(is_g1 || is_g1g || is_ws || is_c3 || is_wsk)) {
        set_error("conv_igemm: fp16 operands are implemented by the lean kernels (1x1: 131 / 132 / 138, gated 141 - 156, weights-stationary 184 - 186; 3x3: 161 / 163 / 167 / 169 / 171), got variant %d", v);
        return ISB_ERR_INVALID;
    }
    if (aa.splits > 1 && !((is_g1 || (is_g1g && v < 152)) && aa.part)) {
        set_error("conv_igemm: split-K is implemented by the gemm1x1 variants (131-150) and needs a partial buffer");
        return ISB_ERR_INVALID;
    }
    if (v < 131 && ((v >= 11 && v <= 69) || (v >= 101 && v <= 109)) && (a.gate || !a.zeros)) {
        set_error("conv_igemm: the plain LDS-DMA variants take no SE gate and need the zero line");
        return ISB_ERR_INVALID;
    }
    if (((v >= 81 && v <= 99) || (v >= 111 && v <= 119)) && (!a.gate || !a.zeros)) {
        set_error("conv_igemm: variants 81-99 and 111-119 are the gated LDS-DMA kernels");
        return ISB_ERR_INVALID;
    }
    if (((v >= 101 && v <= 109) || (v >= 111 && v <= 119)) && a.Cin % 64 != 0) {
        set_error("conv_igemm: the 64-wide k-tile variants need Cin %% 64 == 0 (Cin=%d)", a.Cin);
        return ISB_ERR_INVALID;
    }
    int rc;
#ifdef ISB_BUILD_PROBES
    if (is_wsk) rc = launch_conv_wsk(a, aa, st);
    else
#else
    if (is_wsk) {
        set_error("conv_igemm: variant 157 (gated projection with stationary weights: 2x slower than the tile kernels) exists in probe builds only");
        return ISB_ERR_INVALID;
    }
#endif
    if (is_ws) rc = launch_conv_ws(a, aa, v, st);
    else if (is_g1) rc = launch_tiles_gemm1x1(v, a, aa, st);
    else if (is_g1g) rc = launch_tiles_gemm1x1_gate(v, a, aa, st);
    else if (is_c3) rc = launch_tiles_conv3x3(v, a, aa, st);
    else rc = launch_tiles_igemm(v, a, aa, st);
    if (rc != ISB_OK) return rc;
    ISB_LAUNCHED("conv_igemm", st);
    return ISB_OK;
}

// public entry: one convolution, or (splits > 1) a split-K GEMM into f32 partials followed by the reduction
int launch_conv_igemm(const ConvArgs& a, hipStream_t st) {
    if (a.splits <= 1) return launch_conv_igemm_impl(a, st);
    ConvArgs b = a;
    const int nkt = a.Cin / CK;
    const int per = cdiv(nkt, a.splits);
    b.splits = cdiv(nkt, per);                  // no empty split: every workgroup owns at least one k-tile
    if (b.splits <= 1) {
        b.splits = 0;
        return launch_conv_igemm_impl(b, st);
    }
    if (a.out_f32 || a.KH != 1 || a.stride != 1) {
        set_error("conv_igemm: split-K needs a 1x1 stride-1 convolution with bf16 output");
        return ISB_ERR_INVALID;
    }
    const int rc = launch_conv_igemm_impl(b, st);
    if (rc != ISB_OK) return rc;
    return launch_splitk_reduce(b, st);
}

}  // namespace isb
