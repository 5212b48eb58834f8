// SE-gated 1x1 projections (tile variants 141 - 149, 152 - 156, 191 - 197): gemm1x1_dma_kernel<.., GATE> (k_gemm1x1.h) and
// the loader-wave kernel of the 8 x 8 stages.
#include "k_gemm1x1.h"

namespace isb {

// -------------------------------------------------------------------------------------------
// SE-gated projection with LOADER WAVES (round 3; variants 155 / 156): the 8 x 8 stages' 2304 -> 384 and 3840 -> 640 GEMMs.
// Their k loops are bound by what a CU receives from L2 (EXPERIMENTS.md: FLOP per byte of the tile against 127): the 64 x 192
// tiles of gemm1x1_dma_kernel cap the matrix pipe at 38 %. A CU's share of these layers is 24 576 outputs; the tile that
// moves the fewest bytes for them is 128 x 192 (320 rows per k-step instead of 2 x 256), ONE workgroup per CU -- which the
// tile kernel cannot use, because with nobody beside it every DMA-issue stall (~90 cycles per 1-KiB piece, in the issuing
// wave's stream), every landing wait and its epilogue are exposed. Here the workgroup has 8 CONSUMER waves (fragment reads,
// gate, MFMAs: nothing else in their stream) and 4 LOADER waves that only issue LDS-DMA, NB - 1 k-steps ahead through a ring
// of NB buffers, and wait (counted vmcnt) for the k-step the consumers take next; all twelve meet at ONE barrier per k-step.
// The loaders end after the loop (an ended wave is no longer a party to s_barrier); the consumers run the shared epilogue.
// Same LDS images, fragment reads, gate arithmetic and k order as gemm1x1_dma_kernel<.., GATE = 1>: bit-identical (tested).
// -------------------------------------------------------------------------------------------
// WGM x WGN consumer waves of 32 x (32 TN) sub-tiles + 4 loader waves; NP pair buffers in the ring
template <int WGM, int WGN, int TN, int NP, bool F16>
__global__ __launch_bounds__(64 * (WGM * WGN + 4)) void gemm1x1_lw_kernel(ConvArgs p) {
    T16<F16>::enter();
    constexpr int NCW = WGM * WGN, NLW = 4;                            // consumer / loader waves
    constexpr int NTH = 64 * (NCW + NLW);
    constexpr int BM = 32 * WGM, BN = 32 * TN * WGN;
    constexpr int PIECES = ((BM + BN) / 16 + NLW - 1) / NLW * NLW, PPL = PIECES / NLW;   // 1-KiB pieces per k-step (B rows padded
    constexpr int BUF = PIECES * 16 * ROWB;                            // up to whole rounds of the loaders: immediate vmcnt), per loader
    // the loop runs on PAIRS of k-steps (one barrier per 64 channels: the twelve waves' meeting costs as much as a k-step's
    // MFMAs): ring of NP pair buffers, NP - 1 pairs in flight
    constexpr int PBUF = 2 * BUF;
    constexpr int GATE_OFF = NP * PBUF;
    unsigned char* const lds = conv_lds_dyn;
    const int bias_off = p.grid_bias_off;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    int m0, n0;
    if (!conv_tile_origin(p, BM, BN, m0, n0)) return;
    const int npair = p.Cin / (2 * CK);                                // launcher: Cin % 64 == 0
    const int ohw = p.OH * p.OW;
    // gate rows of the tile's samples + the bias row: staged by everybody with ordinary loads / one DMA, published by the first barrier
    {
        const int s_first = m0 / ohw;
        const int ns = min(m0 + BM - 1, p.M - 1) / ohw - s_first + 1;
        const float* src = p.gate + (size_t)s_first * p.Cin;
        float* dst = reinterpret_cast<float*>(lds + GATE_OFF);
        for (int idx = tid * 4; idx < ns * p.Cin; idx += NTH * 4)
            *reinterpret_cast<float4*>(dst + idx) = *reinterpret_cast<const float4*>(src + idx);
    }
    if (wave >= NCW) {
        // ---------------------------------------------------------------- loader waves
        const int lw = wave - NCW;
        uint32_t voff[PPL];
        uint32_t ldst[PPL];
#pragma unroll
        for (int s = 0; s < PPL; ++s) {
            const int piece = lw + NLW * s;                           // rows 16 piece .. + 15 of the [A rows | B rows] image
            const int row = 16 * piece + (lane >> 2);
            const int logical = (lane & 3) ^ ((row >> 2) & 3);
            if (row < BM) voff[s] = (uint32_t)min(m0 + row, p.M - 1) * (uint32_t)(p.Cin * 2) + logical * 16;
            else voff[s] = (uint32_t)min(n0 + row - BM, p.Cout - 1) * (uint32_t)(p.Cin * 2) + logical * 16;
            ldst[s] = piece * 1024;
        }
        if (lw == 0) {
#pragma unroll
            for (int o = 0; o < BN / 4; o += 64)
                if (lane + o < BN / 4)
                    dma16_s(p.bias, (uint32_t)min(n0 + (lane + o) * 4, p.Cout - 4) * 4, (uint32_t)(uintptr_t)(lds_ptr_t)lds + bias_off + o * 16);
        }
        const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)lds;
        const unsigned char* a_base = reinterpret_cast<const unsigned char*>(p.in);
        const unsigned char* b_base = reinterpret_cast<const unsigned char*>(p.w);
        auto issue = [&](int pr) {                                    // pair pr (k-steps 2 pr, 2 pr + 1) into pair buffer pr % NP
            const uint32_t boff = lds0 + (uint32_t)(pr % NP) * PBUF;
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int s = 0; s < PPL; ++s) {
                    const bool is_a = 16 * (lw + NLW * s) < BM;
                    dma16_s((is_a ? a_base : b_base) + (size_t)(2 * pr + half) * (CK * 2), voff[s], boff + half * BUF + ldst[s]);
                }
        };
        // prologue: NP - 1 pairs in flight; pair 0 landed before the first barrier
        for (int t = 0; t < NP - 1; ++t)
            if (t < npair) issue(t);
        if (min(NP - 1, npair) >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPL) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                               // gates, bias, pair 0 visible to the consumers
        for (int pr = 0; pr < npair; ++pr) {
            // pair buffer (pr - 1) % NP was released by the barrier that ended iteration pr - 1 (at pr = 0 it is still empty)
            if (pr + NP - 1 < npair) issue(pr + NP - 1);
            // pair pr + 1 has to have landed before the barrier; the newer one (NP = 3) may fly
            const int newest = min(npair - 1, pr + NP - 1);
            if (newest - (pr + 1) >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPL) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;                                                        // the consumers finish alone
    }
    // -------------------------------------------------------------------- consumer waves
    const int wm = wave / WGN, wn = wave % WGN;
    f32x16 acc[1][TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][j][e] = 0.f;
    int a_sw[2], b_sw[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        a_sw[ks] = swz(wm * 32 + r, 2 * ks + h);
        b_sw[ks] = BM * ROWB + swz(wn * TN * 32 + r, 2 * ks + h);
    }
    const int s_first = m0 / ohw;
    const int g_row = (min(m0 + wm * 32 + r, p.M - 1) / ohw - s_first) * p.Cin + 8 * h;
    __syncthreads();                                                   // (the loaders' first barrier; also publishes this wave's gate rows)
    for (int pr = 0; pr < npair; ++pr) {
        const unsigned char* bufp = lds + (pr % NP) * PBUF;
        // the four 16-channel steps of the pair: all A fragments and gate rows first (one LDS round trip), then per step the B
        // fragments and the MFMAs
        uint4 af[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            af[q] = *reinterpret_cast<const uint4*>(bufp + (q >> 1) * BUF + a_sw[q & 1]);
            const float* gs = reinterpret_cast<const float*>(lds + GATE_OFF) + g_row + pr * (2 * CK) + q * 16;
            const float4 g0 = *reinterpret_cast<const float4*>(gs), g1 = *reinterpret_cast<const float4*>(gs + 4);
            af[q] = T16<F16>::gate8(af[q], g0, g1);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint4 bfr[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const uint4*>(bufp + (q >> 1) * BUF + b_sw[q & 1] + j * 2048);
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[0][j] = T16<F16>::mfma32(bfr[j], af[q], acc[0][j]);
        }
        __builtin_amdgcn_s_barrier();
    }
    conv_epilogue<1, TN, WGM, WGN, true, F16>(p, acc, lds, m0, n0, wm, wn, r, h, tid, bias_off);
}

int launch_tiles_gemm1x1_gate(int v, const ConvArgs& a, ConvArgs& aa, hipStream_t st) {
    // gated tile GEMM: LDS = ring of NBUF k-step buffers + the f32 gate rows of the tile's samples | epilogue staging, + bias row
#define ISB_G1G(TM, TN, WGM, WGN, NBUF_, F16_)                                                                   \
    do {                                                                                                         \
        constexpr int BM_ = 32 * TM * WGM, BN_ = 32 * TN * WGN;                                                  \
        const int ohw = a.OH * a.OW;                                                                             \
        if (!a.gate || a.KH != 1 || a.stride != 1 || a.pad != 0 || (ohw % BM_ != 0 && BM_ % ohw != 0) ||        \
            (NBUF_ == 3 && (a.Cin % 96 != 0 || a.splits > 1))) {                                                 \
            set_error("conv_igemm: variants 141-149 / 191-197 are gated 1x1 GEMMs on sample-aligned tiles (19x: Cin %% 96 == 0)"); \
            return ISB_ERR_INVALID;                                                                              \
        }                                                                                                        \
        const int ns = BM_ > ohw ? BM_ / ohw : 1;                                                                \
        const int ring = NBUF_ * (BM_ + BN_) * ROWB + ns * a.Cin * 4;                                            \
        const int stage = BM_ * (BN_ * 2 + 16);                                                                  \
        aa.grid_bias_off = ring > stage ? ring : stage;                                                          \
        const int bytes = aa.grid_bias_off + BN_ * 4;                                                            \
        auto kern = gemm1x1_dma_kernel<TM, TN, WGM, WGN, true, false, NBUF_, F16_>;                              \
        static DevMax attr_bytes;                                                                               \
        if (attr_bytes.below(bytes)) {                                                                                \
            ISB_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));  \
            attr_bytes.set(bytes);                                                                                  \
        }                                                                                                        \
        const dim3 g = conv_grid(aa, BM_, BN_);                                                                  \
        hipLaunchKernelGGL(kern, g, dim3(64 * WGM * WGN), bytes, st, aa);                                        \
    } while (0)
    // both 16-bit operand types / bf16 only
#define ISB_G1GH(TM, TN, WGM, WGN)                                                                               \
    do {                                                                                                         \
        if (a.probe & 2) ISB_G1G_STAMPS(TM, TN, WGM, WGN);                                                       \
        else if (a.f16) ISB_G1G(TM, TN, WGM, WGN, 2, true);                                                      \
        else ISB_G1G(TM, TN, WGM, WGN, 2, false);                                                                \
    } while (0)
#define ISB_G1GB(TM, TN, WGM, WGN, NBUF_)                                                                        \
    do {                                                                                                         \
        if (a.f16) { set_error("conv_igemm: tile variant %d has no fp16 form", v); return ISB_ERR_INVALID; }     \
        if ((a.probe & 2) && NBUF_ == 2) ISB_G1G_STAMPS(TM, TN, WGM, WGN);                                       \
        else ISB_G1G(TM, TN, WGM, WGN, NBUF_, false);                                                            \
    } while (0)
#ifdef ISB_BUILD_PROBES
#define ISB_G1G_STAMPS(TM, TN, WGM, WGN)                                                                         \
    do {                                                                                                         \
        constexpr int BM_ = 32 * TM * WGM, BN_ = 32 * TN * WGN;                                                  \
        const int ohw = a.OH * a.OW;                                                                             \
        if (!a.gate || a.f16 || (ohw % BM_ != 0 && BM_ % ohw != 0)) { set_error("conv_igemm: stamps: gated bf16 sample-aligned tiles"); return ISB_ERR_INVALID; } \
        const int ns = BM_ > ohw ? BM_ / ohw : 1;                                                                \
        const int ring = 2 * (BM_ + BN_) * ROWB + ns * a.Cin * 4;                                                \
        const int stage = BM_ * (BN_ * 2 + 16);                                                                  \
        aa.grid_bias_off = ring > stage ? ring : stage;                                                          \
        const int bytes = aa.grid_bias_off + BN_ * 4;                                                            \
        auto kern2 = gemm1x1_dma_kernel<TM, TN, WGM, WGN, true, true, 2>;                                        \
        ISB_HIP(hipFuncSetAttribute((const void*)kern2, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));     \
        const dim3 g = conv_grid(aa, BM_, BN_);                                                                  \
        hipLaunchKernelGGL(kern2, g, dim3(64 * WGM * WGN), bytes, st, aa);                                       \
    } while (0)
#else
#define ISB_G1G_STAMPS(TM, TN, WGM, WGN) do { set_error("conv_igemm: s_memtime stamps need a -DISB_BUILD_PROBES build"); return ISB_ERR_INVALID; } while (0)
#endif
    switch (v) {
        case 141: ISB_G1GH(1, 3, 4, 2); break;      // 128 x 192, SE gate on the A fragments
        case 142: ISB_G1GB(1, 2, 4, 2, 2); break;   // 128 x 128
        case 143: ISB_G1GH(1, 7, 4, 1); break;      // 128 x 224
        case 144: ISB_G1GH(1, 5, 4, 2); break;      // 128 x 320
        case 146: ISB_G1GH(1, 3, 2, 2); break;      //  64 x 192
        case 147: ISB_G1GH(1, 2, 2, 2); break;      //  64 x 128
        case 149: {                                  //  64 x 128, split-K, squeeze-excite FC2 folded in
            constexpr int BM_ = 64, BN_ = 128;
            const int nkt_ = a.Cin / CK, per_ = aa.splits > 1 ? cdiv(nkt_, aa.splits) : nkt_;
            if (a.KH != 1 || a.stride != 1 || a.pad != 0 || (a.OH * a.OW) % BM_ != 0 || aa.splits < 2 || per_ * CK > 256 ||
                !a.se_part || !a.se_b1 || !a.se_w2t || !a.se_b2 || a.se_cse < 1 || a.se_cse > 160 || a.se_nparts < 1 ||
                a.se_nparts > SE_MAX_PARTS) {
                set_error("conv_igemm: variant 149 is a split-K gated 1x1 GEMM (k-range <= 256 channels per split) with FC1 partials");
                return ISB_ERR_INVALID;
            }
            const int ring = 2 * (BM_ + BN_) * ROWB + (256 + 160) * 4 + 4 * 64 * 16;
            const int stage = BM_ * (BN_ * 2 + 16);
            aa.grid_bias_off = ring > stage ? ring : stage;
            const int bytes = aa.grid_bias_off + BN_ * 4;
            auto kern = gemm1x1_dma_kernel<1, 2, 2, 2, 2>;
            auto kern_h = gemm1x1_dma_kernel<1, 2, 2, 2, 2, false, 2, true>;
            static DevOnce attr_set;
            if (attr_set.need()) {
                ISB_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
                ISB_HIP(hipFuncSetAttribute((const void*)kern_h, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
                attr_set.mark();
            }
            const dim3 g = conv_grid(aa, BM_, BN_);
            if (a.f16) hipLaunchKernelGGL(kern_h, g, dim3(256), bytes, st, aa);
            else hipLaunchKernelGGL(kern, g, dim3(256), bytes, st, aa);
            break;
        }
        case 155: case 156: {                                // loader-wave GEMMs (gemm1x1_lw_kernel), one workgroup per CU:
            // 155: 128 x 192, 156: 128 x 320 (the gated projections of the 8 x 8 stages)
            const int bm = 128, bn = v == 155 ? 192 : 320;
            const int np_ = v == 155 ? 3 : 2;
            const int pieces = (((bm + bn) / 16 + 3) / 4) * 4;
            const int ohw_ = a.OH * a.OW;
            if (!a.gate || a.KH != 1 || a.stride != 1 || a.pad != 0 || (ohw_ % bm != 0 && bm % ohw_ != 0) || a.splits > 1 || a.out_f32 ||
                a.Cout % 64 != 0 || a.Cin % 64 != 0 || a.Cin < 256 || (size_t)a.M * a.Cin * 2 >= 0xffffffffull) {
                set_error("conv_igemm: variants 155 / 156 are gated 1x1 GEMMs (Cin %% 64 == 0, >= 256) on sample-aligned 128-row tiles, bf16 / fp16 output");
                return ISB_ERR_INVALID;
            }
            const int ns = bm > ohw_ ? bm / ohw_ : 1;
            const int ring = np_ * 2 * pieces * 16 * ROWB + ns * a.Cin * 4;
            const int stage = bm * (bn * 2 + 16);
            aa.grid_bias_off = ring > stage ? ring : stage;
            const int bytes = aa.grid_bias_off + bn * 4;
            if (bytes > 160 * 1024) {
                set_error("conv_igemm: variant %d needs %d bytes of LDS (K = %d)", v, bytes, a.Cin);
                return ISB_ERR_INVALID;
            }
            const dim3 g = conv_grid(aa, bm, bn);
#define ISB_LW_GO(WGM_, WGN_, TN_, NP_, F16_)                                                                            \
    do {                                                                                                                 \
        auto kern = gemm1x1_lw_kernel<WGM_, WGN_, TN_, NP_, F16_>;                                                       \
        static DevMax attr_bytes;                                                                                       \
        if (attr_bytes.below(bytes)) {                                                                                        \
            ISB_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));          \
            attr_bytes.set(bytes);                                                                                          \
        }                                                                                                                \
        hipLaunchKernelGGL(kern, g, dim3(64 * (WGM_ * WGN_ + 4)), bytes, st, aa);                                        \
    } while (0)
            if (v == 155) { if (a.f16) ISB_LW_GO(4, 2, 3, 3, true); else ISB_LW_GO(4, 2, 3, 3, false); }
            else { if (a.f16) ISB_LW_GO(4, 2, 5, 2, true); else ISB_LW_GO(4, 2, 5, 2, false); }
#undef ISB_LW_GO
            break;
        }
#ifdef ISB_BUILD_PROBES
        case 145: ISB_G1GB(2, 2, 4, 2, 2); break;   // 256 x 128
        case 148: ISB_G1GB(2, 7, 4, 1, 2); break;   // 256 x 224
        case 152: ISB_G1GB(1, 7, 8, 1, 2); break;   // 256 x 224, eight waves
        case 153: ISB_G1GB(1, 6, 8, 1, 2); break;   // 256 x 192, eight waves
        case 191: ISB_G1GB(1, 3, 4, 2, 3); break;   // three k-step buffers: 128 x 192
        case 193: ISB_G1GB(1, 7, 4, 1, 3); break;   // 128 x 224
        case 194: ISB_G1GB(1, 5, 4, 2, 3); break;   // 128 x 320
        case 196: ISB_G1GB(1, 3, 2, 2, 3); break;   //  64 x 192
        case 197: ISB_G1GB(1, 2, 2, 2, 3); break;   //  64 x 128
#endif
        default:
            set_error("conv_igemm: tile variant %d is not in this build (gated 1x1: 141 - 144, 146, 147, 149, 155, 156)", v);
            return ISB_ERR_INVALID;
    }
#undef ISB_G1G
#undef ISB_G1GH
#undef ISB_G1GB
#undef ISB_G1G_STAMPS
    return ISB_OK;
}

}  // namespace isb
