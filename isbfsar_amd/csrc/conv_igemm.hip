// General implicit-GEMM convolution kernels (any tap count / stride / padding) on v_mfma_f32_32x32x16_bf16:
//   conv_igemm_kernel     : register-staged tiles (any shape, SE gate applied at the LDS store); the fallback for gated layers
//                           whose tiles do not align with samples and for callers without a zero line
//   conv_igemm_dma_kernel : tiles global -> LDS by LDS-DMA, per-k-step address arithmetic, 16-byte zero line for padding;
//                           what the detector's / ResNet trunk's strided 1x1 and odd-shaped layers run on
// The lean kernels (conv_gemm1x1*.hip, conv_3x3.hip, conv_fused_mb.hip, conv_ws.hip) take every shape the pose backbone has.
#include "conv_tiles.h"

namespace isb {

template <int TM, int TN, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_igemm_kernel(ConvArgs p) {
    constexpr int NT = 64 * WGM * WGN;                    // threads: 4 or 8 waves
    constexpr int BM = 32 * TM * WGM;
    constexpr int BN = 32 * TN * WGN;
    constexpr int A_ROWS = NT / 4;                        // tile rows staged per pass (4 x 16-B chunks per row)
    constexpr int A_PASS = (BM + A_ROWS - 1) / A_ROWS;
    constexpr int B_CHUNKS = BN * 4;
    constexpr int B_PASS = (B_CHUNKS + NT - 1) / NT;
    constexpr int BUF = (BM + BN) * ROWB;
    constexpr int CROW = BN * 2 + 16;                     // staged C tile row (bf16) + pad: conflict-free b64 writes
    constexpr int LDS_BYTES = (2 * BUF > BM * CROW || BM * CROW > 65536) ? 2 * BUF : BM * CROW;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int r = lane & 31, h = lane >> 5;
    int m0, n0;
    if (!conv_tile_origin(p, BM, BN, m0, n0)) return;
    const int chunk = tid & 3;

    // ---- per-thread A rows (fixed over the k loop): pixel coordinates, 32-bit element offsets
    int a_off[A_PASS], a_iy[A_PASS], a_ix[A_PASS], a_gate[A_PASS];
    const int ohw = p.OH * p.OW;
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
        const int m = m0 + (tid >> 2) + A_ROWS * i;
        const bool ok = m < p.M && (tid >> 2) + A_ROWS * i < BM;
        const int mm = ok ? m : 0;
        const int b = mm / ohw, rem = mm - b * ohw;
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        a_iy[i] = ok ? oy * p.stride - p.pad : -100000;   // rows past M never pass the bounds test
        a_ix[i] = ox * p.stride - p.pad;
        a_off[i] = ok ? ((b * p.H + a_iy[i]) * p.W + a_ix[i]) * p.Cin + chunk * 8 : 0;
        a_gate[i] = b * p.Cin + chunk * 8;
    }
    uint4 rA[1][A_PASS], rB[1][B_PASS];
    float4 gA[A_PASS][2];
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) gA[i][0] = gA[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    using St0 = std::integral_constant<int, 0>;

    auto gload = [&](int kt, auto stage) {
        constexpr int SG = decltype(stage)::value;
        const int k0 = kt * CK;
        const int tap = k0 / p.Cin;
        const int c0 = k0 - tap * p.Cin;
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
        const int tap_off = (ky * p.W + kx) * p.Cin + c0;
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) {
            const int iy = a_iy[i] + ky, ix = a_ix[i] + kx;
            uint4 v = zero4;
            if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
                v = *reinterpret_cast<const uint4*>(p.in + (a_off[i] + tap_off));
                if (p.gate) {      // squeeze-excite gate (1x1 convs only): fetched now, applied when the tile is
                                   // written to LDS, so neither load is waited for before the MFMAs of this step
                    const float* g = p.gate + (a_gate[i] + c0);
                    gA[i][0] = *reinterpret_cast<const float4*>(g);
                    gA[i][1] = *reinterpret_cast<const float4*>(g + 4);
                }
            }
            rA[SG][i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_PASS; ++i) {
            const int cidx = tid + NT * i;
            uint4 v = zero4;
            if (cidx < B_CHUNKS) {
                const int n = n0 + (cidx >> 2);
                if (n < p.Cout) v = *reinterpret_cast<const uint4*>(p.w + (size_t)n * p.K + k0 + (cidx & 3) * 8);
            }
            rB[SG][i] = v;
        }
    };
    auto lstore = [&](int buf, auto stage) {
        constexpr int SG = decltype(stage)::value;
        unsigned char* As = lds + buf * BUF;
        unsigned char* Bs = As + BM * ROWB;
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) {
            uint4 v = rA[SG][i];
            if (p.gate) {
                v = gate_bf16x8(v, gA[i][0], gA[i][1]);
            }
            if ((tid >> 2) + A_ROWS * i < BM) *reinterpret_cast<uint4*>(As + swz((tid >> 2) + A_ROWS * i, chunk)) = v;
        }
#pragma unroll
        for (int i = 0; i < B_PASS; ++i) {
            const int cidx = tid + NT * i;
            if (cidx < B_CHUNKS) *reinterpret_cast<uint4*>(Bs + swz(cidx >> 2, cidx & 3)) = rB[SG][i];
        }
    };

    // accumulators hold the TRANSPOSED tile: D = W_tile (rows n) x X_tile^T (cols m), so a lane owns one
    // pixel (m = lane&31) and 4 consecutive channels per register quad -> 8-byte packed bf16 pieces
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nkt = p.K / CK;
    auto compute = [&](int cur) {
        const unsigned char* As = lds + cur * BUF;
        const unsigned char* Bs = As + BM * ROWB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(As + swz((wm * TM + i) * 32 + r, 2 * ks + h)));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bfr[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Bs + swz((wn * TN + j) * 32 + r, 2 * ks + h)));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
    };
    gload(0, St0{});
    lstore(0, St0{});
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) gload(kt + 1, St0{});            // in flight during this tile's MFMAs
        compute(cur);
        if (kt + 1 < nkt) lstore(cur ^ 1, St0{});
        __syncthreads();
    }

    conv_epilogue<TM, TN, WGM, WGN>(p, acc, lds, m0, n0, wm, wn, r, h, tid);
}


template <int TM, int TN, int WGM, int WGN, int NB, bool GATE = false, int KT = 32>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_igemm_dma_kernel(ConvArgs p) {
    constexpr int NW = WGM * WGN;
    constexpr int BM = 32 * TM * WGM;
    constexpr int BN = 32 * TN * WGN;
    constexpr int ROWB = KT * 2;                          // bytes per LDS row (shadows the 64-B global)
    constexpr int CPRW = ROWB / 16;                       // 16-B chunks per row
    constexpr int RPP = 1024 / ROWB;                      // tile rows per 1-KiB DMA piece
    constexpr int A_INST = BM / RPP, B_INST = BN / RPP;   // pieces per tile
    constexpr int A_PW = (A_INST + NW - 1) / NW, B_PW = (B_INST + NW - 1) / NW;
    constexpr int BUF = (BM + BN) * ROWB;
    constexpr int CROW = BN * 2 + 16;
    constexpr int LDS_PLAIN = (NB * BUF > BM * CROW || BM * CROW > 65536) ? NB * BUF : BM * CROW;
    constexpr int LDS_BYTES = LDS_PLAIN;
    // counted vmcnt waits (NB > 2) need the same number of DMA instructions in every wave: waves with no
    // piece left in a pass copy the zero line into a 1-KiB dump area behind the ring
    constexpr bool PAD_DMA = NB > 2 && (A_INST % NW != 0 || B_INST % NW != 0);
    constexpr int DUMP_OFF = NB * BUF;
    constexpr int GATE_OFF = DUMP_OFF + 1024;               // GATE: f32 gate rows of the tile's samples (dynamic LDS)
    constexpr int BIAS_OFF = LDS_BYTES + (PAD_DMA ? 1024 : 0);   // bias row behind everything else (plain, un-gated variants)
    __shared__ __attribute__((aligned(16))) unsigned char lds_static[GATE ? 16 : BIAS_OFF + BN * 4];
    unsigned char* const lds = GATE ? conv_lds_dyn : lds_static;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int r = lane & 31, h = lane >> 5;
    int m0, n0;
    if (!conv_tile_origin(p, BM, BN, m0, n0)) return;

    // lane -> (tile row, logical chunk) of the 1-KiB piece it fills: LDS position lane*16 holds
    // physical chunk lane&3 of row lane>>2; the source chunk is the inverse swizzle of that
    int a_off[A_PW], a_iy[A_PW], a_ix[A_PW], b_off[B_PW];
    const int ohw = p.OH * p.OW;
#pragma unroll
    for (int s = 0; s < A_PW; ++s) {
        const int row = RPP * (wave + NW * s) + lane / CPRW;
        const int logical = KT == 32 ? ((lane & 3) ^ ((row >> 2) & 3)) : ((lane & 7) ^ ((row >> 1) & 7));
        const int m = m0 + row;
        const bool ok = m < p.M && row < BM;
        const int mm = ok ? m : 0;
        const int b = mm / ohw, rem = mm - b * ohw;
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        a_iy[s] = ok ? oy * p.stride - p.pad : -100000;
        a_ix[s] = ox * p.stride - p.pad;
        a_off[s] = ok ? ((b * p.H + a_iy[s]) * p.W + a_ix[s]) * p.Cin + logical * 8 : 0;
    }
#pragma unroll
    for (int s = 0; s < B_PW; ++s) {
        const int row = RPP * (wave + NW * s) + lane / CPRW;
        const int logical = KT == 32 ? ((lane & 3) ^ ((row >> 2) & 3)) : ((lane & 7) ^ ((row >> 1) & 7));
        const int n = n0 + row;
        b_off[s] = (n < p.Cout && row < BN) ? n * p.K + logical * 8 : -1;
    }

    if constexpr (!GATE) {              // the tile's bias row rides along with the first k-step
        if (wave == 0) {
#pragma unroll
            for (int o = 0; o < BN / 4; o += 64)
                if (lane + o < BN / 4) {
                    const float* src = p.bias + min(n0 + (lane + o) * 4, p.Cout - 4);
                    // low 32 bits of a generic LDS pointer = the LDS byte address (an addrspacecast here trips a
                    // backend verifier error in ROCm 7.2)
                    dma16_at(src, (uint32_t)(uintptr_t)lds + (BIAS_OFF + o * 16));
                }
        }
    }
    auto dma = [&](int kt, int buf) {
        const int k0 = kt * KT;
        const int tap = k0 / p.Cin;
        const int c0 = k0 - tap * p.Cin;
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
        const int tap_off = (ky * p.W + kx) * p.Cin + c0;
        unsigned char* base = lds + buf * BUF;
#pragma unroll
        for (int s = 0; s < A_PW; ++s) {
            const int q = wave + NW * s;
            if (q < A_INST) {
                const int iy = a_iy[s] + ky, ix = a_ix[s] + kx;
                const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const uint16_t* src = ok ? p.in + (a_off[s] + tap_off) : p.zeros;
                dma16(src, base + q * 1024);
            } else if constexpr (PAD_DMA) {
                dma16(p.zeros, lds + DUMP_OFF);
            }
        }
#pragma unroll
        for (int s = 0; s < B_PW; ++s) {
            const int q = wave + NW * s;
            if (q < B_INST) {
                const uint16_t* src = b_off[s] >= 0 ? p.w + (b_off[s] + k0) : p.zeros;
                dma16(src, base + BM * ROWB + q * 1024);
            } else if constexpr (PAD_DMA) {
                dma16(p.zeros, lds + DUMP_OFF);
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // SE gate (GATE): the f32 gate rows of the samples this tile covers sit in LDS; the A fragment is scaled
    // as it leaves LDS, bf16(f32(x) * g) exactly as the register-staged kernel does at its LDS store
    int g_row[TM];
    if constexpr (GATE) {
        const int s_first = m0 / ohw;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = min(m0 + (wm * TM + i) * 32 + r, p.M - 1);
            g_row[i] = (m / ohw - s_first) * p.Cin + 8 * h;
        }
    }
    auto compute = [&](int cur, int kt) {
        const unsigned char* As = lds + cur * BUF;
        const unsigned char* Bs = As + BM * ROWB;
#pragma unroll
        for (int ks = 0; ks < KT / 16; ++ks) {
            bf16x8 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                uint4 v = *reinterpret_cast<const uint4*>(As + swz_kt<KT>((wm * TM + i) * 32 + r, 2 * ks + h));
                if constexpr (GATE) {
                    const float* gs = reinterpret_cast<const float*>(lds + GATE_OFF) + g_row[i] + kt * KT + ks * 16;
                    const float4 g0 = *reinterpret_cast<const float4*>(gs), g1 = *reinterpret_cast<const float4*>(gs + 4);
                    v = gate_bf16x8(v, g0, g1);
                }
                af[i] = __builtin_bit_cast(bf16x8, v);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bfr[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Bs + swz_kt<KT>((wn * TN + j) * 32 + r, 2 * ks + h)));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
    };

    const int nkt = p.K / KT;
    auto stage_gate = [&]() {      // after the first DMAs are in flight; the barrier that publishes tile 0 publishes this too
        if constexpr (GATE) {
            const int s_first = m0 / ohw;
            const int ns = min(m0 + BM - 1, p.M - 1) / ohw - s_first + 1;
            const float* src = p.gate + (size_t)s_first * p.Cin;
            float* dst = reinterpret_cast<float*>(lds + GATE_OFF);
            for (int idx = tid * 4; idx < ns * p.Cin; idx += 64 * NW * 4)
                *reinterpret_cast<float4*>(dst + idx) = *reinterpret_cast<const float4*>(src + idx);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // written before the (raw) barrier that publishes tile 0
        }
    };
    if constexpr (NB == 2) {
        dma(0, 0);
        stage_gate();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nkt) dma(kt + 1, cur ^ 1);     // in flight during this tile's MFMAs
            compute(cur, kt);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else if constexpr (NB == 3) {
        // three buffers, two tiles in flight: tile kt+2 is requested before tile kt is consumed and is
        // waited for one iteration later with a COUNTED vmcnt (the newest tile stays in flight across
        // the barrier). The barrier after the MFMAs also orders the next DMA (into the buffer just
        // consumed) behind every wave's reads.
        constexpr int PER = A_PW + B_PW;
        dma(0, 0);
        stage_gate();
        if (nkt > 1) {
            dma(1, 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt < nkt; ++kt) {
            if (kt + 2 < nkt) dma(kt + 2, (kt + 2) % 3);
            compute(kt % 3, kt);
            if (kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    } else {
        // ring of NB buffers, D = NB-2 tiles in flight beyond the one being consumed. One barrier per
        // tile, placed BEFORE the MFMAs; the DMA issued in iteration kt overwrites the buffer of tile
        // kt-2, which every wave finished before it arrived at barrier kt-1 (hence NB = D + 2).
        // Waits are counted (never vmcnt(0) in steady state) and the barrier is the raw s_barrier:
        // __syncthreads() would drain the DMA queue.
        constexpr int D = NB - 2;
        constexpr int PER = A_PW + B_PW;                    // DMA instructions per wave per tile
        if constexpr (GATE) {      // plain loads first: the counted waits below then cover them (loads return in order)
            stage_gate();
        }
#pragma unroll
        for (int t = 0; t < D; ++t)
            if (t < nkt) dma(t, t);
        for (int kt = 0; kt < nkt; ++kt) {
            if (kt + D < nkt) dma(kt + D, (kt + D) % NB);
            const int ahead = min(D, nkt - 1 - kt);         // tiles issued after tile kt
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            compute(kt % NB, kt);
        }
        __syncthreads();                                    // all MFMA reads done before the tile staging reuses LDS
    }
    conv_epilogue<TM, TN, WGM, WGN, !GATE>(p, acc, lds, m0, n0, wm, wn, r, h, tid, BIAS_OFF);
}

int launch_tiles_igemm(int v, const ConvArgs& a, ConvArgs& aa, hipStream_t st) {
#define ISB_CONV_LAUNCH(TM, TN, WGM, WGN)                                                              \
    do {                                                                                               \
        const dim3 g = conv_grid(aa, 32 * TM * WGM, 32 * TN * WGN);                                    \
        hipLaunchKernelGGL((conv_igemm_kernel<TM, TN, WGM, WGN>), g, dim3(64 * WGM * WGN), 0, st, aa); \
    } while (0)
#define ISB_CONV_LAUNCH_DMA(TM, TN, WGM, WGN)                                                                 \
    do {                                                                                                      \
        const dim3 g = conv_grid(aa, 32 * TM * WGM, 32 * TN * WGN);                                           \
        hipLaunchKernelGGL((conv_igemm_dma_kernel<TM, TN, WGM, WGN, 2>), g, dim3(64 * WGM * WGN), 0, st, aa); \
    } while (0)
    switch (v) {
        // ---- the set launch_conv_igemm selects
        case 1: ISB_CONV_LAUNCH(2, 2, 2, 2); break;         // 128 x 128, 4 waves
        case 3: ISB_CONV_LAUNCH(2, 1, 2, 2); break;         // 128 x  64
        case 5: ISB_CONV_LAUNCH(2, 1, 4, 1); break;         // 256 x  32
        case 75: ISB_CONV_LAUNCH(1, 1, 2, 2); break;        //  64 x  64 (small M)
        case 14: ISB_CONV_LAUNCH_DMA(1, 7, 4, 1); break;    // 128 x 224
        case 54: ISB_CONV_LAUNCH_DMA(1, 3, 4, 2); break;    // 128 x 192, 8 waves of 32 x 96
        case 55: ISB_CONV_LAUNCH_DMA(1, 2, 4, 2); break;    // 128 x 128, 8 waves of 32 x 64
        case 57: ISB_CONV_LAUNCH_DMA(1, 2, 8, 1); break;    // 256 x  64
        case 59: ISB_CONV_LAUNCH_DMA(1, 1, 8, 1); break;    // 256 x  32
        case 64: ISB_CONV_LAUNCH_DMA(1, 1, 2, 2); break;    //  64 x  64: small-M launches (single frames) need many workgroups
#ifdef ISB_BUILD_PROBES
        // ---- tile shapes measured and not selected (EXPERIMENTS.md; tools/conv_sweep.py)
        case 2: ISB_CONV_LAUNCH(1, 3, 4, 1); break;         // 128 x  96
        case 4: ISB_CONV_LAUNCH(1, 7, 4, 1); break;         // 128 x 224
        case 6: ISB_CONV_LAUNCH(2, 2, 4, 2); break;         // 256 x 128, 8 waves
        case 7: ISB_CONV_LAUNCH(4, 2, 2, 4); break;         // 256 x 256, 8 waves
        case 8: ISB_CONV_LAUNCH(2, 3, 4, 2); break;         // 256 x 192, 8 waves
        case 9: ISB_CONV_LAUNCH(2, 1, 4, 2); break;         // 256 x  64, 8 waves
        case 41: ISB_CONV_LAUNCH(1, 6, 4, 1); break;        // 128 x 192: full-width tiles read the A operand once
        case 42: ISB_CONV_LAUNCH(1, 6, 4, 2); break;        // 128 x 384, 8 waves
        case 43: ISB_CONV_LAUNCH(1, 5, 4, 2); break;        // 128 x 320, 8 waves
        case 44: ISB_CONV_LAUNCH(1, 3, 4, 2); break;        // 128 x 192, 8 waves
        case 45: ISB_CONV_LAUNCH(1, 2, 4, 2); break;        // 128 x 128, 8 waves of 32 x 64
        case 47: ISB_CONV_LAUNCH(1, 4, 4, 2); break;        // 128 x 256, 8 waves
        case 48: ISB_CONV_LAUNCH(1, 7, 4, 2); break;        // 128 x 448, 8 waves
        case 71: ISB_CONV_LAUNCH(1, 3, 8, 2); break;        // 256 x 192, 16 waves
        case 72: ISB_CONV_LAUNCH(1, 3, 4, 4); break;        // 128 x 384, 16 waves
        case 73: ISB_CONV_LAUNCH(1, 5, 4, 4); break;        // 128 x 640, 16 waves
        case 74: ISB_CONV_LAUNCH(1, 2, 8, 2); break;        // 256 x 128, 16 waves
        case 76: ISB_CONV_LAUNCH(1, 2, 2, 2); break;        //  64 x 128 (small M)
        case 11: ISB_CONV_LAUNCH_DMA(2, 2, 2, 2); break;
        case 12: ISB_CONV_LAUNCH_DMA(1, 3, 4, 1); break;
        case 13: ISB_CONV_LAUNCH_DMA(2, 1, 2, 2); break;
        case 15: ISB_CONV_LAUNCH_DMA(2, 1, 4, 1); break;
        case 16: ISB_CONV_LAUNCH_DMA(2, 2, 4, 2); break;
        case 17: ISB_CONV_LAUNCH_DMA(4, 2, 2, 4); break;
        case 18: ISB_CONV_LAUNCH_DMA(2, 3, 4, 2); break;
        case 19: ISB_CONV_LAUNCH_DMA(2, 1, 4, 2); break;
        case 51: ISB_CONV_LAUNCH_DMA(1, 6, 4, 1); break;
        case 52: ISB_CONV_LAUNCH_DMA(1, 6, 4, 2); break;
        case 53: ISB_CONV_LAUNCH_DMA(1, 5, 4, 2); break;
        case 56: ISB_CONV_LAUNCH_DMA(1, 4, 4, 2); break;
        case 58: ISB_CONV_LAUNCH_DMA(1, 3, 8, 1); break;    // 256 x 96
        case 60: ISB_CONV_LAUNCH_DMA(1, 2, 8, 2); break;    // 256 x 128, 16 waves of 32 x 64
        case 61: ISB_CONV_LAUNCH_DMA(1, 3, 8, 2); break;    // 256 x 192, 16 waves of 32 x 96
        case 62: ISB_CONV_LAUNCH_DMA(1, 2, 4, 4); break;    // 128 x 256, 16 waves
        case 63: ISB_CONV_LAUNCH_DMA(1, 3, 4, 4); break;    // 128 x 384, 16 waves
        case 65: ISB_CONV_LAUNCH_DMA(1, 2, 2, 2); break;    //  64 x 128
#define ISB_CONV_LAUNCH_NB(TM, TN, WGM, WGN, NB)                                                                 \
    do {                                                                                                         \
        const dim3 g = conv_grid(aa, 32 * TM * WGM, 32 * TN * WGN);                                              \
        hipLaunchKernelGGL((conv_igemm_dma_kernel<TM, TN, WGM, WGN, NB>), g, dim3(64 * WGM * WGN), 0, st, aa);   \
    } while (0)
        case 31: ISB_CONV_LAUNCH_NB(2, 2, 2, 2, 3); break;  // 128 x 128, 4 waves, 3 buffers / 2 tiles in flight
        case 33: ISB_CONV_LAUNCH_NB(2, 1, 2, 2, 3); break;  // 128 x  64
        case 36: ISB_CONV_LAUNCH_NB(2, 2, 4, 2, 3); break;  // 256 x 128, 8 waves
        case 37: ISB_CONV_LAUNCH_NB(4, 2, 2, 4, 3); break;  // 256 x 256, 8 waves
        case 21: ISB_CONV_LAUNCH_NB(2, 2, 2, 2, 4); break;  // 128 x 128, 4 waves, 4-buffer ring
        case 23: ISB_CONV_LAUNCH_NB(2, 1, 2, 2, 4); break;  // 128 x  64
        case 26: ISB_CONV_LAUNCH_NB(2, 2, 4, 2, 4); break;  // 256 x 128, 8 waves
        case 27: ISB_CONV_LAUNCH_NB(4, 2, 2, 4, 4); break;  // 256 x 256, 8 waves
        case 28: ISB_CONV_LAUNCH_NB(4, 1, 2, 4, 4); break;  // 256 x 128 as 2x4 waves of 128x32
#undef ISB_CONV_LAUNCH_NB
    // gated LDS-DMA kernels: ring of NB tile buffers + dump KiB + the f32 gate rows of the samples a tile covers
#define ISB_CONV_LAUNCH_GATE(TM, TN, WGM, WGN, NB, KT)                                                              \
    do {                                                                                                            \
        constexpr int BM_ = 32 * TM * WGM, BN_ = 32 * TN * WGN;                                                     \
        const int ohw = a.OH * a.OW;                                                                                \
        if (ohw % BM_ != 0 && BM_ % ohw != 0) {                                                                     \
            set_error("conv_igemm: gated tile of %d rows does not align with %d-pixel samples", BM_, ohw);          \
            return ISB_ERR_INVALID;                                                                                 \
        }                                                                                                           \
        const int ns = BM_ > ohw ? BM_ / ohw : 1;                                                                   \
        const int ring = NB * (BM_ + BN_) * (KT * 2) + 1024 + ns * a.Cin * 4;                                       \
        const int stage = BM_ * (BN_ * 2 + 16);                                                                     \
        const int bytes = ring > stage ? ring : stage;                                                              \
        auto kern = conv_igemm_dma_kernel<TM, TN, WGM, WGN, NB, true, KT>;                                   \
        static DevMax attr_bytes;                                                                                  \
        if (attr_bytes.below(bytes)) {                                                                                   \
            ISB_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));     \
            attr_bytes.set(bytes);                                                                                     \
        }                                                                                                           \
        const dim3 g = conv_grid(aa, BM_, BN_);                                                                     \
        hipLaunchKernelGGL(kern, g, dim3(64 * WGM * WGN), bytes, st, aa);                                           \
    } while (0)
        case 81: ISB_CONV_LAUNCH_GATE(1, 2, 4, 2, 4, 32); break;   // 128 x 128, 8 waves, 4-buffer ring
        case 82: ISB_CONV_LAUNCH_GATE(1, 3, 4, 2, 4, 32); break;   // 128 x 192
        case 83: ISB_CONV_LAUNCH_GATE(1, 7, 4, 1, 4, 32); break;   // 128 x 224
        case 84: ISB_CONV_LAUNCH_GATE(1, 5, 4, 2, 4, 32); break;   // 128 x 320
        case 85: ISB_CONV_LAUNCH_GATE(1, 2, 2, 2, 4, 32); break;   //  64 x 128, 4 waves
        case 86: ISB_CONV_LAUNCH_GATE(1, 3, 2, 2, 4, 32); break;   //  64 x 192
        case 91: ISB_CONV_LAUNCH_GATE(1, 2, 4, 2, 2, 32); break;   // the same tiles with two buffers
        case 92: ISB_CONV_LAUNCH_GATE(1, 3, 4, 2, 2, 32); break;
        case 93: ISB_CONV_LAUNCH_GATE(1, 7, 4, 1, 2, 32); break;
        case 94: ISB_CONV_LAUNCH_GATE(1, 5, 4, 2, 2, 32); break;
        case 95: ISB_CONV_LAUNCH_GATE(1, 2, 2, 2, 2, 32); break;
        case 96: ISB_CONV_LAUNCH_GATE(1, 3, 2, 2, 2, 32); break;
        case 111: ISB_CONV_LAUNCH_GATE(1, 3, 4, 2, 2, 64); break;  // 64-wide k-tiles (128-B rows): 128 x 192
        case 112: ISB_CONV_LAUNCH_GATE(1, 2, 4, 2, 2, 64); break;  // 128 x 128
        case 113: ISB_CONV_LAUNCH_GATE(1, 7, 4, 1, 2, 64); break;  // 128 x 224
        case 114: ISB_CONV_LAUNCH_GATE(1, 5, 4, 2, 2, 64); break;  // 128 x 320
        case 115: ISB_CONV_LAUNCH_GATE(1, 3, 2, 2, 2, 64); break;  //  64 x 192
        case 116: ISB_CONV_LAUNCH_GATE(1, 2, 2, 2, 2, 64); break;  //  64 x 128
#undef ISB_CONV_LAUNCH_GATE
#define ISB_CONV_LAUNCH_DMA64(TM, TN, WGM, WGN)                                                                        \
    do {                                                                                                               \
        const dim3 g = conv_grid(aa, 32 * TM * WGM, 32 * TN * WGN);                                                    \
        hipLaunchKernelGGL((conv_igemm_dma_kernel<TM, TN, WGM, WGN, 2, false, 64>), g, dim3(64 * WGM * WGN), 0, st, aa); \
    } while (0)
        case 101: ISB_CONV_LAUNCH_DMA64(1, 3, 4, 2); break;   // 128 x 192, 64-wide k-tiles
        case 102: ISB_CONV_LAUNCH_DMA64(1, 2, 4, 2); break;   // 128 x 128
        case 103: ISB_CONV_LAUNCH_DMA64(2, 2, 4, 2); break;   // 256 x 128
        case 104: ISB_CONV_LAUNCH_DMA64(1, 2, 8, 1); break;   // 256 x  64
        case 105: ISB_CONV_LAUNCH_DMA64(1, 7, 4, 1); break;   // 128 x 224
        case 106: ISB_CONV_LAUNCH_DMA64(1, 3, 2, 2); break;   //  64 x 192
        case 107: ISB_CONV_LAUNCH_DMA64(1, 2, 2, 2); break;   //  64 x 128
        case 108: ISB_CONV_LAUNCH_DMA64(2, 3, 4, 2); break;   // 256 x 192
#undef ISB_CONV_LAUNCH_DMA64
#endif  // ISB_BUILD_PROBES
        default:
            set_error("conv_igemm: tile variant %d is not in this build (general kernels: 1, 3, 5, 75, 14, 54, 55, 57, 59, 64; the rest needs -DISB_BUILD_PROBES)", v);
            return ISB_ERR_INVALID;
    }
#undef ISB_CONV_LAUNCH
#undef ISB_CONV_LAUNCH_DMA
    return ISB_OK;
}


}  // namespace isb
