// Whole Fused-MBConv block (3x3 expand + 1x1 project) in one launch: launch_fused_mb. bool F16: fp16 operands / output (ConvArgs.f16).
#include "conv_tiles.h"

namespace isb {

// -------------------------------------------------------------------------------------------
// Whole Fused-MBConv block in one launch: 3x3 expand + folded BN + SiLU -> (bf16) -> 1x1 project + folded BN
// (+ residual). The expanded tensor E (4x the block's input, the largest tensors of the network: 537 MB per layer at
// 64x64) never leaves the chip: a workgroup owns 128 pixels x ALL expanded channels, so after the 3x3 k loop the
// bias/SiLU'd bf16 E tile goes to LDS in the A-operand layout of a second GEMM whose B operand (the projection
// weights, 4-6 KiB per 32-channel block) streams through two small LDS buffers.
//   8 waves as 4 (pixels) x 2 (channels); first GEMM: wave tile 32 x 32*TN (Cexp = 64*TN), lean 3x3 loop as above;
//   second GEMM: wave tile 32 pixels x 64 output channels (Cout2 <= 128), K = Cexp fully unrolled.
// Both sums run in the same order as the separate kernels, so the result is bit-identical to the two-launch path.
// -------------------------------------------------------------------------------------------
// WGM: 32-pixel blocks per workgroup (4 -> 128 pixels, 8 waves; 2 -> 64 pixels, 4 waves: two workgroups per CU
// when the E tile of 384 channels would otherwise fill the LDS); Cexp = 64 * TN; WPR = projected channels rounded up
// to 64 / 96 / 128 (rows of the projection-weight buffers)
// HALO (stride 1, 64 input channels, 64-wide maps: the body blocks of stage 2): what bounds the k loop is the operand
// bytes per k-step (tools/kstep_probe.py) and a third of them are im2col rows of A. The tile's 128 pixels are two image
// rows; their input HALO (4 rows x 66 pixels x 128 B = 33 KiB, out-of-image pixels zero-filled by the buffer bounds
// check) is copied to LDS once and the A fragments of the 18 k-steps are read from it -- a tap shift is just another
// 128-byte LDS row, chunk slot = chunk ^ ((pixel >> 1) & 7): a ds_read_b128 is served 16 lanes at a time from 64 banks (256 B), and
// 16 consecutive pixels of one chunk then sit in 16 different 16-byte slots (round 4; with `pixel & 7` pixels p and p + 8 shared a
// slot: SQ_LDS_BANK_CONFLICT 19 % of the LDS-active cycles of this kernel, every A read two passes). The k loop
// then streams only the weights: 16 instead of 24 KiB per k-step. Same (tap, channel) order: bit-identical.
// REGE (round 5; projections to <= 64 channels): the E tile never exists. After the k loop a lane holds, for its pixel, sixteen
// channels of every 32-channel tile in its accumulators -- bias, SiLU and ONE rounding (what the two-launch path stores) leave them
// as eight 16-bit values per k16 step, and those ARE the B fragment of the projection's transposed MFMA once the contraction
// index is ordered as the accumulator rows are (k slot (h, t) <-> channel 32 j + 16 s + 8 (t / 4) + 4 h + t % 4; the projection
// weights are packed in that order at load time, launch_fmb_pack_w2). So a wave multiplies ITS 128 channels straight from
// registers -- no 64-KiB E image in LDS, no A-fragment reads, and the eight publish() round trips of the streamed projection
// weights become ONE: the packed weights (32 KiB) land in the k loop's buffers by LDS-DMA while the SiLUs run. The two waves
// that share a pixel block then exchange one partial tile through LDS. Sums: a wave's 128 channels in k order, then
// (channels 0 - 127) + (128 - 255): the two-launch path's bits up to f32 association (tested against it with that tolerance).
template <int WGM, int TN, int WPR, bool HALO = false, bool F16 = false, bool REGE = false>
__global__ __launch_bounds__(128 * WGM) void fused_mb_kernel(ConvArgs p) {
    T16<F16>::enter();
    constexpr int WGN = 2, NW = WGM * WGN;
    constexpr int TN2 = (WPR / 32 + 1) / 2;                  // 32-channel tiles of the projection per wave
    constexpr int BM = 32 * WGM, BN = 64 * TN;               // BN = Cexp
    constexpr int B_INST = BN / 16;
    constexpr int B_PW = (B_INST + NW - 1) / NW;
    constexpr int BUF = (BM + BN) * ROWB;
    constexpr int NKB = BN / 32;                             // k-blocks of the second GEMM
    constexpr int E_BYTES = NKB * BM * ROWB;                 // E tile: NKB blocks of [128 rows][64 B], swizzled like A tiles
    constexpr int WP_ROWS = WPR, WP_BUF = REGE ? 0 : WP_ROWS * ROWB;    // projection weights of one k-block (rows past Cout2 unused)
    constexpr int W2_PW = (WP_ROWS / 16 + NW - 1) / NW;
    constexpr int STAGE2 = BM * (64 * TN2 * 2 + 16);         // epilogue staging of the output tile
    constexpr int HW_ = 64, HWD = HW_ + 2, HALO_BYTES = 4 * HWD * 128;     // HALO: 264 pixel rows of 128 B = 33 pieces
    constexpr int BBUF = BN * ROWB;                                         // HALO: one k-step of weights
    constexpr int KREG = HALO ? HALO_BYTES + 2 * BBUF : 2 * BUF;
    static_assert(!REGE || (TN2 == 1 && WGM == 4), "REGE: projections to <= 64 channels, 8 waves");
    constexpr int C2T = (WPR + 31) / 32;                     // REGE: 32-channel tiles of the projection (each wave multiplies all of them)
    constexpr int W2P_BYTES = C2T * NKB * 2 * 1024;          // REGE: the packed projection weights, one 1-KiB fragment per (tile, k16 step)
    constexpr int XCH_OFF = W2P_BYTES > STAGE2 ? W2P_BYTES : STAGE2;      // REGE: partial tiles of the channel halves, 4 KiB per wave
    constexpr int REGE_BYTES = XCH_OFF + 2 * WGM * 4096;
    constexpr int REG_A0 = REGE ? (KREG > REGE_BYTES ? KREG : REGE_BYTES) : (KREG > E_BYTES ? KREG : E_BYTES);
    constexpr int REG_A = REG_A0 > STAGE2 ? REG_A0 : STAGE2;
    constexpr int WP_OFF = REG_A, BIAS1_OFF = WP_OFF + 2 * WP_BUF, BIAS2_OFF = BIAS1_OFF + BN * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds[BIAS2_OFF + 128 * 4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * BM;
    // tuning probe (ConvArgs.probe & 2, isb_debug_fused_mb): s_memtime at the phase boundaries, wave 0 of workgroups 256 .. 319 (steady
    // state: the first round of workgroups starts on an empty chip) -> p.part[64][8]
    const bool stamped = (p.probe & 2) && p.part && blockIdx.x >= 256 && blockIdx.x < 320;
    uint64_t st[7] = {0, 0, 0, 0, 0, 0, 0};
    if (stamped) st[0] = __builtin_amdgcn_s_memtime();

    const int ohw = p.OH * p.OW;
    const uint32_t pix = (uint32_t)p.Cin * 2u;
    const uint32_t shift = (uint32_t)(p.pad * (p.W + 1)) * pix;
    const uint32_t nrec = (uint32_t)((size_t)p.B * p.H * p.W * p.Cin * 2) + shift;
    i32x4_t rsrc;
    {
        const uint64_t base = (uint64_t)(uintptr_t)p.in - shift;
        rsrc.x = (int)(uint32_t)base;
        rsrc.y = (int)(uint32_t)(base >> 32);
        rsrc.z = (int)nrec;
        rsrc.w = 0x00020000;
    }
    uint32_t a_voff, a_mask, b_voff[B_PW], w2_voff[W2_PW];
    {
        const int row = 16 * wave + (lane >> 2);             // A: one piece per wave (8 pieces = 128 rows)
        const int logical = (lane & 3) ^ ((row >> 2) & 3);
        const int m = m0 + row;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int b = mm / ohw, rem = mm - b * ohw;
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        a_voff = (uint32_t)((b * p.H + oy * p.stride) * p.W + ox * p.stride) * pix + logical * 16;
        uint32_t mk = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = iy0 + t / 3, ix = ix0 + t % 3;
            if (ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) mk |= 1u << t;
        }
        a_mask = mk;
    }
#pragma unroll
    for (int s = 0; s < W2_PW; ++s) {                        // projection weights: 16 rows of each k-block per piece
        const int row = 16 * (wave + NW * s) + (lane >> 2);
        const int logical = (lane & 3) ^ ((row >> 2) & 3);
        w2_voff[s] = (uint32_t)min(row, p.Cout2 - 1) * (uint32_t)(BN * 2) + logical * 16;
    }
#pragma unroll
    for (int s = 0; s < B_PW; ++s) {
        const int row = 16 * (wave + NW * s) + (lane >> 2);
        const int logical = (lane & 3) ^ ((row >> 2) & 3);
        b_voff[s] = (uint32_t)min(row, BN - 1) * (uint32_t)(p.K * 2) + logical * 16;
    }
    const unsigned char* b_base = reinterpret_cast<const unsigned char*>(p.w);
    const uint32_t ldsA = (uint32_t)(uintptr_t)(lds_ptr_t)lds;
    const uint32_t lds0 = ldsA + wave * 1024;
    if (wave == 0) {
#pragma unroll
        for (int o = 0; o < BN / 4; o += 64)
            if (lane + o < BN / 4) dma16_s(p.bias, (uint32_t)(lane + o) * 16, ldsA + BIAS1_OFF + o * 16);
    }
    if (wave == 1 && lane < 32) dma16_s(p.bias2, (uint32_t)min(lane * 4, p.Cout2 - 4) * 4, ldsA + BIAS2_OFF);
    int tap = 0, c0 = 0;
    uint32_t tap_soff = 0;
    constexpr int B_LDS0 = HALO ? HALO_BYTES : BM * ROWB;     // weights of buffer 0; buffer 1 follows B_STRIDE later
    constexpr int B_STRIDE = HALO ? BBUF : BUF;
    if constexpr (HALO) {
        // the halo, once: piece i = 8 pixel rows; lane = (pixel row i * 8 + lane / 8, chunk slot lane & 7)
        const int b = m0 / ohw, y0 = (m0 - b * ohw) / HW_;
        for (int i = wave; i < HALO_BYTES / 1024; i += NW) {
            const int hp = i * 8 + (lane >> 3);
            const int hy = hp / HWD, hx = hp - hy * HWD;
            const int y = y0 - 1 + hy, x = hx - 1;
            const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)HW_;
            const uint32_t chunk = (uint32_t)((lane & 7) ^ ((hp >> 1) & 7));
            // relative to the shifted base (in - (W + 1) pixels): pixel (y, x) sits at ((b H + y + 1) W + x + 1) pixels
            const uint32_t voff = ok ? (uint32_t)((b * p.H + y + 1) * HW_ + x + 1) * 128u + chunk * 16u : 0x80000000u;
            dma16_buf(rsrc, voff, 0u, ldsA + i * 1024);
        }
    }
    auto dma = [&](auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        if constexpr (!HALO) {
            const uint32_t soff = tap_soff + (uint32_t)c0 * 2u;
            dma16_buf(rsrc, ((a_mask >> tap) & 1u) ? a_voff : 0x80000000u, soff, lds0 + buf * BUF);
        }
#pragma unroll
        for (int s = 0; s < B_PW; ++s)
            if (wave + NW * s < B_INST) dma16_s(b_base, b_voff[s], lds0 + (B_LDS0 + buf * B_STRIDE + NW * s * 1024));
        b_base += CK * 2;
        c0 += CK;
        if (c0 == p.Cin) {
            c0 = 0;
            ++tap;
            tap_soff += (tap % 3 == 0) ? (uint32_t)(p.W - 2) * pix : pix;
        }
    };

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    int a_sw[2], b_sw[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        a_sw[ks] = swz(wm * 32 + r, 2 * ks + h);
        b_sw[ks] = B_LDS0 + swz(wn * TN * 32 + r, 2 * ks + h);
    }
    // HALO: window origin of the lane's pixel in the halo; k-step kt = (tap kt / 2, channel half kt % 2 = buffer)
    const int hq = wm * 32 + r;
    const int hp0 = (hq / HW_) * HWD + (hq % HW_);
    int h_tap = 0, h_off = 0;                                 // tap index and its pixel offset ky * 66 + kx
    // (round 5: issuing all ten fragment reads of a k-step ahead of its eight MFMAs -- a second register set, sched_group_barrier --
    // changes nothing, 201.9 vs 199.8 us per 128 frames: the k-step is set by the 16 KiB of weights a workgroup pulls in, 24 B/clk per CU
    // with two workgroups, not by the wave's exposed LDS round trips; EXPERIMENTS.md)
    auto compute = [&](auto bufc) {
        constexpr int buf = decltype(bufc)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af;
            if constexpr (HALO) {
                const int hp = hp0 + h_off;
                af = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + hp * 128 + (((4 * buf + 2 * ks + h) ^ ((hp >> 1) & 7)) << 4)));
            } else {
                af = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + a_sw[ks] + buf * BUF));
            }
            bf16x8 bfr[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bfr[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + b_sw[ks] + (buf * B_STRIDE + j * 2048)));
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[j] = T16<F16>::mfma32(bfr[j], af, acc[j]);
        }
        if constexpr (HALO && buf == 1) {                     // both halves of the tap done
            ++h_tap;
            h_off += (h_tap % 3 == 0) ? HWD - 2 : 1;
        }
    };
    auto publish = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    const int nkt = p.K / CK;
    dma(std::integral_constant<int, 0>{});
    publish();
    if (stamped) st[1] = __builtin_amdgcn_s_memtime();
    int kt = 0;
    for (; kt + 2 <= nkt; kt += 2) {
        dma(std::integral_constant<int, 1>{});
        compute(std::integral_constant<int, 0>{});
        publish();
        if (kt + 2 < nkt) dma(std::integral_constant<int, 0>{});
        compute(std::integral_constant<int, 1>{});
        publish();
    }
    if (kt < nkt) {
        compute(std::integral_constant<int, 0>{});
        __syncthreads();
    }

    if constexpr (REGE) {
        // the packed projection weights -> the (free) k-loop region, in flight while the SiLUs run
        {
            constexpr int NPIECE = W2P_BYTES / 1024;
#pragma unroll
            for (int i = 0; i < (NPIECE + NW - 1) / NW; ++i)
                if (wave + NW * i < NPIECE) dma16_s(p.w2p, (uint32_t)((wave + NW * i) * 1024 + lane * 16), ldsA + (wave + NW * i) * 1024);
        }
        if (stamped) st[2] = __builtin_amdgcn_s_memtime();
        // E = T16(silu(acc + bias)): the lane's sixteen channels of tile j as two B fragments (k16 steps s = 0, 1)
        uint4 ef[TN][2];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = (wn * TN + j) * 32 + 8 * q + 4 * h;
                const float4 bs = *reinterpret_cast<const float4*>(lds + BIAS1_OFF + nl * 4);
                float v0 = acc[j][4 * q] + bs.x, v1 = acc[j][4 * q + 1] + bs.y, v2 = acc[j][4 * q + 2] + bs.z, v3 = acc[j][4 * q + 3] + bs.w;
                if (p.act) { v0 = silu_fast(v0); v1 = silu_fast(v1); v2 = silu_fast(v2); v3 = silu_fast(v3); }
                const uint32_t lo = T16<F16>::pack2(v0, v1), hi = T16<F16>::pack2(v2, v3);
                if (q == 0) { ef[j][0].x = lo; ef[j][0].y = hi; }
                else if (q == 1) { ef[j][0].z = lo; ef[j][0].w = hi; }
                else if (q == 2) { ef[j][1].x = lo; ef[j][1].y = hi; }
                else { ef[j][1].z = lo; ef[j][1].w = hi; }
            }
        if (stamped) st[3] = __builtin_amdgcn_s_memtime();
        publish();                                           // the packed weights have landed (and every wave has left the k loop's buffers)
        if (stamped) st[4] = __builtin_amdgcn_s_memtime();
        // partial projection over the wave's channels: acc2[c] (channels 32 c ..) += W2frag(c, tile, s) . Efrag(tile, s)
        f32x16 pacc[C2T];
#pragma unroll
        for (int c = 0; c < C2T; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) pacc[c][e] = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int sI = 0; sI < 2; ++sI) {
                uint4 wf[C2T];
#pragma unroll
                for (int c = 0; c < C2T; ++c)
                    wf[c] = *reinterpret_cast<const uint4*>(lds + ((c * NKB + wn * TN + j) * 2 + sI) * 1024 + lane * 16);
#pragma unroll
                for (int c = 0; c < C2T; ++c) pacc[c] = T16<F16>::mfma32(wf[c], ef[j][sI], pacc[c]);
            }
        // the wave that shares this pixel block holds the other half of the channels: it gets the partial tile it finishes,
        // this wave gets the one IT finishes (tile wn; C2T == 2)
        static_assert(C2T <= 2, "REGE: one partial tile per wave");
        __syncthreads();                                     // every wave is done with the packed weights (the exchange area is behind them, the staging area over them)
        {
            float4* xo = reinterpret_cast<float4*>(lds + XCH_OFF + wave * 4096) + lane;
            const f32x16& give = pacc[C2T == 1 ? 0 : 1 - wn];
            if (C2T == 2 || wn == 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) xo[q * 64] = make_float4(give[4 * q], give[4 * q + 1], give[4 * q + 2], give[4 * q + 3]);
            }
        }
        __syncthreads();
        f32x16 acc2r[1][1];
        {
            const float4* xi = reinterpret_cast<const float4*>(lds + XCH_OFF + (wave ^ 1) * 4096) + lane;      // waves (wm, 0) and (wm, 1) are neighbours
            const f32x16& own = pacc[C2T == 1 ? 0 : wn];
            if (C2T == 2 || wn == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 o = xi[q * 64];
                    // (channels 0 - 127) + (128 - 255): the same two operands in both waves' sums
                    acc2r[0][0][4 * q] = own[4 * q] + o.x; acc2r[0][0][4 * q + 1] = own[4 * q + 1] + o.y;
                    acc2r[0][0][4 * q + 2] = own[4 * q + 2] + o.z; acc2r[0][0][4 * q + 3] = own[4 * q + 3] + o.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) acc2r[0][0][e] = 0.f;      // (C2T == 1: wave (wm, 1) has no tile to finish; its rows lie past Cout2)
            }
        }
        if (stamped) st[5] = __builtin_amdgcn_s_memtime();
        ConvArgs p2 = p;
        p2.Cout = p.Cout2;
        p2.act = 0;
        conv_epilogue<1, 1, WGM, 2, true, F16>(p2, acc2r, lds, m0, 0, wm, wn, r, h, tid, BIAS2_OFF);
        if (stamped && tid == 0) {
            st[6] = __builtin_amdgcn_s_memtime();
            uint64_t* o = reinterpret_cast<uint64_t*>(p.part) + (size_t)(blockIdx.x - 256) * 8;
#pragma unroll
            for (int i = 0; i < 6; ++i) o[i] = st[i + 1] - st[i];
            o[6] = st[6] - st[0];
            o[7] = 1;
        }
        return;
    }
    // ---- E tile: bias + SiLU, one bf16 rounding (what the two-launch path stores), into the A layout of GEMM 2
    const unsigned char* w2_base = reinterpret_cast<const unsigned char*>(p.w2);
    auto dma_w2 = [&](auto bufc) {                           // next 32 expanded channels of the projection weights
        constexpr int buf = decltype(bufc)::value;
#pragma unroll
        for (int s = 0; s < W2_PW; ++s)
            if (wave + NW * s < WP_ROWS / 16) dma16_s(w2_base, w2_voff[s], lds0 + WP_OFF + buf * WP_BUF + NW * s * 1024);
        w2_base += CK * 2;
    };
    if (stamped) st[2] = __builtin_amdgcn_s_memtime();
    dma_w2(std::integral_constant<int, 0>{});
    {
        const int ml = wm * 32 + r;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = (wn * TN + j) * 32 + 8 * q + 4 * h;
                const float4 bs = *reinterpret_cast<const float4*>(lds + BIAS1_OFF + nl * 4);
                float v0 = acc[j][4 * q] + bs.x, v1 = acc[j][4 * q + 1] + bs.y, v2 = acc[j][4 * q + 2] + bs.z, v3 = acc[j][4 * q + 3] + bs.w;
                if (p.act) { v0 = silu_fast(v0); v1 = silu_fast(v1); v2 = silu_fast(v2); v3 = silu_fast(v3); }
                uint2 pk;
                pk.x = (uint32_t)T16<F16>::from_f32(v0) | ((uint32_t)T16<F16>::from_f32(v1) << 16);
                pk.y = (uint32_t)T16<F16>::from_f32(v2) | ((uint32_t)T16<F16>::from_f32(v3) << 16);
                *reinterpret_cast<uint2*>(lds + (wn * TN + j) * (BM * ROWB) + swz(ml, q) + 8 * h) = pk;
            }
    }
    if (stamped) st[3] = __builtin_amdgcn_s_memtime();
    publish();
    if (stamped) st[4] = __builtin_amdgcn_s_memtime();

    // ---- GEMM 2: out[128, Cout2] = E[128, Cexp] . w2[Cout2, Cexp]^T ; wave = 32 pixels x 32 * TN2 channels
    f32x16 acc2[1][TN2];
#pragma unroll
    for (int j = 0; j < TN2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[0][j][e] = 0.f;
    const bool n_live = wn * 32 * TN2 < p.Cout2;             // a wave whose channels all lie past Cout2 only keeps the barriers
    const bool t1_live = (wn * TN2 + 1) * 32 < p.Cout2;     // second tile of the wave (TN2 == 2)
    auto gemm2_step = [&](auto kbc) {
        constexpr int kb = decltype(kbc)::value;
        constexpr int buf = kb & 1;
        if constexpr (kb + 1 < NKB) dma_w2(std::integral_constant<int, buf ^ 1>{});
        if (n_live) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 af = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + a_sw[ks] + kb * (BM * ROWB)));
#pragma unroll
                for (int j = 0; j < TN2; ++j) {
                    if (j == 1 && !t1_live) continue;        // rows past WP_ROWS are not staged
                    const bf16x8 bw = __builtin_bit_cast(
                        bf16x8, *reinterpret_cast<const uint4*>(lds + WP_OFF + buf * WP_BUF + swz((wn * TN2 + j) * 32 + r, 2 * ks + h)));
                    acc2[0][j] = T16<F16>::mfma32(bw, af, acc2[0][j]);
                }
            }
        }
        publish();
    };
    gemm2_step(std::integral_constant<int, 0>{});
    gemm2_step(std::integral_constant<int, 1>{});
    gemm2_step(std::integral_constant<int, 2>{});
    gemm2_step(std::integral_constant<int, 3>{});
    if constexpr (NKB > 4) {
        gemm2_step(std::integral_constant<int, 4>{});
        gemm2_step(std::integral_constant<int, 5>{});
        gemm2_step(std::integral_constant<int, 6>{});
        gemm2_step(std::integral_constant<int, 7>{});
    }
    if constexpr (NKB > 8) {
        gemm2_step(std::integral_constant<int, 8>{});
        gemm2_step(std::integral_constant<int, 9>{});
        gemm2_step(std::integral_constant<int, 10>{});
        gemm2_step(std::integral_constant<int, 11>{});
    }
    static_assert(NKB == 4 || NKB == 8 || NKB == 12, "Cexp = 128, 256 or 384");

    if (stamped) st[5] = __builtin_amdgcn_s_memtime();
    ConvArgs p2 = p;                                         // epilogue of the projection: bias2, no activation, residual
    p2.Cout = p.Cout2;
    p2.act = 0;
    conv_epilogue<1, TN2, WGM, 2, true, F16>(p2, acc2, lds, m0, 0, wm, wn, r, h, tid, BIAS2_OFF);
    if (stamped && tid == 0) {
        st[6] = __builtin_amdgcn_s_memtime();
        uint64_t* o = reinterpret_cast<uint64_t*>(p.part) + (size_t)(blockIdx.x - 256) * 8;
#pragma unroll
        for (int i = 0; i < 6; ++i) o[i] = st[i + 1] - st[i];
        o[6] = st[6] - st[0];
        o[7] = 1;
    }
}

__global__ void fmb_pack_w2_kernel(const uint16_t* w2, uint4* dst, int Cout2, int Cexp, int total) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int lane = i & 63, sI = (i >> 6) & 1, nkb = Cexp / 32;
    const int j = (i >> 7) % nkb, c = (i >> 7) / nkb;
    const int row = 32 * c + (lane & 31), hh = lane >> 5;
    uint32_t v[4] = {0, 0, 0, 0};
    if (row < Cout2) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const uint32_t wv = w2[(size_t)row * Cexp + 32 * j + 16 * sI + 8 * (t >> 2) + 4 * hh + (t & 3)];
            v[t >> 1] |= wv << (16 * (t & 1));
        }
    }
    dst[i] = make_uint4(v[0], v[1], v[2], v[3]);
}

size_t fmb_w2p_bytes(int Cout2, int Cexp) { return (size_t)cdiv(Cout2, 32) * (Cexp / 32) * 2 * 1024; }

int launch_fmb_pack_w2(const uint16_t* w2, void* dst, int Cout2, int Cexp, hipStream_t st) {
    if (Cexp % 32 != 0 || Cout2 < 1) {
        set_error("fmb_pack_w2: Cout2=%d Cexp=%d", Cout2, Cexp);
        return ISB_ERR_INVALID;
    }
    const int total = (int)(fmb_w2p_bytes(Cout2, Cexp) / 16);
    hipLaunchKernelGGL(fmb_pack_w2_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w2, reinterpret_cast<uint4*>(dst), Cout2, Cexp, total);
    ISB_LAUNCHED("fmb_pack_w2", st);
    return ISB_OK;
}

int launch_fused_mb(const ConvArgs& a, hipStream_t st) {
    const bool same1 = a.stride == 1 && a.pad == 1, same2 = a.stride == 2 && a.pad == 0;
    if (a.gate || a.KH != 3 || a.KW != 3 || !(same1 || same2) || a.Cin % 32 != 0 || a.K != 9 * a.Cin || !a.w2 || !a.bias2 ||
        a.Cout2 < 4 || a.Cout2 > 128 || a.Cout2 % 4 != 0 || a.out_f32 ||
        (size_t)a.B * a.H * a.W * a.Cin * 2 + (size_t)(a.W + 1) * a.Cin * 2 >= 0x7ffffff0ull) {
        set_error("fused_mb: needs an un-gated 3x3 expand and a projection to <= 128 channels");
        return ISB_ERR_INVALID;
    }
    ConvArgs aa = a;
    aa.grid_mode = 0;
    aa.exp = exp_flags();
    const int wpr = a.Cout2 <= 64 ? 64 : a.Cout2 <= 96 ? 96 : 128;
#define ISB_FMB(WGM, TN, WPR, HALO_)                                                                                         \
    do {                                                                                                                     \
        if (a.f16) hipLaunchKernelGGL((fused_mb_kernel<WGM, TN, WPR, HALO_, true>), dim3(cdiv(a.M, 32 * WGM)), dim3(128 * WGM), 0, st, aa); \
        else hipLaunchKernelGGL((fused_mb_kernel<WGM, TN, WPR, HALO_, false>), dim3(cdiv(a.M, 32 * WGM)), dim3(128 * WGM), 0, st, aa);     \
    } while (0)
    // the projection from the accumulators (packed weights given; <= 64 projected channels)
#define ISB_FMB_R(WGM, TN, HALO_)                                                                                            \
    do {                                                                                                                     \
        if (a.f16) hipLaunchKernelGGL((fused_mb_kernel<WGM, TN, 64, HALO_, true, true>), dim3(cdiv(a.M, 32 * WGM)), dim3(128 * WGM), 0, st, aa); \
        else hipLaunchKernelGGL((fused_mb_kernel<WGM, TN, 64, HALO_, false, true>), dim3(cdiv(a.M, 32 * WGM)), dim3(128 * WGM), 0, st, aa);     \
    } while (0)
    const bool rege = a.w2p != nullptr && wpr == 64 && a.Cout2 > 32;
    // the shapes EfficientNetV2-L has: 32 -> 128 -> 64 (stride 2), 64 -> 256 -> 64 (body of stage 1: halo-tile A operand on its
    // 64-wide maps), 64 -> 256 -> 96 (stride 2). (384 expanded channels measured slower than two launches: EXPERIMENTS.md.)
    if (a.Cout == 128 && wpr == 64) { if (rege) ISB_FMB_R(4, 2, false); else ISB_FMB(4, 2, 64, false); }
    else if (a.Cout == 256 && wpr == 64 && same1 && a.Cin == 64 && a.W == 64 && a.H % 2 == 0) { if (rege) ISB_FMB_R(4, 4, true); else ISB_FMB(4, 4, 64, true); }
    else if (a.Cout == 256 && wpr == 64) { if (rege) ISB_FMB_R(4, 4, false); else ISB_FMB(4, 4, 64, false); }
    else if (a.Cout == 256 && wpr == 96) ISB_FMB(4, 4, 96, false);
#ifdef ISB_BUILD_PROBES
    else if (a.f16) { set_error("fused_mb: the probe shapes are bf16 only"); return ISB_ERR_INVALID; }
    else if (a.Cout == 128 && wpr == 96) hipLaunchKernelGGL((fused_mb_kernel<4, 2, 96>), dim3(cdiv(a.M, 128)), dim3(512), 0, st, aa);
    else if (a.Cout == 128) hipLaunchKernelGGL((fused_mb_kernel<4, 2, 128>), dim3(cdiv(a.M, 128)), dim3(512), 0, st, aa);
    else if (a.Cout == 256) hipLaunchKernelGGL((fused_mb_kernel<4, 4, 128>), dim3(cdiv(a.M, 128)), dim3(512), 0, st, aa);
    else if (a.Cout == 384 && wpr == 64) hipLaunchKernelGGL((fused_mb_kernel<2, 6, 64>), dim3(cdiv(a.M, 64)), dim3(256), 0, st, aa);
    else if (a.Cout == 384 && wpr == 96) hipLaunchKernelGGL((fused_mb_kernel<2, 6, 96>), dim3(cdiv(a.M, 64)), dim3(256), 0, st, aa);
    else if (a.Cout == 384) hipLaunchKernelGGL((fused_mb_kernel<2, 6, 128>), dim3(cdiv(a.M, 64)), dim3(256), 0, st, aa);
#endif
    else {
        set_error("fused_mb: no kernel for %d expanded / %d projected channels in this build (128 -> 64, 256 -> 64, 256 -> 96)", a.Cout, a.Cout2);
        return ISB_ERR_INVALID;
    }
#undef ISB_FMB
#undef ISB_FMB_R
    ISB_LAUNCHED("fused_mb", st);
    return ISB_OK;
}

}  // namespace isb
