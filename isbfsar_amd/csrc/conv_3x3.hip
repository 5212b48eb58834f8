// 3x3 convolutions of the convolution family (tile variants 161 - 171): the lean buffer-addressed kernel (optionally with the A
// operand from an LDS halo) and the row-ring kernel of the first stage. bool F16: operands / residual / output in IEEE fp16
// instead of bf16 (ConvArgs.f16), same loops.
#include "conv_tiles.h"

namespace isb {

// HALO (stride 1, 96 input channels, 32-wide maps, 128-pixel tiles = four image rows: the body blocks of stage 3): the
// A fragments come from the tile's input halo in LDS (6 rows x 34 pixels, 256-byte pixel rows of which 192 B are used,
// chunk slot = chunk ^ (pixel & 15): 16 consecutive pixels of one chunk in 16 different 16-byte slots of the 64 banks -- with
// `pixel & 7` every A read took two passes, SQ_LDS_BANK_CONFLICT 23 % of the LDS-active cycles), copied once, like
// fused_mb_kernel<.., HALO>; the k loop streams only the weights
// (12 instead of 20 KiB per k-step). Same (tap, channel) order: bit-identical.
template <int TM, int TN, int WGM, int WGN, bool HALO = false, bool F16 = false>
__global__ __launch_bounds__(64 * WGM * WGN) void conv3x3_dma_kernel(ConvArgs p) {
    T16<F16>::enter();
    constexpr int NW = WGM * WGN;
    constexpr int BM = 32 * TM * WGM;
    constexpr int BN = 32 * TN * WGN;
    constexpr int A_INST = BM / 16, B_INST = BN / 16;
    constexpr int A_PW = (A_INST + NW - 1) / NW, B_PW = (B_INST + NW - 1) / NW;
    constexpr int BUF = (BM + BN) * ROWB;
    constexpr int CROW = BN * 2 + 16;
    constexpr int HW_ = 32, HWD = HW_ + 2, HALO_BYTES = 6 * HWD * 256;       // HALO: 204 pixel rows of 256 B = 51 pieces
    constexpr int BBUF = BN * ROWB;
    constexpr int KREG = HALO ? HALO_BYTES + 2 * BBUF : 2 * BUF;
    constexpr int LDS_BYTES = (KREG > BM * CROW || BM * CROW > 65536) ? KREG : BM * CROW;
    constexpr int B_LDS0 = HALO ? HALO_BYTES : BM * ROWB, B_STRIDE = HALO ? BBUF : BUF;
    static_assert(!HALO || (BM == 128 && TM == 1), "the halo path walks 128-pixel tiles, one 32-pixel block per wave row");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES + BN * 4];
    constexpr int bias_off = LDS_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int r = lane & 31, h = lane >> 5;
    int m0, n0;
    if (!conv_tile_origin(p, BM, BN, m0, n0)) return;

    // A: per-lane window origin (relative to the shifted base) + validity of the 9 taps
    const int ohw = p.OH * p.OW;
    const uint32_t pix = (uint32_t)p.Cin * 2u;                                  // bytes per pixel
    const uint32_t shift = (uint32_t)(p.pad * (p.W + 1)) * pix;                 // base' = in - shift
    const uint32_t nrec = (uint32_t)((size_t)p.B * p.H * p.W * p.Cin * 2) + shift;
    i32x4_t rsrc;
    {
        const uint64_t base = (uint64_t)(uintptr_t)p.in - shift;
        rsrc.x = (int)(uint32_t)base;
        rsrc.y = (int)(uint32_t)(base >> 32);                                   // stride 0, no swizzle
        rsrc.z = (int)nrec;
        rsrc.w = 0x00020000;
    }
    uint32_t a_voff[A_PW], a_mask[A_PW], b_voff[B_PW];
#pragma unroll
    for (int s = 0; s < A_PW; ++s) {
        const int row = 16 * (wave + NW * s) + (lane >> 2);
        const int logical = (lane & 3) ^ ((row >> 2) & 3);
        const int m = m0 + row;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int b = mm / ohw, rem = mm - b * ohw;
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        a_voff[s] = (uint32_t)((b * p.H + oy * p.stride) * p.W + ox * p.stride) * pix + logical * 16;
        uint32_t mk = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = iy0 + t / 3, ix = ix0 + t % 3;
            if (ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) mk |= 1u << t;
        }
        a_mask[s] = mk;
    }
#pragma unroll
    for (int s = 0; s < B_PW; ++s) {
        const int row = 16 * (wave + NW * s) + (lane >> 2);
        const int logical = (lane & 3) ^ ((row >> 2) & 3);
        b_voff[s] = (uint32_t)min(n0 + row, p.Cout - 1) * (uint32_t)(p.K * 2) + logical * 16;
    }
    const unsigned char* b_base = reinterpret_cast<const unsigned char*>(p.w);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)lds + wave * 1024;
    if (wave == 0) {
#pragma unroll
        for (int o = 0; o < BN / 4; o += 64)
            if (lane + o < BN / 4)
                dma16_s(p.bias, (uint32_t)min(n0 + (lane + o) * 4, p.Cout - 4) * 4,
                        (uint32_t)(uintptr_t)(lds_ptr_t)lds + bias_off + o * 16);
    }
    // scalar k-step state: tap index, channel offset inside the tap, byte offset of the tap's pixel
    int tap = 0, c0 = 0;
    uint32_t tap_soff = 0;
    if constexpr (HALO) {
        // the halo, once: piece i = 4 pixel rows of 256 B; lane = (pixel row i * 4 + lane / 16, chunk slot lane & 15)
        const int b = m0 / ohw, y0 = (m0 - b * ohw) / HW_;
        const uint32_t ldsA = (uint32_t)(uintptr_t)(lds_ptr_t)lds;
        for (int i = wave; i < HALO_BYTES / 1024; i += NW) {
            const int hp = i * 4 + (lane >> 4);
            const int hy = hp / HWD, hx = hp - hy * HWD;
            const int y = y0 - 1 + hy, x = hx - 1;
            const int chunk = (lane & 15) ^ (hp & 15);
            const bool ok = chunk < 12 && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)HW_;
            // relative to the shifted base (in - (W + 1) pixels): pixel (y, x) sits at ((b H + y + 1) W + x + 1) pixels
            const uint32_t voff = ok ? (uint32_t)((b * p.H + y + 1) * HW_ + x + 1) * 192u + (uint32_t)chunk * 16u : 0x80000000u;
            dma16_buf(rsrc, voff, 0u, ldsA + i * 1024);
        }
    }
    auto dma = [&](auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        const uint32_t soff = tap_soff + (uint32_t)c0 * 2u;
        if constexpr (!HALO) {
#pragma unroll
            for (int s = 0; s < A_PW; ++s)
                if (wave + NW * s < A_INST) {
                    const uint32_t vo = ((a_mask[s] >> tap) & 1u) ? a_voff[s] : 0x80000000u;   // + soff cannot wrap back in range
                    dma16_buf(rsrc, vo, soff, lds0 + (buf * BUF + NW * s * 1024));
                }
        }
#pragma unroll
        for (int s = 0; s < B_PW; ++s)
            if (wave + NW * s < B_INST) dma16_s(b_base, b_voff[s], lds0 + (B_LDS0 + buf * B_STRIDE + NW * s * 1024));
        b_base += CK * 2;
        c0 += CK;
        if (c0 == p.Cin) {                      // next tap: one pixel to the right, or back two and down a row
            c0 = 0;
            ++tap;
            tap_soff += (tap % 3 == 0) ? (uint32_t)(p.W - 2) * pix : pix;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    int a_sw[2], b_sw[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        a_sw[ks] = swz(wm * TM * 32 + r, 2 * ks + h);
        b_sw[ks] = B_LDS0 + swz(wn * TN * 32 + r, 2 * ks + h);
    }
    // HALO: window origin of the lane's pixel in the halo; k-step = (tap, 32-channel block 0..2 of the 96)
    const int hq = wm * 32 + r;
    const int hp0 = (hq / HW_) * HWD + (hq % HW_);
    int h_tap = 0, h_off = 0, h_cb = 0;
    auto compute = [&](auto bufc) {
        constexpr int buf = decltype(bufc)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[TM], bfr[TN];
            if constexpr (HALO) {
                const int hp = hp0 + h_off;
                af[0] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + hp * 256 + (((4 * h_cb + 2 * ks + h) ^ (hp & 15)) << 4)));
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + a_sw[ks] + (buf * BUF + i * 2048)));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bfr[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + b_sw[ks] + (buf * B_STRIDE + j * 2048)));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = T16<F16>::mfma32(bfr[j], af[i], acc[i][j]);
        }
        if constexpr (HALO) {
            if (++h_cb == 3) {                                // the tap's 96 channels done
                h_cb = 0;
                ++h_tap;
                h_off += (h_tap % 3 == 0) ? HWD - 2 : 1;
            }
        }
    };
    auto publish = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    const int nkt = p.K / CK;
    dma(std::integral_constant<int, 0>{});
    publish();
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
    conv_epilogue<TM, TN, WGM, WGN, true, F16>(p, acc, lds, m0, n0, wm, wn, r, h, tid, bias_off);
}

// -------------------------------------------------------------------------------------------
// 3x3 stride-1 convolution 32 -> 32 channels on 128-wide images (the first stage: four layers on the largest
// activations). As an implicit GEMM it has N = 32: every k-step of the kernels above copies 16 KiB of im2col rows for
// two MFMAs per wave -- LDS-DMA issue bound, and every input pixel crosses the L2 -> LDS path nine times. Here a
// workgroup walks down a band of image rows, two output rows (256 pixels) per step, with the input rows it needs in
// an LDS RING of six rows: four feed the current step, the two the next step adds are in flight meanwhile, so every
// input pixel is copied ONCE (out-of-image pixels zero-filled by the buffer bounds check). The A fragments of all
// nine taps are read from the ring directly -- lane r's pixel shifted by the tap is just another 64-byte LDS row --
// and the weights (18 KiB) live in registers as 18 B fragments per lane for the whole band.
// LDS rows are pixels (144 per image row: x = -1 .. 142, nine 1-KiB pieces); chunk slot = logical chunk ^
// ((pixel >> 2) & 3): the 16 lanes a ds_read_b128 serves at a time (16 consecutive pixels, one chunk) hit 16 different 16-byte
// slots of the 64 banks for any tap shift (round 4; `(pixel >> 1) & 3` let pixels p and p + 8 collide: SQ_LDS_BANK_CONFLICT 52 %
// of this kernel's LDS-active cycles). A step's 256 outputs are
// consecutive NHWC pixels, so the shared epilogue (bias, SiLU, residual, 16-byte row stores) applies unchanged.
// Sums run in the (tap, channel) order of the implicit-GEMM kernels: bit-identical results.
// -------------------------------------------------------------------------------------------
constexpr int HALO_ROWB = 144 * 64;                  // bytes per ring row
constexpr int HALO_RING = 6 * HALO_ROWB;
constexpr int HALO_BIAS = HALO_RING + 256 * 64;      // the 32 biases (f32): read per step -- as 16 registers they pushed the kernel past 128 and
constexpr int HALO_LDS = HALO_BIAS + 128;            // the second workgroup off the CU. + the step's output tile (residual in, result out: in place)
template <bool F16>
__global__ __launch_bounds__(512, 4) void conv3x3_c32_rows_kernel(ConvArgs p, int band) {
    T16<F16>::enter();
    constexpr int W_ = 128;
    unsigned char* const lds = conv_lds_dyn;
    unsigned char* const Cs = lds + HALO_RING;           // [256 pixels][64 B], chunk-swizzled like the ring
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int bands = p.H / band;
    const int b = blockIdx.x / bands, ys = (blockIdx.x - b * bands) * band, ye = ys + band;

    const uint32_t nbytes = (uint32_t)((size_t)p.B * p.H * W_ * 64);
    i32x4_t rsrc, rres;
    {
        const uint64_t base = (uint64_t)(uintptr_t)p.in, rb = (uint64_t)(uintptr_t)p.res;
        rsrc.x = (int)(uint32_t)base; rsrc.y = (int)(uint32_t)(base >> 32); rsrc.z = (int)nbytes; rsrc.w = 0x00020000;
        rres.x = (int)(uint32_t)rb; rres.y = (int)(uint32_t)(rb >> 32); rres.z = (int)nbytes; rres.w = 0x00020000;
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)lds;
    // one lane's share of a row piece: pixel hx = 16 * piece + lane / 4 (image x = hx - 1), chunk by the swizzle
    auto load_rows = [&](int y_first, int nrows) {       // image rows y_first .. y_first + nrows - 1 -> their ring slots
        for (int pi = wave; pi < nrows * 9; pi += 8) {
            const int row = pi / 9, piece = pi - row * 9;
            const int y = y_first + row;
            const int hx = piece * 16 + (lane >> 2), x = hx - 1;
            const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)W_;
            const int chunk = (lane & 3) ^ ((hx >> 2) & 3);
            const uint32_t voff = ok ? (uint32_t)((b * p.H + y) * W_ + x) * 64u + (uint32_t)chunk * 16u : 0x80000000u;
            dma16_buf(rsrc, voff, 0u, lds_base + (uint32_t)(((y + 1) % 6) * HALO_ROWB + piece * 1024));
        }
    };
    // the residual rows of a step's outputs: every wave fetches the two pieces that hold ITS 32 pixels, so the whole
    // epilogue is wave-local (no barrier between the residual's arrival, the in-place result and the row stores)
    auto load_res = [&](int y0) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int px = wave * 32 + k * 16 + (lane >> 2);
            const int chunk = (lane & 3) ^ ((px >> 2) & 3);
            const uint32_t voff = (uint32_t)((b * p.H + y0) * W_ + px) * 64u + (uint32_t)chunk * 16u;
            dma16_buf(rres, voff, 0u, lds_base + (uint32_t)(HALO_RING + (wave * 2 + k) * 1024));
        }
    };
    load_rows(ys - 1, 4);
    const bool has_res = p.res != nullptr;               // wave-uniform
    if (has_res) load_res(ys);
    // weights: lane (r, h) holds output channel r, channels 8h..8h+7 of each 16-channel half of each tap
    bf16x8 bfr[9][2];
    if (tid < 32) reinterpret_cast<float*>(lds + HALO_BIAS)[tid] = p.bias[tid];     // (published by the first step's barrier)
    {
        const uint16_t* wrow = p.w + (size_t)r * 288 + 8 * h;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                bfr[tap][ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(wrow + tap * 32 + ks * 16));
    }
    const int q = wave * 32 + r;                        // output pixel of the step
    const int oy = wave >> 2, ox = q & (W_ - 1);        // waves 0-3: first output row, 4-7: second
    const int swq = (q >> 2) & 3;
    uint16_t* const out16 = reinterpret_cast<uint16_t*>(p.out);
    for (int y0 = ys; y0 < ye; y0 += 2) {
        // the rows of this step have landed: behind them in the queue are only the previous step's two row stores
        // and (has_res) this step's two residual pieces
        if (y0 == ys) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (has_res) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __syncthreads();
        const bool more = y0 + 2 < ye;
        if (more) load_rows(y0 + 3, 2);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int slot = (y0 + oy + ky) % 6;        // input row y0 - 1 + oy + ky
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int hx = ox + kx;
                const int sw = (hx >> 2) & 3;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 af = __builtin_bit_cast(
                        bf16x8, *reinterpret_cast<const uint4*>(lds + slot * HALO_ROWB + hx * 64 + (((2 * ks + h) ^ sw) << 4)));
                    acc = T16<F16>::mfma32(bfr[ky * 3 + kx][ks], af, acc);
                }
            }
        }
        // residual landed? newer than it are only the row pieces just requested (3 for waves 0-1, 2 for the others)
        if (has_res) {
            if (!more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (wave < 2) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        }
        // acc[e]: channel 8*(e>>2) + 4*h + (e&3) of pixel q. bias + SiLU + residual, one rounding, in place in LDS
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            unsigned char* cell = Cs + q * 64 + ((qq ^ swq) << 4) + 8 * h;
            const float4 b4 = *reinterpret_cast<const float4*>(lds + HALO_BIAS + (8 * qq + 4 * h) * 4);
            float v0 = acc[4 * qq] + b4.x, v1 = acc[4 * qq + 1] + b4.y, v2 = acc[4 * qq + 2] + b4.z, v3 = acc[4 * qq + 3] + b4.w;
            if (p.act == 1) { v0 = silu_fast(v0); v1 = silu_fast(v1); v2 = silu_fast(v2); v3 = silu_fast(v3); }
            else if (p.act) { v0 = act_other(p.act, v0); v1 = act_other(p.act, v1); v2 = act_other(p.act, v2); v3 = act_other(p.act, v3); }
            if (has_res) {
                const uint2 rr = *reinterpret_cast<const uint2*>(cell);
                v0 += T16<F16>::lo(rr.x); v1 += T16<F16>::hi(rr.x);
                v2 += T16<F16>::lo(rr.y); v3 += T16<F16>::hi(rr.y);
            }
            uint2 pk;
            pk.x = (uint32_t)T16<F16>::from_f32(v0) | ((uint32_t)T16<F16>::from_f32(v1) << 16);
            pk.y = (uint32_t)T16<F16>::from_f32(v2) | ((uint32_t)T16<F16>::from_f32(v3) << 16);
            *reinterpret_cast<uint2*>(cell) = pk;
        }
        // the wave's 32 pixel rows (2 KiB contiguous in NHWC) as 16-byte pieces
        const size_t m0 = (size_t)(b * p.H + y0) * W_;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int px = wave * 32 + k * 16 + (lane >> 2), cc = lane & 3;
            const uint4 v = *reinterpret_cast<const uint4*>(Cs + px * 64 + cc * 16);
            *reinterpret_cast<uint4*>(out16 + (m0 + px) * 32 + ((cc ^ ((px >> 2) & 3)) << 3)) = v;
        }
        if (more && has_res) load_res(y0 + 2);
    }
}

int launch_tiles_conv3x3(int v, const ConvArgs& a, ConvArgs& aa, hipStream_t st) {
    const bool same1 = a.stride == 1 && a.pad == 1, same2 = a.stride == 2 && (a.pad == 0 || a.pad == 1);
    if (a.gate || a.KH != 3 || a.KW != 3 || !(same1 || same2) ||
        (size_t)a.B * a.H * a.W * a.Cin * 2 + (size_t)(a.W + 1) * a.Cin * 2 >= 0x7ffffff0ull) {
        set_error("conv_igemm: variants 161-171 are un-gated 3x3 convolutions on tensors below 2 GiB");
        return ISB_ERR_INVALID;
    }
    // ISB_C3H: both 16-bit operand types; ISB_C3: bf16 only
#define ISB_C3(TM, TN, WGM, WGN)                                                                                 \
    do {                                                                                                         \
        if (a.f16) { set_error("conv_igemm: tile variant %d has no fp16 form", v); return ISB_ERR_INVALID; }     \
        const dim3 g = conv_grid(aa, 32 * TM * WGM, 32 * TN * WGN);                                              \
        hipLaunchKernelGGL((conv3x3_dma_kernel<TM, TN, WGM, WGN>), g, dim3(64 * WGM * WGN), 0, st, aa);          \
    } while (0)
#define ISB_C3H(TM, TN, WGM, WGN)                                                                                \
    do {                                                                                                         \
        const dim3 g = conv_grid(aa, 32 * TM * WGM, 32 * TN * WGN);                                              \
        if (a.f16) hipLaunchKernelGGL((conv3x3_dma_kernel<TM, TN, WGM, WGN, false, true>), g, dim3(64 * WGM * WGN), 0, st, aa); \
        else hipLaunchKernelGGL((conv3x3_dma_kernel<TM, TN, WGM, WGN>), g, dim3(64 * WGM * WGN), 0, st, aa);     \
    } while (0)
    switch (v) {
        case 171: {                                          // 3x3 32 -> 32 on 128-wide images: rows ring in LDS
            // rows per workgroup: long bands reuse the ring (each input row is copied once), but the launch should still
            // offer two workgroups to every CU
            int band = 2;
            for (int cand = 32; cand > 2; cand >>= 1)
                if (a.H % cand == 0 && (long)a.B * (a.H / cand) >= 512) { band = cand; break; }
            if (a.stride != 1 || a.pad != 1 || a.Cin != 32 || a.Cout != 32 || a.W != 128 || a.H % 2 != 0 || a.out_f32 ||
                (size_t)a.B * a.H * a.W * 64 >= 0x7ffffff0ull) {
                set_error("conv_igemm: variant 171 is the 3x3 stride-1 32 -> 32 convolution on 128-wide images (< 2 GiB)");
                return ISB_ERR_INVALID;
            }
            static DevOnce attr_set;
            if (attr_set.need()) {
                ISB_HIP(hipFuncSetAttribute((const void*)conv3x3_c32_rows_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, HALO_LDS));
                ISB_HIP(hipFuncSetAttribute((const void*)conv3x3_c32_rows_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, HALO_LDS));
                attr_set.mark();
            }
            if (a.f16) hipLaunchKernelGGL(conv3x3_c32_rows_kernel<true>, dim3(a.B * (a.H / band)), dim3(512), HALO_LDS, st, aa, band);
            else hipLaunchKernelGGL(conv3x3_c32_rows_kernel<false>, dim3(a.B * (a.H / band)), dim3(512), HALO_LDS, st, aa, band);
            break;
        }
        case 167: {                                          // 128 x 192, A operand from an LDS halo (96 channels, 32-wide maps)
            if (a.stride != 1 || a.pad != 1 || a.Cin != 96 || a.W != 32 || a.H % 4 != 0 || (a.H * a.W) % 128 != 0) {
                set_error("conv_igemm: variant 167 is the 3x3 stride-1 convolution of 96 channels on 32-wide maps");
                return ISB_ERR_INVALID;
            }
            const dim3 g = conv_grid(aa, 128, 192);
            if (a.f16) hipLaunchKernelGGL((conv3x3_dma_kernel<1, 3, 4, 2, true, true>), g, dim3(512), 0, st, aa);
            else hipLaunchKernelGGL((conv3x3_dma_kernel<1, 3, 4, 2, true>), g, dim3(512), 0, st, aa);
            break;
        }
        case 161: ISB_C3H(1, 3, 4, 2); break;   // 128 x 192
        case 162: ISB_C3H(1, 2, 4, 2); break;   // 128 x 128
        case 163: ISB_C3H(1, 1, 8, 1); break;   // 256 x  32
        case 164: ISB_C3(2, 2, 4, 2); break;    // 256 x 128
        case 165: ISB_C3(1, 2, 8, 1); break;    // 256 x  64
        case 168: ISB_C3(1, 2, 2, 2); break;    //  64 x 128, 4 waves: small-M launches (the detector's 8 x 8 / 16 x 16 maps)
        case 169: ISB_C3H(1, 1, 2, 2); break;   //  64 x  64: single frames
#ifdef ISB_BUILD_PROBES
        case 166: ISB_C3(2, 3, 4, 2); break;    // 256 x 192
#endif
        default:
            set_error("conv_igemm: tile variant %d is not in this build (3x3: 161 - 165, 167 - 169, 171)", v);
            return ISB_ERR_INVALID;
    }
#undef ISB_C3
#undef ISB_C3H
    return ISB_OK;
}

}  // namespace isb
