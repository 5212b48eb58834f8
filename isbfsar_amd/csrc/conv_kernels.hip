// EfficientNetV2-L building blocks on gfx950 (bf16 storage, f32 accumulate):
//   conv_igemm  : 3x3 / 1x1 convolution as an implicit GEMM on v_mfma_f32_32x32x16_bf16
//                 (M = B*OH*OW output pixels, N = Cout, K = KH*KW*Cin), NHWC activations,
//                 folded-BN bias + SiLU + residual epilogue, optional SE gate on the A operand
//   dwconv3x3   : depthwise 3x3 (+bias, SiLU) fused with the squeeze-excite average pool, HBM-bound
//   se_fc1/fc2  : the two squeeze-excite FCs in f32 on the vector ALU
//   stem        : conv3x3/s2 3->32 in f32 on the f32 crop
// The backbone is what the reference runs as `bbone1.engine` (utils/params.py:29, hpe.py:103);
// layer semantics follow the public efficientnetv2-l definition (isbfsar_amd/effnetv2.py).
//
// conv_igemm tiling: a k-tile is 32 input channels of ONE filter tap (Cin % 32 == 0), so a row of
// the A tile is one contiguous 64-B run of the NHWC input (or zeros for padding).  A and B tiles
// sit in LDS as [row][64 B] with the 16-B chunk index XOR-swizzled by (row>>2)&3: with that the
// ds_read_b128 fragment reads of the 32x32x16 MFMA (lane -> row lane&31, chunk 2*ks + lane>>5)
// are bank-conflict free (4-way without it).  Two LDS buffers, next tile's global loads in
// flight during the MFMAs, one barrier per k-tile.  The MFMA computes the TRANSPOSED output tile
// (weights as the A operand) so each lane ends up with 4 consecutive channels of one pixel: bias,
// SiLU and the residual are applied in registers, the bf16 tile is staged through LDS (row stride
// BN*2+16 B: conflict-free 8-byte writes) and leaves as full 16-byte pieces.
#include "conv_common.h"

namespace isb {

// epilogue shared by the register-staged and the LDS-DMA kernels
// BIAS_LDS: the tile's bias row already sits in LDS at byte offset bias_off (requested at kernel start); else it is
// read from global now.
// (8-byte stores straight from registers instead of the LDS-staged 16-byte rows measured 25 % slower.)
template <int TM, int TN, int WGM, int WGN, bool BIAS_LDS = false, bool F16 = false>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x16 (&acc)[TM][TN], unsigned char* lds, int m0, int n0,
                                              int wm, int wn, int r, int h, int tid, int bias_off = 0) {
    constexpr int NT = 64 * WGM * WGN;
    constexpr int BM = 32 * TM * WGM;
    constexpr int BN = 32 * TN * WGN;
    constexpr int CROW = BN * 2 + 16;
    // ---- epilogue. acc[i][j][e]: channel n = n0 + (wn*TN+j)*32 + 8*(e>>2) + 4*h + (e&3), pixel m = m0 + (wm*TM+i)*32 + r
    if (p.splits > 1) {       // split-K partial: raw f32 accumulators to part[split][M][Cout]
        float* out32 = p.part + (size_t)blockIdx.z * p.M * p.Cout;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + (wm * TM + i) * 32 + r;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n0 + (wn * TN + j) * 32 + 8 * q + 4 * h;
                    if (n >= p.Cout) continue;
                    *reinterpret_cast<float4*>(out32 + (size_t)m * p.Cout + n) =
                        make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                }
        }
        return;
    }
    if (p.out_f32) {          // f32 output (last 1x1 conv feeding the f32 pose head): direct stores
        float* out32 = reinterpret_cast<float*>(p.out);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + (wm * TM + i) * 32 + r;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n0 + (wn * TN + j) * 32 + 8 * q + 4 * h;
                    if (n >= p.Cout) continue;
                    const float4 bs = *reinterpret_cast<const float4*>(p.bias + n);
                    float4 v = make_float4(acc[i][j][4 * q] + bs.x, acc[i][j][4 * q + 1] + bs.y, acc[i][j][4 * q + 2] + bs.z,
                                           acc[i][j][4 * q + 3] + bs.w);
                    if (p.act == 1) { v.x = silu_fast(v.x); v.y = silu_fast(v.y); v.z = silu_fast(v.z); v.w = silu_fast(v.w); }
                    else if (p.act) { v.x = act_other(p.act, v.x); v.y = act_other(p.act, v.y); v.z = act_other(p.act, v.z); v.w = act_other(p.act, v.w); }
                    *reinterpret_cast<float4*>(out32 + (size_t)m * p.Cout + n) = v;
                }
        }
        return;
    }
    // bf16 output: bias + SiLU + residual in registers (one rounding), stage the tile in LDS, then
    // write full 16-byte pieces, 256 B contiguous per pixel row. Tiles too wide to stage (> 64 KiB)
    // store their 8-byte packed pieces straight from registers (the L2 merges the partial lines).
    constexpr bool STAGE = BM * CROW <= 65536;
    uint16_t* out16 = reinterpret_cast<uint16_t*>(p.out);
    unsigned char* Cs = lds;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ml = (wm * TM + i) * 32 + r;
        const int m = m0 + ml;
        const bool mok = m < p.M;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = (wn * TN + j) * 32 + 8 * q + 4 * h;
                const int n = n0 + nl;
                float v0 = acc[i][j][4 * q], v1 = acc[i][j][4 * q + 1], v2 = acc[i][j][4 * q + 2], v3 = acc[i][j][4 * q + 3];
                if (n < p.Cout) {
                    float4 bs;
                    if constexpr (BIAS_LDS) bs = *reinterpret_cast<const float4*>(lds + bias_off + nl * 4);
                    else bs = *reinterpret_cast<const float4*>(p.bias + n);
                    v0 += bs.x; v1 += bs.y; v2 += bs.z; v3 += bs.w;
                    const int act_now = p.act_after_res ? 0 : p.act;
                    if (act_now == 1) { v0 = silu_fast(v0); v1 = silu_fast(v1); v2 = silu_fast(v2); v3 = silu_fast(v3); }
                    else if (act_now) { v0 = act_other(act_now, v0); v1 = act_other(act_now, v1); v2 = act_other(act_now, v2); v3 = act_other(act_now, v3); }
                    if (p.res && mok) {       // (requesting all residual pieces up front measured 5-7 % slower, twice)
                        const uint2 rr = *reinterpret_cast<const uint2*>(p.res + (size_t)m * p.Cout + n);
                        v0 += T16<F16>::lo(rr.x); v1 += T16<F16>::hi(rr.x);
                        v2 += T16<F16>::lo(rr.y); v3 += T16<F16>::hi(rr.y);
                    }
                    if (p.act_after_res && p.act) { v0 = act_other(p.act, v0); v1 = act_other(p.act, v1); v2 = act_other(p.act, v2); v3 = act_other(p.act, v3); }
                }
                uint2 pk;
                pk.x = (uint32_t)T16<F16>::from_f32(v0) | ((uint32_t)T16<F16>::from_f32(v1) << 16);
                pk.y = (uint32_t)T16<F16>::from_f32(v2) | ((uint32_t)T16<F16>::from_f32(v3) << 16);
                if constexpr (STAGE) *reinterpret_cast<uint2*>(Cs + ml * CROW + nl * 2) = pk;
                else if (mok && n < p.Cout) *reinterpret_cast<uint2*>(out16 + (size_t)m * (p.out_ld ? p.out_ld : p.Cout) + n) = pk;
            }
    }
    if constexpr (!STAGE) return;
    __syncthreads();
    constexpr int CPR = BN / 8;                            // 16-byte pieces per tile row
    const int ldo = p.out_ld ? p.out_ld : p.Cout;          // (a channel slice of a wider tensor: ConvArgs.out_ld)
#pragma unroll 4
    for (int id = tid; id < BM * CPR; id += NT) {
        const int row = id / CPR, cc = id - row * CPR;
        const int m = m0 + row, n = n0 + cc * 8;
        if (m < p.M && n < p.Cout)
            *reinterpret_cast<uint4*>(out16 + (size_t)m * ldo + n) = *reinterpret_cast<const uint4*>(Cs + row * CROW + cc * 16);
    }
}

// Wave-local epilogue of the lean kernels (bf16 output) -- built, bit-identical, NOT the default (ISB_EPI_SHARED). In-kernel clocks (round 2, s_memtime, 256 frames) put the shared
// epilogue above at 9 200 cycles per workgroup on the 224 -> 1344 expand (37 % of the workgroup's life) and 26 200 on the
// 1344 -> 224 projection (25 %): all waves stage the whole tile, meet at a barrier, then walk a store loop with a division,
// 64-bit address arithmetic and a bounds branch per 16-byte piece -- and a projection first fetches its residual as 28
// scattered 8-byte loads per lane, each waited for in turn. Here every wave finishes its OWN 32 x 32 blocks, one at a time,
// through a private 2.5-KiB LDS area: the residual block arrives as two coalesced 16-byte loads per lane one block ahead,
// (all blocks requested up front) is transposed through the area into the accumulator layout, bias / SiLU / residual / one rounding happen in registers
// exactly as above (same operations, same order: bit-identical), the packed block goes back through the same area and
// leaves as two 16-byte stores per lane (rows past M masked). No barrier; whole 32-column blocks past Cout are skipped
// (wave-uniform).
// The caller guarantees that every wave of the workgroup has left the k loop (the loop's last barrier): the areas overlay
// the operand buffers.
#ifndef ISB_EPI_SHARED
#define ISB_EPI_SHARED 1          // build-time A/B switch: 1 = the lean kernels keep the shared epilogue above (the default:
#endif                            // same-session A/B, 256 frames: shared 639 TFLOP/s on the convolution family, wave-local 609)
constexpr int WL_SROW = 80, WL_AREA = 32 * WL_SROW;
template <int TM, int TN, int WGM, int WGN>
__device__ __forceinline__ void conv_epilogue_wl(const ConvArgs& p, f32x16 (&acc)[TM][TN], unsigned char* lds, int m0, int n0,
                                                 int wm, int wn, int r, int h, int lane, int wave, int bias_off) {
    unsigned char* const st = lds + wave * WL_AREA;
    uint16_t* const out16 = reinterpret_cast<uint16_t*>(p.out);
    const int row16 = lane >> 2, cc = lane & 3;
    const bool has_res = p.res != nullptr;                // wave-uniform
    auto res_load = [&](int b, uint4 (&rr)[2]) {          // block b = i * TN + j: rows 16 k2 + row16, channels 8 cc .. + 7
        const int i = b / TN, j = b - i * TN;
        const int nb = n0 + (wn * TN + j) * 32;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const int m = min(m0 + (wm * TM + i) * 32 + 16 * k2 + row16, p.M - 1);
            rr[k2] = nb < p.Cout ? *reinterpret_cast<const uint4*>(p.res + (size_t)m * p.Cout + nb + cc * 8) : make_uint4(0, 0, 0, 0);
        }
    };
    // every residual block of the wave is requested at once: coalesced 64-byte row segments, ONE memory round trip for
    // the whole epilogue (2 x TM x TN registers of 16 bytes)
    uint4 rall[TM * TN][2];
    if (has_res) {
#pragma unroll
        for (int b = 0; b < TM * TN; ++b) res_load(b, rall[b]);
    }
#pragma unroll
    for (int b = 0; b < TM * TN; ++b) {
        const int i = b / TN, j = b - i * TN;
        const int nl0 = (wn * TN + j) * 32, nb = n0 + nl0;
        if (nb >= p.Cout) continue;                                      // tile overhang past the last channel: nothing to store
        if (has_res) {
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) *reinterpret_cast<uint4*>(st + (16 * k2 + row16) * WL_SROW + cc * 16) = rall[b][k2];
        }
        uint2 pk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bs = *reinterpret_cast<const float4*>(lds + bias_off + (nl0 + 8 * q + 4 * h) * 4);
            float v0 = acc[i][j][4 * q] + bs.x, v1 = acc[i][j][4 * q + 1] + bs.y, v2 = acc[i][j][4 * q + 2] + bs.z, v3 = acc[i][j][4 * q + 3] + bs.w;
            if (p.act == 1) { v0 = silu_fast(v0); v1 = silu_fast(v1); v2 = silu_fast(v2); v3 = silu_fast(v3); }
            else if (p.act) { v0 = act_other(p.act, v0); v1 = act_other(p.act, v1); v2 = act_other(p.act, v2); v3 = act_other(p.act, v3); }
            if (has_res) {
                const uint2 rr = *reinterpret_cast<const uint2*>(st + r * WL_SROW + q * 16 + h * 8);
                v0 += bf2f_((uint16_t)(rr.x & 0xffff)); v1 += bf2f_((uint16_t)(rr.x >> 16));
                v2 += bf2f_((uint16_t)(rr.y & 0xffff)); v3 += bf2f_((uint16_t)(rr.y >> 16));
            }
            pk[q].x = (uint32_t)f2bf_(v0) | ((uint32_t)f2bf_(v1) << 16);
            pk[q].y = (uint32_t)f2bf_(v2) | ((uint32_t)f2bf_(v3) << 16);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<uint2*>(st + r * WL_SROW + q * 16 + h * 8) = pk[q];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const uint4 v = *reinterpret_cast<const uint4*>(st + (16 * k2 + row16) * WL_SROW + cc * 16);
            const int m = m0 + (wm * TM + i) * 32 + 16 * k2 + row16;
            if (m < p.M) *reinterpret_cast<uint4*>(out16 + (size_t)m * (p.out_ld ? p.out_ld : p.Cout) + nb + cc * 8) = v;
        }
    }
}

// workgroup -> output tile. Mode 0: 2-D grid. Modes 1/2: 1-D grid; hardware hands consecutive workgroup ids
// to the 8 XCDs round-robin, so id % 8 names the XCD (and its private L2) a workgroup runs on. Within an XCD
// the N tiles of one M tile are consecutive: the A rows are fetched into that L2 once and re-read from it.
__device__ __forceinline__ bool conv_tile_origin(const ConvArgs& p, int BM, int BN, int& m0, int& n0) {
    if (p.grid_mode == 0) {
        m0 = blockIdx.x * BM;
        n0 = blockIdx.y * BN;
        return true;
    }
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int nt = slot % p.grid_n, ml = slot / p.grid_n;
    const int per = (p.grid_m + 7) >> 3;
    const int mt = p.grid_mode == 1 ? ml * 8 + xcd : xcd * per + ml;
    m0 = mt * BM;
    n0 = nt * BN;
    return mt < p.grid_m;
}

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

// -------------------------------------------------------------------------------------------
// Fused MBConv front half: 1x1 expand (this GEMM) -> folded BN + SiLU -> depthwise 3x3 (stride 1)
// -> folded BN + SiLU -> D, plus the squeeze-excite average pool -- without the expanded tensor E ever
// leaving the chip. The M tile is exactly ONE sample (BM == H*W pixels in raster order), so after the
// bias/SiLU'd bf16 E tile has been staged in LDS as [pixel][channel] the 3x3 neighbourhood of every
// output is in LDS. thread = (8-channel chunk, pixel quad) as in dwconv3x3_pool_kernel.
// -------------------------------------------------------------------------------------------
template <int TM, int TN, int WGM, int WGN>
__device__ __forceinline__ void conv_epilogue_dw(const ConvArgs& p, f32x16 (&acc)[TM][TN], unsigned char* lds, int m0, int n0,
                                                 int wm, int wn, int r, int h, int tid) {
    constexpr int NT = 64 * WGM * WGN;
    constexpr int BM = 32 * TM * WGM;
    constexpr int BN = 32 * TN * WGN;
    constexpr int CROW = BN * 2 + 16;
    constexpr int CH = BN / 8;                             // 8-channel chunks in the slab
    constexpr int NQ = BM / 4;                             // pixel quads of the sample
    static_assert(CH * NQ == NT, "one (chunk, quad) item per thread");
    unsigned char* Cs = lds;
    float* red = reinterpret_cast<float*>(lds + BM * CROW);   // [NQ][BN] partial pool sums
    // stage E = silu(acc + bias) as bf16 [pixel][channel]
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ml = (wm * TM + i) * 32 + r;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = (wn * TN + j) * 32 + 8 * q + 4 * h;
                const float4 bs = *reinterpret_cast<const float4*>(p.bias + n0 + nl);
                const float v0 = silu_fast(acc[i][j][4 * q] + bs.x), v1 = silu_fast(acc[i][j][4 * q + 1] + bs.y);
                const float v2 = silu_fast(acc[i][j][4 * q + 2] + bs.z), v3 = silu_fast(acc[i][j][4 * q + 3] + bs.w);
                uint2 pk;
                pk.x = (uint32_t)f2bf_(v0) | ((uint32_t)f2bf_(v1) << 16);
                pk.y = (uint32_t)f2bf_(v2) | ((uint32_t)f2bf_(v3) << 16);
                *reinterpret_cast<uint2*>(Cs + ml * CROW + nl * 2) = pk;
            }
    }
    __syncthreads();
    // depthwise 3x3, stride 1, pad 1 over the staged sample
    const int W = p.OW, Hh = p.OH;
    const int cl = tid % CH, pq = tid / CH;
    const int c = n0 + cl * 8;                             // global expanded channel
    const int qpr = W >> 2;
    const int oy = pq / qpr, ox0 = (pq - oy * qpr) * 4;
    const int b = m0 / BM;
    float w[9][8], a4[4][8], psum[8];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float4 w0 = *reinterpret_cast<const float4*>(p.dw_w + (size_t)t * p.Cout + c), w1 = *reinterpret_cast<const float4*>(p.dw_w + (size_t)t * p.Cout + c + 4);
        w[t][0] = w0.x; w[t][1] = w0.y; w[t][2] = w0.z; w[t][3] = w0.w; w[t][4] = w1.x; w[t][5] = w1.y; w[t][6] = w1.z; w[t][7] = w1.w;
    }
    {
        const float4 s0 = *reinterpret_cast<const float4*>(p.dw_bias + c), s1 = *reinterpret_cast<const float4*>(p.dw_bias + c + 4);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            a4[o][0] = s0.x; a4[o][1] = s0.y; a4[o][2] = s0.z; a4[o][3] = s0.w; a4[o][4] = s1.x; a4[o][5] = s1.y; a4[o][6] = s1.z; a4[o][7] = s1.w;
        }
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy - 1 + ky;
        if ((unsigned)iy >= (unsigned)Hh) continue;
#pragma unroll
        for (int col = 0; col < 6; ++col) {
            const int ix = ox0 - 1 + col;
            if ((unsigned)ix >= (unsigned)W) continue;
            const uint4 v = *reinterpret_cast<const uint4*>(Cs + (iy * W + ix) * CROW + cl * 16);
            float x[8];
            x[0] = bf2f_((uint16_t)(v.x & 0xffff)); x[1] = bf2f_((uint16_t)(v.x >> 16));
            x[2] = bf2f_((uint16_t)(v.y & 0xffff)); x[3] = bf2f_((uint16_t)(v.y >> 16));
            x[4] = bf2f_((uint16_t)(v.z & 0xffff)); x[5] = bf2f_((uint16_t)(v.z >> 16));
            x[6] = bf2f_((uint16_t)(v.w & 0xffff)); x[7] = bf2f_((uint16_t)(v.w >> 16));
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const int kx = col - o;
                if (kx >= 0 && kx < 3) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) a4[o][e] = fmaf(x[e], w[ky * 3 + kx][e], a4[o][e]);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) psum[e] = 0.f;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        uint32_t pk[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint16_t lo = f2bf_(silu_fast(a4[o][2 * e])), hi = f2bf_(silu_fast(a4[o][2 * e + 1]));
            pk[e] = (uint32_t)lo | ((uint32_t)hi << 16);
            psum[2 * e] += bf2f_(lo);
            psum[2 * e + 1] += bf2f_(hi);
        }
        *reinterpret_cast<uint4*>(p.dw_out + (((size_t)(b * Hh + oy) * W + ox0 + o) * p.Cout + c)) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[pq * BN + cl * 8 + e] = psum[e];
    __syncthreads();
    if (tid < BN) {
        float t = 0.f;
        for (int s2 = 0; s2 < NQ; ++s2) t += red[s2 * BN + tid];
        p.pooled[(size_t)b * p.Cout + n0 + tid] = t / (float)BM;
    }
}

// -------------------------------------------------------------------------------------------
// LDS-DMA variant (convolutions without an SE gate): tiles go global -> LDS directly
// (global_load_lds_dwordx4: 1 KiB per wave-instruction, destination = wave-uniform base + lane*16,
// so the XOR swizzle is applied to the per-lane SOURCE address). No staging registers, no
// ds_write instructions; padding taps and rows past M read a 16-byte zero line. The DMA of tile
// k+1 is in flight while tile k feeds the MFMAs; __syncthreads() waits for it (vmcnt) before the swap.
// -------------------------------------------------------------------------------------------


// One LDS-DMA instruction (1 KiB per wave: lane l's 16 bytes land at ldst + 16 l), issued through inline asm.
// Through __builtin_amdgcn_global_load_lds the compiler sees an LDS store and orders EVERY later LDS read behind
// it (s_waitcnt vmcnt(0) before the first ds_read of each k-step): the tile just requested was waited for before
// the MFMAs of the current one, and nothing overlapped. Hidden in asm, completion is ours to track: every
// consumer below waits with an explicit (counted) s_waitcnt vmcnt before the barrier that publishes a tile.
__device__ __forceinline__ void dma16_at(const void* gsrc, uint32_t lds_addr) {    // LDS destination as a byte address
    const uint32_t la = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(la) : "memory");
}
__device__ __forceinline__ void dma16(const void* gsrc, unsigned char* ldst) {
    const uint32_t la = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_ptr_t)ldst);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(la) : "memory");   // m0 is RESERVED on amdgcn:
    // the compiler writes it immediately before each of its own uses and keeps nothing live in it across statements, so
    // no clobber is declared (naming a reserved register in the clobber list is itself diagnosed as undefined behaviour)
}


// k-tile width KT: 32 (64-B LDS rows, 16 rows per 1-KiB DMA piece) or 64 (128-B rows = whole cache lines per
// pixel row, 8 rows per piece, half as many barriers and line requests per byte; needs Cin % 64 == 0)
template <int KT>
__device__ __forceinline__ int swz_kt(int row, int chunk) {
    if constexpr (KT == 32) return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);
    else return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);     // conflict-free for ds_read_b128's 16-lane groups
}

template <int TM, int TN, int WGM, int WGN, int NB, bool DW = false, bool GATE = false, int KT = 32>
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
    constexpr int LDS_DW = BM * CROW + (BM / 4) * BN * 4;   // staged E tile + partial pool sums
    constexpr int LDS_BYTES = DW ? (LDS_DW > NB * BUF ? LDS_DW : NB * BUF) : LDS_PLAIN;
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

    if constexpr (!GATE && !DW) {              // the tile's bias row rides along with the first k-step
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
    if constexpr (DW) conv_epilogue_dw<TM, TN, WGM, WGN>(p, acc, lds, m0, n0, wm, wn, r, h, tid);
    else conv_epilogue<TM, TN, WGM, WGN, !GATE>(p, acc, lds, m0, n0, wm, wn, r, h, tid, BIAS_OFF);
}

// -------------------------------------------------------------------------------------------
// 1x1 convolutions without an SE gate = plain GEMMs out[M,Cout] = in[M,Cin] . w[Cout,Cin]^T. PMC counters on the
// general kernel above (profiles/README.md) show its waves spend 5x more issue cycles on bookkeeping (tap
// arithmetic, 64-bit addresses, zero-line selects: ~190 scalar+vector instructions per k-step) than on the 6 MFMAs
// the step exists for. Here the k loop is stripped to what a GEMM needs:
//   * a lane's source offset never changes (row * Cin * 2 + swizzled chunk, rows past the edge clamped: their
//     results are never stored), so each DMA is "scalar base + 32-bit lane offset" and the scalar base just
//     advances 64 bytes per step: no vector instruction per DMA;
//   * the k loop is unrolled by two, so buffer offsets are immediates of ds_read_b128 and the fragment addresses
//     are four registers computed once;
//   * per step: 3 DMA + 8 ds_read + 6 MFMA + ~10 scalar instructions (128 x 192 tile).
// -------------------------------------------------------------------------------------------

// GATE: 0 = none, 1 = gate rows read from p.gate, 2 = gate of the workgroup's k-range computed here from the
// squeeze-excite FC1 partials (single-frame split-K launches: se_fc2_kernel's arithmetic, in its order)
// STAMPS (tuning probe, isb_debug_conv variant 9000 + v): wave 0 of the first 64 workgroups sums s_memtime intervals over
// its k loop -- waiting for the DMA (vmcnt), waiting at the barrier, the rest (fragment reads + MFMAs) -- and writes
// {prologue, DMA wait, barrier wait, whole k loop, epilogue, k-steps} to p.part[workgroup]
// NBUF = 3: three k-step buffers, requests run TWO steps ahead and the wait before a step is a counted vmcnt (the pieces of
// the step after it may still fly) instead of vmcnt(0).
template <int TM, int TN, int WGM, int WGN, int GATE = 0, bool STAMPS = false, int NBUF = 2, bool F16 = false>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm1x1_dma_kernel(ConvArgs p) {
    uint64_t st_t0 = 0, st_wait = 0, st_bar = 0, st_loop0 = 0, st_loop1 = 0;
    if constexpr (STAMPS) st_t0 = __builtin_amdgcn_s_memtime();
    constexpr int NW = WGM * WGN;
    constexpr int BM = 32 * TM * WGM;
    constexpr int BN = 32 * TN * WGN;
    constexpr int A_INST = BM / 16, B_INST = BN / 16;
    constexpr int A_PW = (A_INST + NW - 1) / NW, B_PW = (B_INST + NW - 1) / NW;
    constexpr int BUF = (BM + BN) * ROWB;
    constexpr int CROW = BN * 2 + 16;
    constexpr int LDS_BYTES = (NBUF * BUF > BM * CROW || BM * CROW > 65536) ? NBUF * BUF : BM * CROW;
    constexpr int GATE_OFF = NBUF * BUF;                       // GATE: f32 gate rows of the tile's samples (dynamic LDS)
    // the tile's bias row is requested with the first k-step and sits behind everything else in LDS (the epilogue's
    // staging area overlays the k-loop buffers): no global-load latency between the last MFMA and the first store
    __shared__ __attribute__((aligned(16))) unsigned char lds_static[GATE ? 16 : LDS_BYTES + BN * 4];
    unsigned char* const lds = GATE ? conv_lds_dyn : lds_static;
    const int bias_off = GATE ? p.grid_bias_off : LDS_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int r = lane & 31, h = lane >> 5;
    int m0, n0;
    if (!conv_tile_origin(p, BM, BN, m0, n0)) return;

    uint32_t a_voff[A_PW], b_voff[B_PW];
#pragma unroll
    for (int s = 0; s < A_PW; ++s) {
        const int row = 16 * (wave + NW * s) + (lane >> 2);
        const int logical = (lane & 3) ^ ((row >> 2) & 3);
        a_voff[s] = (uint32_t)min(m0 + row, p.M - 1) * (uint32_t)(p.Cin * 2) + logical * 16;
    }
#pragma unroll
    for (int s = 0; s < B_PW; ++s) {
        const int row = 16 * (wave + NW * s) + (lane >> 2);
        const int logical = (lane & 3) ^ ((row >> 2) & 3);
        b_voff[s] = (uint32_t)min(n0 + row, p.Cout - 1) * (uint32_t)(p.Cin * 2) + logical * 16;
    }
    // k range of this workgroup: all of K, or one split of it (blockIdx.z)
    int kt_first = 0, nkt = p.Cin / CK;
    if (p.splits > 1) {
        const int per = (nkt + p.splits - 1) / p.splits;
        kt_first = min((int)blockIdx.z * per, nkt);
        nkt = min(per, nkt - kt_first);
    }
    // GATE == 2: lane = 4 channels of the k-range, wave = a quarter of the FC2 inputs; the weight rows are requested
    // before anything else so that they travel while the first tiles do
    float4 wpre[GATE == 2 ? 40 : 1];
    int se_jb = 0, se_je = 0;
    bool se_cok = false;
    if constexpr (GATE == 2) {
        static_assert(NW == 4, "the folded FC2 splits its inputs over 4 waves");
        const int c = kt_first * CK + lane * 4;
        se_cok = lane * 4 < nkt * CK;
        const int jq = (p.se_cse + 3) >> 2;
        se_jb = wave * jq;
        se_je = min(p.se_cse, se_jb + jq);
#pragma unroll
        for (int q = 0; q < 40; ++q)
            wpre[q] = (se_cok && se_jb + q < se_je) ? *reinterpret_cast<const float4*>(p.se_w2t + (size_t)(se_jb + q) * p.Cin + c)
                                                    : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const unsigned char* a_base = reinterpret_cast<const unsigned char*>(p.in) + kt_first * (CK * 2);
    const unsigned char* b_base = reinterpret_cast<const unsigned char*>(p.w) + kt_first * (CK * 2);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)lds + wave * 1024;
    if (wave == 0) {
#pragma unroll
        for (int o = 0; o < BN / 4; o += 64)       // 64 lanes x 4 floats per instruction
            if (lane + o < BN / 4)
                dma16_s(p.bias, (uint32_t)min(n0 + (lane + o) * 4, p.Cout - 4) * 4,
                        (uint32_t)(uintptr_t)(lds_ptr_t)lds + bias_off + o * 16);
    }
    auto dma = [&](auto bufc) {                 // requests the NEXT 32 channels, then advances the scalar bases
        constexpr int buf = decltype(bufc)::value;
#pragma unroll
        for (int s = 0; s < A_PW; ++s)
            if (wave + NW * s < A_INST) dma16_s(a_base, a_voff[s], lds0 + (buf * BUF + NW * s * 1024));
#pragma unroll
        for (int s = 0; s < B_PW; ++s)
            if (wave + NW * s < B_INST) dma16_s(b_base, b_voff[s], lds0 + (buf * BUF + BM * ROWB + NW * s * 1024));
        a_base += CK * 2;
        b_base += CK * 2;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment addresses: row blocks of 32 are 2048 bytes apart and leave the swizzle untouched -> immediates
    int a_sw[2], b_sw[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        a_sw[ks] = swz(wm * TM * 32 + r, 2 * ks + h);
        b_sw[ks] = BM * ROWB + swz(wn * TN * 32 + r, 2 * ks + h);
    }
    // SE gate: the A fragment is scaled as it leaves LDS, bf16(f32(x) * g), exactly what the register-staged
    // kernel does at its LDS store. g_row: float offset of this lane's sample row (+ its 8-channel half)
    dma(std::integral_constant<int, 0>{});      // first tile in flight while the gate rows are staged
    int g_row[TM];
    int kt_now = kt_first;
    if constexpr (GATE == 2) {
        float* gate_s = reinterpret_cast<float*>(lds + GATE_OFF);            // [256] gate of this k-range
        float* mids = gate_s + 256;                                          // [160]
        float4* red = reinterpret_cast<float4*>(mids + 160);                 // [4][64]
        const int sample = m0 / (p.OH * p.OW);
        if (tid < p.se_cse) {
            float pv[SE_MAX_PARTS];
#pragma unroll
            for (int kc = 0; kc < SE_MAX_PARTS; ++kc)
                pv[kc] = kc < p.se_nparts ? p.se_part[((size_t)kc * p.B + sample) * p.se_cse + tid] : 0.f;
            float v = p.se_b1[tid];
#pragma unroll
            for (int kc = 0; kc < SE_MAX_PARTS; ++kc)
                if (kc < p.se_nparts) v += pv[kc];
            mids[tid] = v / (1.0f + expf(-v));
        }
        __syncthreads();
        float4 ga = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < 40; ++q) {
            if (se_jb + q >= se_je) break;
            const float mv = mids[se_jb + q];
            ga.x = fmaf(mv, wpre[q].x, ga.x);
            ga.y = fmaf(mv, wpre[q].y, ga.y);
            ga.z = fmaf(mv, wpre[q].z, ga.z);
            ga.w = fmaf(mv, wpre[q].w, ga.w);
        }
        red[wave * 64 + lane] = ga;
        __syncthreads();
        if (wave == 0 && se_cok) {
            float4 v = red[lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float4 u = red[w * 64 + lane];
                v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
            }
            const float4 bias = *reinterpret_cast<const float4*>(p.se_b2 + kt_first * CK + lane * 4);
            v.x = 1.0f / (1.0f + expf(-(v.x + bias.x)));
            v.y = 1.0f / (1.0f + expf(-(v.y + bias.y)));
            v.z = 1.0f / (1.0f + expf(-(v.z + bias.z)));
            v.w = 1.0f / (1.0f + expf(-(v.w + bias.w)));
            *reinterpret_cast<float4*>(gate_s + lane * 4) = v;
        }
        kt_now = 0;                                 // gate_s is indexed from the start of the k-range; published below
#pragma unroll
        for (int i = 0; i < TM; ++i) g_row[i] = 8 * h;
    } else if constexpr (GATE == 1) {
        const int ohw = p.OH * p.OW;
        const int s_first = m0 / ohw;
#pragma unroll
        for (int i = 0; i < TM; ++i) g_row[i] = (min(m0 + (wm * TM + i) * 32 + r, p.M - 1) / ohw - s_first) * p.Cin + 8 * h;
        const int ns = min(m0 + BM - 1, p.M - 1) / ohw - s_first + 1;
        const float* src = p.gate + (size_t)s_first * p.Cin;
        float* dst = reinterpret_cast<float*>(lds + GATE_OFF);
        for (int idx = tid * 4; idx < ns * p.Cin; idx += 64 * NW * 4)
            *reinterpret_cast<float4*>(dst + idx) = *reinterpret_cast<const float4*>(src + idx);
    }
    auto compute = [&](auto bufc) {
        constexpr int buf = decltype(bufc)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                uint4 v = *reinterpret_cast<const uint4*>(lds + a_sw[ks] + (buf * BUF + i * 2048));
                if constexpr (GATE) {
                    const float* gs = reinterpret_cast<const float*>(lds + GATE_OFF) + g_row[i] + kt_now * CK + ks * 16;
                    const float4 g0 = *reinterpret_cast<const float4*>(gs), g1 = *reinterpret_cast<const float4*>(gs + 4);
                    v = T16<F16>::gate8(v, g0, g1);
                }
                af[i] = v;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bfr[j] = *reinterpret_cast<const uint4*>(lds + b_sw[ks] + (buf * BUF + j * 2048));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = T16<F16>::mfma32(bfr[j], af[i], acc[i][j]);
        }
        ++kt_now;
    };
    auto publish = [&]() {
        if constexpr (STAMPS) {
            const uint64_t a = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint64_t b = __builtin_amdgcn_s_memtime();
            __syncthreads();
            const uint64_t c = __builtin_amdgcn_s_memtime();
            st_wait += b - a;
            st_bar += c - b;
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    };

    if constexpr (NBUF == 3) {
        // pieces this wave requests per k-step (wave-uniform): the counted wait needs it as an immediate
        int n_req = 0;
#pragma unroll
        for (int s = 0; s < A_PW; ++s) n_req += (wave + NW * s < A_INST) ? 1 : 0;
#pragma unroll
        for (int s = 0; s < B_PW; ++s) n_req += (wave + NW * s < B_INST) ? 1 : 0;
        static_assert(A_PW + B_PW <= 8, "counted waits for up to 8 pieces per wave and k-step");
        auto wait_landed = [&](bool next_in_flight) {       // the step about to be read has landed; only the step after it may fly
            if (!next_in_flight) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            else switch (n_req) {
                case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
                case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
            __builtin_amdgcn_s_barrier();
        };
        // the gate rows / bias were staged with ordinary stores and the bias DMA above: make them visible once
        if (nkt > 1) dma(std::integral_constant<int, 1>{});
        if (nkt > 1) {                                      // step 0 landed, step 1 may fly
            switch (n_req) {
                case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
                case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if constexpr (STAMPS) { st_loop0 = __builtin_amdgcn_s_memtime(); st_wait = 0; st_bar = 0; }
        for (int kt = 0; kt < nkt; kt += 3) {               // step kt in buffer 0, kt + 1 in buffer 1, kt + 2 in buffer 2 (nkt % 3 == 0: launcher)
            dma(std::integral_constant<int, 2>{});
            compute(std::integral_constant<int, 0>{});
            wait_landed(true);
            const bool more = kt + 3 < nkt;
            if (more) dma(std::integral_constant<int, 0>{});
            compute(std::integral_constant<int, 1>{});
            wait_landed(more);
            if (more) dma(std::integral_constant<int, 1>{});
            compute(std::integral_constant<int, 2>{});
            wait_landed(more);
        }
        __syncthreads();
    } else {
    publish();
    if constexpr (STAMPS) { st_loop0 = __builtin_amdgcn_s_memtime(); st_wait = 0; st_bar = 0; }
    int kt = 0;
    for (; kt + 2 <= nkt; kt += 2) {            // straight-line body: tile kt in buffer 0, tile kt+1 in buffer 1
        dma(std::integral_constant<int, 1>{});
        compute(std::integral_constant<int, 0>{});
        publish();
        if (kt + 2 < nkt) dma(std::integral_constant<int, 0>{});
        compute(std::integral_constant<int, 1>{});
        publish();
    }
    if (kt < nkt) {                             // odd tail: requested into buffer 0 and published above
        compute(std::integral_constant<int, 0>{});
        __syncthreads();
    }
    }
    if constexpr (STAMPS) st_loop1 = __builtin_amdgcn_s_memtime();
    if (ISB_EPI_SHARED || p.splits > 1 || p.out_f32) conv_epilogue<TM, TN, WGM, WGN, true, F16>(p, acc, lds, m0, n0, wm, wn, r, h, tid, bias_off);
    else conv_epilogue_wl<TM, TN, WGM, WGN>(p, acc, lds, m0, n0, wm, wn, r, h, lane, wave, bias_off);
    if constexpr (STAMPS) {
        const uint64_t t_end = __builtin_amdgcn_s_memtime();
        const int wgid = blockIdx.x + gridDim.x * blockIdx.y;
        if (tid == 0 && wgid < 64) {
            uint64_t* o = reinterpret_cast<uint64_t*>(p.part) + (size_t)wgid * 8;
            o[0] = st_loop0 - st_t0; o[1] = st_wait; o[2] = st_bar; o[3] = st_loop1 - st_loop0; o[4] = t_end - st_loop1; o[5] = (uint64_t)nkt;
        }
    }
}

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

// -------------------------------------------------------------------------------------------
// 3x3 convolutions (stride 1 pad 1, or stride 2 with TF-SAME bottom/right padding) with the same lean k loop as
// gemm1x1_dma_kernel. The A operand is addressed as a RAW BUFFER: a lane's byte offset is fixed (its output pixel,
// window origin), the filter tap and channel block are one SCALAR offset per k-step, and padding costs no data
// movement at all -- a lane whose tap falls outside the image (9-bit mask, computed once) submits an out-of-range
// offset and the hardware returns zeros (buffer_load ... lds). The buffer base sits pad*(W+1) pixels before the
// tensor so that window origins of border pixels are non-negative offsets.
// -------------------------------------------------------------------------------------------
typedef int i32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16_buf(i32x4_t rsrc, uint32_t voff, uint32_t soff, uint32_t lds_addr) {
    const uint32_t la = __builtin_amdgcn_readfirstlane(lds_addr);
    const uint32_t so = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(voff), "s"(rsrc), "s"(la), "s"(so) : "memory");
}

// HALO (stride 1, 96 input channels, 32-wide maps, 128-pixel tiles = four image rows: the body blocks of stage 3): the
// A fragments come from the tile's input halo in LDS (6 rows x 34 pixels, 256-byte pixel rows of which 192 B are used,
// chunk slot = chunk ^ (pixel & 7)), copied once, like fused_mb_kernel<.., HALO>; the k loop streams only the weights
// (12 instead of 20 KiB per k-step). Same (tap, channel) order: bit-identical.
template <int TM, int TN, int WGM, int WGN, bool HALO = false>
__global__ __launch_bounds__(64 * WGM * WGN) void conv3x3_dma_kernel(ConvArgs p) {
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
            const int chunk = (lane & 15) ^ (hp & 7);
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
                af[0] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + hp * 256 + (((4 * h_cb + 2 * ks + h) ^ (hp & 7)) << 4)));
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
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
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
    if (ISB_EPI_SHARED || p.out_f32) conv_epilogue<TM, TN, WGM, WGN, true>(p, acc, lds, m0, n0, wm, wn, r, h, tid, bias_off);
    else conv_epilogue_wl<TM, TN, WGM, WGN>(p, acc, lds, m0, n0, wm, wn, r, h, lane, wave, bias_off);
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
// ((pixel >> 1) & 3): eight consecutive pixels hit all 32 banks for any tap shift. A step's 256 outputs are
// consecutive NHWC pixels, so the shared epilogue (bias, SiLU, residual, 16-byte row stores) applies unchanged.
// Sums run in the (tap, channel) order of the implicit-GEMM kernels: bit-identical results.
// -------------------------------------------------------------------------------------------
constexpr int HALO_ROWB = 144 * 64;                  // bytes per ring row
constexpr int HALO_RING = 6 * HALO_ROWB;
constexpr int HALO_LDS = HALO_RING + 256 * 64;       // + the step's output tile (residual in, result out: in place)
__global__ __launch_bounds__(512, 2) void conv3x3_c32_rows_kernel(ConvArgs p, int band) {
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
            const int chunk = (lane & 3) ^ ((hx >> 1) & 3);
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
            const int chunk = (lane & 3) ^ ((px >> 1) & 3);
            const uint32_t voff = (uint32_t)((b * p.H + y0) * W_ + px) * 64u + (uint32_t)chunk * 16u;
            dma16_buf(rres, voff, 0u, lds_base + (uint32_t)(HALO_RING + (wave * 2 + k) * 1024));
        }
    };
    load_rows(ys - 1, 4);
    const bool has_res = p.res != nullptr;               // wave-uniform
    if (has_res) load_res(ys);
    // weights: lane (r, h) holds output channel r, channels 8h..8h+7 of each 16-channel half of each tap
    bf16x8 bfr[9][2];
    float4 bias4[4];
    {
        const uint16_t* wrow = p.w + (size_t)r * 288 + 8 * h;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                bfr[tap][ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(wrow + tap * 32 + ks * 16));
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) bias4[qq] = *reinterpret_cast<const float4*>(p.bias + 8 * qq + 4 * h);
    }
    const int q = wave * 32 + r;                        // output pixel of the step
    const int oy = wave >> 2, ox = q & (W_ - 1);        // waves 0-3: first output row, 4-7: second
    const int swq = (q >> 1) & 3;
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
                const int sw = (hx >> 1) & 3;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 af = __builtin_bit_cast(
                        bf16x8, *reinterpret_cast<const uint4*>(lds + slot * HALO_ROWB + hx * 64 + (((2 * ks + h) ^ sw) << 4)));
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ky * 3 + kx][ks], af, acc, 0, 0, 0);
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
            float v0 = acc[4 * qq] + bias4[qq].x, v1 = acc[4 * qq + 1] + bias4[qq].y, v2 = acc[4 * qq + 2] + bias4[qq].z,
                  v3 = acc[4 * qq + 3] + bias4[qq].w;
            if (p.act == 1) { v0 = silu_fast(v0); v1 = silu_fast(v1); v2 = silu_fast(v2); v3 = silu_fast(v3); }
            else if (p.act) { v0 = act_other(p.act, v0); v1 = act_other(p.act, v1); v2 = act_other(p.act, v2); v3 = act_other(p.act, v3); }
            if (has_res) {
                const uint2 rr = *reinterpret_cast<const uint2*>(cell);
                v0 += bf2f_((uint16_t)(rr.x & 0xffff)); v1 += bf2f_((uint16_t)(rr.x >> 16));
                v2 += bf2f_((uint16_t)(rr.y & 0xffff)); v3 += bf2f_((uint16_t)(rr.y >> 16));
            }
            uint2 pk;
            pk.x = (uint32_t)f2bf_(v0) | ((uint32_t)f2bf_(v1) << 16);
            pk.y = (uint32_t)f2bf_(v2) | ((uint32_t)f2bf_(v3) << 16);
            *reinterpret_cast<uint2*>(cell) = pk;
        }
        // the wave's 32 pixel rows (2 KiB contiguous in NHWC) as 16-byte pieces
        const size_t m0 = (size_t)(b * p.H + y0) * W_;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int px = wave * 32 + k * 16 + (lane >> 2), cc = lane & 3;
            const uint4 v = *reinterpret_cast<const uint4*>(Cs + px * 64 + cc * 16);
            *reinterpret_cast<uint4*>(out16 + (m0 + px) * 32 + ((cc ^ ((px >> 1) & 3)) << 3)) = v;
        }
        if (more && has_res) load_res(y0 + 2);
    }
}

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
// 128-byte LDS row, chunk slot = chunk ^ (pixel & 7) keeps eight consecutive pixels on 32 different banks. The k loop
// then streams only the weights: 16 instead of 24 KiB per k-step. Same (tap, channel) order: bit-identical.
template <int WGM, int TN, int WPR, bool HALO = false>
__global__ __launch_bounds__(128 * WGM) void fused_mb_kernel(ConvArgs p) {
    constexpr int WGN = 2, NW = WGM * WGN;
    constexpr int TN2 = (WPR / 32 + 1) / 2;                  // 32-channel tiles of the projection per wave
    constexpr int BM = 32 * WGM, BN = 64 * TN;               // BN = Cexp
    constexpr int B_INST = BN / 16;
    constexpr int B_PW = (B_INST + NW - 1) / NW;
    constexpr int BUF = (BM + BN) * ROWB;
    constexpr int NKB = BN / 32;                             // k-blocks of the second GEMM
    constexpr int E_BYTES = NKB * BM * ROWB;                 // E tile: NKB blocks of [128 rows][64 B], swizzled like A tiles
    constexpr int WP_ROWS = WPR, WP_BUF = WP_ROWS * ROWB;    // projection weights of one k-block (rows past Cout2 unused)
    constexpr int W2_PW = (WP_ROWS / 16 + NW - 1) / NW;
    constexpr int STAGE2 = BM * (64 * TN2 * 2 + 16);         // epilogue staging of the output tile
    constexpr int HW_ = 64, HWD = HW_ + 2, HALO_BYTES = 4 * HWD * 128;     // HALO: 264 pixel rows of 128 B = 33 pieces
    constexpr int BBUF = BN * ROWB;                                         // HALO: one k-step of weights
    constexpr int KREG = HALO ? HALO_BYTES + 2 * BBUF : 2 * BUF;
    constexpr int REG_A0 = KREG > E_BYTES ? KREG : E_BYTES;
    constexpr int REG_A = REG_A0 > STAGE2 ? REG_A0 : STAGE2;
    constexpr int WP_OFF = REG_A, BIAS1_OFF = WP_OFF + 2 * WP_BUF, BIAS2_OFF = BIAS1_OFF + BN * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds[BIAS2_OFF + 128 * 4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * BM;

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
            const uint32_t chunk = (uint32_t)((lane & 7) ^ (hp & 7));
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
    auto compute = [&](auto bufc) {
        constexpr int buf = decltype(bufc)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af;
            if constexpr (HALO) {
                const int hp = hp0 + h_off;
                af = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + hp * 128 + (((4 * buf + 2 * ks + h) ^ (hp & 7)) << 4)));
            } else {
                af = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + a_sw[ks] + buf * BUF));
            }
            bf16x8 bfr[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bfr[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + b_sw[ks] + (buf * B_STRIDE + j * 2048)));
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af, acc[j], 0, 0, 0);
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

    // ---- E tile: bias + SiLU, one bf16 rounding (what the two-launch path stores), into the A layout of GEMM 2
    const unsigned char* w2_base = reinterpret_cast<const unsigned char*>(p.w2);
    auto dma_w2 = [&](auto bufc) {                           // next 32 expanded channels of the projection weights
        constexpr int buf = decltype(bufc)::value;
#pragma unroll
        for (int s = 0; s < W2_PW; ++s)
            if (wave + NW * s < WP_ROWS / 16) dma16_s(w2_base, w2_voff[s], lds0 + WP_OFF + buf * WP_BUF + NW * s * 1024);
        w2_base += CK * 2;
    };
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
                pk.x = (uint32_t)f2bf_(v0) | ((uint32_t)f2bf_(v1) << 16);
                pk.y = (uint32_t)f2bf_(v2) | ((uint32_t)f2bf_(v3) << 16);
                *reinterpret_cast<uint2*>(lds + (wn * TN + j) * (BM * ROWB) + swz(ml, q) + 8 * h) = pk;
            }
    }
    publish();

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
                    acc2[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw, af, acc2[0][j], 0, 0, 0);
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

    ConvArgs p2 = p;                                         // epilogue of the projection: bias2, no activation, residual
    p2.Cout = p.Cout2;
    p2.act = 0;
    if (ISB_EPI_SHARED) conv_epilogue<1, TN2, WGM, 2, true>(p2, acc2, lds, m0, 0, wm, wn, r, h, tid, BIAS2_OFF);
    else conv_epilogue_wl<1, TN2, WGM, 2>(p2, acc2, lds, m0, 0, wm, wn, r, h, lane, wave, BIAS2_OFF);
}

static int launch_conv_igemm_impl(const ConvArgs& a, hipStream_t st);

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

static bool fused_mb_halo() {       // ISB_FMB_HALO=0: im2col A operand (A/B switch)
    static const bool on = [] { const char* e = getenv("ISB_FMB_HALO"); return !e || atoi(e) != 0; }();
    return on;
}

int launch_fused_mb(const ConvArgs& a, hipStream_t st) {
    const bool same1 = a.stride == 1 && a.pad == 1, same2 = a.stride == 2 && a.pad == 0;
    if (a.gate || a.KH != 3 || a.KW != 3 || !(same1 || same2) || a.Cin % 32 != 0 || a.K != 9 * a.Cin || !a.w2 || !a.bias2 ||
        a.Cout2 < 4 || a.Cout2 > 128 || a.Cout2 % 4 != 0 || a.out_f32 ||
        (a.Cout != 128 && a.Cout != 256 && a.Cout != 384) ||
        (size_t)a.B * a.H * a.W * a.Cin * 2 + (size_t)(a.W + 1) * a.Cin * 2 >= 0x7ffffff0ull) {
        set_error("fused_mb: needs an un-gated 3x3 expand to 128/256/384 channels and a projection to <= 128 channels");
        return ISB_ERR_INVALID;
    }
    ConvArgs aa = a;
    aa.grid_mode = 0;
    const int wpr = a.Cout2 <= 64 ? 64 : a.Cout2 <= 96 ? 96 : 128;
#define ISB_FMB(WGM, TN, WPR) hipLaunchKernelGGL((fused_mb_kernel<WGM, TN, WPR>), dim3(cdiv(a.M, 32 * WGM)), dim3(128 * WGM), 0, st, aa)
    if (a.Cout == 128) {
        if (wpr == 64) ISB_FMB(4, 2, 64); else if (wpr == 96) ISB_FMB(4, 2, 96); else ISB_FMB(4, 2, 128);
    } else if (a.Cout == 256) {
        if (wpr == 64 && same1 && a.Cin == 64 && a.W == 64 && a.H % 2 == 0 && fused_mb_halo())
            hipLaunchKernelGGL((fused_mb_kernel<4, 4, 64, true>), dim3(cdiv(a.M, 128)), dim3(512), 0, st, aa);
        else if (wpr == 64) ISB_FMB(4, 4, 64); else if (wpr == 96) ISB_FMB(4, 4, 96); else ISB_FMB(4, 4, 128);
    } else {                                                 // 384 expanded channels: 64-pixel tiles, two workgroups per CU
        if (wpr == 64) ISB_FMB(2, 6, 64); else if (wpr == 96) ISB_FMB(2, 6, 96); else ISB_FMB(2, 6, 128);
    }
#undef ISB_FMB
    ISB_LAUNCHED("fused_mb", st);
    return ISB_OK;
}

// split-K reduction: out[m][n] = bf16( sum_s part[s][m][n] (in split order) + bias[n] (+ res[m][n]) )
__global__ __launch_bounds__(256) void splitk_reduce_kernel(ConvArgs p) {
    const size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x;       // one thread = 4 consecutive channels
    const size_t total4 = (size_t)p.M * p.Cout / 4;
    if (i4 >= total4) return;
    const size_t e = i4 * 4;
    const int n = (int)(e % p.Cout);
    float4 v = *reinterpret_cast<const float4*>(p.part + e);
    for (int s = 1; s < p.splits; ++s) {
        const float4 u = *reinterpret_cast<const float4*>(p.part + (size_t)s * p.M * p.Cout + e);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    const float4 bs = *reinterpret_cast<const float4*>(p.bias + n);
    v.x += bs.x; v.y += bs.y; v.z += bs.z; v.w += bs.w;
    if (p.act) { v.x = silu_fast(v.x); v.y = silu_fast(v.y); v.z = silu_fast(v.z); v.w = silu_fast(v.w); }
    if (p.res) {
        const uint2 rr = *reinterpret_cast<const uint2*>(p.res + e);
        if (p.f16) { v.x += T16<true>::lo(rr.x); v.y += T16<true>::hi(rr.x); v.z += T16<true>::lo(rr.y); v.w += T16<true>::hi(rr.y); }
        else { v.x += T16<false>::lo(rr.x); v.y += T16<false>::hi(rr.x); v.z += T16<false>::lo(rr.y); v.w += T16<false>::hi(rr.y); }
    }
    uint2 pk;
    if (p.f16) { pk.x = T16<true>::pack2(v.x, v.y); pk.y = T16<true>::pack2(v.z, v.w); }
    else {
        pk.x = (uint32_t)f2bf_(v.x) | ((uint32_t)f2bf_(v.y) << 16);
        pk.y = (uint32_t)f2bf_(v.z) | ((uint32_t)f2bf_(v.w) << 16);
    }
    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.out) + e) = pk;
}

int launch_splitk_reduce(const ConvArgs& a, hipStream_t st) {
    if (a.splits < 2 || !a.part || a.Cout % 4 != 0 || a.out_f32) {
        set_error("splitk_reduce: needs splits >= 2, a partial buffer and bf16 output");
        return ISB_ERR_INVALID;
    }
    const size_t total4 = (size_t)a.M * a.Cout / 4;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdivz(total4, 256)), dim3(256), 0, st, a);
    ISB_LAUNCHED("splitk_reduce", st);
    return ISB_OK;
}

static int conv_grid_mode() {
    static const int mode = [] {
        const char* e = getenv("ISB_CONV_GRID");     // tuning override; 1 measured 4 % faster than the 2-D grid
        return e ? atoi(e) : 1;
    }();
    return mode;
}

static dim3 conv_grid(ConvArgs& a, int BM, int BN) {
    a.grid_m = cdiv(a.M, BM);
    a.grid_n = cdiv(a.Cout, BN);
    const unsigned z = a.splits > 1 ? (unsigned)a.splits : 1u;      // split-K (gemm1x1 kernels only)
    if (a.grid_mode == 0 || a.grid_n == 1) {
        a.grid_mode = 0;
        return dim3(a.grid_m, a.grid_n, z);
    }
    return dim3(8 * ((a.grid_m + 7) / 8) * a.grid_n, 1, z);
}

extern "C" int isb_wsreg_verified(void);      // wsreg_guard.cpp: 1 only if the build checked the staging registers in the disassembly
static int wsreg_on() {             // ISB_WSREG: weights-stationary kernel for the short-K expand convolutions. Default 184 (K <= 224; K = 384: 186);
                                    // 0 = tile kernels only, 181 / 182 / 183 = the earlier forms (A/B switch; 181 / 182 also take K = 384)
    // fail closed: without the build's verdict the default is 0 (tile kernels); an explicit ISB_WSREG still selects a form
    static const int on = [] { const char* e = getenv("ISB_WSREG"); const int v = e ? atoi(e) : (isb_wsreg_verified() ? 184 : 0); return v == 1 ? 184 : v; }();
    return on;
}

static bool conv3_halo() {          // ISB_C3_HALO=0: im2col A operand (A/B switch)
    static const bool on = [] { const char* e = getenv("ISB_C3_HALO"); return !e || atoi(e) != 0; }();
    return on;
}

static int launch_conv_igemm_impl(const ConvArgs& a, hipStream_t st) {
    ConvArgs aa = a;
    aa.grid_mode = conv_grid_mode();
    if (a.Cin % 32 != 0 || a.Cout % 32 != 0 || a.K != a.KH * a.KW * a.Cin || a.M <= 0) {
        set_error("conv_igemm: unsupported shape Cin=%d Cout=%d K=%d M=%d", a.Cin, a.Cout, a.K, a.M);
        return ISB_ERR_INVALID;
    }
    if (a.gate && (a.KH != 1 || a.stride != 1)) {
        set_error("conv_igemm: SE gate only on 1x1 convolutions");
        return ISB_ERR_INVALID;
    }
    // tile variants: 0 = pick by Cout
    int v = a.variant;
    if (v == 0) {
        // chosen by measurement on MI355X (tools/conv_sweep.py, tools/conv_probe*.py, profiles/README.md):
        //   * 1x1 stride-1 convolutions are plain GEMMs and run on the lean gemm1x1 kernels (131-139 without,
        //     141-148 with an SE gate), 8 waves of 32 x 64..96 sub-tiles: high occupancy beats big tiles here;
        //   * 3x3 convolutions run on conv3x3_dma_kernel (161-166: raw-buffer A operand, hardware zero padding);
        //     the general LDS-DMA kernels remain for shapes outside its contract;
        //   * a single frame (M <= 2048) needs many small workgroups: 64-row tiles.
        const bool g1 = a.KH == 1 && a.stride == 1 && a.pad == 0 && a.zeros;
        const int ohw = a.OH * a.OW;
        // (stride 2: TF-SAME on an even input = pad 0, bottom / right overhang; or PyTorch's symmetric pad 1 -- the lean 3x3 kernel
        // addresses its taps from a shifted buffer base and a per-lane validity mask, whatever the padding)
        static const bool c3_s2p1 = [] { const char* e = getenv("ISB_C3_S2P1"); return !e || atoi(e) != 0; }();   // A/B switch
        const bool c3 = !a.gate && a.KH == 3 && a.KW == 3 && ((a.stride == 1 && a.pad == 1) || (a.stride == 2 && (a.pad == 0 || (a.pad == 1 && c3_s2p1)))) &&
                        (size_t)a.B * a.H * a.W * a.Cin * 2 + (size_t)(a.W + 1) * a.Cin * 2 < 0x7ffffff0ull;
        if (a.M <= 2048 && a.Cout >= 64) {
            if (g1 && !a.gate) v = 138;
            else if (g1 && a.gate && ohw % 64 == 0) v = 147;
            else v = (!a.gate && a.zeros) ? 64 : 75;
        } else if (a.M <= 8192 && a.Cout >= 128 && a.Cout % 128 == 0 && !a.gate && !a.f16 && (c3 || g1) &&
                   (long)cdiv(a.M, 128) * cdiv(a.Cout, 128) < 384) {
            // a few thousand rows (the detector's 8 x 8 / 16 x 16 maps at 64 frames, the ResNet-50's 7 x 7 and 14 x 14 maps): 128-row
            // tiles would leave most CUs idle -- 64 x 128 tiles of 4 waves double the workgroups
            v = c3 ? 168 : 150;
        } else if (g1 && !a.gate && !a.res && !a.out_f32 && a.act <= 1 && a.splits <= 1 && wsreg_on() && (!a.f16 || (a.Cin == 384 && wsreg_on() >= 184)) &&
                   (a.Cin == 96 || a.Cin == 192 || a.Cin == 224 || (a.Cin == 384 && wsreg_on() != 183 && wsreg_on() != 1840)) && a.Cout % 32 == 0 && a.M >= 64 * 128 &&
                   (size_t)a.M * a.Cin * 2 < 0xffffffffull) {
            v = wsreg_on();                                                      // weights-stationary persistent GEMM (short-K expand convolutions)
            if (v >= 184) v = a.Cin == 384 ? ((a.M >= 16384 || !a.act) ? 186 : 185) : 184;   // (1840: K <= 224 only, A/B switch); K = 384: eight waves pay from 256 tiles on                          // two waves per SIMD: 2 workgroups x 4 waves, or (K = 384) 1 x 8
        } else if (g1 && !a.gate) {
            if (a.Cout % 192 == 0) v = 131;            // 128 x 192
            else if (a.Cout == 64) v = 135;            // 256 x  64
            else if (a.Cout == 224) v = 137;           // 128 x 224
            else if (a.Cout == 32) v = 59;
            else v = 132;                              // 128 x 128
        } else if (g1 && a.gate && (ohw % 128 == 0 || 128 % ohw == 0)) {
            // the 8 x 8 stages: 128-row tiles leave at most one workgroup per CU (a half-batch lane: half the CUs);
            // 64 x 192 tiles of 4 waves measured +14 % (384 outputs) and +41 % (640 outputs) at 8192 rows
            const long wgs128 = (long)cdiv(a.M, 128) * cdiv(a.Cout, a.Cout % 320 == 0 ? 320 : 192);
            static const bool lw_on = [] { const char* e = getenv("ISB_LW"); return !e || atoi(e) != 0; }();   // A/B switch
            if (lw_on && ohw == 64 && a.M >= 4096 && a.Cin % 64 == 0 && a.Cin >= 256 && a.splits <= 1 && !a.out_f32 &&
                (a.Cout % 192 == 0 || a.Cout % 320 == 0) && (size_t)a.M * a.Cin * 2 < 0xffffffffull &&
                2 * a.Cin * 4 + (a.Cout % 192 == 0 || a.M < 16384 ? 6 * 320 : 4 * 448) * 64 + 2048 <= 160 * 1024) {
                // the 8 x 8 stages: 128-row tiles with loader waves, one workgroup per CU (gemm1x1_lw_kernel; bit-identical).
                // 2304 -> 384: 53.7 vs 60.3 us at 256 frames, 39.6 vs 46.7 at 128; 3840 -> 640: 113.7 vs 123.0 / 76.0 vs 84.1
                v = (a.Cout % 192 == 0 || a.M < 16384) ? 155 : 156;
            } else
            if (ohw % 64 == 0 && a.Cout % 64 == 0 && ((a.Cout % 192 == 0 && wgs128 <= 256) || (a.Cout % 320 == 0 && wgs128 < 256))) v = 146;
            else if (a.Cout % 320 == 0) v = 144;       // 128 x 320
            else if (a.Cout == 224) v = 143;           // 128 x 224
            else if (a.Cout % 192 == 0) v = 141;       // 128 x 192
            else v = 142;                              // 128 x 128
        } else if (c3) {
            if (a.Cout == 32 && a.Cin == 32 && a.stride == 1 && a.W == 128 && a.H % 2 == 0 && !a.out_f32 && a.act <= 1)
                v = 171;                              // rows ring in LDS, +50 % over the implicit GEMM (bit-identical)
            else if (a.Cout == 32) v = 163;           // 256 x  32   (lean 3x3, buffer-addressed A operand)
            else if (a.Cout % 192 == 0 && a.stride == 1 && a.Cin == 96 && a.W == 32 && a.H % 4 == 0 && conv3_halo())
                v = 167;                              // 128 x 192, halo-tile A operand
            else if (a.Cout % 192 == 0) v = 161;      // 128 x 192
            else {
                // the detector's / the ResNet trunk's plain 3x3 layers (same k order in every tile shape: bit-identical). At 256 frames:
                // 64 outputs on 256 x 64 tiles instead of half-empty 128 x 128 ones (32 -> 64 @128: 417 vs 584 us, 64 -> 64 @64: 157 vs
                // 227); widening layers from 512 outputs on 256 x 128 (256 -> 512 @16: 193 vs 227, 512 -> 1024 @8: 174 vs 218)
                static const bool sel = [] { const char* e = getenv("ISB_C3_SEL"); return !e || atoi(e) != 0; }();   // A/B switch
                if (sel && a.Cout == 64 && a.M >= 32768) v = 165;                            // 256 x  64
                else if (sel && a.Cout >= 512 && a.Cout % 128 == 0 && a.Cout > a.Cin && a.M >= 16384) v = 164;   // 256 x 128
                else v = 162;                         // 128 x 128
            }
        } else if (!a.gate && a.zeros) {
            if (a.Cout == 32) v = 59;                 // 256 x  32, 8 waves
            else if (a.Cout % 192 == 0) v = 54;       // 128 x 192, 8 waves of 32 x 96
            else if (a.Cout % 128 == 0) v = 55;       // 128 x 128, 8 waves of 32 x 64
            else if (a.Cout == 64) v = 57;            // 256 x  64
            else if (a.Cout == 224) v = 14;
            else v = 55;
        } else {
            if (a.Cout % 320 == 0) v = 43;            // 128 x 320, 8 waves
            else if (a.Cout % 192 == 0) v = 44;       // 128 x 192, 8 waves
            else if (a.Cout == 224) v = 4;            // 128 x 224
            else if (a.Cout % 128 == 0) v = 1;
            else if (a.Cout % 96 == 0) v = 2;
            else if (a.Cout % 64 == 0) v = 3;
            else v = 5;
        }
    }
    const bool is_g1 = (v >= 131 && v <= 153) || (v >= 191 && v <= 197);
    if (a.act_after_res && (a.act < 2 || a.out_f32 || aa.splits > 1 || v == 171 || (v >= 181 && v <= 188) || !ISB_EPI_SHARED)) {
        set_error("conv_igemm: act_after_res takes act 2-4 on the kernels with the shared bf16 epilogue (variant %d)", v);
        return ISB_ERR_INVALID;
    }
    if (a.out_ld && (a.out_ld < a.Cout || a.out_ld % 8 != 0 || a.out_f32 || aa.splits > 1 || v == 171 || v == 149 || (v >= 181 && v <= 188))) {
        set_error("conv_igemm: out_ld (a channel slice of a wider 16-bit tensor) needs out_ld >= Cout, a multiple of 8, and a kernel with the shared epilogue (variant %d)", v);
        return ISB_ERR_INVALID;
    }
    if (a.f16 && !(v == 131 || v == 132 || v == 138 || v == 141 || v == 144 || v == 146 || v == 147 || v == 149 || v == 155 || v == 156 || v == 185 || v == 186)) {
        set_error("conv_igemm: fp16 operands are implemented by variants 131 / 132 / 138, 141 / 144 / 146 / 147 / 149 and 185 / 186 (got %d)", v);
        return ISB_ERR_INVALID;
    }
    if (aa.splits > 1 && !(is_g1 && aa.part)) {
        set_error("conv_igemm: split-K is implemented by the gemm1x1 variants (131-149) and needs a partial buffer");
        return ISB_ERR_INVALID;
    }
    const bool is_dma = (v >= 11 && v <= 39) || (v >= 51 && v <= 69);
    const bool is_gdma = (v >= 81 && v <= 99) || (v >= 111 && v <= 119);
    const bool is_dma64 = v >= 101 && v <= 109;
    if ((is_dma64 || (v >= 111 && v <= 119)) && a.Cin % 64 != 0) {
        set_error("conv_igemm: the 64-wide k-tile variants need Cin %% 64 == 0 (Cin=%d)", a.Cin);
        return ISB_ERR_INVALID;
    }
    if (is_dma64 && (a.gate || !a.zeros)) {
        set_error("conv_igemm: variants 101-109 take no SE gate and need the zero line");
        return ISB_ERR_INVALID;
    }
    if (is_dma && (a.gate || !a.zeros)) {
        set_error("conv_igemm: the plain LDS-DMA variants take no SE gate and need the zero line");
        return ISB_ERR_INVALID;
    }
    if (is_gdma && (!a.gate || !a.zeros)) {
        set_error("conv_igemm: variants 81-99 and 111-119 are the gated LDS-DMA kernels");
        return ISB_ERR_INVALID;
    }
    // gated LDS-DMA kernels: ring of NB tile buffers + dump KiB + the f32 gate rows of the samples a tile covers
#define ISB_CONV_LAUNCH_GATE(TM, TN, WGM, WGN, NB, KT)                                                                \
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
        auto kern = conv_igemm_dma_kernel<TM, TN, WGM, WGN, NB, false, true, KT>;                                   \
        static int attr_bytes = 0;                                                                                  \
        if (bytes > attr_bytes) {                                                                                   \
            ISB_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));     \
            attr_bytes = bytes;                                                                                     \
        }                                                                                                           \
        const dim3 g = conv_grid(aa, BM_, BN_);                                                                     \
        hipLaunchKernelGGL(kern, g, dim3(64 * WGM * WGN), bytes, st, aa);                                           \
    } while (0)
#define ISB_CONV_LAUNCH(TM, TN, WGM, WGN)                                                              \
    do {                                                                                               \
        const dim3 g = conv_grid(aa, 32 * TM * WGM, 32 * TN * WGN);                                 \
        hipLaunchKernelGGL((conv_igemm_kernel<TM, TN, WGM, WGN>), g, dim3(64 * WGM * WGN), 0, st, aa); \
    } while (0)
    switch (v) {
        case 1: ISB_CONV_LAUNCH(2, 2, 2, 2); break;      // 128 x 128, 4 waves
        case 2: ISB_CONV_LAUNCH(1, 3, 4, 1); break;      // 128 x  96
        case 3: ISB_CONV_LAUNCH(2, 1, 2, 2); break;      // 128 x  64
        case 4: ISB_CONV_LAUNCH(1, 7, 4, 1); break;      // 128 x 224
        case 5: ISB_CONV_LAUNCH(2, 1, 4, 1); break;      // 256 x  32
        case 6: ISB_CONV_LAUNCH(2, 2, 4, 2); break;      // 256 x 128, 8 waves
        case 7: ISB_CONV_LAUNCH(4, 2, 2, 4); break;      // 256 x 256, 8 waves
        case 8: ISB_CONV_LAUNCH(2, 3, 4, 2); break;      // 256 x 192, 8 waves
        case 9: ISB_CONV_LAUNCH(2, 1, 4, 2); break;      // 256 x  64, 8 waves
        case 41: ISB_CONV_LAUNCH(1, 6, 4, 1); break;     // 128 x 192: full-width tiles read the A operand once
        case 42: ISB_CONV_LAUNCH(1, 6, 4, 2); break;     // 128 x 384, 8 waves
        case 43: ISB_CONV_LAUNCH(1, 5, 4, 2); break;     // 128 x 320, 8 waves
        case 44: ISB_CONV_LAUNCH(1, 3, 4, 2); break;     // 128 x 192, 8 waves
        case 45: ISB_CONV_LAUNCH(1, 2, 4, 2); break;     // 128 x 128, 8 waves of 32 x 64
        case 47: ISB_CONV_LAUNCH(1, 4, 4, 2); break;     // 128 x 256, 8 waves
        case 48: ISB_CONV_LAUNCH(1, 7, 4, 2); break;     // 128 x 448, 8 waves
        case 71: ISB_CONV_LAUNCH(1, 3, 8, 2); break;     // 256 x 192, 16 waves
        case 75: ISB_CONV_LAUNCH(1, 1, 2, 2); break;     //  64 x  64 (small M)
        case 76: ISB_CONV_LAUNCH(1, 2, 2, 2); break;     //  64 x 128 (small M)
        case 72: ISB_CONV_LAUNCH(1, 3, 4, 4); break;     // 128 x 384, 16 waves
        case 73: ISB_CONV_LAUNCH(1, 5, 4, 4); break;     // 128 x 640, 16 waves
        case 74: ISB_CONV_LAUNCH(1, 2, 8, 2); break;     // 256 x 128, 16 waves
#define ISB_CONV_LAUNCH_DMA(TM, TN, WGM, WGN)                                                              \
    do {                                                                                                   \
        const dim3 g = conv_grid(aa, 32 * TM * WGM, 32 * TN * WGN);                                     \
        hipLaunchKernelGGL((conv_igemm_dma_kernel<TM, TN, WGM, WGN, 2>), g, dim3(64 * WGM * WGN), 0, st, aa); \
    } while (0)
        case 11: ISB_CONV_LAUNCH_DMA(2, 2, 2, 2); break;
        case 12: ISB_CONV_LAUNCH_DMA(1, 3, 4, 1); break;
        case 13: ISB_CONV_LAUNCH_DMA(2, 1, 2, 2); break;
        case 14: ISB_CONV_LAUNCH_DMA(1, 7, 4, 1); break;
        case 15: ISB_CONV_LAUNCH_DMA(2, 1, 4, 1); break;
        case 16: ISB_CONV_LAUNCH_DMA(2, 2, 4, 2); break;
        case 17: ISB_CONV_LAUNCH_DMA(4, 2, 2, 4); break;
        case 18: ISB_CONV_LAUNCH_DMA(2, 3, 4, 2); break;
        case 19: ISB_CONV_LAUNCH_DMA(2, 1, 4, 2); break;
        case 51: ISB_CONV_LAUNCH_DMA(1, 6, 4, 1); break;
        case 52: ISB_CONV_LAUNCH_DMA(1, 6, 4, 2); break;
        case 53: ISB_CONV_LAUNCH_DMA(1, 5, 4, 2); break;
        case 54: ISB_CONV_LAUNCH_DMA(1, 3, 4, 2); break;
        case 55: ISB_CONV_LAUNCH_DMA(1, 2, 4, 2); break;
        case 56: ISB_CONV_LAUNCH_DMA(1, 4, 4, 2); break;
        case 57: ISB_CONV_LAUNCH_DMA(1, 2, 8, 1); break;    // 256 x 64
        case 58: ISB_CONV_LAUNCH_DMA(1, 3, 8, 1); break;    // 256 x 96
        case 59: ISB_CONV_LAUNCH_DMA(1, 1, 8, 1); break;    // 256 x 32
        case 60: ISB_CONV_LAUNCH_DMA(1, 2, 8, 2); break;    // 256 x 128, 16 waves of 32 x 64
        case 64: ISB_CONV_LAUNCH_DMA(1, 1, 2, 2); break;    //  64 x  64: small-M launches (single frames) need many workgroups
        case 65: ISB_CONV_LAUNCH_DMA(1, 2, 2, 2); break;    //  64 x 128
        case 61: ISB_CONV_LAUNCH_DMA(1, 3, 8, 2); break;    // 256 x 192, 16 waves of 32 x 96
        case 62: ISB_CONV_LAUNCH_DMA(1, 2, 4, 4); break;    // 128 x 256, 16 waves
        case 63: ISB_CONV_LAUNCH_DMA(1, 3, 4, 4); break;    // 128 x 384, 16 waves
#undef ISB_CONV_LAUNCH_DMA
#define ISB_CONV_LAUNCH_RING(TM, TN, WGM, WGN)                                                                \
    do {                                                                                                      \
        const dim3 g = conv_grid(aa, 32 * TM * WGM, 32 * TN * WGN);                                        \
        hipLaunchKernelGGL((conv_igemm_dma_kernel<TM, TN, WGM, WGN, 4>), g, dim3(64 * WGM * WGN), 0, st, aa); \
    } while (0)
#define ISB_CONV_LAUNCH_3B(TM, TN, WGM, WGN)                                                                  \
    do {                                                                                                      \
        const dim3 g = conv_grid(aa, 32 * TM * WGM, 32 * TN * WGN);                                        \
        hipLaunchKernelGGL((conv_igemm_dma_kernel<TM, TN, WGM, WGN, 3>), g, dim3(64 * WGM * WGN), 0, st, aa); \
    } while (0)
        case 31: ISB_CONV_LAUNCH_3B(2, 2, 2, 2); break;     // 128 x 128, 4 waves, 3 buffers / 2 tiles in flight
        case 33: ISB_CONV_LAUNCH_3B(2, 1, 2, 2); break;     // 128 x  64
        case 36: ISB_CONV_LAUNCH_3B(2, 2, 4, 2); break;     // 256 x 128, 8 waves
        case 37: ISB_CONV_LAUNCH_3B(4, 2, 2, 4); break;     // 256 x 256, 8 waves
#undef ISB_CONV_LAUNCH_3B
        case 21: ISB_CONV_LAUNCH_RING(2, 2, 2, 2); break;   // 128 x 128, 4 waves, 4-buffer ring
        case 23: ISB_CONV_LAUNCH_RING(2, 1, 2, 2); break;   // 128 x  64
        case 26: ISB_CONV_LAUNCH_RING(2, 2, 4, 2); break;   // 256 x 128, 8 waves
        case 27: ISB_CONV_LAUNCH_RING(4, 2, 2, 4); break;   // 256 x 256, 8 waves
        case 28: ISB_CONV_LAUNCH_RING(4, 1, 2, 4); break;   // 256 x 128 as 2x4 waves of 128x32
#undef ISB_CONV_LAUNCH_RING
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
        hipLaunchKernelGGL((conv_igemm_dma_kernel<TM, TN, WGM, WGN, 2, false, false, 64>), g, dim3(64 * WGM * WGN), 0, st, aa); \
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
#define ISB_CONV_LAUNCH_G1(TM, TN, WGM, WGN)                                                                     \
    do {                                                                                                         \
        if (a.gate || a.KH != 1 || a.stride != 1 || a.pad != 0) {                                                \
            set_error("conv_igemm: variants 131-139 are un-gated 1x1 stride-1 GEMMs");                           \
            return ISB_ERR_INVALID;                                                                              \
        }                                                                                                        \
        const dim3 g = conv_grid(aa, 32 * TM * WGM, 32 * TN * WGN);                                              \
        if (a.probe & 2) hipLaunchKernelGGL((gemm1x1_dma_kernel<TM, TN, WGM, WGN, 0, true>), g, dim3(64 * WGM * WGN), 0, st, aa); \
        else hipLaunchKernelGGL((gemm1x1_dma_kernel<TM, TN, WGM, WGN>), g, dim3(64 * WGM * WGN), 0, st, aa);     \
    } while (0)
    // the variants the fp16 stages select exist in both operand types
#define ISB_CONV_LAUNCH_G1H(TM, TN, WGM, WGN)                                                                    \
    do {                                                                                                         \
        if (!a.f16) { ISB_CONV_LAUNCH_G1(TM, TN, WGM, WGN); break; }                                             \
        if (a.gate || a.KH != 1 || a.stride != 1 || a.pad != 0 || (a.probe & 2)) {                               \
            set_error("conv_igemm: variants 131-139 are un-gated 1x1 stride-1 GEMMs");                           \
            return ISB_ERR_INVALID;                                                                              \
        }                                                                                                        \
        const dim3 g = conv_grid(aa, 32 * TM * WGM, 32 * TN * WGN);                                              \
        hipLaunchKernelGGL((gemm1x1_dma_kernel<TM, TN, WGM, WGN, 0, false, 2, true>), g, dim3(64 * WGM * WGN), 0, st, aa); \
    } while (0)
        case 181: case 182: case 183: case 184: case 185: case 186: case 187: case 188: {    // conv_ws.hip
            const int rc = launch_conv_ws(a, aa, v, st);
            if (rc != ISB_OK) return rc;
            break;
        }
        case 131: ISB_CONV_LAUNCH_G1H(1, 3, 4, 2); break;  // 128 x 192
        case 132: ISB_CONV_LAUNCH_G1H(1, 2, 4, 2); break;  // 128 x 128
        case 133: ISB_CONV_LAUNCH_G1(2, 3, 4, 2); break;   // 256 x 192 (8 waves of 64 x 96)
        case 134: ISB_CONV_LAUNCH_G1(2, 2, 4, 2); break;   // 256 x 128
        case 135: ISB_CONV_LAUNCH_G1(1, 2, 8, 1); break;   // 256 x  64
        case 136: ISB_CONV_LAUNCH_G1(1, 3, 8, 1); break;   // 256 x  96
        case 137: ISB_CONV_LAUNCH_G1(1, 7, 4, 1); break;   // 128 x 224
        case 138: ISB_CONV_LAUNCH_G1H(1, 1, 2, 2); break;  //  64 x  64
        case 139: ISB_CONV_LAUNCH_G1(2, 4, 4, 2); break;   // 256 x 256 (8 waves of 64 x 128)
        case 140: ISB_CONV_LAUNCH_G1(1, 3, 2, 2); break;   //  64 x 192, 4 waves (four workgroups per CU: more independent phases)
        case 150: ISB_CONV_LAUNCH_G1(1, 2, 2, 2); break;   //  64 x 128, 4 waves
#undef ISB_CONV_LAUNCH_G1H
#undef ISB_CONV_LAUNCH_G1
#define ISB_CONV_LAUNCH_C3(TM, TN, WGM, WGN)                                                                     \
    do {                                                                                                         \
        const bool same1 = a.stride == 1 && a.pad == 1, same2 = a.stride == 2 && (a.pad == 0 || a.pad == 1);    \
        if (a.gate || a.KH != 3 || a.KW != 3 || !(same1 || same2) ||                                             \
            (size_t)a.B * a.H * a.W * a.Cin * 2 + (size_t)(a.W + 1) * a.Cin * 2 >= 0x7ffffff0ull) {             \
            set_error("conv_igemm: variants 161-169 are un-gated 3x3 convolutions on tensors below 2 GiB");      \
            return ISB_ERR_INVALID;                                                                              \
        }                                                                                                        \
        const dim3 g = conv_grid(aa, 32 * TM * WGM, 32 * TN * WGN);                                              \
        hipLaunchKernelGGL((conv3x3_dma_kernel<TM, TN, WGM, WGN>), g, dim3(64 * WGM * WGN), 0, st, aa);          \
    } while (0)
        case 171: {                                          // 3x3 32 -> 32 on 128-wide images: rows ring in LDS
            // rows per workgroup: long bands reuse the ring (each input row is copied once), but the launch should still
            // offer two workgroups to every CU
            int band = 2;
            for (int cand = 32; cand > 2; cand >>= 1)
                if (a.H % cand == 0 && (long)a.B * (a.H / cand) >= 512) { band = cand; break; }
            if (a.gate || a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.Cin != 32 || a.Cout != 32 || a.W != 128 ||
                a.H % 2 != 0 || a.out_f32 || (size_t)a.B * a.H * a.W * 64 >= 0x7ffffff0ull) {
                set_error("conv_igemm: variant 171 is the 3x3 stride-1 32 -> 32 convolution on 128-wide images (< 2 GiB)");
                return ISB_ERR_INVALID;
            }
            static bool attr_set = false;
            if (!attr_set) {
                ISB_HIP(hipFuncSetAttribute((const void*)conv3x3_c32_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, HALO_LDS));
                attr_set = true;
            }
            hipLaunchKernelGGL(conv3x3_c32_rows_kernel, dim3(a.B * (a.H / band)), dim3(512), HALO_LDS, st, aa, band);
            break;
        }
        case 167: {                                          // 128 x 192, A operand from an LDS halo (96 channels, 32-wide maps)
            if (a.gate || a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.Cin != 96 || a.W != 32 || a.H % 4 != 0 ||
                (a.H * a.W) % 128 != 0 || (size_t)a.B * a.H * a.W * a.Cin * 2 + (size_t)(a.W + 1) * a.Cin * 2 >= 0x7ffffff0ull) {
                set_error("conv_igemm: variant 167 is the 3x3 stride-1 convolution of 96 channels on 32-wide maps");
                return ISB_ERR_INVALID;
            }
            const dim3 g = conv_grid(aa, 128, 192);
            hipLaunchKernelGGL((conv3x3_dma_kernel<1, 3, 4, 2, true>), g, dim3(512), 0, st, aa);
            break;
        }
        case 161: ISB_CONV_LAUNCH_C3(1, 3, 4, 2); break;   // 128 x 192
        case 162: ISB_CONV_LAUNCH_C3(1, 2, 4, 2); break;   // 128 x 128
        case 163: ISB_CONV_LAUNCH_C3(1, 1, 8, 1); break;   // 256 x  32
        case 164: ISB_CONV_LAUNCH_C3(2, 2, 4, 2); break;   // 256 x 128
        case 165: ISB_CONV_LAUNCH_C3(1, 2, 8, 1); break;   // 256 x  64
        case 166: ISB_CONV_LAUNCH_C3(2, 3, 4, 2); break;   // 256 x 192
        case 168: ISB_CONV_LAUNCH_C3(1, 2, 2, 2); break;   //  64 x 128, 4 waves: small-M launches (the detector's 8 x 8 / 16 x 16 maps)
        case 169: ISB_CONV_LAUNCH_C3(1, 1, 2, 2); break;   //  64 x  64
#undef ISB_CONV_LAUNCH_C3
#define ISB_CONV_LAUNCH_G1G(TM, TN, WGM, WGN) ISB_CONV_LAUNCH_G1GN(TM, TN, WGM, WGN, 2)
#define ISB_CONV_LAUNCH_G1GN(TM, TN, WGM, WGN, NBUF_) ISB_CONV_LAUNCH_G1GT(TM, TN, WGM, WGN, NBUF_, false)
    // the variants the fp16 stages select exist in both operand types
#define ISB_CONV_LAUNCH_G1GH(TM, TN, WGM, WGN)                                                                   \
    do {                                                                                                         \
        if (a.f16) ISB_CONV_LAUNCH_G1GT(TM, TN, WGM, WGN, 2, true);                                              \
        else ISB_CONV_LAUNCH_G1GT(TM, TN, WGM, WGN, 2, false);                                                   \
    } while (0)
#define ISB_CONV_LAUNCH_G1GT(TM, TN, WGM, WGN, NBUF_, F16_)                                                             \
    do {                                                                                                         \
        constexpr int BM_ = 32 * TM * WGM, BN_ = 32 * TN * WGN;                                                  \
        const int ohw = a.OH * a.OW;                                                                             \
        if (!a.gate || a.KH != 1 || a.stride != 1 || a.pad != 0 || (ohw % BM_ != 0 && BM_ % ohw != 0) ||        \
            (NBUF_ == 3 && (a.Cin % 96 != 0 || a.splits > 1))) {                                                 \
            set_error("conv_igemm: variants 141-149 / 191-197 are gated 1x1 GEMMs on sample-aligned tiles (19x: Cin %% 96 == 0)"); \
            return ISB_ERR_INVALID;                                                                              \
        }                                                                                                        \
        const int ns = BM_ > ohw ? BM_ / ohw : 1;                                                                \
        const int ring = NBUF_ * (BM_ + BN_) * ROWB + ns * a.Cin * 4;                                                \
        const int stage = BM_ * (BN_ * 2 + 16);                                                                  \
        aa.grid_bias_off = ring > stage ? ring : stage;                                                          \
        const int bytes = aa.grid_bias_off + BN_ * 4;                                                            \
        auto kern = gemm1x1_dma_kernel<TM, TN, WGM, WGN, true, false, NBUF_, F16_>;                                            \
        static int attr_bytes = 0;                                                                               \
        if (bytes > attr_bytes) {                                                                                \
            ISB_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));  \
            attr_bytes = bytes;                                                                                  \
        }                                                                                                        \
        const dim3 g = conv_grid(aa, BM_, BN_);                                                                  \
        if ((a.probe & 2) && !F16_) {                                                                            \
            auto kern2 = gemm1x1_dma_kernel<TM, TN, WGM, WGN, true, true, NBUF_>;                                       \
            ISB_HIP(hipFuncSetAttribute((const void*)kern2, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)); \
            hipLaunchKernelGGL(kern2, g, dim3(64 * WGM * WGN), bytes, st, aa);                                   \
        } else                                                                                                   \
        hipLaunchKernelGGL(kern, g, dim3(64 * WGM * WGN), bytes, st, aa);                                        \
    } while (0)
        case 149: {                                          //  64 x 128, split-K, squeeze-excite FC2 folded in
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
            static bool attr_set = false;
            if (!attr_set) {
                ISB_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
                ISB_HIP(hipFuncSetAttribute((const void*)kern_h, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
                attr_set = true;
            }
            const dim3 g = conv_grid(aa, BM_, BN_);
            if (a.f16) hipLaunchKernelGGL(kern_h, g, dim3(256), bytes, st, aa);
            else hipLaunchKernelGGL(kern, g, dim3(256), bytes, st, aa);
            break;
        }
        case 141: ISB_CONV_LAUNCH_G1GH(1, 3, 4, 2); break;  // 128 x 192, SE gate on the A fragments
        case 142: ISB_CONV_LAUNCH_G1G(1, 2, 4, 2); break;   // 128 x 128
        case 143: ISB_CONV_LAUNCH_G1G(1, 7, 4, 1); break;   // 128 x 224
        case 144: ISB_CONV_LAUNCH_G1GH(1, 5, 4, 2); break;  // 128 x 320
        case 145: ISB_CONV_LAUNCH_G1G(2, 2, 4, 2); break;   // 256 x 128
        case 146: ISB_CONV_LAUNCH_G1GH(1, 3, 2, 2); break;  //  64 x 192
        case 147: ISB_CONV_LAUNCH_G1GH(1, 2, 2, 2); break;  //  64 x 128
        case 148: ISB_CONV_LAUNCH_G1G(2, 7, 4, 1); break;   // 256 x 224
        case 152: ISB_CONV_LAUNCH_G1G(1, 7, 8, 1); break;   // 256 x 224, eight waves
        case 191: ISB_CONV_LAUNCH_G1GN(1, 3, 4, 2, 3); break;   // three k-step buffers: 128 x 192
        case 193: ISB_CONV_LAUNCH_G1GN(1, 7, 4, 1, 3); break;   // 128 x 224
        case 194: ISB_CONV_LAUNCH_G1GN(1, 5, 4, 2, 3); break;   // 128 x 320
        case 196: ISB_CONV_LAUNCH_G1GN(1, 3, 2, 2, 3); break;   //  64 x 192
        case 197: ISB_CONV_LAUNCH_G1GN(1, 2, 2, 2, 3); break;   //  64 x 128
        case 153: ISB_CONV_LAUNCH_G1G(1, 6, 8, 1); break;   // 256 x 192, eight waves
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
        static int attr_bytes = 0;                                                                                       \
        if (bytes > attr_bytes) {                                                                                        \
            ISB_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));          \
            attr_bytes = bytes;                                                                                          \
        }                                                                                                                \
        hipLaunchKernelGGL(kern, g, dim3(64 * (WGM_ * WGN_ + 4)), bytes, st, aa);                                        \
    } while (0)
            if (v == 155) { if (a.f16) ISB_LW_GO(4, 2, 3, 3, true); else ISB_LW_GO(4, 2, 3, 3, false); }
            else { if (a.f16) ISB_LW_GO(4, 2, 5, 2, true); else ISB_LW_GO(4, 2, 5, 2, false); }
#undef ISB_LW_GO
            break;
        }
#undef ISB_CONV_LAUNCH_G1G
#undef ISB_CONV_LAUNCH_G1GH
#undef ISB_CONV_LAUNCH_G1GN
#undef ISB_CONV_LAUNCH_G1GT
        default:
            set_error("conv_igemm: unknown tile variant %d", v);
            return ISB_ERR_INVALID;
    }
#undef ISB_CONV_LAUNCH
    ISB_LAUNCHED("conv_igemm", st);
    return ISB_OK;
}

int launch_conv_expand_dw(const ConvArgs& a, hipStream_t st) {
    const int hw = a.OH * a.OW;
    if (a.KH != 1 || a.stride != 1 || a.gate || a.res || !a.zeros || !a.dw_w || !a.dw_bias || !a.dw_out || !a.pooled || a.OH != a.OW ||
        a.Cin % 32 != 0) {
        set_error("conv_expand_dw: needs an un-gated stride-1 1x1 expand conv with depthwise weights");
        return ISB_ERR_INVALID;
    }
    ConvArgs aa = a;
    aa.grid_mode = 0;
    if (hw == 256 && a.Cout % 64 == 0) {           // one 16x16 sample per tile: 256 x 64, 8 waves
        dim3 g(a.B, a.Cout / 64);
        hipLaunchKernelGGL((conv_igemm_dma_kernel<1, 2, 8, 1, 2, true>), g, dim3(512), 0, st, aa);
    } else if (hw == 64 && a.Cout % 128 == 0) {    // one 8x8 sample per tile: 64 x 128, 4 waves
        dim3 g(a.B, a.Cout / 128);
        hipLaunchKernelGGL((conv_igemm_dma_kernel<1, 2, 2, 2, 2, true>), g, dim3(256), 0, st, aa);
    } else {
        set_error("conv_expand_dw: unsupported shape hw=%d Cout=%d", hw, a.Cout);
        return ISB_ERR_INVALID;
    }
    ISB_LAUNCHED("conv_expand_dw", st);
    return ISB_OK;
}

// =====================================================================================
// depthwise 3x3 (+ folded-BN bias + SiLU) fused with the squeeze-excite average pool.
// WG = one sample x a slab of CH 8-channel chunks x all output pixels; thread = (chunk, pixel-quad):
// 4 horizontally adjacent outputs share their input columns ((4-1)*S+3 columns x 3 rows of 16-B loads
// instead of 36). The per-(sample, channel) mean is reduced inside the WG in a fixed order (no atomics:
// results do not depend on scheduling) and written straight to pooled[b][c].
// weights tap-major bf16 [9][C] (BN scale folded); the 9-tap sums run on v_dot2c_f32_bf16 with the other half
// of the weight pair zeroed = exact f32 FMAs fed by the packed activations
// =====================================================================================
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// acc += x.lo * w.lo + x.hi * w.hi on bf16 pairs (v_dot2c_f32_bf16). With one half of w zero this is the exact
// f32 FMA of ONE channel straight from the packed activations: no bf16 -> f32 unpacking, no operand shuffles.
__device__ __forceinline__ float dot2_bf16(uint32_t x, uint32_t w, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, x), __builtin_bit_cast(bf16x2_t, w), acc, false);
}

// sum over the 16 lanes of a DPP row (row_ror 8, 4, 2, 1), the total in every lane
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}

// IN_F16 / OUT_F16: input + taps / output in fp16 instead of bf16 (DwArgs.in_f16 / out_f16)
template <int S, bool FC1 = false, bool IN_F16 = false, bool OUT_F16 = false>
__global__ __launch_bounds__(256) void dwconv3x3_pool_kernel(DwArgs p) {
    __shared__ float red[32][129];
    __shared__ __attribute__((aligned(16))) float pmean[FC1 ? 128 : 4];
    constexpr int NCOL = 3 * S + 3;
    const int nq = (p.OH * p.OW) >> 2;                    // pixel quads per sample
    const int PQ = nq >= 32 ? 32 : nq;                    // quad slots in the WG
    const int CH = 256 / PQ;                              // chunks per WG (8 or 16)
    const int cl = threadIdx.x % CH, pq = threadIdx.x / CH;
    const int b = blockIdx.y;
    const int c = (blockIdx.x * CH + cl) * 8;
    const bool cok = c < p.C;
    // FC1: the slab's squeeze-excite weights do not depend on anything computed here -- request them first, they
    // arrive while the taps run (one frame = one workgroup per CU: the 80 registers cost no occupancy)
    const int sub = threadIdx.x & 31, grp = threadIdx.x >> 5;
    bool fc_ok = false;
    float4 wv[FC1 ? 20 : 1];
    if constexpr (FC1) {
        const int c0 = blockIdx.x * CH * 8, nch = min(CH * 8, p.C - c0);
        fc_ok = sub * 4 < nch;
#pragma unroll
        for (int q = 0; q < 20; ++q) {
            const int j = grp + 8 * q;
            wv[q] = (fc_ok && j < p.cse) ? *reinterpret_cast<const float4*>(p.se_w1 + (size_t)j * p.C + c0 + sub * 4)
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    uint32_t wlo[9][4], whi[9][4];                        // tap weights of the even / odd channel of each pair
    float bias[8], psum[8];
    // bf16 (1, 0) and (0, 1) in registers: as a literal 0x3f800000 becomes the INLINE constant 1.0, which a packed
    // bf16 operand reads as (1, 0) -- the pool would sum the wrong channel
    uint32_t one_lo, one_hi;
    if constexpr (OUT_F16) asm volatile("v_mov_b32 %0, 0x3c00\n\tv_lshlrev_b32 %1, 16, %0" : "=v"(one_lo), "=v"(one_hi));
    else asm volatile("v_mov_b32 %0, 0x3f80\n\tv_lshlrev_b32 %1, 16, %0" : "=v"(one_lo), "=v"(one_hi));
#pragma unroll
    for (int e = 0; e < 8; ++e) psum[e] = 0.f;
    if (cok) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const uint4 wv = *reinterpret_cast<const uint4*>(p.w + (size_t)t * p.C + c);
            const uint32_t wp[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { wlo[t][e] = wp[e] & 0xffffu; whi[t][e] = wp[e] & 0xffff0000u; }
        }
        const float4 s0 = *reinterpret_cast<const float4*>(p.bias + c), s1 = *reinterpret_cast<const float4*>(p.bias + c + 4);
        bias[0] = s0.x; bias[1] = s0.y; bias[2] = s0.z; bias[3] = s0.w; bias[4] = s1.x; bias[5] = s1.y; bias[6] = s1.z; bias[7] = s1.w;
        const int qpr = p.OW >> 2;                        // quads per output row
        for (int q = pq; q < nq; q += PQ) {
            const int oy = q / qpr, ox0 = (q - oy * qpr) * 4;
            float acc[4][8];
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[o][e] = bias[e];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = oy * S - p.pad + ky;
                const bool yok = (unsigned)iy < (unsigned)p.H;
                const uint16_t* rowp = p.in + ((size_t)(b * p.H + (yok ? iy : 0)) * p.W) * p.C + c;
                uint4 v[NCOL];
#pragma unroll
                for (int col = 0; col < NCOL; ++col) {     // branch-free: every load of the row is in flight at once
                    const int ix = ox0 * S - p.pad + col;
                    const bool ok = yok && (unsigned)ix < (unsigned)p.W;
                    const uint4 t = *reinterpret_cast<const uint4*>(rowp + (size_t)(ok ? ix : 0) * p.C);
                    v[col] = ok ? t : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int col = 0; col < NCOL; ++col) {
                    const uint32_t x[4] = {v[col].x, v[col].y, v[col].z, v[col].w};
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        const int kx = col - o * S;       // tap of output o that reads this column
                        if (kx >= 0 && kx < 3) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                acc[o][2 * e] = T16<IN_F16>::dot2(x[e], wlo[ky * 3 + kx][e], acc[o][2 * e]);
                                acc[o][2 * e + 1] = T16<IN_F16>::dot2(x[e], whi[ky * 3 + kx][e], acc[o][2 * e + 1]);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                uint32_t pk[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint16_t lo = T16<OUT_F16>::from_f32(silu_fast(acc[o][2 * e])), hi = T16<OUT_F16>::from_f32(silu_fast(acc[o][2 * e + 1]));
                    pk[e] = (uint32_t)lo | ((uint32_t)hi << 16);
                    // the pool sees the stored (rounded) activations: x * 1.0 + psum, one instruction per channel
                    psum[2 * e] = T16<OUT_F16>::dot2(pk[e], one_lo, psum[2 * e]);
                    psum[2 * e + 1] = T16<OUT_F16>::dot2(pk[e], one_hi, psum[2 * e + 1]);
                }
                *reinterpret_cast<uint4*>(p.out + (((size_t)(b * p.OH + oy) * p.OW + ox0 + o) * p.C + c)) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            }
        }
    }
    if (p.pooled) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[pq][cl * 8 + e] = psum[e];
        __syncthreads();
        if ((int)threadIdx.x < CH * 8) {
            const int cc = blockIdx.x * CH * 8 + threadIdx.x;
            float mean = 0.f;
            if (cc < p.C) {
                float t = 0.f;
                for (int s2 = 0; s2 < PQ; ++s2) t += red[s2][threadIdx.x];
                mean = t / (float)(p.OH * p.OW);
                p.pooled[(size_t)b * p.C + cc] = mean;
            }
            if constexpr (FC1) pmean[threadIdx.x] = mean;
        }
        if constexpr (FC1) {
            // this slab's share of squeeze-excite FC1: 32 lanes per output row j hold the row's slab (requested at
            // kernel start), multiply with the pooled means and meet in a butterfly. 8 groups x 20 rows: cse <= 160.
            __syncthreads();
            const float4 pv = fc_ok ? *reinterpret_cast<const float4*>(&pmean[sub * 4]) : make_float4(0.f, 0.f, 0.f, 0.f);
            float racc[20], rup[20];
#pragma unroll
            for (int q = 0; q < 20; ++q)      // 16-lane sums on the vector ALU (DPP row rotations), no LDS round trips
                racc[q] = row16_sum(fmaf(pv.w, wv[q].w, fmaf(pv.z, wv[q].z, fmaf(pv.y, wv[q].y, pv.x * wv[q].x))));
#pragma unroll
            for (int q = 0; q < 20; ++q) rup[q] = __shfl_xor(racc[q], 16, 64);     // the other half-row: 20 permutes in flight
#pragma unroll
            for (int q = 0; q < 20; ++q) {
                const int j = grp + 8 * q;
                if (sub == 0 && j < p.cse) p.se_part[((size_t)blockIdx.x * p.B + b) * p.cse + j] = racc[q] + rup[q];
            }
        }
    }
}

// Whole-sample maps of 8 x 8 and 16 x 16 pixels at stride 1 (stages 3-6: 54 of the 61 depthwise launches). On these the kernel
// above moved its bytes at 3.5 TB/s (5.0-5.5 on the two stride-2 launches): a workgroup's working set is 16-32 KiB (one sample x
// 128 / 64 channels), but its threads request it 4.5 times over through the texture path (six 16-byte columns x three rows per
// four outputs), and only a few hundred bytes per thread are ever in flight. Here the slab goes to LDS ONCE, as fully coalesced
// 256- / 128-byte pixel rows (every thread has its 4 / 8 16-byte loads in flight at the start), inside a ring of zero pixels --
// TF-SAME padding becomes data -- and the taps read LDS (conflict-free: the 16 lanes of a ds_read_b128 group cover whole
// pixels). Same thread <-> (pixel quad, channel chunk) map, same tap order, same accumulators and pool order as
// dwconv3x3_pool_kernel<1>: bit-identical (tested). 8 x 8: 3.5 -> 4.7 TB/s.
template <bool F16, int HW, bool FC1 = false>
__global__ __launch_bounds__(256) void dwconv3x3_map_kernel(DwArgs p) {
    constexpr int NQ = HW * HW / 4;                        // pixel quads per sample
    constexpr int PQ = NQ >= 32 ? 32 : NQ;                 // quad slots in the workgroup (16 on 8 x 8 maps)
    constexpr int CH = 256 / PQ;                           // 8-channel chunks per workgroup: 16 (128 channels) / 8 (64 channels)
    constexpr int TW = HW + 2;                             // tile width with the zero ring
    constexpr int NLD = HW * HW * CH / 256;                // 16-byte loads per thread: 4 / 8
    constexpr int QPR = HW / 4;                            // quads per row
    __shared__ __attribute__((aligned(16))) uint4 tile[TW * TW * CH];     // [y + 1][x + 1][chunk]
    __shared__ float red[PQ][CH * 8 + 1];
    __shared__ __attribute__((aligned(16))) float pmean[FC1 ? 128 : 4];
    const int tid = threadIdx.x;
    const int cl = tid % CH, pq = tid / CH;
    const int b = blockIdx.y, c0 = blockIdx.x * (CH * 8);
    const int c = c0 + cl * 8;
    const bool cok = c < p.C;
    // the slab: pixel px = idx / CH, chunk idx % CH -- CH consecutive threads fetch one pixel's contiguous bytes
    uint4 ld[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int idx = tid + 256 * k, px = idx / CH, ch = idx % CH;
        ld[k] = (c0 + ch * 8 < p.C) ? *reinterpret_cast<const uint4*>(p.in + ((size_t)b * (HW * HW) + px) * p.C + c0 + ch * 8) : make_uint4(0, 0, 0, 0);
    }
    for (int i = tid; i < (4 * HW + 4) * CH; i += 256) {   // the ring of zero pixels: top row, bottom row, left / right columns
        const int q = i / CH, ch = i % CH;
        const int y = q < TW ? 0 : (q < 2 * TW ? TW - 1 : 1 + ((q - 2 * TW) >> 1)), x = q < TW ? q : (q < 2 * TW ? q - TW : ((q - 2 * TW) & 1) * (TW - 1));
        tile[(y * TW + x) * CH + ch] = make_uint4(0, 0, 0, 0);
    }
    uint32_t wlo[9][4], whi[9][4];
    float bias[8], psum[8];
    uint32_t one_lo, one_hi;
    if constexpr (F16) asm volatile("v_mov_b32 %0, 0x3c00\n\tv_lshlrev_b32 %1, 16, %0" : "=v"(one_lo), "=v"(one_hi));
    else asm volatile("v_mov_b32 %0, 0x3f80\n\tv_lshlrev_b32 %1, 16, %0" : "=v"(one_lo), "=v"(one_hi));
#pragma unroll
    for (int e = 0; e < 8; ++e) psum[e] = 0.f;
    if (cok) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const uint4 wv = *reinterpret_cast<const uint4*>(p.w + (size_t)t * p.C + c);
            const uint32_t wp[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { wlo[t][e] = wp[e] & 0xffffu; whi[t][e] = wp[e] & 0xffff0000u; }
        }
        const float4 s0 = *reinterpret_cast<const float4*>(p.bias + c), s1 = *reinterpret_cast<const float4*>(p.bias + c + 4);
        bias[0] = s0.x; bias[1] = s0.y; bias[2] = s0.z; bias[3] = s0.w; bias[4] = s1.x; bias[5] = s1.y; bias[6] = s1.z; bias[7] = s1.w;
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int idx = tid + 256 * k, px = idx / CH, ch = idx % CH;
        tile[((px / HW + 1) * TW + (px % HW) + 1) * CH + ch] = ld[k];
    }
    __syncthreads();
    if (cok) {
        for (int q = pq; q < NQ; q += PQ) {
            const int oy = q / QPR, ox0 = (q - oy * QPR) * 4;
            float acc[4][8];
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[o][e] = bias[e];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                uint4 v[6];
#pragma unroll
                for (int col = 0; col < 6; ++col) v[col] = tile[((oy + ky) * TW + ox0 + col) * CH + cl];
#pragma unroll
                for (int col = 0; col < 6; ++col) {
                    const uint32_t x[4] = {v[col].x, v[col].y, v[col].z, v[col].w};
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        const int kx = col - o;
                        if (kx >= 0 && kx < 3) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                acc[o][2 * e] = T16<F16>::dot2(x[e], wlo[ky * 3 + kx][e], acc[o][2 * e]);
                                acc[o][2 * e + 1] = T16<F16>::dot2(x[e], whi[ky * 3 + kx][e], acc[o][2 * e + 1]);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                uint32_t pk[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint16_t lo = T16<F16>::from_f32(silu_fast(acc[o][2 * e])), hi = T16<F16>::from_f32(silu_fast(acc[o][2 * e + 1]));
                    pk[e] = (uint32_t)lo | ((uint32_t)hi << 16);
                    psum[2 * e] = T16<F16>::dot2(pk[e], one_lo, psum[2 * e]);
                    psum[2 * e + 1] = T16<F16>::dot2(pk[e], one_hi, psum[2 * e + 1]);
                }
                *reinterpret_cast<uint4*>(p.out + (((size_t)b * (HW * HW) + oy * HW + ox0 + o) * p.C + c)) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            }
        }
    }
    if (p.pooled) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[pq][cl * 8 + e] = psum[e];
        // FC1: this slab's rows of the squeeze-excite weights, requested before the pool sums meet (same thread <-> (row, channel)
        // map, same sums as dwconv3x3_pool_kernel's: bit-identical partials)
        const int sub = tid & 31, grp = tid >> 5;
        bool fc_ok = false;
        float4 wv[FC1 ? 20 : 1];
        if constexpr (FC1) {
            fc_ok = sub * 4 < min(CH * 8, p.C - c0);
#pragma unroll
            for (int q = 0; q < 20; ++q) {
                const int j = grp + 8 * q;
                wv[q] = (fc_ok && j < p.cse) ? *reinterpret_cast<const float4*>(p.se_w1 + (size_t)j * p.C + c0 + sub * 4)
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        __syncthreads();
        if (tid < CH * 8) {
            const int cc = c0 + tid;
            float mean = 0.f;
            if (cc < p.C) {
                float t = 0.f;
                for (int s2 = 0; s2 < PQ; ++s2) t += red[s2][tid];
                mean = t / (float)(HW * HW);
                p.pooled[(size_t)b * p.C + cc] = mean;
            }
            if constexpr (FC1) pmean[tid] = mean;
        }
        if constexpr (FC1) {
            __syncthreads();
            const float4 pv = fc_ok ? *reinterpret_cast<const float4*>(&pmean[sub * 4]) : make_float4(0.f, 0.f, 0.f, 0.f);
            float racc[20], rup[20];
#pragma unroll
            for (int q = 0; q < 20; ++q)
                racc[q] = row16_sum(fmaf(pv.w, wv[q].w, fmaf(pv.z, wv[q].z, fmaf(pv.y, wv[q].y, pv.x * wv[q].x))));
#pragma unroll
            for (int q = 0; q < 20; ++q) rup[q] = __shfl_xor(racc[q], 16, 64);
#pragma unroll
            for (int q = 0; q < 20; ++q) {
                const int j = grp + 8 * q;
                if (sub == 0 && j < p.cse) p.se_part[((size_t)blockIdx.x * p.B + b) * p.cse + j] = racc[q] + rup[q];
            }
        }
    }
}

static bool dw_map8_on() {          // ISB_DW_MAP8=0: the general kernel on 8 x 8 / 16 x 16 maps too (A/B switch)
    static const bool on = [] { const char* e = getenv("ISB_DW_MAP8"); return !e || atoi(e) != 0; }();
    return on;
}

int dw_slabs(const DwArgs& a) {
    const int nq = (a.OH * a.OW) >> 2;
    return cdiv(a.C / 8, 256 / std::min(32, std::max(nq, 1)));
}

int launch_dwconv3x3(const DwArgs& a, hipStream_t st) {
    if (a.C % 8 != 0 || a.OW % 4 != 0 || ((a.OH * a.OW) >> 2) < 8 || 256 % std::min(32, (a.OH * a.OW) >> 2) != 0) {
        set_error("dwconv3x3: unsupported shape C=%d OH=%d OW=%d", a.C, a.OH, a.OW);
        return ISB_ERR_INVALID;
    }
    dim3 grid(dw_slabs(a), a.B);
    if (a.se_w1) {
        if (!a.pooled || !a.se_part || a.cse < 1 || a.cse > 160 || a.C % 8 != 0 || (int)grid.x > SE_MAX_PARTS) {
            set_error("dwconv3x3: the folded SE FC1 needs pooled + se_part, cse <= 160 and at most %d slabs (C=%d)", SE_MAX_PARTS, a.C);
            return ISB_ERR_INVALID;
        }
    }
    // fp16 forms: stride 1 fp16 -> fp16 (the blocks inside the fp16 stages), stride 2 bf16 -> fp16 (the block that enters them)
    const int form = (a.in_f16 ? 1 : 0) | (a.out_f16 ? 2 : 0);
    if ((form == 3 && a.stride != 1) || (form == 2 && a.stride != 2) || form == 1) {
        set_error("dwconv3x3: fp16 forms are stride 1 fp16 -> fp16 and stride 2 bf16 -> fp16 (in_f16=%d out_f16=%d stride=%d)", a.in_f16, a.out_f16, a.stride);
        return ISB_ERR_INVALID;
    }
    if (a.stride == 1 && a.H == a.W && (a.H == 8 || a.H == 16) && a.OH == a.H && a.OW == a.W && a.pad == 1 && (form == 0 || form == 3) &&
        !a.general && dw_map8_on()) {
        // grid.x = dw_slabs(a): slabs of 128 (8 x 8 maps) / 64 (16 x 16 maps) channels, as in the general kernel
#define ISB_DW_MAP(F16_, HW_)                                                                                            \
    do {                                                                                                                 \
        if (a.se_w1) hipLaunchKernelGGL((dwconv3x3_map_kernel<F16_, HW_, true>), grid, dim3(256), 0, st, a);             \
        else hipLaunchKernelGGL((dwconv3x3_map_kernel<F16_, HW_, false>), grid, dim3(256), 0, st, a);                    \
    } while (0)
        if (a.H == 8) { if (form == 3) ISB_DW_MAP(true, 8); else ISB_DW_MAP(false, 8); }
        else { if (form == 3) ISB_DW_MAP(true, 16); else ISB_DW_MAP(false, 16); }
#undef ISB_DW_MAP
        ISB_LAUNCHED("dwconv3x3_map", st);
        return ISB_OK;
    }
    if (a.se_w1) {
        if (form == 3) hipLaunchKernelGGL((dwconv3x3_pool_kernel<1, true, true, true>), grid, dim3(256), 0, st, a);
        else if (form == 2) hipLaunchKernelGGL((dwconv3x3_pool_kernel<2, true, false, true>), grid, dim3(256), 0, st, a);
        else if (a.stride == 1) hipLaunchKernelGGL((dwconv3x3_pool_kernel<1, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((dwconv3x3_pool_kernel<2, true>), grid, dim3(256), 0, st, a);
    } else if (form == 3) hipLaunchKernelGGL((dwconv3x3_pool_kernel<1, false, true, true>), grid, dim3(256), 0, st, a);
    else if (form == 2) hipLaunchKernelGGL((dwconv3x3_pool_kernel<2, false, false, true>), grid, dim3(256), 0, st, a);
    else if (a.stride == 1) hipLaunchKernelGGL((dwconv3x3_pool_kernel<1, false>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dwconv3x3_pool_kernel<2, false>), grid, dim3(256), 0, st, a);
    ISB_LAUNCHED("dwconv3x3_pool", st);
    return ISB_OK;
}

// =====================================================================================
// squeeze-excite FCs in f32 on the vector ALU. Two tiny GEMMs (B x cse x C, 0.1 GFLOP) between a
// global pool and the gated projection: what costs is latency and parallelism, not FLOPs.
//   se_fc1_part: part[kc][b][j] = sum_{c in chunk kc} pooled[b][c] * W1[j][c]
//                WG = 16 samples x 16 outputs x one 256-channel chunk, operands staged in LDS once
//                (grid = cse/16 x B/16 x C/256 workgroups: every CU gets work, each makes ONE round trip)
//   se_fc2:      mid[b][j]  = silu(b1[j] + sum_kc part[kc][b][j])           (prologue, fixed order)
//                gate[b][c] = sigmoid(b2[c] + sum_j mid[b][j] * W2T[j][c])
//                WG = 8 samples x 256 channels; lane = 4 channels (16-B weight loads), the 4 waves split j,
//                partial sums meet in LDS and are added in wave order
// every sum runs in a fixed order that depends on neither scheduling nor the batch size
// =====================================================================================
constexpr int SE_CHUNK = 256;      // channels per fc1 workgroup
constexpr int SE_ROW = SE_CHUNK + 4;   // LDS row stride in floats (1040 B: consecutive rows land in consecutive 16-B slots)

__global__ __launch_bounds__(256) void se_fc1_part_kernel(SeFcArgs p) {
    __shared__ __attribute__((aligned(16))) float Ps[16][SE_ROW];
    __shared__ __attribute__((aligned(16))) float Ws[16][SE_ROW];
    const int t = threadIdx.x;
    const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 16, c0 = blockIdx.z * SE_CHUNK;
    {
        const int row = t >> 4, q = (t & 15) * 4;
        const int b = b0 + row, j = j0 + row;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int cl = pass * 64 + q, c = c0 + cl;
            float4 pv = make_float4(0.f, 0.f, 0.f, 0.f), wv = pv;
            if (c < p.C) {
                if (b < p.B) pv = *reinterpret_cast<const float4*>(p.pooled + (size_t)b * p.C + c);
                if (j < p.cse) wv = *reinterpret_cast<const float4*>(p.w1 + (size_t)j * p.C + c);
            }
            *reinterpret_cast<float4*>(&Ps[row][cl]) = pv;
            *reinterpret_cast<float4*>(&Ws[row][cl]) = wv;
        }
    }
    __syncthreads();
    const int s = t >> 4, jl = t & 15;
    float acc = 0.f;
#pragma unroll 8
    for (int c4 = 0; c4 < SE_CHUNK / 4; ++c4) {
        const float4 pv = *reinterpret_cast<const float4*>(&Ps[s][c4 * 4]);
        const float4 wv = *reinterpret_cast<const float4*>(&Ws[jl][c4 * 4]);
        acc = fmaf(pv.x, wv.x, acc);
        acc = fmaf(pv.y, wv.y, acc);
        acc = fmaf(pv.z, wv.z, acc);
        acc = fmaf(pv.w, wv.w, acc);
    }
    const int b = b0 + s, j = j0 + jl;
    if (b < p.B && j < p.cse) p.part[((size_t)blockIdx.z * p.B + b) * p.cse + j] = acc;
}

// PRE (launches of a few samples: one workgroup per CU, latency is everything): the wave's W2T rows are requested
// before the FC1 partials are even read, so the kernel makes one memory round trip instead of six
template <bool PRE>
__global__ __launch_bounds__(256) void se_fc2_kernel(SeFcArgs p) {
    __shared__ __attribute__((aligned(16))) float mids[160][8];          // [j][sample]
    __shared__ __attribute__((aligned(16))) float4 red[4][8][64];        // [wave][sample][lane]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int b0 = blockIdx.y * 8;
    float4 wpre[PRE ? 40 : 1];
    if constexpr (PRE) {
        const int c = blockIdx.x * 256 + lane * 4;
        const int jq = (p.cse + 3) >> 2, jb = wave * jq, je = min(p.cse, jb + jq);
#pragma unroll
        for (int q = 0; q < 40; ++q)
            wpre[q] = (c < p.C && jb + q < je) ? *reinterpret_cast<const float4*>(p.w2t + (size_t)(jb + q) * p.C + c)
                                              : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int nkc = p.nparts > 0 ? p.nparts : (p.C + SE_CHUNK - 1) / SE_CHUNK;
    for (int i = t; i < 8 * p.cse; i += 256) {
        const int sm = i / p.cse, j = i - sm * p.cse;
        float v = 0.f;
        if (b0 + sm < p.B) {
            float pv[SE_MAX_PARTS];            // all partials requested at once (one memory round trip), added in order
#pragma unroll
            for (int kc = 0; kc < SE_MAX_PARTS; ++kc) pv[kc] = kc < nkc ? p.part[((size_t)kc * p.B + b0 + sm) * p.cse + j] : 0.f;
            v = p.b1[j];
#pragma unroll
            for (int kc = 0; kc < SE_MAX_PARTS; ++kc)
                if (kc < nkc) v += pv[kc];
            v = v / (1.0f + expf(-v));
        }
        mids[j][sm] = v;
    }
    __syncthreads();
    const int c = blockIdx.x * 256 + lane * 4;
    const bool cok = c < p.C;
    const int jq = (p.cse + 3) >> 2;
    const int jb = wave * jq, je = min(p.cse, jb + jq);
    float4 acc[8];
#pragma unroll
    for (int sm = 0; sm < 8; ++sm) acc[sm] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int ns = min(8, p.B - b0);
    if constexpr (PRE) {
#pragma unroll
        for (int q = 0; q < 40; ++q) {
            if (jb + q >= je) break;
            const float4 wv = wpre[q];
            const float4 m0 = *reinterpret_cast<const float4*>(&mids[jb + q][0]);
            const float4 m1 = *reinterpret_cast<const float4*>(&mids[jb + q][4]);
            const float mm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
            for (int sm = 0; sm < 8; ++sm) {
                if (sm >= ns) break;           // a single frame pays for one sample, not eight
                acc[sm].x = fmaf(mm[sm], wv.x, acc[sm].x);
                acc[sm].y = fmaf(mm[sm], wv.y, acc[sm].y);
                acc[sm].z = fmaf(mm[sm], wv.z, acc[sm].z);
                acc[sm].w = fmaf(mm[sm], wv.w, acc[sm].w);
            }
        }
    } else if (cok) {
#pragma unroll 8
        for (int j = jb; j < je; ++j) {
            const float4 wv = *reinterpret_cast<const float4*>(p.w2t + (size_t)j * p.C + c);
            const float4 m0 = *reinterpret_cast<const float4*>(&mids[j][0]);
            const float4 m1 = *reinterpret_cast<const float4*>(&mids[j][4]);
            const float mm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
            for (int sm = 0; sm < 8; ++sm) {
                acc[sm].x = fmaf(mm[sm], wv.x, acc[sm].x);
                acc[sm].y = fmaf(mm[sm], wv.y, acc[sm].y);
                acc[sm].z = fmaf(mm[sm], wv.z, acc[sm].z);
                acc[sm].w = fmaf(mm[sm], wv.w, acc[sm].w);
            }
        }
    }
#pragma unroll
    for (int sm = 0; sm < 8; ++sm) red[wave][sm][lane] = acc[sm];
    __syncthreads();
    if (!cok) return;
    const float4 bias = *reinterpret_cast<const float4*>(p.b2 + c);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int sm = wave * 2 + q;
        if (b0 + sm >= p.B) continue;
        float4 v = red[0][sm][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float4 u = red[w][sm][lane];
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        v.x = 1.0f / (1.0f + expf(-(v.x + bias.x)));
        v.y = 1.0f / (1.0f + expf(-(v.y + bias.y)));
        v.z = 1.0f / (1.0f + expf(-(v.z + bias.z)));
        v.w = 1.0f / (1.0f + expf(-(v.w + bias.w)));
        *reinterpret_cast<float4*>(p.gate + (size_t)(b0 + sm) * p.C + c) = v;
    }
}

int launch_se_fcs(const SeFcArgs& a, hipStream_t st) {
    if (a.cse > 160 || a.C % 4 != 0 || a.C > 15 * SE_CHUNK || !a.part || a.nparts > SE_MAX_PARTS) {
        set_error("se_fcs: unsupported shape cse=%d C=%d parts=%d", a.cse, a.C, a.nparts);
        return ISB_ERR_INVALID;
    }
    if (a.nparts <= 0)
        hipLaunchKernelGGL(se_fc1_part_kernel, dim3(cdiv(a.cse, 16), cdiv(a.B, 16), cdiv(a.C, SE_CHUNK)), dim3(256), 0, st, a);
    if (a.B <= 8) hipLaunchKernelGGL(se_fc2_kernel<true>, dim3(cdiv(a.C, 256), 1), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(se_fc2_kernel<false>, dim3(cdiv(a.C, 256), cdiv(a.B, 8)), dim3(256), 0, st, a);
    ISB_LAUNCHED("se_fcs", st);
    return ISB_OK;
}

// =====================================================================================
// stem: conv3x3 stride 2 (TF SAME on an even input: pad bottom/right), 3 -> 32, bias, SiLU.
// f32 crop [B,256,256,3] -> bf16 [B,128,128,32]. thread = one output pixel, all 32 channels;
// weights [32][3][3][3] f32 (scale folded) are wave-uniform -> scalar loads.
// =====================================================================================
__global__ __launch_bounds__(256) void stem_kernel(StemArgs p) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int OH = p.H / 2, OW = p.W / 2;
    if (idx >= (size_t)p.B * OH * OW) return;
    const int ox = (int)(idx % OW), oy = (int)((idx / OW) % OH), b = (int)(idx / ((size_t)OW * OH));
    float x[27];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = 2 * oy + ky, ix = 2 * ox + kx;
            const bool ok = iy < p.H && ix < p.W;
            const float* src = p.in + ((size_t)(b * p.H + (ok ? iy : 0)) * p.W + (ok ? ix : 0)) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) x[(ky * 3 + kx) * 3 + c] = ok ? src[c] : 0.f;
        }
    uint32_t o[16];
#pragma unroll
    for (int co = 0; co < 32; co += 2) {
        float a0 = p.bias[co], a1 = p.bias[co + 1];
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            a0 = fmaf(x[k], p.w[co * 27 + k], a0);
            a1 = fmaf(x[k], p.w[(co + 1) * 27 + k], a1);
        }
        o[co >> 1] = (uint32_t)f2bf_(silu_(a0)) | ((uint32_t)f2bf_(silu_(a1)) << 16);
    }
    uint4* dst = reinterpret_cast<uint4*>(p.out + idx * 32);
    dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
    dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
    dst[2] = make_uint4(o[8], o[9], o[10], o[11]);
    dst[3] = make_uint4(o[12], o[13], o[14], o[15]);
}

int launch_stem(const StemArgs& a, hipStream_t st) {
    const size_t total = (size_t)a.B * (a.H / 2) * (a.W / 2);
    hipLaunchKernelGGL(stem_kernel, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, st, a);
    ISB_LAUNCHED("stem", st);
    return ISB_OK;
}

// f32 -> bf16 (weights at load time), with an optional per-row scale (folded BN)
__global__ void f32_to_bf16_rows_kernel(const float* in, const float* row_scale, uint16_t* out, size_t rows, size_t cols, int f16) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * cols) return;
    const float s = row_scale ? row_scale[i / cols] : 1.f;
    out[i] = f16 ? f2h_(in[i] * s) : f2bf_(in[i] * s);
}

int launch_f32_to_bf16_rows(const float* in, const float* row_scale, uint16_t* out, size_t rows, size_t cols, hipStream_t st, int f16) {
    hipLaunchKernelGGL(f32_to_bf16_rows_kernel, dim3((unsigned)cdivz(rows * cols, 256)), dim3(256), 0, st, in, row_scale, out, rows, cols, f16);
    ISB_LAUNCHED("f32_to_bf16_rows", st);
    return ISB_OK;
}

}  // namespace isb
