// The FRONT HALF of a stride-1 MBConv block on 16 x 16 maps in one launch (round 5): 1x1 expand + BN + SiLU -> depthwise 3x3 + BN +
// SiLU -> D (NHWC, what the gated projection reads) + the squeeze-excite pool. The expanded tensor (176 MB per 224 -> 1344 block
// at 256 frames, written once and read once by the two-launch path) never leaves the chip.
//
// mbfront8_kernel (conv_mb8.hip) did this for 8 x 8 maps, where a sample is one 64-row tile; it runs two waves per SIMD (96
// stationary weight registers per wave at K = 384) and is bound by latency, not by its instruction count (EXPERIMENTS.md round 5).
// Here K is 192 / 224 (48 / 56 weight registers), the depthwise taps run on the matrix pipe (dw_mm.h: no 80-register tap stage), and
// a sample is walked in BANDS of two image rows so that a wave's expanded tile is a ring of six padded rows (6.9 KB) instead of a
// whole padded map (20.7 KB): a workgroup of FOUR waves (one per SIMD) owns a 128-channel slice, THREE workgroups share a CU --
// three waves per SIMD. (Six-wave workgroups of 192 channels, which divide 768 / 1152 / 1344 evenly, were the first form: the
// hardware then places ONE workgroup per CU -- a second one would need four waves on some SIMD -- and the launch ran in two rounds.)
//   step s = 0..7 of a sample: the band's 32 x Cin input tile (LDS-DMA, one buffer, requested behind the previous band's MFMAs)
//                x the wave's 32 channels (weights stationary: K / 4 registers) -> bias, SiLU, one rounding -> ring rows 2s, 2s + 1;
//   step s = 1..8: depthwise outputs of rows 2s - 2, 2s - 1 from ring rows 2s - 3 .. 2s (row -1 / 16: zeros; columns -1 / 16: the
//                ring's zero columns): per 8-channel group three 16-byte fragment reads + three v_mfma_f32_16x16x32 (dw_mm.h),
//                bias as the C operand, SiLU, one rounding, pooled sums, the two D rows through the two ring rows that just died.
// Arithmetic and summation orders are those of the expand GEMM (k ascending in one accumulator) and of dwconv3x3_mm_kernel<16>
// (tap MFMAs, lane sums over the bands in order, then dw_mm.h's butterfly over the 32 (pixel pair, pixel) lanes): bit-identical to that two-launch
// path (tested), which is what batches below the threshold run.
#include "conv_tiles.h"
#include "dw_mm.h"

namespace isb {

namespace {

template <int CIN>
struct Mf16 {
    static constexpr int NK16 = CIN / 16, NKT = CIN / 32;
    static constexpr int NW = 4;                            // waves per workgroup (one per SIMD): a 128-channel slice
    static constexpr int XBUF = NKT * 2048;                 // a band's input tile: [NKT][32 rows][64 B], swizzled (gemm1x1's A image)
    static constexpr int ET_ROW = 18 * 64;                  // one padded row of a wave's ring: [18 pixels][32 ch x 2 B]
    static constexpr int ET_BYTES = 6 * ET_ROW;             // six rows
    static constexpr int ET_OFF = XBUF;
    static constexpr int TBL_OFF = ET_OFF + NW * ET_BYTES;
    static constexpr int TBL_BYTES = 1024;                  // per wave: bias [32] f32 | depthwise bias [32] f32 | taps [9][32] 16-bit
    static constexpr int LDS = TBL_OFF + NW * TBL_BYTES;
    static_assert(3 * LDS <= 160 * 1024, "three workgroups per CU");
};

}  // namespace

template <int CIN, bool F16>
__global__ __launch_bounds__(256, 3) void mbfront16_kernel(MbFront16Args p) {
    using S = Mf16<CIN>;
    constexpr int NK16 = S::NK16, NW = S::NW, ET_ROW = S::ET_ROW;
    T16<F16>::enter();
    unsigned char* const lds = conv_lds_dyn;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int CEXP = p.cexp, NSL = (CEXP + 127) / 128;
    // workgroup -> (slice, sample sequence): ids 8 apart share an XCD, and all slices of a sample sequence sit on ONE XCD, so that its L2
    // fetches a sample's input once (slices dealt round-robin over the XCDs read every sample through every L2: +4.8 GB per pass)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int slice = slot % NSL, q = (slot / NSL) * 8 + xcd, Q = (int)(gridDim.x / (8 * NSL)) * 8;
    if (q >= p.B) return;
    const int cb = min(slice * NW + wave, CEXP / 32 - 1), c0 = cb * 32;
    const bool live = (slice * NW + wave) * 32 < CEXP;     // 1344 channels = 10 slices + 64: the last slice's upper waves only stage tiles and meet the barriers
    unsigned char* const et = lds + S::ET_OFF + wave * S::ET_BYTES;
    unsigned char* const tbl = lds + S::TBL_OFF + wave * S::TBL_BYTES;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)lds;

    // band s of sample smp -> the X tile: piece pc = (k-tile pc >> 1, rows 16 (pc & 1) ..), 1 KiB each
    auto dma_x = [&](int smp, int s) {
        const unsigned char* src = reinterpret_cast<const unsigned char*>(p.x) + ((size_t)smp * 256 + s * 32) * (CIN * 2);
        for (int pc = wave; pc < S::NKT * 2; pc += NW) {
            const int kt = pc >> 1, row = 16 * (pc & 1) + (lane >> 2);
            const int logical = (lane & 3) ^ ((row >> 2) & 3);
            dma16_s(src, (uint32_t)(row * CIN * 2 + kt * 64 + logical * 16), lds0 + (uint32_t)(pc * 1024));
        }
    };
    dma_x(q, 0);
    // the wave's weights, for the whole kernel (fragment-packed: one coalesced 1-KiB load per k16 step)
    uint4 wreg[NK16];
    {
        const uint4* src = p.w1p + (size_t)cb * NK16 * 64 + lane;
#pragma unroll
        for (int s = 0; s < NK16; ++s) wreg[s] = src[s * 64];
    }
    if (lane < 8) *reinterpret_cast<float4*>(tbl + lane * 16) = *reinterpret_cast<const float4*>(p.b1 + c0 + lane * 4);
    else if (lane < 16) *reinterpret_cast<float4*>(tbl + lane * 16) = *reinterpret_cast<const float4*>(p.dwb + c0 + (lane - 8) * 4);
    else if (lane < 16 + 36) {
        const int t = (lane - 16) >> 2, c4 = (lane - 16) & 3;
        *reinterpret_cast<uint4*>(tbl + 256 + t * 64 + c4 * 16) = *reinterpret_cast<const uint4*>(p.dww + (size_t)t * CEXP + c0 + c4 * 8);
    }
    for (int i = lane; i < S::ET_BYTES / 16; i += 64) *reinterpret_cast<uint4*>(et + i * 16) = make_uint4(0, 0, 0, 0);     // the ring, zero columns included
    const int a_sw0 = swz(r, h), a_sw1 = swz(r, 2 + h);
    uint32_t one_lo, one_hi;                                        // (1, 0) / (0, 1) pairs in the storage type (see dwconv3x3_pool_kernel)
    if constexpr (F16) asm volatile("v_mov_b32 %0, 0x3c00\n\tv_lshlrev_b32 %1, 16, %0" : "=v"(one_lo), "=v"(one_hi));
    else asm volatile("v_mov_b32 %0, 0x3f80\n\tv_lshlrev_b32 %1, 16, %0" : "=v"(one_lo), "=v"(one_hi));

    // lane constants. The ring's 16-byte chunk slots are turned by f(y, x) = ((x >> 2) & 1) | ((y & 1) << 1) (y the image row, x the
    // padded column): the sixteen lanes a ds_read_b128 serves together (pixels 2 n + j of two consecutive rows) then fall into
    // sixteen different slots of the 256-byte bank row (four pixels of 64 bytes)
    const DwmmLane wl(lane);
    const int w_d = min(max(wl.d, 0), 2);
    const int mn = lane & 15, mj = lane >> 4, ms = mj >> 1;        // pixel pair, input column / output rows, pixel of the pair
    // E write: lane = pixel r of the band (row r >> 4, column r & 15), 16 channels 4 h + 8 qq + i
    const int e_x = (r & 15) + 1;
    const int e_lane = (r >> 4) * ET_ROW + e_x * 64 + h * 8;
    const int e_f = ((e_x >> 2) & 1) | ((r >> 4) << 1);            // (the band's first row 2 s is even)
    // B fragment: output row ry of the step's two, pixel pair pr; input row index k = ry + ky of the four rows 2s - 3 .. 2s
    const int t_ry = mn >> 3, t_x = 2 * (mn & 7) + mj;             // padded column of the fragment's pixel
    const int t_lane = t_x * 64;
    int t_f[2];
    t_f[0] = ((t_x >> 2) & 1) | (((t_ry + 1) & 1) << 1);           // ky even: image row 2s - 3 + ry + ky is odd iff ry + ky is even
    t_f[1] = ((t_x >> 2) & 1) | ((t_ry & 1) << 1);                 // ky odd
    // D staging: the step's two output rows leave through the interiors of the two ring rows its taps read LAST (rows 2s - 3 and
    // 2s - 2 are dead once the step's MFMAs are done): the lane's output pixel (ry, 2 pr + ms), its 4 channels of group g
    const int d_x = 2 * (mn & 7) + ms;
    const int d_lane = (d_x + 1) * 64 + (mj & 1) * 8;
    const int d_f = (d_x >> 1) & 3;

    uint64_t stp[7] = {0, 0, 0, 0, 0, 0, 0}, st_t0 = 0, st_n = 0;       // tuning probe (MbFront16Args.stamps)
    uint64_t st_r0 = 0;
    if (p.stamps) { st_t0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int smp = q; smp < p.B; smp += Q) {
        // Three workgroups share a CU and a SIMD's issue port goes to the OLDEST wave first: left alone, a CU's first workgroup runs at
        // full speed (4 samples in 71 us), the third crawls (3 samples in 96 us) and the launch ends with CUs a third full (census of
        // the workgroups' clocks, EXPERIMENTS.md round 5). Priority outranks age, so a workgroup steps down as it gets ahead: the one
        // with the most samples still to do is served first and the three arrive together (first form, by samples done: 122.0 -> 114.7 us).
        {
            const int left = (p.B - 1 - smp) / Q;                  // samples this workgroup still has to do after this one
            if (left >= 3) __builtin_amdgcn_s_setprio(3);
            else if (left == 2) __builtin_amdgcn_s_setprio(2);
            else if (left == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        float psum[4][4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) psum[g][i] = 0.f;
        // image row -1: zeros (slot 5), its interior was the previous sample's
        *reinterpret_cast<uint4*>(et + 5 * ET_ROW + 64 + lane * 16) = make_uint4(0, 0, 0, 0);
#pragma unroll 1
        for (int s = 0; s <= 8; ++s) {
            uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
            if (p.stamps) { t0 = __builtin_amdgcn_s_memtime(); ++st_n; }
            t1 = t2 = t3 = t4 = t0;
            if (s < 8) {
                // ---- the band's tile landed (requested behind the previous band's MFMAs). Younger vector-memory operations of this
                // wave may keep flying (vmcnt retires in order): the two D-row stores of the previous band's depthwise step, and at a
                // sample's first band the previous sample's last four D-row stores and its pooled means
                // Round 6: this kernel is no longer the default (mbfront16r_kernel below is; this one stays as form 1, the A/B partner and
                // the bit-identity test's second witness), so its hand-counted waits -- vmcnt(2) behind a band's two D-row stores, vmcnt(5)
                // behind a sample's last four and its pooled means: correct only while the compiler emits exactly those stores --
                // became a plain vmcnt(0): +3 % on a kernel nobody times any more, and nothing left to verify in the disassembly.
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (p.stamps) t1 = __builtin_amdgcn_s_memtime();
                // ---- expand: 32 pixels x the wave's 32 channels; fragment reads two k16 steps ahead (second register set)
                f32x16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
                if (live) {
                    uint4 fa[2][2];
                    auto rd = [&](int pr2, uint4 (&f)[2]) __attribute__((always_inline)) {
                        f[0] = *reinterpret_cast<const uint4*>(lds + pr2 * 2048 + a_sw0);
                        f[1] = *reinterpret_cast<const uint4*>(lds + pr2 * 2048 + a_sw1);
                    };
                    rd(0, fa[0]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int pr2 = 0; pr2 < NK16 / 2; ++pr2) {
                        if (pr2 + 1 < NK16 / 2) rd(pr2 + 1, fa[(pr2 + 1) & 1]);
                        acc = T16<F16>::mfma32(wreg[2 * pr2], fa[pr2 & 1][0], acc);
                        acc = T16<F16>::mfma32(wreg[2 * pr2 + 1], fa[pr2 & 1][1], acc);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (p.stamps) t2 = __builtin_amdgcn_s_memtime();
                __builtin_amdgcn_s_barrier();                       // everybody has read the tile: the next one may land
                if (p.stamps) t3 = __builtin_amdgcn_s_memtime();
                if (s + 1 < 8) dma_x(smp, s + 1);
                else if (smp + Q < p.B) dma_x(smp + Q, 0);
                // ---- E = T16(silu(acc + bias)) -> ring rows 2s, 2s + 1 (slots (2s) % 6, + 1)
                unsigned char* const cell = et + ((2 * s) % 6) * ET_ROW + e_lane;
                if (live)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const float4 bs = *reinterpret_cast<const float4*>(tbl + (8 * qq + 4 * h) * 4);
                    const float v0 = silu_fast(acc[4 * qq] + bs.x), v1 = silu_fast(acc[4 * qq + 1] + bs.y);
                    const float v2 = silu_fast(acc[4 * qq + 2] + bs.z), v3 = silu_fast(acc[4 * qq + 3] + bs.w);
                    uint2 pk;
                    pk.x = T16<F16>::pack2(v0, v1);
                    pk.y = T16<F16>::pack2(v2, v3);
                    *reinterpret_cast<uint2*>(cell + ((qq ^ e_f) << 4)) = pk;
                }
            } else {
                // image row 16: zeros (slot 4)
                *reinterpret_cast<uint4*>(et + 4 * ET_ROW + 64 + lane * 16) = make_uint4(0, 0, 0, 0);
            }
            if (p.stamps) { t4 = __builtin_amdgcn_s_memtime(); stp[0] += t1 - t0; stp[1] += t2 - t1; stp[2] += t3 - t2; stp[3] += t4 - t3; }
            if (s == 0 || !live) continue;
            // ---- depthwise 3x3 + bias on the matrix pipe: output rows 2s - 2, 2s - 1 from ring rows 2s - 3 .. 2s
            int rowoff[3];                                          // the lane's three input rows: slot (2s + 3 + ry + ky) % 6
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int s_a = ((2 * s + 3 + ky) % 6) * ET_ROW, s_b = ((2 * s + 4 + ky) % 6) * ET_ROW;       // (wave-uniform)
                rowoff[ky] = (t_ry ? s_b : s_a) + t_lane;
            }
            f32x4 a4[4];
            {
                uint4 bf[2][3];
                auto rdb = [&](int g, uint4 (&f)[3]) __attribute__((always_inline)) {
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) f[ky] = *reinterpret_cast<const uint4*>(et + rowoff[ky] + ((g ^ t_f[ky & 1]) << 4));
                };
                rdb(0, bf[0]);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint4 af[3];
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
                        af[ky] = wl.place((uint32_t)*reinterpret_cast<const uint16_t*>(tbl + 256 + (ky * 3 + w_d) * 64 + (g * 8 + wl.c) * 2));
                    const float4 db = *reinterpret_cast<const float4*>(tbl + 128 + (g * 8 + 4 * (mj & 1)) * 4);
                    if (g + 1 < 4) rdb(g + 1, bf[(g + 1) & 1]);
                    f32x4 c4 = f32x4{db.x, db.y, db.z, db.w};
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) c4 = mfma16<F16>(af[ky], bf[g & 1][ky], c4);
                    a4[g] = c4;
                }
            }
            if (p.stamps) { t5 = __builtin_amdgcn_s_memtime(); stp[4] += t5 - t4; }
            // SiLU, one rounding, pooled sums (the pool sees the stored activations), the two D rows through the dead ring rows
            const int d_r0 = ((2 * s + 3) % 6) * ET_ROW, d_r1 = ((2 * s + 4) % 6) * ET_ROW;       // slots of image rows 2s - 3, 2s - 2
            const int d_row = t_ry ? d_r1 : d_r0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint32_t pk0 = T16<F16>::pack2(silu_fast(a4[g][0]), silu_fast(a4[g][1]));
                const uint32_t pk1 = T16<F16>::pack2(silu_fast(a4[g][2]), silu_fast(a4[g][3]));
                psum[g][0] = T16<F16>::dot2(pk0, one_lo, psum[g][0]);
                psum[g][1] = T16<F16>::dot2(pk0, one_hi, psum[g][1]);
                psum[g][2] = T16<F16>::dot2(pk1, one_lo, psum[g][2]);
                psum[g][3] = T16<F16>::dot2(pk1, one_hi, psum[g][3]);
                *reinterpret_cast<uint2*>(et + d_row + d_lane + ((g ^ d_f) << 4)) = make_uint2(pk0, pk1);
            }
            {
                uint16_t* const drow = p.d + ((size_t)smp * 256 + (2 * s - 2) * 16) * CEXP + c0 + (lane & 3) * 8;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int px = (lane >> 2) + 16 * i, xx = lane >> 2;
                    const uint4 v = *reinterpret_cast<const uint4*>(et + (i ? d_r1 : d_r0) + (xx + 1) * 64 + (((lane & 3) ^ ((xx >> 1) & 3)) << 4));
                    *reinterpret_cast<uint4*>(drow + (size_t)px * CEXP) = v;
                }
            }
            if (p.stamps) stp[5] += __builtin_amdgcn_s_memtime() - t5;
        }
        uint64_t tp = 0;
        if (p.stamps) tp = __builtin_amdgcn_s_memtime();
        // ---- pooled means: the lanes' sums over the eight bands -> dw_mm.h's butterfly over the 32 (pixel pair, pixel) lanes
        // (dwconv3x3_mm_kernel's order), / 256; lane (pair g < 4, first pixel) stores group g's four channels of its channel half
        {
            const int xaddr = (lane ^ 32) << 2;
            float4 tot[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) tot[g] = dwmm_pool_sum4(psum[g], xaddr, 1.0f / 256.0f);
            const float4 mine = sel4(mn < 2, sel4(mn == 0, tot[0], tot[1]), sel4(mn == 2, tot[2], tot[3]));
            if (mn < 4 && ms == 0 && live) *reinterpret_cast<float4*>(p.pooled + (size_t)smp * CEXP + c0 + mn * 8 + 4 * (mj & 1)) = mine;
        }
        if (p.stamps) stp[6] += __builtin_amdgcn_s_memtime() - tp;
    }
    if (p.stamps && (p.exp & 0x10000) && tid == 0 && blockIdx.x < 1000) {       // census: every workgroup's loop start / end on the 100-MHz clock
        p.stamps[(size_t)blockIdx.x * 2] = st_r0;
        p.stamps[(size_t)blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime();
    } else if (p.stamps && blockIdx.x < 32 && lane == 0) {
        uint64_t* o = p.stamps + ((size_t)blockIdx.x * 6 + wave) * 10;       // (slots for six waves: four in use)
        o[0] = __builtin_amdgcn_s_memtime() - st_t0; o[1] = st_n;
#pragma unroll
        for (int i = 0; i < 7; ++i) o[2 + i] = stp[i];
        o[9] = __builtin_amdgcn_s_memrealtime() - st_r0;          // 100 MHz ticks of the same interval as o[0]: the clock the chip held
    }
}

// =====================================================================================
// Round 6: the same front half with the waves of a workgroup in DIFFERENT ROLES -- four waves per SIMD.
//
// mbfront16_kernel above keeps three waves per SIMD (164 registers: 56 stationary weight registers + the depthwise stage's fragments and
// sums in the same wave) and every wave walks expand -> SiLU -> taps -> SiLU in turn: its clocks say each phase takes 3-4x its issue cost
// (EXPERIMENTS.md round 5: vector ALU 52 % busy, matrix pipe 19 %). Here a workgroup is EIGHT waves on one 128-channel slice:
//   waves 0-3  PRODUCERS  (one per SIMD): the expand GEMM of a band (32 pixels x the wave's 32 channels, weights stationary in 56
//              registers, bias in 16), SiLU, one rounding, the two E rows into the channel block's ring in LDS; they also issue the
//              LDS-DMA of the next band's input tile (two buffers) and have no other vector-memory operation, so their wait for it is a
//              plain vmcnt(0) -- no counted wait anywhere in this kernel;
//   waves 4-7  CONSUMERS  (one per SIMD): depthwise 3x3 on the matrix pipe from the ring (weight fragments stationary in 48 registers:
//              the Toeplitz fragments depend on the lane and the tap, not on the data), SiLU, one rounding, pooled sums, the two D rows.
// Neither role needs the other's registers: 128 suffice and TWO such workgroups share a CU -- four waves per SIMD, two of them in an
// MFMA + SiLU stream and two in a fragment-read + MFMA + SiLU + store stream, two ticks apart. One workgroup barrier per band ("tick"):
// at tick T the producers write band T's rows while the consumers compute the output rows of band T - 2 from rows that are complete.
// The ring of a channel block: ten rows -- a zero row, eight slots (image row y lives in slot y & 7; live at a tick are the 7 rows
// 2t - 5 .. 2t + 1, the next sample's first bands among them), a second zero row (rows -1 and 16 of every sample; never written). A
// depthwise step stages its two D rows through the two rows its taps read last (the first step of a sample: through the previous
// sample's last row). The tick bodies are unrolled over the 8 bands of a sample and a lane's row choice (its fragment's pixel pair
// sits in one of two neighbouring rows) is folded into its base address, so every ring offset is an immediate -- except where the
// two rows wrap around the ring (slots 7 -> 0: two fragment reads and two staging steps per sample pay one vector add).
// Arithmetic and orders are the first kernel's (k ascending in one accumulator; tap MFMAs; lane sums over the bands in order, then the
// butterfly over the 32 lanes): bit-identical to it and to the two-launch path (tested).
// Work: units (slice, sample) of an XCD's samples in slice-major order, cut into equal contiguous ranges for the XCD's 64 workgroup
// slots (a range that crosses a slice boundary drains and reloads: at most one of the 64 per boundary).
template <int CIN>
struct Mf16r {
    static constexpr int NK16 = CIN / 16, NKT = CIN / 32;
    static constexpr int XBUF = NKT * 2048;                 // a band's input tile: [NKT][32 rows][64 B], swizzled (gemm1x1's A image)
    static constexpr int ROW = 18 * 64;                     // one padded ring row: [18 pixels][32 ch x 2 B]
    static constexpr int RING = 10 * ROW;                   // physical rows: 0 = zeros (image row -1), 1 + (y & 7) = image row y, 9 = zeros (image row 16)
    static constexpr int RING_OFF = 2 * XBUF;
    static constexpr int TBL_OFF = RING_OFF + 4 * RING;
    static constexpr int TBL_BYTES = 1024;                  // per channel block: depthwise bias [32] f32 at 128 | taps [9][32] 16-bit at 256
    static constexpr int LDS = TBL_OFF + 4 * TBL_BYTES;
    static_assert(2 * LDS <= 160 * 1024, "two workgroups per CU");
    static constexpr int slot(int y) { return y < 0 ? 0 : (y > 15 ? 9 * ROW : (1 + (y & 7)) * ROW); }
    static constexpr bool wraps(int y) { return slot(y + 1) - slot(y) != ROW; }       // rows y, y + 1 are not neighbours in the ring
};

template <int CIN, bool F16, bool STAMP = false>
__global__ __launch_bounds__(512, 4) void mbfront16r_kernel(MbFront16Args p) {
    using S = Mf16r<CIN>;
    constexpr int NK16 = S::NK16, ROW = S::ROW;
    T16<F16>::enter();
    unsigned char* const lds = conv_lds_dyn;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave < 4;
    const int cbw = wave & 3;                                          // channel block of the slice this wave works on
    const int CEXP = p.cexp, NSL = (CEXP + 127) / 128;
    // this workgroup's units: XCD x (ids 8 apart share one) owns samples [x Bx, x Bx + nx); its slots cut NSL * nx units evenly
    const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3, nslots = (int)(gridDim.x >> 3);
    const int Bx = (p.B + 7) / 8, smp0 = xcd * Bx, nx = min(Bx, p.B - smp0);
    if (nx <= 0) return;
    const int U = NSL * nx;
    // ranges differ by one unit (1344 channels at 256 frames: 5.5 units per slot, alternately 5 and 6). The slots of an XCD are dealt to
    // its 32 CUs in order, so slots k and k + 32 share a CU: give them neighbouring ranges (2 k, 2 k + 1) -- one of 5 and one of 6 units
    // per CU instead of 5 + 5 on one CU and 6 + 6 on the next (placement is the hardware's business: a speed matter only)
    const int half = (nslots + 1) >> 1;
    const int rng = slot_id < half ? 2 * slot_id : 2 * (slot_id - half) + 1;
    int u = (int)((long long)rng * U / nslots);
    const int u1 = rng < nslots ? (int)((long long)(rng + 1) * U / nslots) : u;
    if (u >= u1) return;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)lds;
    // tuning probe (STAMP instantiation only; MbFront16Args.stamps: [32 workgroups][8 waves][4] = cycles in the tick loops, of those waiting
    // at the tick barrier (arrival -> release), ticks, 100-MHz ticks of the loops)
    uint64_t st_loop = 0, st_wait = 0, st_n = 0, st_real = 0;
    unsigned char* const ring = lds + S::RING_OFF + cbw * S::RING;
    unsigned char* const tbl = lds + S::TBL_OFF + cbw * S::TBL_BYTES;

    // band (sample smp, band bnd) -> X buffer buf: piece pc = (k-tile pc >> 1, rows 16 (pc & 1) ..), 1 KiB each; issued by the producers
    auto dma_x = [&](int smp, int bnd, int buf) {
        const unsigned char* src = reinterpret_cast<const unsigned char*>(p.x) + ((size_t)smp * 256 + bnd * 32) * (CIN * 2);
        for (int pc = cbw; pc < S::NKT * 2; pc += 4) {
            const int kt = pc >> 1, row = 16 * (pc & 1) + (lane >> 2);
            const int logical = (lane & 3) ^ ((row >> 2) & 3);
            dma16_s(src, (uint32_t)(row * CIN * 2 + kt * 64 + logical * 16), lds0 + (uint32_t)(buf * S::XBUF + pc * 1024));
        }
    };
    // the tick barrier: a role's LDS writes of the tick are done (the producers: and their pieces of the next input tile have landed)
    auto tick_barrier = [&](bool prod) __attribute__((always_inline)) {
        if constexpr (STAMP) {
            if (prod) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const uint64_t tb = __builtin_amdgcn_s_memtime();
            asm volatile("s_barrier" ::: "memory");
            st_wait += __builtin_amdgcn_s_memtime() - tb;
            ++st_n;
        } else {
            if (prod) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    };
    // the rings, zero columns and zero rows included (only the interiors of rows 1-8 are ever written afterwards): once per workgroup
    for (int i = tid; i < 4 * S::RING / 16; i += 512) *reinterpret_cast<uint4*>(lds + S::RING_OFF + i * 16) = make_uint4(0, 0, 0, 0);

    // a segment: n samples of one slice (the role branch is OUTSIDE the segment loop: inside it the compiler hoists both roles' lane
    // constants above the branch and spills 47 registers)
#define ISB_MBF16R_SEGMENT                                                                                                  \
        const int slice = u / nx, i0 = u % nx, n = min(nx - i0, u1 - u);                                                      \
        u += n;                                                                                                               \
        const int sbase = smp0 + i0;                                                                                          \
        const int cb = min(slice * 4 + cbw, CEXP / 32 - 1), c0 = cb * 32;                                                     \
        const bool live = (slice * 4 + cbw) * 32 < CEXP;    /* 1344 channels = 10 slices + 64: the last slice's upper waves only keep the barriers (producers: and stage tiles) */
    if (producer) {
        while (u < u1) {
            ISB_MBF16R_SEGMENT
            // ---------------------------------------------------------------- producer
            if (p.exp & 0x1) __builtin_amdgcn_s_setprio(2);          // (A/B switch, ISB_EXP bit 0: producers outrank consumers)
            const int r = lane & 31, h = lane >> 5;
            uint4 wreg[NK16];
            {
                const uint4* src = p.w1p + (size_t)cb * NK16 * 64 + lane;
#pragma unroll
                for (int s = 0; s < NK16; ++s) wreg[s] = src[s * 64];
            }
            float4 bias[4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) bias[qq] = *reinterpret_cast<const float4*>(p.b1 + c0 + 8 * qq + 4 * h);
            const int a_sw0 = swz(r, h), a_sw1 = swz(r, 2 + h);
            // E write: lane = pixel r of the band (row r >> 4, column r & 15), 16 channels 4 h + 8 qq + i; the 16-byte chunk slots of a
            // pixel are turned by f(x) = (x >> 1) & 3 (x the padded column): the consumers' fragment reads (ds_read_b128, pixels 2 n + j of
            // two neighbouring rows) stay conflict-free and these 8-byte writes fall 2-way instead of the first kernel's 4-way
            // (f = ((x >> 2) & 1) | ((y & 1) << 1) there; SQ_LDS_BANK_CONFLICT was 41 % of the LDS cycles)
            const int e_x = (r & 15) + 1, e_rr = r >> 4;
            const int e_f = (e_x >> 1) & 3;
            unsigned char* const e_cell = ring + e_rr * ROW + e_x * 64 + h * 8;       // + the band's first slot (2 t) & 7: rows 2 t, 2 t + 1 are neighbours
            dma_x(sbase, 0, 0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            uint64_t st_t0 = 0, st_r0 = 0;
            if constexpr (STAMP) { st_t0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
#pragma unroll 1
            for (int j = 0; j < n; ++j) {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    if (t < 7) dma_x(sbase + j, t + 1, (t + 1) & 1);
                    else if (j + 1 < n) dma_x(sbase + j + 1, 0, 0);
                    if (live) {
                        const unsigned char* const xb = lds + (t & 1) * S::XBUF;
                        f32x16 acc;
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
                        // fragment reads two k16 pairs (four MFMAs, 128 matrix cycles) ahead of their use, in three register sets: an LDS round
                        // trip under load is longer than the two MFMAs one set ahead covers
                        uint4 fa[3][2];
                        auto rd = [&](int pr2, uint4 (&f)[2]) __attribute__((always_inline)) {
                            f[0] = *reinterpret_cast<const uint4*>(xb + pr2 * 2048 + a_sw0);
                            f[1] = *reinterpret_cast<const uint4*>(xb + pr2 * 2048 + a_sw1);
                        };
                        rd(0, fa[0]);
                        rd(1, fa[1]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int pr2 = 0; pr2 < NK16 / 2; ++pr2) {
                            if (pr2 + 2 < NK16 / 2) rd(pr2 + 2, fa[(pr2 + 2) % 3]);
                            acc = T16<F16>::mfma32(wreg[2 * pr2], fa[pr2 % 3][0], acc);
                            acc = T16<F16>::mfma32(wreg[2 * pr2 + 1], fa[pr2 % 3][1], acc);
                            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        // E = T16(silu(acc + bias)) -> ring rows 2 t, 2 t + 1
                        unsigned char* const cell = e_cell + S::slot(2 * t);
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) {
                            const f32x2_t v01 = silu_fast2(f32x2_t{acc[4 * qq], acc[4 * qq + 1]} + f32x2_t{bias[qq].x, bias[qq].y});
                            const f32x2_t v23 = silu_fast2(f32x2_t{acc[4 * qq + 2], acc[4 * qq + 3]} + f32x2_t{bias[qq].z, bias[qq].w});
                            uint2 pk;
                            pk.x = T16<F16>::pack2(v01.x, v01.y);
                            pk.y = T16<F16>::pack2(v23.x, v23.y);
                            *reinterpret_cast<uint2*>(cell + ((qq ^ e_f) << 4)) = pk;
                        }
                    }
                    // the next band's tile has landed (this wave's pieces; the barrier joins the other producers'), the E rows are written
                    tick_barrier(true);
                }
            }
            tick_barrier(true);                                        // the consumers' last two steps
            tick_barrier(true);
            if constexpr (STAMP) { st_loop += __builtin_amdgcn_s_memtime() - st_t0; st_real += __builtin_amdgcn_s_memrealtime() - st_r0; }
        }
    } else {
        while (u < u1) {
            ISB_MBF16R_SEGMENT
            // ---------------------------------------------------------------- consumer
            // the consumers' tick is the longer one (fragment reads -> 12 MFMAs -> SiLU -> D rows: stamps say the producers wait 31 % of a tick at
            // the barrier with equal priorities); served first they stop being the pole: 118 -> 115 us per 256 frames (ISB_EXP bit 1: off)
            if (!(p.exp & 0x2)) __builtin_amdgcn_s_setprio(1);
            // the block's depthwise bias and taps in the channel block's LDS table, then the weight fragments for the whole segment
            if (lane < 8) *reinterpret_cast<float4*>(tbl + 128 + lane * 16) = *reinterpret_cast<const float4*>(p.dwb + c0 + lane * 4);
            else if (lane >= 16 && lane < 16 + 36) {
                const int t = (lane - 16) >> 2, c4 = (lane - 16) & 3;
                *reinterpret_cast<uint4*>(tbl + 256 + t * 64 + c4 * 16) = *reinterpret_cast<const uint4*>(p.dww + (size_t)t * CEXP + c0 + c4 * 8);
            }
            const DwmmLane wl(lane);
            const int w_d = min(max(wl.d, 0), 2);
            const int mn = lane & 15, mj = lane >> 4, ms = mj >> 1;    // pixel pair, input column / output rows, pixel of the pair
            uint4 af[4][3];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
                    af[g][ky] = wl.place((uint32_t)*reinterpret_cast<const uint16_t*>(tbl + 256 + (ky * 3 + w_d) * 64 + (g * 8 + wl.c) * 2));
            uint32_t one_lo, one_hi;                                    // (1, 0) / (0, 1) pairs in the storage type
            if constexpr (F16) asm volatile("v_mov_b32 %0, 0x3c00\n\tv_lshlrev_b32 %1, 16, %0" : "=v"(one_lo), "=v"(one_hi));
            else asm volatile("v_mov_b32 %0, 0x3f80\n\tv_lshlrev_b32 %1, 16, %0" : "=v"(one_lo), "=v"(one_hi));
            // B fragment: output row ry of the step's two, pixel pair; input row index ry + ky of the four rows 2s - 3 .. 2s
            const int t_ry = mn >> 3, t_x = 2 * (mn & 7) + mj;         // padded column of the fragment's pixel
            // the lane's fragment address for group g, its row choice (t_ry) folded in: + the offset of image row 2s - 3 + ky
            unsigned char* t_base[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) t_base[g] = ring + t_ry * ROW + t_x * 64 + ((g ^ ((t_x >> 1) & 3)) << 4);
            const int wrap_fix = -t_ry * 8 * ROW;                       // where rows y, y + 1 wrap around the ring: the second one is 7 rows back, not 1 on
            // D staging: the step's two output rows leave through the interiors of the two ring rows its taps read LAST
            const int d_x = 2 * (mn & 7) + ms;
            // (the second row's pixels sit pairwise swapped, x ^ 1: the 8-byte writes of the two rows then fall into different halves of the
            // 128-byte bank window, 2-way instead of 4-way; the 16-byte readback stays conflict-free)
            unsigned char* d_base[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) d_base[g] = ring + t_ry * ROW + ((d_x ^ t_ry) + 1) * 64 + (mj & 1) * 8 + ((g ^ ((d_x >> 1) & 3)) << 4);
            const int o_sw = ((lane & 3) ^ (((lane >> 2) >> 1) & 3)) << 4;
            unsigned char* const o_base[2] = {ring + ((lane >> 2) + 1) * 64 + o_sw, ring + (((lane >> 2) ^ 1) + 1) * 64 + o_sw};   // the D rows as 16-byte pieces
            const float* const dwb_l = reinterpret_cast<const float*>(tbl + 128) + 4 * (mj & 1);
            float psum[4][4];
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            uint64_t st_t0 = 0, st_r0 = 0;
            if constexpr (STAMP) { st_t0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
            tick_barrier(false);                                        // the producers' first two bands
            tick_barrier(false);
#pragma unroll 1
            for (int j = 0; j < n; ++j) {
                const int smp = sbase + j;
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int i = 0; i < 4; ++i) psum[g][i] = 0.f;
#pragma unroll
                for (int s = 1; s <= 8; ++s) {                          // output rows 2s - 2, 2s - 1 from ring rows 2s - 3 .. 2s
                    if (live) {
                        f32x4 a4[4];
                        {
                            uint4 bf[2][3];
                            auto rdb = [&](int g, uint4 (&f)[3]) __attribute__((always_inline)) {
#pragma unroll
                                for (int ky = 0; ky < 3; ++ky) {
                                    // the lane's input row 2s - 3 + ry + ky
                                    const unsigned char* a = t_base[g] + S::slot(2 * s - 3 + ky);
                                    if (S::wraps(2 * s - 3 + ky)) a += wrap_fix;
                                    f[ky] = *reinterpret_cast<const uint4*>(a);
                                }
                            };
                            rdb(0, bf[0]);
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                f32x4 c4 = *reinterpret_cast<const f32x4*>(dwb_l + g * 8);       // the bias: the chain's C operand
                                if (g + 1 < 4) rdb(g + 1, bf[(g + 1) & 1]);
#pragma unroll
                                for (int ky = 0; ky < 3; ++ky) c4 = mfma16<F16>(af[g][ky], bf[g & 1][ky], c4);
                                a4[g] = c4;
                            }
                        }
                        // SiLU, one rounding, pooled sums (the pool sees the stored activations), the two D rows through the dead ring rows
                        const int d_y0 = s == 1 ? 15 : 2 * s - 3;           // (step 1: through the slot of the previous sample's last row, dead since its step 8)
                        const int d_r0 = S::slot(d_y0), d_r1 = S::slot(2 * s - 2);       // staging rows of output rows 2s - 2, 2s - 1
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x2_t v01 = silu_fast2(f32x2_t{a4[g][0], a4[g][1]}), v23 = silu_fast2(f32x2_t{a4[g][2], a4[g][3]});
                            const uint32_t pk0 = T16<F16>::pack2(v01.x, v01.y);
                            const uint32_t pk1 = T16<F16>::pack2(v23.x, v23.y);
                            psum[g][0] = T16<F16>::dot2(pk0, one_lo, psum[g][0]);
                            psum[g][1] = T16<F16>::dot2(pk0, one_hi, psum[g][1]);
                            psum[g][2] = T16<F16>::dot2(pk1, one_lo, psum[g][2]);
                            psum[g][3] = T16<F16>::dot2(pk1, one_hi, psum[g][3]);
                            unsigned char* da = d_base[g] + d_r0;
                            if (d_r1 - d_r0 != ROW) da += wrap_fix;
                            *reinterpret_cast<uint2*>(da) = make_uint2(pk0, pk1);
                        }
                        {
                            uint16_t* const drow = p.d + ((size_t)smp * 256 + (2 * s - 2) * 16) * CEXP + c0 + (lane & 3) * 8;
#pragma unroll
                            for (int i = 0; i < 2; ++i) {
                                const int px = (lane >> 2) + 16 * i;
                                const uint4 v = *reinterpret_cast<const uint4*>(o_base[i] + (i ? d_r1 : d_r0));
                                *reinterpret_cast<uint4*>(drow + (size_t)px * CEXP) = v;
                            }
                        }
                        if (s == 8) {
                            // pooled means: the lanes' sums over the eight bands -> dw_mm.h's butterfly over the 32 (pixel pair, pixel) lanes,
                            // / 256; lane (pair g < 4, first pixel) stores group g's four channels of its channel half
                            const int xaddr = (lane ^ 32) << 2;
                            float4 tot[4];
#pragma unroll
                            for (int g = 0; g < 4; ++g) tot[g] = dwmm_pool_sum4(psum[g], xaddr, 1.0f / 256.0f);
                            const float4 mine = sel4(mn < 2, sel4(mn == 0, tot[0], tot[1]), sel4(mn == 2, tot[2], tot[3]));
                            if (mn < 4 && ms == 0) *reinterpret_cast<float4*>(p.pooled + (size_t)smp * CEXP + c0 + mn * 8 + 4 * (mj & 1)) = mine;
                        }
                    }
                    tick_barrier(false);
                }
            }
            if constexpr (STAMP) { st_loop += __builtin_amdgcn_s_memtime() - st_t0; st_real += __builtin_amdgcn_s_memrealtime() - st_r0; }
        }
    }
#undef ISB_MBF16R_SEGMENT
    if constexpr (STAMP) {
        if (p.stamps && blockIdx.x < 32 && lane == 0) {
            uint64_t* o = p.stamps + ((size_t)blockIdx.x * 8 + wave) * 4;
            o[0] = st_loop; o[1] = st_wait; o[2] = st_n; o[3] = st_real;
        }
    }
}

int launch_mbfront16(const MbFront16Args& a, hipStream_t st) {
    if (a.B < 1 || !a.x || !a.w1p || !a.b1 || !a.dww || !a.dwb || !a.d || !a.pooled || (a.cin != 192 && a.cin != 224) || a.cexp % 32 != 0 ||
        a.cexp < 128) {
        set_error("mbfront16: bad arguments (B=%d cin=%d cexp=%d; built for 192 / 224 inputs)", a.B, a.cin, a.cexp);
        return ISB_ERR_INVALID;
    }
    const int nsl = cdiv(a.cexp, 128);
    const int Qx = std::max(1, std::min(cdiv(a.B, 8), 768 / (8 * nsl)));     // sample sequences per XCD: three workgroups per CU
    MbFront16Args aa = a;
    aa.exp = exp_flags();
    // round 6: producer / consumer waves, four waves per SIMD (mbfront16r_kernel). a.form: 0 = the library's choice, 1 = the first
    // kernel (every wave does everything, three waves per SIMD), 2 = roles. ISB_MBF16_FORM overrides the choice (A/B runs, tests).
    static const int env_form = [] { const char* e = getenv("ISB_MBF16_FORM"); return e ? atoi(e) : 0; }();
    const int form = a.form ? a.form : (env_form ? env_form : 2);
    if (form == 2 && a.stamps) {
        if (!(a.cin == 224 && a.f16)) { set_error("mbfront16r: the stamped instantiation is <224, fp16>"); return ISB_ERR_INVALID; }
        const int nslots = std::min(64, nsl * cdiv(a.B, 8));
        static DevOnce attr_set;
        if (attr_set.need()) {
            ISB_HIP(hipFuncSetAttribute((const void*)mbfront16r_kernel<224, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Mf16r<224>::LDS));
            attr_set.mark();
        }
        hipLaunchKernelGGL((mbfront16r_kernel<224, true, true>), dim3(8 * nslots), dim3(512), Mf16r<224>::LDS, st, aa);
        ISB_LAUNCHED("mbfront16r (stamped)", st);
        return ISB_OK;
    }
    if (form == 2) {
        const int nslots = std::min(64, nsl * cdiv(a.B, 8));
#define ISB_MBF16R(CIN_, F16_)                                                                                               \
    do {                                                                                                                     \
        static DevOnce attr_set;                                                                                             \
        if (attr_set.need()) {                                                                                               \
            ISB_HIP(hipFuncSetAttribute((const void*)mbfront16r_kernel<CIN_, F16_>, hipFuncAttributeMaxDynamicSharedMemorySize, Mf16r<CIN_>::LDS)); \
            attr_set.mark();                                                                                                 \
            if (getenv("ISB_OCC")) {                                                                                         \
                int nb = -1;                                                                                                 \
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)mbfront16r_kernel<CIN_, F16_>, 512, Mf16r<CIN_>::LDS); \
                fprintf(stderr, "[isb] mbfront16r<%d>: %d workgroups per CU by the occupancy API (LDS %d B)\n", CIN_, nb, Mf16r<CIN_>::LDS); \
            }                                                                                                                \
        }                                                                                                                    \
        hipLaunchKernelGGL((mbfront16r_kernel<CIN_, F16_>), dim3(8 * nslots), dim3(512), Mf16r<CIN_>::LDS, st, aa);          \
    } while (0)
        if (a.cin == 224) { if (a.f16) ISB_MBF16R(224, true); else ISB_MBF16R(224, false); }
        else { if (a.f16) ISB_MBF16R(192, true); else ISB_MBF16R(192, false); }
#undef ISB_MBF16R
        ISB_LAUNCHED("mbfront16r", st);
        return ISB_OK;
    }
#define ISB_MBF16(CIN_, F16_)                                                                                                \
    do {                                                                                                                     \
        static DevOnce attr_set;                                                                                             \
        if (attr_set.need()) {                                                                                               \
            ISB_HIP(hipFuncSetAttribute((const void*)mbfront16_kernel<CIN_, F16_>, hipFuncAttributeMaxDynamicSharedMemorySize, Mf16<CIN_>::LDS)); \
            attr_set.mark();                                                                                                 \
            if (getenv("ISB_OCC")) {                                                                                         \
                int nb = -1;                                                                                                 \
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)mbfront16_kernel<CIN_, F16_>, 256, Mf16<CIN_>::LDS); \
                fprintf(stderr, "[isb] mbfront16<%d>: %d workgroups per CU by the occupancy API (LDS %d B)\n", CIN_, nb, Mf16<CIN_>::LDS);  \
            }                                                                                                                \
        }                                                                                                                    \
        hipLaunchKernelGGL((mbfront16_kernel<CIN_, F16_>), dim3(8 * nsl * Qx), dim3(256), Mf16<CIN_>::LDS, st, aa);                \
    } while (0)
    if (a.cin == 224) { if (a.f16) ISB_MBF16(224, true); else ISB_MBF16(224, false); }
    else { if (a.f16) ISB_MBF16(192, true); else ISB_MBF16(192, false); }
#undef ISB_MBF16
    ISB_LAUNCHED("mbfront16", st);
    return ISB_OK;
}

}  // namespace isb
