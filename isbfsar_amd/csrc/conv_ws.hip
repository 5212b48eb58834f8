// The weights-stationary GEMMs of the MBConv expand convolutions (every instantiation is a fully unrolled tile pass).
// launch_conv_ws is called by launch_conv_igemm for tile variants 184 - 186 (181 - 183, 187, 188: -DISB_BUILD_PROBES builds).
#include "conv_common.h"

namespace isb {

// -------------------------------------------------------------------------------------------
// Weights-stationary 1x1 GEMM for the MBConv expand convolutions with a SHORT K (Cin = 96 / 192 / 224 / 384: 3 - 12
// k-steps). What the tile kernels above pay for on these layers is not arithmetic: a 128 x 192 tile has 42 MFMAs per
// wave (K = 224), and around them a prologue (first-tile latency), 20 KiB of operand DMA + 64 KiB of LDS fragment reads
// per k-step (three quarters of both are WEIGHTS, re-fetched by every M tile), a barrier per k-step, and an epilogue of
// ~1600 vector-issue cycles per wave (SiLU: two quarter-rate transcendentals per element) during which the matrix pipes
// idle. Measured (round 2, 224 -> 1344 at 256 frames): 75 us = 529 TFLOP/s, 21 % of peak.
// Here ONE persistent workgroup of 6 waves per CU owns a slice of 192 output channels for its whole life:
//   * wave w keeps the weights of its 32 channels for ALL of K in registers as MFMA A-operand fragments (K / 4 VGPRs):
//     the weights never touch LDS -- no weight DMA, no weight fragment reads, no weight re-fetch per tile;
//   * the activations arrive as whole 96-row M tiles (96 x K bf16, 18 - 72 KiB) in two LDS buffers by LDS-DMA: tile j + 1
//     travels while tile j is consumed, so there is no prologue per tile and ONE barrier per tile (A is re-read
//     Cout / 192 times in all, out of the XCD's L2: neighbouring workgroup ids walk the same tile sequence);
//   * between two barriers a wave runs its 6 NK MFMAs back to back (6 ds_read_b128 per k-step) and its epilogue (bias,
//     SiLU, one bf16 rounding, wave-local staging of a 32 x 32 block, 16-byte stores) with no synchronisation at all;
//   * the two waves that share a SIMD (w and w + 4: waves go to the SIMDs in the order 0, 2, 1, 3, 0, 2) run these two
//     phases in OPPOSITE order -- waves 0-3 multiply tile j and then finish it, waves 4-5 first finish tile j - 1 and then
//     multiply tile j -- so one wave's vector work (the SiLU epilogue) runs beside the other's matrix work on the same
//     SIMD. (In-kernel s_memtime stamps of the first, per-k-step-barrier form: k loop 4900 cycles, epilogue 2650, and
//     1850 more at the next barrier waiting for the slowest epilogue: every wave of the CU sat in the same phase.)
// LDS-DMA completion is tracked with a counted s_waitcnt vmcnt: per tile a wave issues its NK pieces of the NEXT tile and
// then, in either phase order, 6 stores; when the next tile is needed exactly those 6 stores are younger: vmcnt(6).
// Stores are never masked (rows past M are clamped onto row M - 1, whose values they repeat) and waves 4-5 issue six
// dummy pieces where the stores of the tile "before the first" would sit, so the count is exact from the first tile on.
// Sums run in the k order of gemm1x1_dma_kernel and bias / SiLU / rounding are the same code: bit-identical results.
// -------------------------------------------------------------------------------------------
constexpr int WS_SROW = 80;                              // staging row: 64 B + 16 (8-byte writes spread over the banks)
constexpr int WS_STAGE = 32 * WS_SROW;
// NW waves per workgroup (each owns 32 output channels: slice = 32 NW), TMB 32-row blocks per M tile (BM = 32 TMB rows,
// a multiple of the 16 NW rows one round of DMA pieces covers)
constexpr int ws_lds_bytes(int nk, int nw, int tmb) { return 2 * nk * (32 * tmb * ROWB) + nw * WS_STAGE + 1024; }

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

#ifdef ISB_BUILD_PROBES
// Two shapes are instantiated:
//   NW = 6, TMB = 3 (variant 181): 96 x 192 tiles, six waves on four SIMDs, the SIMD partners (waves w and w + 4) run
//                   multiply / finish in opposite order;
//   NW = 4, TMB = 4 (variant 182): 128 x 128 tiles, ONE wave per SIMD: nothing on a SIMD competes with the wave, its
//                   multiply phase runs at the matrix pipe's rate and its finish phase at the vector ALU's.
template <int NK, bool ACT, bool STAMPS = false, int NW = 6, int TMB = 3>
__global__ __launch_bounds__(64 * NW, NW == 6 ? 2 : 1) void gemm1x1_wsreg_kernel(ConvArgs p) {
    constexpr int K = 32 * NK;
    constexpr int BM = 32 * TMB, BN = 32 * NW;
    constexpr int CHUNK = BM * ROWB;                     // one k-step of an M tile
    constexpr int TILE = NK * CHUNK;                     // one M tile, chunk (k-step) major
    constexpr int PIECES = BM / (16 * NW);               // 1-KiB DMA pieces per wave and k-step
    constexpr int NS = 2 * TMB;                          // global stores per wave and tile
    static_assert(BM % (16 * NW) == 0, "a round of DMA pieces covers 16 NW rows");
    constexpr int STAGE_OFF = 2 * TILE, DUMP_OFF = STAGE_OFF + NW * WS_STAGE;
    unsigned char* const lds = conv_lds_dyn;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // workgroup -> (sequence of M tiles, slice of 32 NW channels). Workgroup ids go round-robin over the XCDs, so id % 8
    // names an XCD; in XCD-major order consecutive workgroups are the slices of one sequence: they read the same
    // activation rows at the same time, out of that XCD's L2 (placement is for speed only)
    const int nsl = p.grid_n, Q = p.grid_m;              // slices; tile sequences
    const int g = blockIdx.x, idx = (g & 7) * (gridDim.x >> 3) + (g >> 3);
    const int q = idx / nsl, slice = idx - q * nsl;
    const int n_mt = (p.M + BM - 1) / BM;
    if (q >= Q || q >= n_mt) return;
    const int nw0 = slice * BN + 32 * wave;
    const bool live = nw0 < p.Cout;                      // the last slice may be narrower than 32 NW channels: such a wave
                                                         // only moves its share of the activations and keeps the barriers
    const bool late = NW == 6 && wave >= 4;              // the SIMD partners of waves 0 and 1: epilogue first, then multiply

    // the wave's weights, all of K, as A-operand fragments: lane (r, h) holds channel nw0 + r, k = 16 ks + 8 h .. + 7
    bf16x8 bfr[2 * NK];
    float4 bias4[4];
    {
        const int nrow = live ? nw0 : 0;
        const uint16_t* wrow = p.w + (size_t)(nrow + r) * K + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 2 * NK; ++ks) bfr[ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(wrow + 16 * ks));
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) bias4[qq] = *reinterpret_cast<const float4*>(p.bias + nrow + 8 * qq + 4 * h);
    }
    // the compiler's own waits for these ordinary loads must happen HERE, before any asm DMA is in flight (it does not
    // see them in its vmcnt model)
#pragma unroll
    for (int ks = 0; ks < 2 * NK; ++ks) asm volatile("" ::"v"(bfr[ks]));
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) asm volatile("" ::"v"(bias4[qq].x), "v"(bias4[qq].y), "v"(bias4[qq].z), "v"(bias4[qq].w));

    // activation pieces: in round j of a k-step wave w fills rows 16 (NW j + w) .. + 15 of the chunk; lane -> (row, swizzled chunk)
    const uint32_t lds_a = (uint32_t)(uintptr_t)(lds_ptr_t)lds;
    const unsigned char* a_bytes = reinterpret_cast<const unsigned char*>(p.in);
    auto issue_tile = [&](int tile, int buf) {  // the wave's pieces of an M tile (past the last tile: row M - 1, never used)
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            const int rowl = 16 * (NW * j + wave) + (lane >> 2);
            const uint32_t lchunk = (uint32_t)(((lane & 3) ^ ((rowl >> 2) & 3)) * 16);
            const uint32_t voff = (uint32_t)min(tile * BM + rowl, p.M - 1) * (uint32_t)(K * 2) + lchunk;
#pragma unroll
            for (int s = 0; s < NK; ++s) dma16_s(a_bytes + s * 64, voff, lds_a + (buf * TILE + s * CHUNK + (NW * j + wave) * 1024));
        }
    };
    unsigned char* const stage = lds + STAGE_OFF + wave * WS_STAGE;
    uint16_t* const out16 = reinterpret_cast<uint16_t*>(p.out);
    const int a_sw0 = swz(r, h), a_sw1 = swz(r, 2 + h);
    f32x16 acc[TMB];

    auto multiply = [&](int buf) {              // acc = tile (in LDS buffer buf) x the wave's weights
#pragma unroll
        for (int i = 0; i < TMB; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        const unsigned char* At = lds + buf * TILE;
        // the fragments of k-step s + 1 are requested BEFORE the MFMAs of k-step s (two register sets): in-kernel clocks of
        // the read-then-multiply order showed 65 cycles per MFMA for a wave alone on its SIMD -- after the barrier every wave
        // of the CU reads its 8 KiB at once, and each k-step waited for that burst before its first MFMA
        bf16x8 af[2][2][TMB];
        auto load_frags = [&](int set, int s2) {
#pragma unroll
            for (int i = 0; i < TMB; ++i) {
                af[set][0][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(At + s2 * CHUNK + a_sw0 + i * 2048));
                af[set][1][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(At + s2 * CHUNK + a_sw1 + i * 2048));
            }
        };
        load_frags(0, 0);
#pragma unroll
        for (int s = 0; s < NK; ++s) {
            if (s + 1 < NK) load_frags((s + 1) & 1, s + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < TMB; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[2 * s + ks], af[s & 1][ks][i], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // epilogue, wave-local. acc[i][e]: pixel 32 i + r of the tile, channel nw0 + 8 (e >> 2) + 4 h + (e & 3)
    auto finish = [&](int tile) {
        const int m0 = tile * BM;
#pragma unroll
        for (int i = 0; i < TMB; ++i) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                float v0 = acc[i][4 * qq] + bias4[qq].x, v1 = acc[i][4 * qq + 1] + bias4[qq].y;
                float v2 = acc[i][4 * qq + 2] + bias4[qq].z, v3 = acc[i][4 * qq + 3] + bias4[qq].w;
                if constexpr (ACT) { v0 = silu_fast(v0); v1 = silu_fast(v1); v2 = silu_fast(v2); v3 = silu_fast(v3); }
                uint2 pk;           // two v_cvt_pk_bf16_f32 (the same round-to-nearest-even as the scalar casts elsewhere)
                pk.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{v0, v1}, bf16x2_t));
                pk.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{v2, v3}, bf16x2_t));
                *reinterpret_cast<uint2*>(stage + r * WS_SROW + qq * 16 + h * 8) = pk;
            }
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const int row = 16 * k2 + (lane >> 2), cc = lane & 3;
                const uint4 v = *reinterpret_cast<const uint4*>(stage + row * WS_SROW + cc * 16);
                const int m = min(m0 + 32 * i + row, p.M - 1);         // rows past M repeat row M - 1: same bytes, same address
                *reinterpret_cast<uint4*>(out16 + (size_t)m * p.Cout + nw0 + cc * 8) = v;
            }
        }
    };
    auto dummy_stores = [&]() {                 // queue entries where a tile's NS stores would sit (keeps vmcnt counts exact)
#pragma unroll
        for (int i = 0; i < NS; ++i) dma16_s(p.bias, (uint32_t)(lane & 7) * 16u, lds_a + DUMP_OFF);
    };
    // probe: waves 0 and NW - 2 of the first 64 workgroups stamp s_memtime into the dump KiB for tiles 1..4:
    // 8 slots per tile and wave: 0 tile landed, 1 barrier passed, 2 first phase done, 3 second phase done
    const bool stamp_wg = STAMPS && (wave == 0 || wave == NW - 2) && g < 64;
    int tile_no = 0;
    auto stamp = [&](int slot) {
        if constexpr (!STAMPS) return;
        if (stamp_wg && tile_no >= 1 && tile_no <= 4) {
            const uint64_t tnow = __builtin_amdgcn_s_memtime();
            if (lane == 0) *reinterpret_cast<uint64_t*>(lds + DUMP_OFF + (((tile_no - 1) * 2 + (wave != 0)) * 8 + slot) * 8) = tnow;
        }
    };

    // tuning probes: static priority for one half, or a raised priority while a wave is in its vector phase
    if ((p.probe & 4) && late) __builtin_amdgcn_s_setprio(1);
    if ((p.probe & 8) && !late) __builtin_amdgcn_s_setprio(1);
    const bool dyn_prio = (p.probe & 16) != 0;
    issue_tile(q, 0);
    if (late) dummy_stores();                   // the stores of the tile "before the first"
    int buf = 0;
    for (int t = q; t < n_mt; t += Q, buf ^= 1, ++tile_no) {
        if (t == q && !late) wait_vm<0>();      // first tile of the early waves: nothing younger than its pieces
        else wait_vm<NS>();                     // the NS stores (or dummies) issued after the tile's pieces may still fly
        stamp(0);
        __builtin_amdgcn_s_barrier();           // everybody's pieces have landed; everybody is done with the other buffer
        stamp(1);
        issue_tile(t + Q, buf ^ 1);
        if (!live) {
            dummy_stores();
        } else if (!late) {
            multiply(buf);
            stamp(2);
            if (dyn_prio) __builtin_amdgcn_s_setprio(2);
            finish(t);
            if (dyn_prio) __builtin_amdgcn_s_setprio(0);
            stamp(3);
        } else {
            if (dyn_prio) __builtin_amdgcn_s_setprio(2);
            if (t != q) finish(t - Q);
            if (dyn_prio) __builtin_amdgcn_s_setprio(0);
            stamp(2);
            multiply(buf);
            stamp(3);
        }
    }
    if (late && live) finish(n_mt - 1 - (n_mt - 1 - q) % Q);   // the last tile of this sequence
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the run-ahead pieces still target this workgroup's LDS
    if (STAMPS && stamp_wg) {
        __builtin_amdgcn_s_waitcnt(0);
        if (wave == 0) {
            const uint4 v = *reinterpret_cast<const uint4*>(lds + DUMP_OFF + lane * 16);
            *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(p.part) + (size_t)g * 1024 + lane * 16) = v;
        }
    }
}

#endif  // ISB_BUILD_PROBES

// -------------------------------------------------------------------------------------------
// Weights-stationary GEMM, one wave per SIMD, epilogue software-pipelined into the next tile's MFMA stream (variant 183).
// What the stamps of variants 181 / 182 showed: (i) two waves on a SIMD do not overlap matrix and vector phases -- a wave
// streaming MFMAs back to back keeps the SIMD's issue port and its partner's SiLU epilogue crawls (5 700 cycles beside it,
// 2 250 alone); (ii) a wave ALONE on its SIMD multiplies at the pipe's rate (1 790 cycles for 56 MFMAs) once its fragment
// reads run one k-step ahead, but then spends 3 100 cycles in its own epilogue (a lone wave issues one vector instruction
// per 4 cycles) and 1 260 issuing fourteen 1-KiB LDS-DMA pieces. So here the ONE wave per SIMD does everything itself, in
// the order the hardware can overlap: between two MFMAs of tile j (32 cycles of matrix pipe, 8 of them issue) sit the
// vector instructions that finish tile j - 1 (a second set of accumulators), placed with sched_group_barrier; the next
// tile's activations are requested as ordinary 16-byte global loads right after the barrier (registers: a lone wave has
// 512), and written to the other LDS buffer after the MFMAs -- no LDS-DMA issue cost and no hand-counted vmcnt: the
// compiler's own waits are exact.
//   workgroup = 4 waves (one per SIMD) x 32 channels = a 128-channel slice, 128-row tiles, one workgroup per CU.
// Sums run in the k order of gemm1x1_dma_kernel and bias / SiLU / rounding are the same code: bit-identical results.
// -------------------------------------------------------------------------------------------
// The staging registers of gemm1x1_wspipe_kernel, named literally: a[200:255] (set 0: one wave per SIMD, fourteen pieces) or
// v[228:255] (set 1: two waves per SIMD, seven pieces; no accumulation registers at all there, so that the MFMAs accumulate in vector registers and the epilogue reads them without copies). A load that is still in flight must never be copied or moved by the
// register allocator, so these registers are kept out of its sight: the compiler sees them only as clobbers of the requests
// (tests/test_abi.py checks in the built library that no other instruction of these kernels names them).
template <int SET, int X, int OFF>
__device__ __forceinline__ void wsp_request(const void* src) {
    static_assert(X >= 0 && X < (SET == 0 ? 14 : 7), "staging pieces");
    if constexpr (SET == 0 && X == 0) asm volatile("global_load_dwordx4 a[200:203], %0, off offset:%1" :: "v"(src), "n"(OFF) : "a200", "a201", "a202", "a203");
    if constexpr (SET == 0 && X == 1) asm volatile("global_load_dwordx4 a[204:207], %0, off offset:%1" :: "v"(src), "n"(OFF) : "a204", "a205", "a206", "a207");
    if constexpr (SET == 0 && X == 2) asm volatile("global_load_dwordx4 a[208:211], %0, off offset:%1" :: "v"(src), "n"(OFF) : "a208", "a209", "a210", "a211");
    if constexpr (SET == 0 && X == 3) asm volatile("global_load_dwordx4 a[212:215], %0, off offset:%1" :: "v"(src), "n"(OFF) : "a212", "a213", "a214", "a215");
    if constexpr (SET == 0 && X == 4) asm volatile("global_load_dwordx4 a[216:219], %0, off offset:%1" :: "v"(src), "n"(OFF) : "a216", "a217", "a218", "a219");
    if constexpr (SET == 0 && X == 5) asm volatile("global_load_dwordx4 a[220:223], %0, off offset:%1" :: "v"(src), "n"(OFF) : "a220", "a221", "a222", "a223");
    if constexpr (SET == 0 && X == 6) asm volatile("global_load_dwordx4 a[224:227], %0, off offset:%1" :: "v"(src), "n"(OFF) : "a224", "a225", "a226", "a227");
    if constexpr (SET == 0 && X == 7) asm volatile("global_load_dwordx4 a[228:231], %0, off offset:%1" :: "v"(src), "n"(OFF) : "a228", "a229", "a230", "a231");
    if constexpr (SET == 0 && X == 8) asm volatile("global_load_dwordx4 a[232:235], %0, off offset:%1" :: "v"(src), "n"(OFF) : "a232", "a233", "a234", "a235");
    if constexpr (SET == 0 && X == 9) asm volatile("global_load_dwordx4 a[236:239], %0, off offset:%1" :: "v"(src), "n"(OFF) : "a236", "a237", "a238", "a239");
    if constexpr (SET == 0 && X == 10) asm volatile("global_load_dwordx4 a[240:243], %0, off offset:%1" :: "v"(src), "n"(OFF) : "a240", "a241", "a242", "a243");
    if constexpr (SET == 0 && X == 11) asm volatile("global_load_dwordx4 a[244:247], %0, off offset:%1" :: "v"(src), "n"(OFF) : "a244", "a245", "a246", "a247");
    if constexpr (SET == 0 && X == 12) asm volatile("global_load_dwordx4 a[248:251], %0, off offset:%1" :: "v"(src), "n"(OFF) : "a248", "a249", "a250", "a251");
    if constexpr (SET == 0 && X == 13) asm volatile("global_load_dwordx4 a[252:255], %0, off offset:%1" :: "v"(src), "n"(OFF) : "a252", "a253", "a254", "a255");
    if constexpr (SET == 1 && X == 0) asm volatile("global_load_dwordx4 v[228:231], %0, off offset:%1" :: "v"(src), "n"(OFF) : "v228", "v229", "v230", "v231");
    if constexpr (SET == 1 && X == 1) asm volatile("global_load_dwordx4 v[232:235], %0, off offset:%1" :: "v"(src), "n"(OFF) : "v232", "v233", "v234", "v235");
    if constexpr (SET == 1 && X == 2) asm volatile("global_load_dwordx4 v[236:239], %0, off offset:%1" :: "v"(src), "n"(OFF) : "v236", "v237", "v238", "v239");
    if constexpr (SET == 1 && X == 3) asm volatile("global_load_dwordx4 v[240:243], %0, off offset:%1" :: "v"(src), "n"(OFF) : "v240", "v241", "v242", "v243");
    if constexpr (SET == 1 && X == 4) asm volatile("global_load_dwordx4 v[244:247], %0, off offset:%1" :: "v"(src), "n"(OFF) : "v244", "v245", "v246", "v247");
    if constexpr (SET == 1 && X == 5) asm volatile("global_load_dwordx4 v[248:251], %0, off offset:%1" :: "v"(src), "n"(OFF) : "v248", "v249", "v250", "v251");
    if constexpr (SET == 1 && X == 6) asm volatile("global_load_dwordx4 v[252:255], %0, off offset:%1" :: "v"(src), "n"(OFF) : "v252", "v253", "v254", "v255");
}
template <int SET, int X, int OFF, int WAITN>
__device__ __forceinline__ void wsp_to_lds(uint32_t lds_addr) {
    static_assert(X >= 0 && X < (SET == 0 ? 14 : 7), "staging pieces");
    if constexpr (SET == 0 && X == 0) asm volatile("s_waitcnt vmcnt(%1)\n\tds_write_b128 %0, a[200:203] offset:%2" :: "v"(lds_addr), "n"(WAITN), "n"(OFF) : "memory");
    if constexpr (SET == 0 && X == 1) asm volatile("s_waitcnt vmcnt(%1)\n\tds_write_b128 %0, a[204:207] offset:%2" :: "v"(lds_addr), "n"(WAITN), "n"(OFF) : "memory");
    if constexpr (SET == 0 && X == 2) asm volatile("s_waitcnt vmcnt(%1)\n\tds_write_b128 %0, a[208:211] offset:%2" :: "v"(lds_addr), "n"(WAITN), "n"(OFF) : "memory");
    if constexpr (SET == 0 && X == 3) asm volatile("s_waitcnt vmcnt(%1)\n\tds_write_b128 %0, a[212:215] offset:%2" :: "v"(lds_addr), "n"(WAITN), "n"(OFF) : "memory");
    if constexpr (SET == 0 && X == 4) asm volatile("s_waitcnt vmcnt(%1)\n\tds_write_b128 %0, a[216:219] offset:%2" :: "v"(lds_addr), "n"(WAITN), "n"(OFF) : "memory");
    if constexpr (SET == 0 && X == 5) asm volatile("s_waitcnt vmcnt(%1)\n\tds_write_b128 %0, a[220:223] offset:%2" :: "v"(lds_addr), "n"(WAITN), "n"(OFF) : "memory");
    if constexpr (SET == 0 && X == 6) asm volatile("s_waitcnt vmcnt(%1)\n\tds_write_b128 %0, a[224:227] offset:%2" :: "v"(lds_addr), "n"(WAITN), "n"(OFF) : "memory");
    if constexpr (SET == 0 && X == 7) asm volatile("s_waitcnt vmcnt(%1)\n\tds_write_b128 %0, a[228:231] offset:%2" :: "v"(lds_addr), "n"(WAITN), "n"(OFF) : "memory");
    if constexpr (SET == 0 && X == 8) asm volatile("s_waitcnt vmcnt(%1)\n\tds_write_b128 %0, a[232:235] offset:%2" :: "v"(lds_addr), "n"(WAITN), "n"(OFF) : "memory");
    if constexpr (SET == 0 && X == 9) asm volatile("s_waitcnt vmcnt(%1)\n\tds_write_b128 %0, a[236:239] offset:%2" :: "v"(lds_addr), "n"(WAITN), "n"(OFF) : "memory");
    if constexpr (SET == 0 && X == 10) asm volatile("s_waitcnt vmcnt(%1)\n\tds_write_b128 %0, a[240:243] offset:%2" :: "v"(lds_addr), "n"(WAITN), "n"(OFF) : "memory");
    if constexpr (SET == 0 && X == 11) asm volatile("s_waitcnt vmcnt(%1)\n\tds_write_b128 %0, a[244:247] offset:%2" :: "v"(lds_addr), "n"(WAITN), "n"(OFF) : "memory");
    if constexpr (SET == 0 && X == 12) asm volatile("s_waitcnt vmcnt(%1)\n\tds_write_b128 %0, a[248:251] offset:%2" :: "v"(lds_addr), "n"(WAITN), "n"(OFF) : "memory");
    if constexpr (SET == 0 && X == 13) asm volatile("s_waitcnt vmcnt(%1)\n\tds_write_b128 %0, a[252:255] offset:%2" :: "v"(lds_addr), "n"(WAITN), "n"(OFF) : "memory");
    if constexpr (SET == 1 && X == 0) asm volatile("s_waitcnt vmcnt(%1)\n\tds_write_b128 %0, v[228:231] offset:%2" :: "v"(lds_addr), "n"(WAITN), "n"(OFF) : "memory");
    if constexpr (SET == 1 && X == 1) asm volatile("s_waitcnt vmcnt(%1)\n\tds_write_b128 %0, v[232:235] offset:%2" :: "v"(lds_addr), "n"(WAITN), "n"(OFF) : "memory");
    if constexpr (SET == 1 && X == 2) asm volatile("s_waitcnt vmcnt(%1)\n\tds_write_b128 %0, v[236:239] offset:%2" :: "v"(lds_addr), "n"(WAITN), "n"(OFF) : "memory");
    if constexpr (SET == 1 && X == 3) asm volatile("s_waitcnt vmcnt(%1)\n\tds_write_b128 %0, v[240:243] offset:%2" :: "v"(lds_addr), "n"(WAITN), "n"(OFF) : "memory");
    if constexpr (SET == 1 && X == 4) asm volatile("s_waitcnt vmcnt(%1)\n\tds_write_b128 %0, v[244:247] offset:%2" :: "v"(lds_addr), "n"(WAITN), "n"(OFF) : "memory");
    if constexpr (SET == 1 && X == 5) asm volatile("s_waitcnt vmcnt(%1)\n\tds_write_b128 %0, v[248:251] offset:%2" :: "v"(lds_addr), "n"(WAITN), "n"(OFF) : "memory");
    if constexpr (SET == 1 && X == 6) asm volatile("s_waitcnt vmcnt(%1)\n\tds_write_b128 %0, v[252:255] offset:%2" :: "v"(lds_addr), "n"(WAITN), "n"(OFF) : "memory");
}
// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N)
template <int I0, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I0 < N) {
        f(std::integral_constant<int, I0>{});
        static_for<I0 + 1, N>(f);
    }
}

// TMB = 32-row blocks per WAVE, NWM = waves along the rows (the workgroup has 4 NWM waves: 4 channel blocks x NWM row groups),
// WPC = workgroups per CU; waves per SIMD = NWM WPC (1: staging in a[200:255], 2: in v[228:255]).
// M16: the same kernel on v_mfma_f32_16x16x32_bf16 (four MFMAs of 16 cycles per 32 x 32 block and k-step instead of two of 32;
// same registers, same LDS reads). The chip holds a higher clock on that shape (MI355X_MICROARCH.md, DVFS give-back (7)); its
// f32 sums run in another order, so results agree with the 32x32x16 kernels to the last bf16 bit only almost always.
// F16: operands and output in fp16 instead of bf16 (ConvArgs.f16; 32x32x16 form only)
template <int NK, bool ACT, bool STAMPS = false, int TMB = 4, int WPC = (TMB == 4 ? 1 : 2), int NWM = 1, bool M16 = false, bool F16 = false>
__global__ __launch_bounds__(256 * NWM, WPC) void gemm1x1_wspipe_kernel(ConvArgs p) {
    T16<F16>::enter();
    static_assert(!(M16 && F16), "the fp16 form exists on the 32x32x16 MFMA");
    constexpr int K = 32 * NK, NW = 4, NWT = NW * NWM;
    constexpr int BM = 32 * TMB * NWM, BN = 32 * NW;
    constexpr int CHUNK = BM * ROWB, TILE = NK * CHUNK;
    constexpr int PPK = BM / 16;                         // 16-row pieces per k-step, dealt to the NWT waves in turn: piece P = x NWT + wave
    static_assert(PPK % NWT == 0 || NWT % PPK == 0, "pieces and waves");
    constexpr int JJ = PPK > NWT ? PPK / NWT : 1;        // row pieces per wave and k-step (128-row tiles of 4 waves: 2)
    constexpr int SR = NWT > PPK ? NWT / PPK : 1;        // k-steps covered by one round of pieces (64-row tiles of 8 waves: 2)
    constexpr int NLD = NK * PPK / NWT;                  // 16-byte loads per lane and tile
    constexpr int MPS = (M16 ? 4 : 2) * TMB;             // MFMAs per k-step and wave
    constexpr int SLOTS = MPS * NK;                      // MFMAs per tile and wave
    constexpr int NH = 16 * TMB;                         // epilogue halves per tile and wave
    constexpr int RSET = NWM * WPC == 1 ? 0 : 1;         // which literal staging registers (one or two waves per SIMD)
    static_assert(NLD * NWT == NK * PPK, "whole pieces");
    static_assert(NLD <= (RSET == 0 ? 14 : 7), "staging pieces");
    constexpr int SPP = SLOTS / NLD;                     // slots per staged piece: 4 (8 with M16)
    static_assert(SLOTS == (M16 ? 8 : 4) * NLD, "one piece every fourth (eighth) slot");
    constexpr int STAGE_OFF = 2 * TILE;                  // per wave: two staging blocks (row blocks alternate)
    unsigned char* const lds = conv_lds_dyn;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int nsl = p.grid_n, Q = p.grid_m;
    const int g = blockIdx.x, idx = (g & 7) * (gridDim.x >> 3) + (g >> 3);
    const int q = idx / nsl, slice = idx - q * nsl;
    const int n_mt = (p.M + BM - 1) / BM;
    if (q >= Q || q >= n_mt) return;
    const int wn = wave & 3, wmh = wave >> 2;             // channel block, row group
    const int nw0 = slice * BN + 32 * wn;
    const bool live = nw0 < p.Cout;

    auto stamp_wg = [&](int slot) {             // probe: wave 0 of the first 64 workgroups, whole-kernel marks in slots 120..
        if constexpr (!STAMPS) return;
        if (wave == 0 && g < 64) {
            const uint64_t tnow = __builtin_amdgcn_s_memtime();
            if (lane == 0) reinterpret_cast<uint64_t*>(p.part)[(size_t)g * 128 + slot] = tnow;
        }
    };
    stamp_wg(120);
    if constexpr (STAMPS) {                     // the constant 100 MHz counter beside the shader clock: the kernel's clock rate
        if (wave == 0 && g < 64) {
            const uint64_t rt = __builtin_amdgcn_s_memrealtime();
            if (lane == 0) reinterpret_cast<uint64_t*>(p.part)[(size_t)g * 128 + 125] = rt;
        }
        if (wave == 0 && lane == 0) reinterpret_cast<uint64_t*>(p.part)[8192 + 2 * g] = __builtin_amdgcn_s_memrealtime();   // every workgroup: entry
    }
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    uint4 bfr[2 * NK];
    f32x2 bias2[4][2];
    {
        const int nrow = live ? nw0 : 0;
        // 32x32x16: fragment ks = channel r, k 16 ks + 8 h ..; 16x16x32: fragment 2 s + sn = channel 16 sn + (lane & 15), k 32 s + 8 (lane >> 4) ..
        const uint16_t* wrow = M16 ? p.w + (size_t)(nrow + (lane & 15)) * K + 8 * (lane >> 4) : p.w + (size_t)(nrow + r) * K + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 2 * NK; ++ks)
            bfr[ks] = *reinterpret_cast<const uint4*>(M16 ? wrow + (size_t)(16 * (ks & 1)) * K + 32 * (ks >> 1) : wrow + 16 * ks);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            // the lane's four channels of quad qq: 8 qq + 4 h .. (32x32x16), 16 (qq & 1) + 4 (lane >> 4) .. (16x16x32)
            const float4 b4 = *reinterpret_cast<const float4*>(p.bias + nrow + (M16 ? 16 * (qq & 1) + 4 * (lane >> 4) : 8 * qq + 4 * h));
            bias2[qq][0] = f32x2{b4.x, b4.y};
            bias2[qq][1] = f32x2{b4.z, b4.w};
        }
    }
    // Activation staging. In piece j of k-step s wave w owns rows 16 (NW j + w) .. + 15 of the chunk; lane -> (row, swizzled
    // 16-byte chunk). The loads go into the ACCUMULATION registers a[200:255] (a lone wave has 256 of them beside its 256
    // vector registers; loads and LDS writes may name them directly) and stay in flight for a whole tile: piece x of tile j + 2Q is requested
    // in slot 4x + 3 of tile j and written to LDS in slot 4x + 1 of tile j + Q. The compiler does not count these loads:
    // the wait in front of each LDS write is explicit -- vmcnt(NLD - 1): the NLD - 1 pieces requested after the one being
    // written (and any epilogue stores, which only make the wait stricter) may still fly.
    const unsigned char* a_bytes = reinterpret_cast<const unsigned char*>(p.in);
    // piece x of this wave: k-step s = SX(x) + s_base, row piece jj = JX(x) (rows 16 (NWT jj + wave % PPK) ..)
    uint32_t ld_off[JJ];
    int ld_row[JJ];
    const int s_base = NWT > PPK ? wave / PPK : 0;
#pragma unroll
    for (int j = 0; j < JJ; ++j) {
        ld_row[j] = 16 * (NWT * j + wave % PPK) + (lane >> 2);
        ld_off[j] = (uint32_t)(((lane & 3) ^ ((M16 ? -(ld_row[j] >> 2) : (ld_row[j] >> 2)) & 3)) * 16) + (uint32_t)(s_base * 64);
    }
    const uint32_t lds_wr = (uint32_t)(uintptr_t)(lds_ptr_t)lds + (uint32_t)((wave % PPK) * 1024 + lane * 16 + s_base * CHUNK);
#define ISB_WSP_SX(X) (NWT > PPK ? (X) * SR : (X) / JJ)
#define ISB_WSP_LOAD(X, tile)                                                                                      \
    wsp_request<RSET, (X), ISB_WSP_SX(X) * 64>(a_bytes + (size_t)min((tile) * BM + ld_row[(X) % JJ], p.M - 1) * (K * 2) + ld_off[(X) % JJ])
#define ISB_WSP_STORE(X, buf, WAITN)                                                                               \
    wsp_to_lds<RSET, (X), ISB_WSP_SX(X) * CHUNK, (WAITN)>(lds_wr + (uint32_t)((buf) * TILE + NWT * ((X) % JJ) * 1024))

    unsigned char* const stage = lds + STAGE_OFF + wave * (2 * WS_STAGE);
    uint16_t* const out16 = reinterpret_cast<uint16_t*>(p.out);
    // fragment of (k16 half ks | row half sm) of a 32-row block: 32x32x16 reads row r, chunk 2 ks + h; 16x16x32 row 16 sm + (lane & 15),
    // chunk lane >> 4, with the chunk slots turned by -(row >> 2) so that its 16-lane read groups stay on 64 different banks
    const int r16 = lane & 15, g16 = lane >> 4;
    const int a_sw0 = (M16 ? r16 * ROWB + ((g16 ^ ((-(r16 >> 2)) & 3)) << 4) : swz(r, h)) + wmh * TMB * 2048;
    const int a_sw1 = (M16 ? (16 + r16) * ROWB + ((g16 ^ ((-((16 + r16) >> 2)) & 3)) << 4) : swz(r, 2 + h)) + wmh * TMB * 2048;
    f32x16 acc[TMB], accp[TMB];                 // tile being multiplied / tile being finished

    // The epilogue of the previous tile in 64 halves of a value pair -- quad = 4 values of row block i (pixel 32 i + r,
    // channels nw0 + 8 qq + 4 h ..), pair = 2 of them; first half: bias, scale, 2 x v_exp; second half: + 1, 2 x v_rcp, times x,
    // one rounding (the arithmetic of silu_fast, two values per packed instruction) -- then an 8-byte write into the wave's
    // staging rows; some slots after the last quad of a row block its 32 x 32 block leaves as 16-byte pieces.
    f32x2 xa[2], ex[2];
    uint32_t pk[2];
    auto half_pair = [&](int hidx) {
        const int quad = hidx >> 2, sub = hidx & 3, i = quad >> 2, qq = quad & 3, pr = sub & 1;
        if (sub < 2) {
            f32x2 a = f32x2{accp[i][4 * qq + 2 * pr], accp[i][4 * qq + 2 * pr + 1]} + bias2[qq][pr];
            xa[pr] = a;
            if constexpr (ACT) {
                const f32x2 e = a * -1.4426950408889634f;
                ex[pr] = f32x2{__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
            }
        } else {
            f32x2 o = xa[pr];
            if constexpr (ACT) {
                const f32x2 d = ex[pr] + 1.0f;
                o = o * f32x2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
            }
            pk[pr] = T16<F16>::pack2(o.x, o.y);
            if (pr == 1) {
                const int prow = M16 ? 16 * (qq >> 1) + r16 : r, chb = M16 ? (16 * (qq & 1) + 4 * g16) * 2 : qq * 16 + h * 8;
                *reinterpret_cast<uint2*>(stage + (i & 1) * WS_STAGE + prow * WS_SROW + chb) = make_uint2(pk[0], pk[1]);
            }
        }
    };
    auto send_block = [&](int i, int m0) {
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const int row = 16 * k2 + (lane >> 2), cc = lane & 3;
            const uint4 v = *reinterpret_cast<const uint4*>(stage + (i & 1) * WS_STAGE + row * WS_SROW + cc * 16);
            const int m = min(m0 + 32 * (wmh * TMB + i) + row, p.M - 1);   // rows past M repeat row M - 1: same bytes, same address
            *reinterpret_cast<uint4*>(out16 + (size_t)m * p.Cout + nw0 + cc * 8) = v;
        }
    };
    // halves [NH m / SLOTS, NH (m + 1) / SLOTS) ride behind MFMA m; row block i leaves 3 halves after its last one
    auto finish_slot = [&](int m, int m0) {
#pragma unroll
        for (int hh = (NH * m) / SLOTS; hh < (NH * (m + 1)) / SLOTS; ++hh) {
            half_pair(hh);
            if (hh >= 18 && ((hh - 18) & 15) == 0) send_block((hh - 18) >> 4, m0);
        }
    };

    uint4 af[2][2][TMB];
    auto load_frag = [&](int s2, int ms, const unsigned char* At) {
        af[s2 & 1][ms / TMB][ms % TMB] = *reinterpret_cast<const uint4*>(At + s2 * CHUNK + ((ms / TMB) ? a_sw1 : a_sw0) + (ms % TMB) * 2048);
    };
    // one tile: SLOTS x { MFMA | fragment read of the next k-step | a share of the previous tile's epilogue |
    //                      every 4th slot: one staged piece to LDS, one piece of the tile after next requested }
    auto tile_pass = [&](int buf, int t, auto with_f, int m0_prev) {
        constexpr bool WITH_F = decltype(with_f)::value;
        const unsigned char* At = lds + buf * TILE;
#pragma unroll
        for (int i = 0; i < TMB; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
        for (int ms = 0; ms < 2 * TMB; ++ms) load_frag(0, ms, At);
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, SLOTS>([&](auto mc) {
            constexpr int m = decltype(mc)::value, s = m / MPS, q = m % MPS;
            if constexpr (M16) {
                // slot q of the k-step: channel half sn = q & 1 of fragment ms = q >> 1 (row block ms % TMB, row half ms / TMB)
                constexpr int sn = q & 1, ms = q >> 1, i = ms % TMB, qq = 2 * (ms / TMB) + sn;
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                f32x4 c = {acc[i][4 * qq], acc[i][4 * qq + 1], acc[i][4 * qq + 2], acc[i][4 * qq + 3]};
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bfr[2 * s + sn]), __builtin_bit_cast(bf16x8, af[s & 1][ms / TMB][i]), c, 0, 0, 0);
                acc[i][4 * qq] = c[0]; acc[i][4 * qq + 1] = c[1]; acc[i][4 * qq + 2] = c[2]; acc[i][4 * qq + 3] = c[3];
                if constexpr (s + 1 < NK && sn == 1) load_frag(s + 1, ms, At);
            } else {
                constexpr int ms = q;
                acc[ms % TMB] = T16<F16>::mfma32(bfr[2 * s + ms / TMB], af[s & 1][ms / TMB][ms % TMB], acc[ms % TMB]);
                if constexpr (s + 1 < NK) load_frag(s + 1, ms, At);
            }
            if constexpr (m % SPP == SPP / 4) ISB_WSP_STORE((m / SPP), buf ^ 1, NLD - 1);
            if constexpr (m % SPP == 3 * SPP / 4) ISB_WSP_LOAD((m / SPP), t + 2 * Q);
            if constexpr (WITH_F) finish_slot(m, m0_prev);
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (WITH_F) send_block(TMB - 1, m0_prev);
    };

    // prologue: tile q straight into buffer 0, tile q + Q requested (its pieces are written during the first pass)
    static_for<0, NLD>([&](auto xc) { ISB_WSP_LOAD(decltype(xc)::value, q); });
    static_for<0, NLD>([&](auto xc) { ISB_WSP_STORE(decltype(xc)::value, 0, 0); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    static_for<0, NLD>([&](auto xc) { ISB_WSP_LOAD(decltype(xc)::value, q + Q); });

    int buf = 0;
    bool first = true;
    int t_prev = q;
    int tile_no = 0;
    auto stamp = [&](int slot) {                // probe: wave 0 of the first 64 workgroups, tiles 1..4, 8 slots per tile
        if constexpr (!STAMPS) return;
        if (wave == 0 && g < 64 && tile_no >= 1 && tile_no <= 4) {
            const uint64_t tnow = __builtin_amdgcn_s_memtime();
            if (lane == 0) reinterpret_cast<uint64_t*>(p.part)[(size_t)g * 128 + (tile_no - 1) * 8 + slot] = tnow;
        }
    };
    stamp_wg(121);
    for (int t = q; t < n_mt; t += Q, buf ^= 1, ++tile_no) {
        stamp(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's pieces of tile t are in LDS
        __builtin_amdgcn_s_barrier();           // ... and everybody's; everybody is done reading the other buffer
        stamp(1);
        if (live) {
            if (first) tile_pass(buf, t, std::false_type{}, 0);
            else tile_pass(buf, t, std::true_type{}, t_prev * BM);
        } else {                                // a wave past Cout in the last slice only stages its rows
            static_for<0, NLD>([&](auto xc) {
                ISB_WSP_STORE(decltype(xc)::value, buf ^ 1, NLD - 1);
                ISB_WSP_LOAD(decltype(xc)::value, t + 2 * Q);
            });
        }
        stamp(2);
        if (live) {
#pragma unroll
            for (int i = 0; i < TMB; ++i) accp[i] = acc[i];
        }
        stamp(3);
        first = false;
        t_prev = t;
    }
    stamp_wg(122);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the run-ahead requests name this wave's registers
    if (live) {                                 // the last tile's epilogue
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) {
            half_pair(hh);
            if (hh >= 18 && ((hh - 18) & 15) == 0) send_block((hh - 18) >> 4, t_prev * BM);
        }
        send_block(TMB - 1, t_prev * BM);
    }
    stamp_wg(123);
    if constexpr (STAMPS) {
        const uint64_t rt = __builtin_amdgcn_s_memrealtime();
        if (wave == 0 && g < 64 && lane == 0) {
            reinterpret_cast<uint64_t*>(p.part)[(size_t)g * 128 + 124] = (uint64_t)tile_no;
            reinterpret_cast<uint64_t*>(p.part)[(size_t)g * 128 + 126] = rt;
        }
        if (wave == 0 && lane == 0) reinterpret_cast<uint64_t*>(p.part)[8192 + 2 * g + 1] = rt;                              // ... and exit
    }
#undef ISB_WSP_SX
#undef ISB_WSP_LOAD
#undef ISB_WSP_STORE
}


int launch_conv_ws(const ConvArgs& a, ConvArgs& aa, int v, hipStream_t st) {
    switch (v) {
#ifdef ISB_BUILD_PROBES
        case 181:                                            // weights-stationary persistent GEMM: 96 x 192 tiles, 6 waves
        case 182: {                                          //                                  or 128 x 128 tiles, 4 waves (one per SIMD)
            const int nw = v == 181 ? 6 : 4, tmb = v == 181 ? 3 : 4;
            const int bm = 32 * tmb, bn = 32 * nw;
            const int n_mt = cdiv(a.M, bm), nsl = cdiv(a.Cout, bn);
            if (a.f16 || a.gate || a.res || a.out_f32 || a.act > 1 || a.KH != 1 || a.stride != 1 || a.pad != 0 || a.splits > 1 || a.Cout % 32 != 0 ||
                (v == 181 && a.Cout % bn != 0) || nsl > 128 ||
                (a.Cin != 96 && a.Cin != 192 && a.Cin != 224 && a.Cin != 384) || (size_t)a.M * a.Cin * 2 >= 0xffffffffull) {
                set_error("conv_igemm: variants 181 / 182 are un-gated bf16 1x1 GEMMs without residual, Cin 96/192/224/384 (181: Cout %% 192 == 0)");
                return ISB_ERR_INVALID;
            }
            // one workgroup per CU (256, a multiple of 8 for the XCD decode); Q tile sequences of nsl slices each
            const int n_wg = 256;
            const int Q = std::max(1, std::min(n_wg / nsl, n_mt));
            aa.grid_n = nsl;
            aa.grid_m = Q;
            const dim3 g(n_wg);
            // at least 84 KiB so that two of these never share a CU (the phase pairing assumes one workgroup per CU)
#define ISB_WS(NK, NW, TMB)                                                                                     \
    do {                                                                                                        \
        const int bytes = std::max(ws_lds_bytes(NK, NW, TMB), 84 * 1024);                                       \
        if (bytes > 160 * 1024) {                                                                               \
            set_error("conv_igemm: weights-stationary tile of %d bytes does not fit the LDS", bytes);           \
            return ISB_ERR_INVALID;                                                                             \
        }                                                                                                       \
        static DevOnce attr_set;                                                                           \
        if (attr_set.need()) {                                                                                        \
            ISB_HIP(hipFuncSetAttribute((const void*)gemm1x1_wsreg_kernel<NK, true, false, NW, TMB>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));  \
            ISB_HIP(hipFuncSetAttribute((const void*)gemm1x1_wsreg_kernel<NK, false, false, NW, TMB>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)); \
            ISB_HIP(hipFuncSetAttribute((const void*)gemm1x1_wsreg_kernel<NK, true, true, NW, TMB>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));   \
            attr_set.mark();                                                                                    \
        }                                                                                                       \
        if (a.probe & 2) hipLaunchKernelGGL((gemm1x1_wsreg_kernel<NK, true, true, NW, TMB>), g, dim3(64 * NW), bytes, st, aa);           \
        else if (a.act) hipLaunchKernelGGL((gemm1x1_wsreg_kernel<NK, true, false, NW, TMB>), g, dim3(64 * NW), bytes, st, aa);           \
        else hipLaunchKernelGGL((gemm1x1_wsreg_kernel<NK, false, false, NW, TMB>), g, dim3(64 * NW), bytes, st, aa);                     \
    } while (0)
            if (v == 181) {
                if (a.Cin == 96) ISB_WS(3, 6, 3); else if (a.Cin == 192) ISB_WS(6, 6, 3); else if (a.Cin == 224) ISB_WS(7, 6, 3); else ISB_WS(12, 6, 3);
            } else {
                if (a.Cin == 96) ISB_WS(3, 4, 4); else if (a.Cin == 192) ISB_WS(6, 4, 4); else if (a.Cin == 224) ISB_WS(7, 4, 4); else ISB_WS(12, 4, 2);
            }
#undef ISB_WS
            break;
        }
        case 183: case 187: case 188:
#endif
        case 184: case 185: case 186: {            // weights-stationary, epilogue pipelined into the MFMA stream:
            // 184: 64-row tiles, two workgroups per CU (K <= 224); 185: 64-row tiles, one workgroup of 4 waves per CU, K = 384 (96
            // weight registers per lane); 186: the same tiles, 8 waves (2 x 32 rows). Probe builds: 183 = 128-row tiles, one wave
            // per SIMD; 187 / 188 = 184 / 186 on the 16x16x32 MFMA
            const int bm = v == 183 ? 128 : 64;
            const int nsl = cdiv(a.Cout, 128), n_mt = cdiv(a.M, bm);
            const bool k_ok = (v == 185 || v == 186 || v == 188) ? a.Cin == 384 : (a.Cin == 96 || a.Cin == 192 || a.Cin == 224);
            if (a.gate || a.res || a.out_f32 || a.act > 1 || a.KH != 1 || a.stride != 1 || a.pad != 0 || a.splits > 1 || a.Cout % 32 != 0 ||
                nsl > 128 || !k_ok || (size_t)a.M * a.Cin * 2 >= 0xffffffffull) {
                set_error("conv_igemm: variants 184 (Cin 96/192/224) and 185 / 186 (Cin 384) are un-gated 1x1 GEMMs without residual");
                return ISB_ERR_INVALID;
            }
            const int n_wg = (v == 184 || v == 187) ? 512 : 256;
            aa.grid_n = nsl;
            aa.grid_m = std::max(1, std::min(n_wg / nsl, n_mt));
            const dim3 g(n_wg);
#define ISB_WSP_GO(NK, ACT, STAMPS, TMB, WPC, NWM, M16_, F16_)                                                  \
    do {                                                                                                        \
        const int bytes = std::max(2 * NK * (32 * TMB * NWM) * 64 + 4 * NWM * 2 * WS_STAGE, WPC == 1 ? 84 * 1024 : 0); \
        static DevOnce attr_set;                                                                           \
        if (attr_set.need()) {                                                                                        \
            ISB_HIP(hipFuncSetAttribute((const void*)gemm1x1_wspipe_kernel<NK, ACT, STAMPS, TMB, WPC, NWM, M16_, F16_>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)); \
            attr_set.mark();                                                                                    \
        }                                                                                                       \
        hipLaunchKernelGGL((gemm1x1_wspipe_kernel<NK, ACT, STAMPS, TMB, WPC, NWM, M16_, F16_>), g, dim3(256 * NWM), bytes, st, aa); \
    } while (0)
            // every instantiation is a fully unrolled tile pass: the product build holds the forms the network selects (SiLU
            // epilogue, both 16-bit operand types); no-activation / stamped / 16x16x32 / 128-row forms are probe builds
#ifdef ISB_BUILD_PROBES
#define ISB_WSP(NK, TMB, WPC, NWM)                                                                              \
    do {                                                                                                        \
        if (a.f16 && a.act && !(a.probe & 2)) ISB_WSP_GO(NK, true, false, TMB, WPC, NWM, false, true);          \
        else if (a.f16) { set_error("conv_igemm: fp16 weights-stationary forms have the SiLU epilogue, no stamps"); return ISB_ERR_INVALID; } \
        else if (a.probe & 2) ISB_WSP_GO(NK, true, true, TMB, WPC, NWM, false, false);                          \
        else if (a.act) ISB_WSP_GO(NK, true, false, TMB, WPC, NWM, false, false);                               \
        else ISB_WSP_GO(NK, false, false, TMB, WPC, NWM, false, false);                                         \
    } while (0)
#else
#define ISB_WSP(NK, TMB, WPC, NWM)                                                                              \
    do {                                                                                                        \
        if (!a.act || (a.probe & 2)) {                                                                          \
            set_error("conv_igemm: variant %d without SiLU / with stamps needs a -DISB_BUILD_PROBES build", v); \
            return ISB_ERR_INVALID;                                                                             \
        }                                                                                                       \
        if (a.f16) ISB_WSP_GO(NK, true, false, TMB, WPC, NWM, false, true);                                     \
        else ISB_WSP_GO(NK, true, false, TMB, WPC, NWM, false, false);                                          \
    } while (0)
#endif
            if (v == 186) ISB_WSP(12, 1, 1, 2);
            else if (v == 185) ISB_WSP(12, 2, 1, 1);
            else if (v == 184) { if (a.Cin == 96) ISB_WSP(3, 2, 2, 1); else if (a.Cin == 192) ISB_WSP(6, 2, 2, 1); else ISB_WSP(7, 2, 2, 1); }
#ifdef ISB_BUILD_PROBES
#define ISB_WSP_ACT(NK, TMB, WPC, NWM, M16_)                                                                    \
    do {                                                                                                        \
        if (!a.act || (a.probe & 2) || a.f16) {                                                                 \
            set_error("conv_igemm: variant %d is built in bf16 with the SiLU epilogue and without stamps only", v); \
            return ISB_ERR_INVALID;                                                                             \
        }                                                                                                       \
        ISB_WSP_GO(NK, true, false, TMB, WPC, NWM, M16_, false);                                                \
    } while (0)
            else if (v == 188) ISB_WSP_ACT(12, 1, 1, 2, true);
            else if (v == 187) { if (a.Cin == 96) ISB_WSP_ACT(3, 2, 2, 1, true); else if (a.Cin == 192) ISB_WSP_ACT(6, 2, 2, 1, true); else ISB_WSP_ACT(7, 2, 2, 1, true); }
            else if (v == 183) { if (a.Cin == 96) ISB_WSP_ACT(3, 4, 1, 1, false); else if (a.Cin == 192) ISB_WSP_ACT(6, 4, 1, 1, false); else ISB_WSP_ACT(7, 4, 1, 1, false); }
#undef ISB_WSP_ACT
#endif
#undef ISB_WSP
#undef ISB_WSP_GO
            break;
        }
        default:
            set_error("conv_ws: tile variant %d is not in this build (weights-stationary: 184, 185, 186)", v);
            return ISB_ERR_INVALID;
    }
    return ISB_OK;
}

}  // namespace isb
