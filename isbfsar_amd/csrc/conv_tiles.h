// Device code shared by the convolution tile kernels (conv_igemm.hip, conv_gemm1x1*.hip, conv_3x3.hip, conv_fused_mb.hip,
// conv_mb8.hip): the common epilogue, the workgroup -> tile map, the LDS-DMA issue helpers; and the host-side contract
// between the dispatcher (conv_dispatch.hip: chooses a tile variant per layer) and the per-family launchers.
//
// conv_igemm tiling: a k-tile is 32 input channels of ONE filter tap (Cin % 32 == 0), so a row of
// the A tile is one contiguous 64-B run of the NHWC input (or zeros for padding).  A and B tiles
// sit in LDS as [row][64 B] with the 16-B chunk index XOR-swizzled by (row>>2)&3: with that the
// ds_read_b128 fragment reads of the 32x32x16 MFMA (lane -> row lane&31, chunk 2*ks + lane>>5)
// are bank-conflict free (4-way without it).  Two LDS buffers, next tile's global loads in
// flight during the MFMAs, one barrier per k-tile.  The MFMA computes the TRANSPOSED output tile
// (weights as the A operand) so each lane ends up with 4 consecutive channels of one pixel: bias,
// SiLU and the residual are applied in registers, the 16-bit tile is staged through LDS (row stride
// BN*2+16 B: conflict-free 8-byte writes) and leaves as full 16-byte pieces.
#pragma once
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

// ---------------------------------------------------------------------------------------------------------------------
// host side: dispatcher <-> per-family launchers. `a` = the caller's arguments, `aa` = a copy whose grid fields the
// launcher fills (conv_grid); v = tile variant. Variants outside the set the three networks select (pose backbone,
// detector, ResNet trunk) exist only in builds with -DISB_BUILD_PROBES (tools/, EXPERIMENTS.md).
// ---------------------------------------------------------------------------------------------------------------------
int launch_tiles_igemm(int v, const ConvArgs& a, ConvArgs& aa, hipStream_t st);         // conv_igemm.hip: 1 - 116
int launch_tiles_gemm1x1(int v, const ConvArgs& a, ConvArgs& aa, hipStream_t st);       // conv_gemm1x1.hip: 131 - 140, 150
int launch_tiles_gemm1x1_gate(int v, const ConvArgs& a, ConvArgs& aa, hipStream_t st);  // conv_gemm1x1_gate.hip: 141 - 149, 152 - 156, 191 - 197
int launch_tiles_conv3x3(int v, const ConvArgs& a, ConvArgs& aa, hipStream_t st);       // conv_3x3.hip: 161 - 171
int launch_conv_wsk(const ConvArgs& a, ConvArgs& aa, hipStream_t st);                   // conv_wsk.hip: 157 (gated projection, weights stationary in registers)

static inline dim3 conv_grid(ConvArgs& a, int BM, int BN) {
    a.grid_m = cdiv(a.M, BM);
    a.grid_n = cdiv(a.Cout, BN);
    const unsigned z = a.splits > 1 ? (unsigned)a.splits : 1u;      // split-K (gemm1x1 kernels only)
    if (a.grid_mode == 0 || a.grid_n == 1) {
        a.grid_mode = 0;
        return dim3(a.grid_m, a.grid_n, z);
    }
    return dim3(8 * ((a.grid_m + 7) / 8) * a.grid_n, 1, z);
}

}  // namespace isb
