// The MBConv blocks of the two 8 x 8 stages of EfficientNetV2-L (stride 1: 24 x 384 -> 2304 -> 384, 1 x 384 -> 2304 -> 640,
// 6 x 640 -> 3840 -> 640; 31 of the backbone's 79 blocks, 31 % of its FLOPs) as ONE launch: mb8_chain_kernel.
//
// At 8 x 8 a sample is 64 pixels = two 32-row MFMA blocks, and nothing in an MBConv block couples samples: the depthwise 3 x 3
// needs no halo beyond the sample, the squeeze-excite pool / FC1 / FC2 / gate are per sample. So ONE workgroup owns ONE sample for
// the whole chain of blocks and never meets another workgroup. Per launch of the five-kernel form (expand GEMM, depthwise + pool,
// FC1, FC2, gated projection) a block costs 35-55 us per GEMM for 11-13 us of work at either floor (first-tile latency, gate rows,
// epilogue, the tail of a single round of workgroups) and sends the expanded tensor (295 KB per sample) through HBM twice. Here:
//
//   residual stream X [64 px][Cin]   lives in LDS for the whole chain (A-operand tiles of the expand GEMM, 4 KiB per 32 channels)
//   phase 1-3, per WAVE and 32-channel block of the expanded tensor, no barrier at all:
//       expand: the wave streams the block's weights global -> REGISTERS (pre-packed in MFMA fragment order: one coalesced 1-KiB
//               load per k16 step, each fragment feeds the two row blocks; the next batch travels under the current one's MFMAs
//               and the next block's first batch under this block's vector phase) -- weights never touch LDS;
//       SiLU -> 16-bit -> the wave's private [10 x 10 padded px][32 ch] tile in LDS (the zero ring is TF-SAME's padding);
//       depthwise 3 x 3 from that tile (the taps of dwconv3x3_map_kernel in its order), SiLU, one rounding;
//       D block -> global scratch AS THE 4-KiB A-TILE the projection will DMA back (L2-resident: written and re-read by this CU);
//       pool: quad partial sums -> 16-term sums in quad order -> mean (the order of the depthwise kernel: same bits);
//   phase 4  squeeze-excite for the one sample in the arithmetic AND ORDER of se_fc1_part_kernel / se_fc2_kernel (256-channel
//            chains, partials in chunk order; four j-quarters in wave order), weights streamed to registers from packed layouts;
//   phase 5  gated projection: 2-3 helper waves DMA the D tiles into an LDS ring (one barrier per four k-steps), the projection
//            waves stream W2 through registers like phase 1, gate the A fragments as they leave LDS (gemm1x1's arithmetic);
//   phase 6  bias + residual (from X in LDS) + one rounding -> X in place (the next block's operand), last block -> global.
//
// Every sum runs in the order of the five-kernel path: the chain's output is BIT-IDENTICAL to it (tests/test_hpe_gpu.py).
// What bounds it: per block a workgroup pulls ALL the block's weights through its CU's L2 port (3.5 MB 16-bit + 1.8 MB f32
// squeeze-excite for 384 channels) for 64 rows -- 128 FLOP per 16-bit weight byte against the 127 a CU's matrix pipe needs per
// byte delivered at 32 B/clk; DESIGN.md section 3 has the arithmetic and the measured stamps.
#include "conv_tiles.h"
#include "dw_mm.h"

namespace isb {

namespace {

constexpr int MB8_NW = 8;                       // waves per workgroup (two per SIMD)
[[maybe_unused]] constexpr int MB8_NT = 64 * MB8_NW;
constexpr int MB8_XBYTES = 20 * 4096;           // residual stream: up to 640 channels = 20 tiles of [64 rows][64 B]
constexpr int MB8_ET_PIX = 100;                 // 10 x 10 padded pixels
constexpr int MB8_ET_BYTES = MB8_ET_PIX * 64;   // per wave: [pixel][32 ch x 2 B]
constexpr int MB8_ET_OFF = MB8_XBYTES;
constexpr int MB8_POOL_OFF = MB8_ET_OFF + MB8_NW * MB8_ET_BYTES;      // f32 [Cexp]: pooled means, later the gate
constexpr int MB8_MID_OFF = MB8_POOL_OFF + 3840 * 4;                  // f32 [160] squeeze-excite hidden units
constexpr int MB8_PART_OFF = MB8_MID_OFF + 160 * 4;                   // f32 [15][160] FC1 partial sums per 256-channel chunk
constexpr int MB8_LDS = MB8_PART_OFF + 15 * 160 * 4;                  // 158 720 B
[[maybe_unused]] constexpr int MB8_RING_OFF = MB8_ET_OFF;        // phase 5: 8 D tiles of 4 KiB (two halves of four k-steps) overlay the E tiles
static_assert(8 * 4096 <= MB8_NW * MB8_ET_BYTES, "the D ring fits the E tiles' region");
static_assert(MB8_LDS <= 160 * 1024, "LDS");

#ifdef ISB_BUILD_PROBES      // (shapes of mb8_chain_kernel)
template <int CIN, int COUT, bool F16>
struct Mb8Shape {
    static constexpr int CEXP = 6 * CIN, CSE = CIN / 4;
    static constexpr int NCB = CEXP / 32;                   // 32-channel blocks of the expanded tensor = k-steps of the projection
    static constexpr int NK16 = CIN / 16;                   // k16 steps of the expand GEMM
    static constexpr int KB = NK16 / 4;                     // ... in four register batches (two in flight)
    static constexpr int CBW = COUT == 384 ? 2 : 4;         // 32-channel blocks per projection wave
    static constexpr int NPW = COUT / 32 / CBW;             // projection waves: 6 (384 outputs) / 5 (640)
    static constexpr int NHELP = MB8_NW - NPW;              // helper waves (D-tile DMA)
    static_assert(NK16 % 4 == 0 && NCB % MB8_NW == 0 && NCB % 4 == 0 && COUT % (32 * CBW) == 0 && NPW < MB8_NW, "shape");
};

template <int I0, int N, class F>
__device__ __forceinline__ void mb8_static_for(F&& f) {
    if constexpr (I0 < N) {
        f(std::integral_constant<int, I0>{});
        mb8_static_for<I0 + 1, N>(f);
    }
}

#endif
__device__ __forceinline__ int et_pix(int m) { return ((m >> 3) + 1) * 10 + (m & 7) + 1; }      // pixel m of the 8 x 8 map in the padded tile

#ifdef ISB_BUILD_PROBES      // the per-sample chain: measured 2x slower than the five launches (EXPERIMENTS.md round 4), kept as a probe
// One MBConv block for the workgroup's sample. X (LDS) in: the block's input, out: its output.
template <int CIN, int COUT, bool F16>
__device__ __forceinline__ void mb8_block(const Mb8Block& bk, unsigned char* lds, unsigned char* dscr, uint16_t* out_g, int tid, uint64_t* stamps) {
    using S = Mb8Shape<CIN, COUT, F16>;
    constexpr int CEXP = S::CEXP, CSE = S::CSE, NCB = S::NCB, NK16 = S::NK16, KB = S::KB, CBW = S::CBW, NPW = S::NPW, NHELP = S::NHELP;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    float* const pooled = reinterpret_cast<float*>(lds + MB8_POOL_OFF);
    float* const mid = reinterpret_cast<float*>(lds + MB8_MID_OFF);
    float* const part = reinterpret_cast<float*>(lds + MB8_PART_OFF);
    // tuning probe (Mb8Args.stamps): wave 0 and the last wave of the first workgroups mark the phase boundaries with s_memtime
    auto stamp = [&](int slot) {
        if (stamps && (wave == 0 || wave == MB8_NW - 1)) {
            const uint64_t t = __builtin_amdgcn_s_memtime();
            if (lane == 0) stamps[(wave == 0 ? 0 : 16) + slot] = t;
        }
    };
    stamp(0);

    // ------------------------------------------------------------------ phases 1-3: per wave, per 32-channel block
    {
        unsigned char* const et = lds + MB8_ET_OFF + wave * MB8_ET_BYTES;
        // the zero ring (36 of the 100 padded pixels; the projection's D ring overlays this region, so once per block)
        for (int i = lane; i < 36 * 4; i += 64) {
            const int q = i >> 2, ch = i & 3;
            const int pix = q < 10 ? q : (q < 20 ? 90 + (q - 10) : (q < 28 ? 10 * (q - 19) : 10 * (q - 27) + 9));
            *reinterpret_cast<uint4*>(et + pix * 64 + ch * 16) = make_uint4(0, 0, 0, 0);
        }
        const int a_sw0 = swz(r, h), a_sw1 = swz(r, 2 + h);            // A fragment of row block 0, k16 half 0 / 1 of a k-step tile
        const int pq = lane >> 2, cl = lane & 3;                        // depthwise: pixel quad (16 per map), 8-channel chunk
        const int oy = pq >> 1, ox0 = (pq & 1) * 4;
        uint32_t one_lo, one_hi;                                        // (1, 0) / (0, 1) pairs in the storage type (see dwconv3x3_pool_kernel)
        if constexpr (F16) asm volatile("v_mov_b32 %0, 0x3c00\n\tv_lshlrev_b32 %1, 16, %0" : "=v"(one_lo), "=v"(one_hi));
        else asm volatile("v_mov_b32 %0, 0x3f80\n\tv_lshlrev_b32 %1, 16, %0" : "=v"(one_lo), "=v"(one_hi));

        uint4 wb[2][KB];
        auto load_batch = [&](auto bufc, int cb, int b) {               // k16 steps b KB .. of channel block cb
            constexpr int buf = decltype(bufc)::value;
            const uint4* src = bk.w1p + ((size_t)cb * NK16 + (size_t)b * KB) * 64 + lane;
#pragma unroll
            for (int s = 0; s < KB; ++s) wb[buf][s] = src[s * 64];
        };
        // every workgroup walks the channel blocks in its own rotation: at any moment the 32 CUs of an XCD then stream 32 x 8
        // DIFFERENT weight regions instead of hammering the same eight through one or two L2 channels (no sum depends on the order
        // of the channel blocks)
        constexpr int NIT = NCB / MB8_NW;
        const int rot = (int)(blockIdx.x >> 3) % NIT;
        auto cb_of = [&](int it) { return wave + MB8_NW * ((it + rot) % NIT); };
        load_batch(std::integral_constant<int, 0>{}, cb_of(0), 0);
        for (int it = 0; it < NIT; ++it) {
            const int cb = cb_of(it);
            // the block's small operands: requested now, used after the MFMAs
            const int c0 = cb * 32;
            float4 bias1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bias1[q] = *reinterpret_cast<const float4*>(bk.b1 + c0 + 8 * q + 4 * h);
            uint4 taps[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) taps[t] = *reinterpret_cast<const uint4*>(bk.dww + (size_t)t * CEXP + c0 + cl * 8);
            const float4 db0 = *reinterpret_cast<const float4*>(bk.dwb + c0 + cl * 8), db1 = *reinterpret_cast<const float4*>(bk.dwb + c0 + cl * 8 + 4);

            f32x16 acc[2];
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[0][e] = acc[1][e] = 0.f;
            auto mma_batch = [&](auto bufc, int b) {
                constexpr int buf = decltype(bufc)::value;
#pragma unroll
                for (int s = 0; s < KB; ++s) {
                    const int k16 = b * KB + s;
                    const unsigned char* xt = lds + (k16 >> 1) * 4096 + ((k16 & 1) ? a_sw1 : a_sw0);
                    const uint4 a0 = *reinterpret_cast<const uint4*>(xt), a1 = *reinterpret_cast<const uint4*>(xt + 2048);
                    acc[0] = T16<F16>::mfma32(wb[buf][s], a0, acc[0]);
                    acc[1] = T16<F16>::mfma32(wb[buf][s], a1, acc[1]);
                }
            };
            const int cbn = it + 1 < NIT ? cb_of(it + 1) : NCB;
            load_batch(std::integral_constant<int, 1>{}, cb, 1);
            mma_batch(std::integral_constant<int, 0>{}, 0);
            load_batch(std::integral_constant<int, 0>{}, cb, 2);
            mma_batch(std::integral_constant<int, 1>{}, 1);
            load_batch(std::integral_constant<int, 1>{}, cb, 3);
            mma_batch(std::integral_constant<int, 0>{}, 2);
            if (cbn < NCB) load_batch(std::integral_constant<int, 0>{}, cbn, 0);     // travels under this block's vector phase
            mma_batch(std::integral_constant<int, 1>{}, 3);

            // ---- E = T16(silu(acc + bias)) -> the wave's padded tile. acc[rb][e]: pixel 32 rb + r, channel c0 + 8 (e >> 2) + 4 h + (e & 3)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                unsigned char* cell = et + et_pix(rb * 32 + r) * 64 + h * 8;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v0 = silu_fast(acc[rb][4 * q] + bias1[q].x), v1 = silu_fast(acc[rb][4 * q + 1] + bias1[q].y);
                    const float v2 = silu_fast(acc[rb][4 * q + 2] + bias1[q].z), v3 = silu_fast(acc[rb][4 * q + 3] + bias1[q].w);
                    uint2 pk;
                    pk.x = (uint32_t)T16<F16>::from_f32(v0) | ((uint32_t)T16<F16>::from_f32(v1) << 16);
                    pk.y = (uint32_t)T16<F16>::from_f32(v2) | ((uint32_t)T16<F16>::from_f32(v3) << 16);
                    *reinterpret_cast<uint2*>(cell + q * 16) = pk;
                }
            }
            // ---- depthwise 3 x 3 + bias + SiLU on the tile: lane = (pixel quad, 8-channel chunk), taps and order of dwconv3x3_map_kernel
            uint32_t wlo[9][4], whi[9][4];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const uint32_t wp[4] = {taps[t].x, taps[t].y, taps[t].z, taps[t].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { wlo[t][e] = wp[e] & 0xffffu; whi[t][e] = wp[e] & 0xffff0000u; }
            }
            const float dbias[8] = {db0.x, db0.y, db0.z, db0.w, db1.x, db1.y, db1.z, db1.w};
            float dacc[4][8], psum[8];
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int e = 0; e < 8; ++e) dacc[o][e] = dbias[e];
#pragma unroll
            for (int e = 0; e < 8; ++e) psum[e] = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                uint4 v[6];
#pragma unroll
                for (int col = 0; col < 6; ++col) v[col] = *reinterpret_cast<const uint4*>(et + ((oy + ky) * 10 + ox0 + col) * 64 + cl * 16);
#pragma unroll
                for (int col = 0; col < 6; ++col) {
                    const uint32_t x[4] = {v[col].x, v[col].y, v[col].z, v[col].w};
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        const int kx = col - o;
                        if (kx >= 0 && kx < 3) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                dacc[o][2 * e] = T16<F16>::dot2(x[e], wlo[ky * 3 + kx][e], dacc[o][2 * e]);
                                dacc[o][2 * e + 1] = T16<F16>::dot2(x[e], whi[ky * 3 + kx][e], dacc[o][2 * e + 1]);
                            }
                        }
                    }
                }
            }
            unsigned char* const dtile = dscr + (size_t)cb * 4096;      // the projection's A tile of k-step cb: [64 rows][64 B], swizzled
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                uint32_t pk[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint16_t lo = T16<F16>::from_f32(silu_fast(dacc[o][2 * e])), hi = T16<F16>::from_f32(silu_fast(dacc[o][2 * e + 1]));
                    pk[e] = (uint32_t)lo | ((uint32_t)hi << 16);
                    psum[2 * e] = T16<F16>::dot2(pk[e], one_lo, psum[2 * e]);
                    psum[2 * e + 1] = T16<F16>::dot2(pk[e], one_hi, psum[2 * e + 1]);
                }
                *reinterpret_cast<uint4*>(dtile + swz(oy * 8 + ox0 + o, cl)) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            }
            // ---- pool: 16 quad sums per channel, added in quad order, / 64 (dwconv3x3_map_kernel's red[][] walk). The interior of
            // the tile is free now: the quad sums go there (pixels 11.., never the ring)
            float* const red = reinterpret_cast<float*>(et + 11 * 64);        // [16][32] f32 = 2 KiB inside the tile's interior rows 1-4
#pragma unroll
            for (int e = 0; e < 8; ++e) red[pq * 32 + cl * 8 + e] = psum[e];
            if (lane < 32) {
                float t = 0.f;
#pragma unroll
                for (int s2 = 0; s2 < 16; ++s2) t += red[s2 * 32 + lane];
                pooled[c0 + lane] = t / 64.0f;
            }
            // (the quad sums overlapped ring-free pixels 11 .. 42; the ring's zeros at 19, 20, 29, 30, 39, 40 must be restored)
            if (lane < 24) {
                const int q = lane >> 2, ch = lane & 3;
                const int pix = 19 + 10 * (q >> 1) + (q & 1);
                *reinterpret_cast<uint4*>(et + pix * 64 + ch * 16) = make_uint4(0, 0, 0, 0);
            }
        }
    }
    stamp(1);
    __syncthreads();                                                    // D tiles written (vmcnt) and pooled complete
    stamp(2);

    // ------------------------------------------------------------------ phase 4: squeeze-excite for this sample
    {
        // FC1 partials: part[kc][j] = chain over the 256 channels of chunk kc (se_fc1_part_kernel's order)
        constexpr int NKC = CEXP / 256;
        for (int id = tid; id < NKC * CSE; id += MB8_NT) {
            const int kc = id / CSE, j = id - kc * CSE;
            const float4* wsrc = reinterpret_cast<const float4*>(bk.se_w1p) + (size_t)kc * 64 * CSE + j;
            const float4* psrc = reinterpret_cast<const float4*>(pooled + kc * 256);
            float a = 0.f;
#pragma unroll 16
            for (int c4 = 0; c4 < 64; ++c4) {
                const float4 wv = wsrc[(size_t)c4 * CSE];
                const float4 pv = psrc[c4];
                a = fmaf(pv.x, wv.x, a);
                a = fmaf(pv.y, wv.y, a);
                a = fmaf(pv.z, wv.z, a);
                a = fmaf(pv.w, wv.w, a);
            }
            part[kc * CSE + j] = a;
        }
        stamp(3);
        __syncthreads();
        if (tid < CSE) {                                                // se_fc2_kernel's prologue
            float v = bk.se_b1[tid];
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc) v += part[kc * CSE + tid];
            mid[tid] = v / (1.0f + expf(-v));
        }
        __syncthreads();
        stamp(4);
        // FC2 + sigmoid: four j-quarters summed separately, then added in quarter order (the four waves of se_fc2_kernel)
        constexpr int JQ = (CSE + 3) >> 2;
        float* const gate = pooled;                                     // every reader of the pooled means is past the barrier above
        for (int slot = tid; slot < CEXP / 4; slot += MB8_NT) {
            const int c = slot * 4;
            float4 pw[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
                const int jb = w * JQ, je = jb + JQ < CSE ? jb + JQ : CSE;
#pragma unroll 8
                for (int j = jb; j < je; ++j) {
                    const float4 wv = *reinterpret_cast<const float4*>(bk.se_w2t + (size_t)j * CEXP + c);
                    const float mv = mid[j];
                    a.x = fmaf(mv, wv.x, a.x);
                    a.y = fmaf(mv, wv.y, a.y);
                    a.z = fmaf(mv, wv.z, a.z);
                    a.w = fmaf(mv, wv.w, a.w);
                }
                pw[w] = a;
            }
            float4 v = pw[0];
#pragma unroll
            for (int w = 1; w < 4; ++w) { v.x += pw[w].x; v.y += pw[w].y; v.z += pw[w].z; v.w += pw[w].w; }
            const float4 b2 = *reinterpret_cast<const float4*>(bk.se_b2 + c);
            v.x = 1.0f / (1.0f + expf(-(v.x + b2.x)));
            v.y = 1.0f / (1.0f + expf(-(v.y + b2.y)));
            v.z = 1.0f / (1.0f + expf(-(v.z + b2.z)));
            v.w = 1.0f / (1.0f + expf(-(v.w + b2.w)));
            *reinterpret_cast<float4*>(gate + c) = v;
        }
    }
    // (FC2 reads `mid` and writes the gate over the pooled means: the quarter loops above finished reading pooled two barriers ago)
    stamp(5);
    __syncthreads();
    stamp(6);

    // ------------------------------------------------------------------ phase 5: gated projection, K = Cexp in halves of four k-steps
    {
        constexpr int NH = NCB / 4;
        const uint32_t ring_lds = (uint32_t)(uintptr_t)(lds_ptr_t)lds + MB8_RING_OFF;
        const float* const gate = pooled;
        if (wave >= NPW) {
            // ---- helper waves: D tiles global -> ring by LDS-DMA, 16 KiB per half dealt to the helpers
            const int hw = wave - NPW;
            auto issue = [&](int hf) {
                for (int pc = hw; pc < 16; pc += NHELP)
                    dma16_s(dscr + (size_t)hf * 16384, (uint32_t)(pc * 1024 + lane * 16), ring_lds + (uint32_t)((hf & 1) * 16384 + pc * 1024));
            };
            issue(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            for (int hf = 0; hf < NH; ++hf) {
                if (hf + 1 < NH) issue(hf + 1);                         // the other half was released by the barrier that ended hf - 1
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        } else {
            // ---- projection waves: CBW 32-channel blocks x both row blocks; W2 global -> registers (fragment-packed per wave)
            f32x16 acc[2][CBW];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int j = 0; j < CBW; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[rb][j][e] = 0.f;
            const uint4* const wsrc = bk.w2p + (size_t)wave * (CEXP / 16) * CBW * 64 + lane;      // [k16][cbw][lane]
            constexpr int QS = CBW == 2 ? 4 : 2, NQ = 8 / QS;          // k16 steps per register batch (two batches in flight), batches per half
            uint4 wq[2][QS][CBW];
            auto load_q = [&](auto bufc, int k16_0) {
                constexpr int buf = decltype(bufc)::value;
#pragma unroll
                for (int s = 0; s < QS; ++s)
#pragma unroll
                    for (int j = 0; j < CBW; ++j) wq[buf][s][j] = wsrc[((size_t)(k16_0 + s) * CBW + j) * 64];
            };
            const int a_sw[2] = {swz(r, h), swz(r, 2 + h)};
            auto mma_q = [&](auto bufc, int hf, int qi) {              // k16 steps QS qi .. of half hf
                constexpr int buf = decltype(bufc)::value;
#pragma unroll
                for (int s = 0; s < QS; ++s) {
                    const int k16h = qi * QS + s;
                    const int ktl = k16h >> 1, ks = k16h & 1;           // tile in the half, k16 half of the tile
                    const unsigned char* at = lds + MB8_RING_OFF + (hf & 1) * 16384 + ktl * 4096 + a_sw[ks];
                    const float* gs = gate + (4 * hf + ktl) * 32 + ks * 16 + 8 * h;
                    const float4 g0 = *reinterpret_cast<const float4*>(gs), g1 = *reinterpret_cast<const float4*>(gs + 4);
                    const uint4 a0 = T16<F16>::gate8(*reinterpret_cast<const uint4*>(at), g0, g1);
                    const uint4 a1 = T16<F16>::gate8(*reinterpret_cast<const uint4*>(at + 2048), g0, g1);
#pragma unroll
                    for (int j = 0; j < CBW; ++j) {
                        acc[0][j] = T16<F16>::mfma32(wq[buf][s][j], a0, acc[0][j]);
                        acc[1][j] = T16<F16>::mfma32(wq[buf][s][j], a1, acc[1][j]);
                    }
                }
            };
            load_q(std::integral_constant<int, 0>{}, 0);
            __builtin_amdgcn_s_barrier();                               // half 0 landed
            for (int hf = 0; hf < NH; ++hf) {
                mb8_static_for<0, NQ>([&](auto qc) {
                    constexpr int qi = decltype(qc)::value;
                    const int next = 8 * hf + (qi + 1) * QS;
                    if (next < CEXP / 16) load_q(std::integral_constant<int, (qi + 1) & 1>{}, next);
                    mma_q(std::integral_constant<int, qi & 1>{}, hf, qi);
                });
                __builtin_amdgcn_s_barrier();
            }
            stamp(7);
            // ---- phase 6: bias + residual + one rounding (conv_epilogue's order) -> X in place, and the chain's last block -> global
            const int b_img = blockIdx.x;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const int m = rb * 32 + r;
#pragma unroll
                for (int j = 0; j < CBW; ++j) {
                    const int kt = wave * CBW + j;                      // output channel block = tile of the next block's X
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = kt * 32 + 8 * q + 4 * h;
                        const float4 bs = *reinterpret_cast<const float4*>(bk.b2 + n);
                        float v0 = acc[rb][j][4 * q] + bs.x, v1 = acc[rb][j][4 * q + 1] + bs.y, v2 = acc[rb][j][4 * q + 2] + bs.z, v3 = acc[rb][j][4 * q + 3] + bs.w;
                        unsigned char* cell = lds + kt * 4096 + swz(m, q) + 8 * h;
                        if (bk.residual) {
                            const uint2 rr = *reinterpret_cast<const uint2*>(cell);
                            v0 += T16<F16>::lo(rr.x); v1 += T16<F16>::hi(rr.x);
                            v2 += T16<F16>::lo(rr.y); v3 += T16<F16>::hi(rr.y);
                        }
                        uint2 pk;
                        pk.x = (uint32_t)T16<F16>::from_f32(v0) | ((uint32_t)T16<F16>::from_f32(v1) << 16);
                        pk.y = (uint32_t)T16<F16>::from_f32(v2) | ((uint32_t)T16<F16>::from_f32(v3) << 16);
                        *reinterpret_cast<uint2*>(cell) = pk;
                        if (out_g) *reinterpret_cast<uint2*>(out_g + ((size_t)b_img * 64 + m) * COUT + n) = pk;
                    }
                }
            }
        }
    }
    stamp(8);
    __syncthreads();                                                    // X complete; ring region free for the next block's E tiles
    stamp(9);
}

template <bool F16>
__global__ __launch_bounds__(MB8_NT) void mb8_chain_kernel(Mb8Args p) {
    T16<F16>::enter();
    unsigned char* const lds = conv_lds_dyn;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    // the sample's input rows -> X tiles: tile kt = channels 32 kt .., [64 rows][64 B] swizzled (gemm1x1's A image)
    {
        const int cin = p.cin0;
        const unsigned char* src = reinterpret_cast<const unsigned char*>(p.x) + (size_t)b * 64 * cin * 2;
        const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)lds;
        const int npieces = (cin / 32) * 4;
        for (int pc = wave; pc < npieces; pc += MB8_NW) {
            const int kt = pc >> 2, row = 16 * (pc & 3) + (lane >> 2);
            const int logical = (lane & 3) ^ ((row >> 2) & 3);
            dma16_s(src, (uint32_t)(row * cin * 2 + kt * 64 + logical * 16), lds0 + (uint32_t)(pc * 1024));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    unsigned char* const dscr = reinterpret_cast<unsigned char*>(p.dscratch) + (size_t)b * p.dscratch_stride;
    for (int i = 0; i < p.nblocks; ++i) {
        const Mb8Block& bk = p.blocks[i];
        // the chain's output = its last block's; store_all (tests): every block's output, out_block_stride elements apart
        uint16_t* const out_g = p.store_all ? p.out + (size_t)i * p.out_block_stride : (i + 1 == p.nblocks ? p.out : nullptr);
        // stamps: blocks 1 (384 -> 384) and nblocks - 2 (640 -> 640) of the first 32 workgroups, 32 slots each
        uint64_t* const stamps = (p.stamps && b < 32 && (i == 1 || i == p.nblocks - 2)) ? p.stamps + ((size_t)b * 2 + (i == 1 ? 0 : 1)) * 32 : nullptr;
        if (bk.cin == 384 && bk.cout == 384) mb8_block<384, 384, F16>(bk, lds, dscr, out_g, tid, stamps);
        else if (bk.cin == 384) mb8_block<384, 640, F16>(bk, lds, dscr, out_g, tid, stamps);
        else mb8_block<640, 640, F16>(bk, lds, dscr, out_g, tid, stamps);
    }
}

#endif  // ISB_BUILD_PROBES
// -------------------------------------------------------------------------------------------------------------------
// The FRONT HALF of a stride-1 MBConv block on 8 x 8 maps in one launch: 1x1 expand + BN + SiLU -> depthwise 3x3 + BN + SiLU -> D
// (NHWC, what the gated projection reads) + the squeeze-excite pool. The expanded tensor never leaves the chip.
// Tiling as in the weights-stationary expand GEMM (conv_ws.hip), where a 64-row tile IS one sample: a workgroup of 4 waves (one per
// SIMD, the whole register file each) owns a 128-channel slice of the expanded tensor for its whole life -- wave w keeps the weights
// of its 32 channels for all of K in registers -- and walks the samples q, q + Q, ...; a sample's 64 x Cin input tile arrives in LDS
// by LDS-DMA, double-buffered, one barrier per sample. What differs from the per-sample chain above (and from rounds 1-2's fused
// fronts on the tile GEMM): N IS tiled across CUs, so the L2 -> CU traffic is the expand GEMM's (activations re-read per slice), and
// the vector work -- two SiLUs and nine taps per expanded element, 4 650 cycles per (sample, 32 channels) against 1 536 of MFMA -- is
// written in ONE scheduling region with the NEXT sample's MFMAs, whose results it does not need: the wave's matrix and vector
// instructions interleave (a second accumulator set). The depthwise stage is wave-local (private padded tile, taps / order / pool of
// dwconv3x3_map_kernel): outputs and pooled means are the bits of the two-launch path.
// -------------------------------------------------------------------------------------------------------------------
template <int CIN>
struct Mf8 {
    static constexpr int NK16 = CIN / 16, NKT = CIN / 32, CEXP = 6 * CIN, NSL = CEXP / 128;
    static constexpr int XBUF = NKT * 4096;
    static constexpr int ET_OFF = XBUF;
    // per wave 1 KiB: bias [32] f32 | depthwise bias [32] f32 | taps [9][32] 16-bit (tap-major)
    static constexpr int TBL_BYTES = 1024;
    static constexpr int TBL_OFF = ET_OFF + 4 * MB8_ET_BYTES;
    static constexpr int LDS = TBL_OFF + 4 * TBL_BYTES;
    static_assert(2 * LDS <= 160 * 1024, "two workgroups per CU");
};

template <int CIN, bool F16>
__global__ __launch_bounds__(256, 2) void mbfront8_kernel(MbFront8Args p) {
    using S = Mf8<CIN>;
    constexpr int NK16 = S::NK16, CEXP = S::CEXP, NSL = S::NSL;
    T16<F16>::enter();
    unsigned char* const lds = conv_lds_dyn;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int slice = blockIdx.x % NSL, q = blockIdx.x / NSL, Q = gridDim.x / NSL;
    if (q >= p.B) return;
    const int cb = slice * 4 + wave, c0 = cb * 32;
    unsigned char* const et = lds + S::ET_OFF + wave * MB8_ET_BYTES;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)lds;

    // one sample's input rows -> X tiles: tile kt = channels 32 kt .., [64 rows][64 B] swizzled (gemm1x1's A image)
    auto dma_x = [&](int smp) {
        const unsigned char* src = reinterpret_cast<const unsigned char*>(p.x) + (size_t)smp * 64 * CIN * 2;
        for (int pc = wave; pc < S::NKT * 4; pc += 4) {
            const int kt = pc >> 2, row = 16 * (pc & 3) + (lane >> 2);
            const int logical = (lane & 3) ^ ((row >> 2) & 3);
            dma16_s(src, (uint32_t)(row * CIN * 2 + kt * 64 + logical * 16), lds0 + (uint32_t)(pc * 1024));
        }
    };
    dma_x(q);
    // the wave's weights, for the whole kernel (fragment-packed: one coalesced 1-KiB load per k16 step)
    uint4 wreg[NK16];
    {
        const uint4* src = p.w1p + (size_t)cb * NK16 * 64 + lane;
#pragma unroll
        for (int s = 0; s < NK16; ++s) wreg[s] = src[s * 64];
    }
    const int pq = lane >> 2, cl = lane & 3;                        // depthwise: pixel quad (16 per map), 8-channel chunk
    const int oy = pq >> 1, ox0 = (pq & 1) * 4;
    // the block's small per-channel operands live in a per-wave LDS table for the whole kernel (the registers belong to the stationary
    // weights; two waves share a SIMD's 512): read back as 16-byte pieces where they are used
    unsigned char* const tbl = lds + S::TBL_OFF + wave * S::TBL_BYTES;
    if (lane < 8) *reinterpret_cast<float4*>(tbl + lane * 16) = *reinterpret_cast<const float4*>(p.b1 + c0 + lane * 4);
    else if (lane < 16) *reinterpret_cast<float4*>(tbl + lane * 16) = *reinterpret_cast<const float4*>(p.dwb + c0 + (lane - 8) * 4);
    else if (lane < 16 + 36) {
        const int t = (lane - 16) >> 2, c4 = (lane - 16) & 3;
        *reinterpret_cast<uint4*>(tbl + 256 + t * 64 + c4 * 16) = *reinterpret_cast<const uint4*>(p.dww + (size_t)t * CEXP + c0 + c4 * 8);
    }
    for (int i = lane; i < 36 * 4; i += 64) {                       // the zero ring of the wave's padded tile, once
        const int qi = i >> 2, ch = i & 3;
        const int pix = qi < 10 ? qi : (qi < 20 ? 90 + (qi - 10) : (qi < 28 ? 10 * (qi - 19) : 10 * (qi - 27) + 9));
        *reinterpret_cast<uint4*>(et + pix * 64 + ch * 16) = make_uint4(0, 0, 0, 0);
    }
    const int a_sw0 = swz(r, h), a_sw1 = swz(r, 2 + h);
    uint32_t one_lo, one_hi;                                        // (1, 0) / (0, 1) pairs in the storage type (see dwconv3x3_pool_kernel)
    if constexpr (F16) asm volatile("v_mov_b32 %0, 0x3c00\n\tv_lshlrev_b32 %1, 16, %0" : "=v"(one_lo), "=v"(one_hi));
    else asm volatile("v_mov_b32 %0, 0x3f80\n\tv_lshlrev_b32 %1, 16, %0" : "=v"(one_lo), "=v"(one_hi));

    // lane constants of the matrix-pipe depthwise (dw_mm.h). The padded tile's 16-byte chunk slots are turned by
    // f(y, x) = ((x >> 2) & 1) | ((y & 1) << 1): the sixteen lanes a ds_read_b128 serves together (pixels 2 n + j of four rows) then
    // fall into sixteen different slots of the 256-byte bank row (four pixels of 64 bytes)
    const DwmmLane wl(lane);
    const int w_d = min(max(wl.d, 0), 2);
    const int mn = lane & 15, mj = lane >> 4, ms = mj >> 1;        // pixel pair, input column / output rows, pixel of the pair
    const int e_f = ((((r & 7) + 1) >> 2) & 1) | ((((r >> 3) + 1) & 1) << 1);      // E write: padded pixel ((r >> 3) + 1 + 4 rb, (r & 7) + 1)
    int b_pix[3], b_f[2];
    {
        const int ry = mn >> 2, xx = 2 * (mn & 3) + mj;             // B fragment: padded pixel (4 t + ry + ky, xx)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) b_pix[ky] = ((ry + ky) * 10 + xx) * 64;
        b_f[0] = ((xx >> 2) & 1) | ((ry & 1) << 1);                 // ky even
        b_f[1] = ((xx >> 2) & 1) | (((ry + 1) & 1) << 1);           // ky odd
    }
    int d_pix, d_f;
    {
        const int yy = (mn >> 2) + 1, xx = 2 * (mn & 3) + ms + 1;   // the lane's output pixel (tile 0), padded coordinates
        d_pix = (yy * 10 + xx) * 64 + (mj & 1) * 8;
        d_f = ((xx >> 2) & 1) | ((yy & 1) << 1);
    }
    uint64_t st_wait = 0, st_body = 0, st_t0 = 0;                  // tuning probe (MbFront8Args.stamps)
    uint64_t st_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};                               // ... phases of the body: expand MFMAs, second barrier, E epilogue, depthwise + stores
    if (p.stamps) st_t0 = __builtin_amdgcn_s_memtime();
    int it = 0;
    for (int smp = q; smp < p.B; smp += Q, ++it) {
        // two workgroups share a CU and the OLDER wave is served first: the workgroup with more samples still to do outranks the
        // other, so that both arrive together (conv_mb16.hip has the census that showed the effect)
        {
            const int left = (p.B - 1 - smp) / Q, total = (p.B - 1 - q) / Q + 1;       // samples still to do after this one, of `total`
            const int bucket = min(3, (4 * left + total - 1) / total);
            if (bucket == 3) __builtin_amdgcn_s_setprio(3);
            else if (bucket == 2) __builtin_amdgcn_s_setprio(2);
            else if (bucket == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        uint64_t ta = 0, tb = 0;
        if (p.stamps) ta = __builtin_amdgcn_s_memtime();
        // the sample's tiles landed: requested after the previous sample's MFMAs, ahead of that sample's five stores (four D rows, the
        // pooled means), which may keep flying (vmcnt retires in order)
        if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (p.stamps) { tb = __builtin_amdgcn_s_memtime(); st_wait += tb - ta; }
        // ---- expand: the sample x the wave's 32 channels
        f32x16 acc[2];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][e] = acc[1][e] = 0.f;
        // fragment reads run one pair of k16 steps (four MFMAs) ahead of the MFMAs that use them, in a second register set: left to
        // itself the compiler emits read -> wait -> MFMA per fragment, and with two waves per SIMD nothing hides the LDS latency
        // (round 5 stamps: 2 930 cycles for these 48 MFMAs)
        {
            uint4 fa[2][4];                                          // [set][k16 of the pair x row block]
            auto rd = [&](int pair, uint4 (&f)[4]) {
                const unsigned char* xt = lds + pair * 4096;
                f[0] = *reinterpret_cast<const uint4*>(xt + a_sw0); f[1] = *reinterpret_cast<const uint4*>(xt + a_sw0 + 2048);
                f[2] = *reinterpret_cast<const uint4*>(xt + a_sw1); f[3] = *reinterpret_cast<const uint4*>(xt + a_sw1 + 2048);
            };
            rd(0, fa[0]);
#pragma unroll
            for (int pr = 0; pr < NK16 / 2; ++pr) {
                if (pr + 1 < NK16 / 2) rd(pr + 1, fa[(pr + 1) & 1]);
                acc[0] = T16<F16>::mfma32(wreg[2 * pr], fa[pr & 1][0], acc[0]);
                acc[1] = T16<F16>::mfma32(wreg[2 * pr], fa[pr & 1][1], acc[1]);
                acc[0] = T16<F16>::mfma32(wreg[2 * pr + 1], fa[pr & 1][2], acc[0]);
                acc[1] = T16<F16>::mfma32(wreg[2 * pr + 1], fa[pr & 1][3], acc[1]);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);  // the next pair's four fragment reads first ...
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);  // ... then this pair's four MFMAs
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        uint64_t tc = 0, td = 0, te = 0;
        if (p.stamps) tc = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();                               // everybody has read the tiles: the next sample's may land
        if (p.stamps) { td = __builtin_amdgcn_s_memtime(); st_ph[0] += tc - tb; st_ph[1] += td - tc; }
        if (smp + Q < p.B) dma_x(smp + Q);
        // ---- E = T16(silu(acc + bias)) -> the wave's padded tile (16-byte chunk slot = chunk ^ f(y, x): see the lane constants)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            unsigned char* cell = et + et_pix(rb * 32 + r) * 64 + h * 8;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const float4 bs = *reinterpret_cast<const float4*>(tbl + (8 * qq + 4 * h) * 4);
                const float v0 = silu_fast(acc[rb][4 * qq] + bs.x), v1 = silu_fast(acc[rb][4 * qq + 1] + bs.y);
                const float v2 = silu_fast(acc[rb][4 * qq + 2] + bs.z), v3 = silu_fast(acc[rb][4 * qq + 3] + bs.w);
                uint2 pk;
                pk.x = T16<F16>::pack2(v0, v1);
                pk.y = T16<F16>::pack2(v2, v3);
                *reinterpret_cast<uint2*>(cell + ((qq ^ e_f) << 4)) = pk;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (p.stamps) { te = __builtin_amdgcn_s_memtime(); st_ph[2] += te - td; }
        // ---- depthwise 3x3 + bias on the MATRIX pipe (dw_mm.h: per 8-channel group and 32-pixel tile three 16-byte fragment reads
        // and three v_mfma_f32_16x16x32 instead of 288 v_dot2), SiLU, pooled sums; arithmetic and orders of dwconv3x3_mm_kernel<8>
        // (all 24 MFMAs first -- the fragments of a group are read while the previous group multiplies -- then the 32 SiLUs as
        // independent streams: one item after the other is a chain of dependent instructions, 5 700 cycles per sample)
        f32x4 a4[4][2];
        {
            uint4 bf[2][6];
            auto rdb = [&](int g, uint4 (&f)[6]) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
                        f[t * 3 + ky] = *reinterpret_cast<const uint4*>(et + b_pix[ky] + ((g ^ b_f[ky & 1]) << 4) + t * (4 * 10 * 64));
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
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4 c4 = f32x4{db.x, db.y, db.z, db.w};
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) c4 = mfma16<F16>(af[ky], bf[g & 1][t * 3 + ky], c4);
                    a4[g][t] = c4;
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 10, 0); // the group's taps and bias and the next group's six fragments: reads first
                __builtin_amdgcn_sched_group_barrier(0x002, 16, 0); // the weight fragments take their places (vector ALU)
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);  // six MFMAs (two chains of three)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        uint64_t tg = 0, th = 0, ti = 0;
        if (p.stamps) { tg = __builtin_amdgcn_s_memtime(); st_ph[4] += tg - te; }
        uint32_t dpk[4][2][2];                                      // [group][tile]: the lane's 4 channels of pixel 2 n + s
        float psum[4][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int i = 0; i < 4; ++i) psum[g][i] = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const uint32_t pk0 = T16<F16>::pack2(silu_fast(a4[g][t][0]), silu_fast(a4[g][t][1]));
                const uint32_t pk1 = T16<F16>::pack2(silu_fast(a4[g][t][2]), silu_fast(a4[g][t][3]));
                psum[g][0] = T16<F16>::dot2(pk0, one_lo, psum[g][0]);      // the pool sees the stored (rounded) activations
                psum[g][1] = T16<F16>::dot2(pk0, one_hi, psum[g][1]);
                psum[g][2] = T16<F16>::dot2(pk1, one_lo, psum[g][2]);
                psum[g][3] = T16<F16>::dot2(pk1, one_hi, psum[g][3]);
                dpk[g][t][0] = pk0;
                dpk[g][t][1] = pk1;
            }
        }
        if (p.stamps) { th = __builtin_amdgcn_s_memtime(); st_ph[5] += th - tg; }
        // D: every fragment read of the sample is done -- the results go into the tile's interior in place (own chunks, wave-private
        // tile) and leave as whole 16-byte pieces: lane = (pixel quad, 8-channel chunk), four stores of 64 bytes per pixel and wave
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                *reinterpret_cast<uint2*>(et + d_pix + ((g ^ d_f) << 4) + t * (4 * 10 * 64)) = make_uint2(dpk[g][t][0], dpk[g][t][1]);
        uint16_t* const drow = p.d + ((size_t)smp * 64) * CEXP + c0 + cl * 8;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int yy = oy + 1, xx = ox0 + o + 1;
            const uint4 v = *reinterpret_cast<const uint4*>(et + (yy * 10 + xx) * 64 + ((cl ^ (((xx >> 2) & 1) | ((yy & 1) << 1))) << 4));
            *reinterpret_cast<uint4*>(drow + (size_t)(oy * 8 + ox0 + o) * CEXP) = v;
        }
        if (p.stamps) { ti = __builtin_amdgcn_s_memtime(); st_ph[6] += ti - th; }
        // pool: the lanes' sums over their two tiles -> dw_mm.h's butterfly over the 32 (pixel pair, pixel) lanes, / 64 (dwconv3x3_mm_kernel's
        // order). ONE 16-byte store (the counted wait above numbers this sample's stores): lane (pair g < 4, first pixel) stores group g's
        // four channels of its channel half
        {
            const int xaddr = (lane ^ 32) << 2;
            float4 tot[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) tot[g] = dwmm_pool_sum4(psum[g], xaddr, 1.0f / 64.0f);
            const float4 mine = sel4(mn < 2, sel4(mn == 0, tot[0], tot[1]), sel4(mn == 2, tot[2], tot[3]));
            if (mn < 4 && ms == 0) *reinterpret_cast<float4*>(p.pooled + (size_t)smp * CEXP + c0 + mn * 8 + 4 * (mj & 1)) = mine;
        }
        if (p.stamps) { const uint64_t tf = __builtin_amdgcn_s_memtime(); st_body += tf - tb; st_ph[3] += tf - te; st_ph[7] += tf - ti; }
    }
    if (p.stamps && blockIdx.x < 32 && lane == 0) {
        // (volatile: 8-byte stores, one per stamp -- the build's check of the counted wait tells them from the loop's 16-byte stores)
        volatile uint64_t* o = p.stamps + ((size_t)blockIdx.x * 4 + wave) * 16;
        o[0] = __builtin_amdgcn_s_memtime() - st_t0; o[1] = st_wait; o[2] = st_body; o[3] = (uint64_t)it;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[4 + i] = st_ph[i];
    }
}

// =====================================================================================
// Round 6: the same front half with the waves of a workgroup in DIFFERENT ROLES (the form of mbfront16r_kernel, conv_mb16.hip).
// mbfront8_kernel above runs two waves per SIMD (96 stationary weight registers + the depthwise stage in the same wave) and each of its
// phases takes 3-4x its issue cost (EXPERIMENTS.md round 5: the SIMD issues 55 % of its cycles, the matrix pipe is busy 33 %). Here ONE
// workgroup of TWELVE waves per CU owns a 128-channel slice:
//   waves 0-3   PRODUCERS (one per SIMD): the expand GEMM of a sample (64 pixels x the wave's 32 channels: 48 MFMAs on 96 stationary weight
//               registers), bias, SiLU, one rounding, the E tile into the channel block's padded 10 x 10 tile in LDS (two tile buffers);
//   waves 4-11  CONSUMERS (two per SIMD, 16 channels each): depthwise 3x3 on the matrix pipe from the tile the producers finished a tick
//               earlier (weight fragments stationary), SiLU, one rounding, pooled sums, the D rows in place through the tile; they also
//               issue the LDS-DMA of the next sample's input tile (two buffers) -- the producers' instruction stream is the pole of a tick
//               (48 MFMAs = 1 536 matrix cycles, then 32 SiLUs per lane), the consumers' vector work fits into the issue slots the
//               producers' MFMAs leave free (an MFMA holds the issue port 8 of its 32 cycles).
// One workgroup barrier per sample. No counted vmcnt wait: a consumer waits for its input-tile pieces (issued a tick earlier) with
// vmcnt(0) BEFORE it issues the tick's stores. Arithmetic and orders are mbfront8_kernel's: bit-identical to it and to the two launches.
// Work: units (slice, sample) of an XCD's samples in slice-major order, cut into equal contiguous ranges for the XCD's 32 workgroups.
template <int CIN>
struct Mf8r {
    static constexpr int NK16 = CIN / 16, NKT = CIN / 32, CEXP = 6 * CIN, NSL = CEXP / 128;
    static constexpr int XBUF = NKT * 4096;                 // a sample's input tile: [NKT][64 rows][64 B], swizzled
    static constexpr int ET = MB8_ET_BYTES;                 // a channel block's padded tile: [100 pixels][32 ch x 2 B]
    static constexpr int ET_OFF = 2 * XBUF;                 // [4 blocks][2 buffers]
    static constexpr int TBL_OFF = ET_OFF + 8 * ET;
    static constexpr int TBL_BYTES = 1024;                  // per channel block: bias [32] f32 | depthwise bias [32] f32 at 128 | taps [9][32] 16-bit at 256
    static constexpr int LDS = TBL_OFF + 4 * TBL_BYTES;
    static_assert(LDS <= 160 * 1024, "one workgroup per CU");
};

template <int CIN, bool F16, bool STAMP = false>
__global__ __launch_bounds__(768, 3) void mbfront8r_kernel(MbFront8Args p) {
    using S = Mf8r<CIN>;
    constexpr int NK16 = S::NK16, CEXP = S::CEXP, NSL = S::NSL, ET = S::ET;
    T16<F16>::enter();
    unsigned char* const lds = conv_lds_dyn;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave < 4;
    // this workgroup's units: XCD x (ids 8 apart share one) owns samples [x Bx, x Bx + nx); its slots cut NSL * nx units evenly
    const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3, nslots = (int)(gridDim.x >> 3);
    const int Bx = (p.B + 7) / 8, smp0 = xcd * Bx, nx = min(Bx, p.B - smp0);
    if (nx <= 0) return;
    const int U = NSL * nx;
    int u = (int)((long long)slot_id * U / nslots);
    const int u1 = (int)((long long)(slot_id + 1) * U / nslots);
    if (u >= u1) return;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)lds;
    // tuning probe (STAMP instantiation; MbFront8Args.stamps: [32 workgroups][12 waves][8]): ticks, then cycles summed over the ticks --
    // producers: MFMAs of row block 0, its epilogue, MFMAs of row block 1, its epilogue, barrier; consumers: input requests, fragment
    // reads + MFMAs, SiLU + D + readback + stores, pool, barrier
    uint64_t stv[6] = {0, 0, 0, 0, 0, 0};
    auto now = [&]() __attribute__((always_inline)) -> uint64_t { if constexpr (STAMP) return __builtin_amdgcn_s_memtime(); else return 0; };
    // the padded tiles, zero rings included (only interiors are ever written afterwards): once per workgroup
    for (int i = tid; i < 8 * ET / 16; i += 768) *reinterpret_cast<uint4*>(lds + S::ET_OFF + i * 16) = make_uint4(0, 0, 0, 0);

#define ISB_MBF8R_SEGMENT                                                                                                   \
        const int slice = u / nx, i0 = u % nx, n = min(nx - i0, u1 - u);                                                      \
        u += n;                                                                                                               \
        const int sbase = smp0 + i0;
    if (producer) {
        const int r = lane & 31, h = lane >> 5;
        const int a_sw0 = swz(r, h), a_sw1 = swz(r, 2 + h);
        unsigned char* const et0 = lds + S::ET_OFF + wave * 2 * ET;
        unsigned char* const tbl = lds + S::TBL_OFF + wave * S::TBL_BYTES;
        const int e_f = ((((r & 7) + 1) >> 2) & 1) | ((((r >> 3) + 1) & 1) << 1);      // E write: padded pixel ((r >> 3) + 1 + 4 rb, (r & 7) + 1)
        const int e_cell = et_pix(r) * 64 + h * 8;                                       // row block rb: + rb * 4 * 10 * 64
        while (u < u1) {
            ISB_MBF8R_SEGMENT
            (void)sbase;
            const int cb = slice * 4 + wave, c0 = cb * 32;
            uint4 wreg[NK16];
            {
                const uint4* src = p.w1p + (size_t)cb * NK16 * 64 + lane;
#pragma unroll
                for (int s = 0; s < NK16; ++s) wreg[s] = src[s * 64];
            }
            if (lane < 8) *reinterpret_cast<float4*>(tbl + lane * 16) = *reinterpret_cast<const float4*>(p.b1 + c0 + lane * 4);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");       // (the consumers: the first sample's input tile has landed)
#pragma unroll 1
            for (int T = 0; T <= n; ++T) {
                uint64_t tq = now();
                if (T < n) {
                    const unsigned char* const xb = lds + (T & 1) * S::XBUF;
                    unsigned char* const et = et0 + (T & 1) * ET;
                    if constexpr (STAMP) ++stv[0];
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb) {
                        f32x16 acc;
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
                        // fragment reads two k16 pairs ahead of their use (three register sets)
                        uint4 fa[3][2];
                        auto rd = [&](int pr, uint4 (&f)[2]) __attribute__((always_inline)) {
                            f[0] = *reinterpret_cast<const uint4*>(xb + pr * 4096 + rb * 2048 + a_sw0);
                            f[1] = *reinterpret_cast<const uint4*>(xb + pr * 4096 + rb * 2048 + a_sw1);
                        };
                        rd(0, fa[0]);
                        rd(1, fa[1]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int pr = 0; pr < NK16 / 2; ++pr) {
                            if (pr + 2 < NK16 / 2) rd(pr + 2, fa[(pr + 2) % 3]);
                            acc = T16<F16>::mfma32(wreg[2 * pr], fa[pr % 3][0], acc);
                            acc = T16<F16>::mfma32(wreg[2 * pr + 1], fa[pr % 3][1], acc);
                            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if constexpr (STAMP) { asm volatile("s_nop 0" ::: "memory"); const uint64_t t2 = now(); stv[1 + 2 * rb] += t2 - tq; tq = t2; }
                        // E = T16(silu(acc + bias)) -> the block's padded tile (16-byte chunk slot = chunk ^ f(y, x))
                        unsigned char* const cell = et + e_cell + rb * (4 * 10 * 64);
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) {
                            const float4 bs = *reinterpret_cast<const float4*>(tbl + (8 * qq + 4 * h) * 4);
                            const f32x2_t v01 = silu_fast2(f32x2_t{acc[4 * qq], acc[4 * qq + 1]} + f32x2_t{bs.x, bs.y});
                            const f32x2_t v23 = silu_fast2(f32x2_t{acc[4 * qq + 2], acc[4 * qq + 3]} + f32x2_t{bs.z, bs.w});
                            uint2 pk;
                            pk.x = T16<F16>::pack2(v01.x, v01.y);
                            pk.y = T16<F16>::pack2(v23.x, v23.y);
                            *reinterpret_cast<uint2*>(cell + ((qq ^ e_f) << 4)) = pk;
                        }
                        if constexpr (STAMP) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const uint64_t t2 = now(); stv[2 + 2 * rb] += t2 - tq; tq = t2; }
                    }
                }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if constexpr (STAMP) { if (T < n) stv[5] += now() - tq; }
            }
        }
    } else {
        const int cw = wave - 4, blk = cw & 3, half = cw >> 2;      // the channel block and which 16 of its channels
        unsigned char* const et0 = lds + S::ET_OFF + blk * 2 * ET;
        unsigned char* const tbl = lds + S::TBL_OFF + blk * S::TBL_BYTES;
        // the next sample's input tile: 4 NKT pieces of 1 KiB over the eight consumer waves
        auto dma_x = [&](int smp, int buf) {
            const unsigned char* src = reinterpret_cast<const unsigned char*>(p.x) + (size_t)smp * 64 * CIN * 2;
            for (int pc = cw; pc < S::NKT * 4; pc += 8) {
                const int kt = pc >> 2, row = 16 * (pc & 3) + (lane >> 2);
                const int logical = (lane & 3) ^ ((row >> 2) & 3);
                dma16_s(src, (uint32_t)(row * CIN * 2 + kt * 64 + logical * 16), lds0 + (uint32_t)(buf * S::XBUF + pc * 1024));
            }
        };
        const DwmmLane wl(lane);
        const int w_d = min(max(wl.d, 0), 2);
        const int mn = lane & 15, mj = lane >> 4, ms = mj >> 1;        // pixel pair, input column / output rows, pixel of the pair
        int b_pix[3], b_f[2];
        {
            const int ry = mn >> 2, xx = 2 * (mn & 3) + mj;             // B fragment: padded pixel (4 t + ry + ky, xx)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) b_pix[ky] = ((ry + ky) * 10 + xx) * 64;
            b_f[0] = ((xx >> 2) & 1) | ((ry & 1) << 1);                 // ky even
            b_f[1] = ((xx >> 2) & 1) | (((ry + 1) & 1) << 1);           // ky odd
        }
        int d_pix, d_f;
        {
            const int yy = (mn >> 2) + 1, xx = 2 * (mn & 3) + ms + 1;   // the lane's output pixel (tile 0), padded coordinates
            d_pix = (yy * 10 + xx) * 64 + (mj & 1) * 8;
            d_f = ((xx >> 2) & 1) | ((yy & 1) << 1);
        }
        // readback: piece i of lane = pixel (lane >> 1) + 32 i, the wave's 16-byte chunk lane & 1 of it (its 16 channels sit at slots
        // (2 half + chunk) ^ f): a store instruction then writes 32 bytes of 32 pixels (one pixel per lane made it 16 bytes of 64 rows)
        int o_off[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int px = (lane >> 1) + 32 * i, yy = (px >> 3) + 1, xx = (px & 7) + 1;
            o_off[i] = (yy * 10 + xx) * 64 + (((2 * half + (lane & 1)) ^ (((xx >> 2) & 1) | ((yy & 1) << 1))) << 4);
        }
        uint32_t one_lo, one_hi;                                        // (1, 0) / (0, 1) pairs in the storage type
        if constexpr (F16) asm volatile("v_mov_b32 %0, 0x3c00\n\tv_lshlrev_b32 %1, 16, %0" : "=v"(one_lo), "=v"(one_hi));
        else asm volatile("v_mov_b32 %0, 0x3f80\n\tv_lshlrev_b32 %1, 16, %0" : "=v"(one_lo), "=v"(one_hi));
        while (u < u1) {
            ISB_MBF8R_SEGMENT
            const int cb = slice * 4 + blk, c0 = cb * 32;
            // the block's depthwise bias and taps in the block's table (both halves write the same values), then this wave's weight fragments
            if (lane < 8) *reinterpret_cast<float4*>(tbl + 128 + lane * 16) = *reinterpret_cast<const float4*>(p.dwb + c0 + lane * 4);
            else if (lane >= 16 && lane < 16 + 36) {
                const int t = (lane - 16) >> 2, c4 = (lane - 16) & 3;
                *reinterpret_cast<uint4*>(tbl + 256 + t * 64 + c4 * 16) = *reinterpret_cast<const uint4*>(p.dww + (size_t)t * CEXP + c0 + c4 * 8);
            }
            uint4 af[2][3];
#pragma unroll
            for (int gl = 0; gl < 2; ++gl)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
                    af[gl][ky] = wl.place((uint32_t)*reinterpret_cast<const uint16_t*>(tbl + 256 + (ky * 3 + w_d) * 64 + ((2 * half + gl) * 8 + wl.c) * 2));
            dma_x(sbase, 0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // a sample's results leave at the TOP of the next tick, ahead of that tick's input requests: the wait at the bottom of a tick is
            // then a plain vmcnt(0) with a whole tick between every request and it (stamps: waiting for the pieces in mid-tick, before the
            // stores, cost the consumers 1 900 cycles of a 6 800-cycle tick -- an input piece lands about 2 000 cycles after its request)
            uint4 dv[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
            float4 pool_v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int xaddr = (lane ^ 32) << 2;
            auto flush = [&](int smp) __attribute__((always_inline)) {
                uint16_t* const drow = p.d + ((size_t)smp * 64 + (lane >> 1)) * CEXP + c0 + half * 16 + (lane & 1) * 8;
#pragma unroll
                for (int i = 0; i < 2; ++i) *reinterpret_cast<uint4*>(drow + (size_t)(32 * i) * CEXP) = dv[i];
                // (lane = pair gl < 2, first pixel: group gl's four channels of its channel half)
                if (mn < 2 && ms == 0) *reinterpret_cast<float4*>(p.pooled + (size_t)smp * CEXP + c0 + half * 16 + mn * 8 + 4 * (mj & 1)) = pool_v;
            };
#pragma unroll 1
            for (int T = 0; T <= n; ++T) {
                uint64_t tq = now();
                if (T >= 2) flush(sbase + T - 2);
                if (T + 1 < n) dma_x(sbase + T + 1, (T + 1) & 1);
                if constexpr (STAMP) { if (T >= 1) { ++stv[0]; const uint64_t t2 = now(); stv[1] += t2 - tq; tq = t2; } }
                if (T >= 1) {
                    unsigned char* const et = et0 + ((T - 1) & 1) * ET;
                    // ---- depthwise 3x3 + bias on the matrix pipe: two tiles of 32 pixels x this wave's two 8-channel groups
                    f32x4 a4[2][2];
                    {
                        uint4 bf[2][6];
                        auto rdb = [&](int t, uint4 (&f)[6]) __attribute__((always_inline)) {
#pragma unroll
                            for (int gl = 0; gl < 2; ++gl)
#pragma unroll
                                for (int ky = 0; ky < 3; ++ky)
                                    f[gl * 3 + ky] = *reinterpret_cast<const uint4*>(et + b_pix[ky] + (((2 * half + gl) ^ b_f[ky & 1]) << 4) + t * (4 * 10 * 64));
                        };
                        rdb(0, bf[0]);
                        rdb(1, bf[1]);
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int gl = 0; gl < 2; ++gl) {
                                f32x4 c4 = *reinterpret_cast<const f32x4*>(tbl + 128 + ((2 * half + gl) * 8 + 4 * (mj & 1)) * 4);
#pragma unroll
                                for (int ky = 0; ky < 3; ++ky) c4 = mfma16<F16>(af[gl][ky], bf[t][gl * 3 + ky], c4);
                                a4[gl][t] = c4;
                            }
                    }
                    if constexpr (STAMP) { asm volatile("s_nop 0" ::: "memory"); const uint64_t t2 = now(); stv[2] += t2 - tq; tq = t2; }
                    // SiLU, one rounding, pooled sums (the pool sees the stored activations), D into the tile's interior in place
                    float psum[2][4];
#pragma unroll
                    for (int gl = 0; gl < 2; ++gl) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) psum[gl][i] = 0.f;
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const f32x2_t v01 = silu_fast2(f32x2_t{a4[gl][t][0], a4[gl][t][1]}), v23 = silu_fast2(f32x2_t{a4[gl][t][2], a4[gl][t][3]});
                            const uint32_t pk0 = T16<F16>::pack2(v01.x, v01.y);
                            const uint32_t pk1 = T16<F16>::pack2(v23.x, v23.y);
                            psum[gl][0] = T16<F16>::dot2(pk0, one_lo, psum[gl][0]);
                            psum[gl][1] = T16<F16>::dot2(pk0, one_hi, psum[gl][1]);
                            psum[gl][2] = T16<F16>::dot2(pk1, one_lo, psum[gl][2]);
                            psum[gl][3] = T16<F16>::dot2(pk1, one_hi, psum[gl][3]);
                            *reinterpret_cast<uint2*>(et + d_pix + (((2 * half + gl) ^ d_f) << 4) + t * (4 * 10 * 64)) = make_uint2(pk0, pk1);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) dv[i] = *reinterpret_cast<const uint4*>(et + o_off[i]);
                    if constexpr (STAMP) { const uint64_t t2 = now(); stv[3] += t2 - tq; tq = t2; }
                    // pooled means: the lanes' sums over their two tiles -> dw_mm.h's butterfly over the 32 (pixel pair, pixel) lanes, / 64
                    // (through the wave's LDS scratch, one lane per channel walking the slots, this was 1 250-1 500 cycles of the tick)
                    pool_v = sel4(mn == 0, dwmm_pool_sum4(psum[0], xaddr, 1.0f / 64.0f), dwmm_pool_sum4(psum[1], xaddr, 1.0f / 64.0f));
                    if constexpr (STAMP) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const uint64_t t2 = now(); stv[4] += t2 - tq; tq = t2; }
                }
                // the pieces requested at the top of the tick have landed (and the previous sample's stores are done)
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if constexpr (STAMP) { if (T >= 1) stv[5] += now() - tq; }
            }
            flush(sbase + n - 1);
        }
    }
#undef ISB_MBF8R_SEGMENT
    if constexpr (STAMP) {
        if (p.stamps && blockIdx.x < 32 && lane == 0) {
            uint64_t* o = p.stamps + ((size_t)blockIdx.x * 12 + wave) * 8;
#pragma unroll
            for (int k = 0; k < 6; ++k) o[k] = stv[k];
        }
    }
}

// 16-bit weights [N][K] row-major -> MFMA fragment order for register streaming: groups of G 32-channel blocks, k16-major inside
// a group: dst[((grp * K/16 + s) * G + g) * 64 + lane] (16 bytes) = W[(grp G + g) 32 + (lane & 31)][16 s + 8 (lane >> 5) .. + 8]
__global__ void mb8_pack_frag_kernel(const uint16_t* w, uint4* dst, int N, int K, int G) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)N * K / 8;
    if (i >= total) return;
    const int lane = (int)(i & 63);
    size_t t = i >> 6;
    const int g = (int)(t % G); t /= G;
    const int nk16 = K / 16;
    const int s = (int)(t % nk16);
    const int grp = (int)(t / nk16);
    const int n = (grp * G + g) * 32 + (lane & 31), k = 16 * s + 8 * (lane >> 5);
    dst[i] = *reinterpret_cast<const uint4*>(w + (size_t)n * K + k);
}

// squeeze-excite FC1 weights f32 [cse][C] -> [C/256][64 c4][cse][4]: thread j of a chunk reads 16 contiguous bytes per step,
// neighbouring threads neighbouring addresses
__global__ void mb8_pack_se1_kernel(const float* w1, float* dst, int cse, int C) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)cse * C) return;
    const int e = (int)(i & 3);
    size_t t = i >> 2;
    const int j = (int)(t % cse); t /= cse;
    const int c4 = (int)(t & 63);
    const int kc = (int)(t >> 6);
    dst[i] = w1[(size_t)j * C + kc * 256 + c4 * 4 + e];
}

}  // namespace

int launch_mb8_pack_frag(const uint16_t* w, void* dst, int N, int K, int G, hipStream_t st) {
    if (N % (32 * G) != 0 || K % 16 != 0) {
        set_error("mb8_pack_frag: N=%d K=%d G=%d", N, K, G);
        return ISB_ERR_INVALID;
    }
    const size_t total = (size_t)N * K / 8;
    hipLaunchKernelGGL(mb8_pack_frag_kernel, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, st, w, reinterpret_cast<uint4*>(dst), N, K, G);
    ISB_LAUNCHED("mb8_pack_frag", st);
    return ISB_OK;
}

int launch_mb8_pack_se1(const float* w1, float* dst, int cse, int C, hipStream_t st) {
    if (C % 256 != 0) {
        set_error("mb8_pack_se1: C=%d is not a multiple of 256", C);
        return ISB_ERR_INVALID;
    }
    hipLaunchKernelGGL(mb8_pack_se1_kernel, dim3((unsigned)cdivz((size_t)cse * C, 256)), dim3(256), 0, st, w1, dst, cse, C);
    ISB_LAUNCHED("mb8_pack_se1", st);
    return ISB_OK;
}

int launch_mbfront8(const MbFront8Args& a, hipStream_t st) {
    if (a.B < 1 || !a.x || !a.w1p || !a.b1 || !a.dww || !a.dwb || !a.d || !a.pooled || a.cin != 384) {
        set_error("mbfront8: bad arguments (B=%d cin=%d; built for 384 -> 2304)", a.B, a.cin);
        return ISB_ERR_INVALID;
    }
    constexpr int NSL = Mf8<384>::NSL, LDSB = Mf8<384>::LDS;
    static DevOnce attr_set;
    if (attr_set.need()) {
        ISB_HIP(hipFuncSetAttribute((const void*)mbfront8_kernel<384, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
        ISB_HIP(hipFuncSetAttribute((const void*)mbfront8_kernel<384, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
        attr_set.mark();
    }
    const int Q = std::max(1, std::min(a.B, 512 / NSL));              // sample sequences: two workgroups per CU
    MbFront8Args aa = a;
    aa.exp = exp_flags();
    // round 6: producer / consumer waves, twelve per workgroup, one workgroup per CU (mbfront8r_kernel). a.form: 0 = the library's choice,
    // 1 = the first kernel, 2 = roles. ISB_MBF8_FORM overrides the choice (A/B runs, tests).
    static const int env_form = [] { const char* e = getenv("ISB_MBF8_FORM"); return e ? atoi(e) : 0; }();
    const int form = a.form ? a.form : (env_form ? env_form : 2);
    if (form == 2 && a.stamps) {
        if (!a.f16) { set_error("mbfront8r: the stamped instantiation is fp16"); return ISB_ERR_INVALID; }
        static DevOnce attr_s;
        if (attr_s.need()) {
            ISB_HIP(hipFuncSetAttribute((const void*)mbfront8r_kernel<384, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Mf8r<384>::LDS));
            attr_s.mark();
        }
        const int nslots = std::min(32, NSL * cdiv(a.B, 8));
        hipLaunchKernelGGL((mbfront8r_kernel<384, true, true>), dim3(8 * nslots), dim3(768), Mf8r<384>::LDS, st, aa);
        ISB_LAUNCHED("mbfront8r (stamped)", st);
        return ISB_OK;
    }
    if (form == 2) {
        static DevOnce attr_r;
        if (attr_r.need()) {
            ISB_HIP(hipFuncSetAttribute((const void*)mbfront8r_kernel<384, false>, hipFuncAttributeMaxDynamicSharedMemorySize, Mf8r<384>::LDS));
            ISB_HIP(hipFuncSetAttribute((const void*)mbfront8r_kernel<384, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Mf8r<384>::LDS));
            attr_r.mark();
        }
        const int nslots = std::min(32, NSL * cdiv(a.B, 8));
        if (a.f16) hipLaunchKernelGGL((mbfront8r_kernel<384, true>), dim3(8 * nslots), dim3(768), Mf8r<384>::LDS, st, aa);
        else hipLaunchKernelGGL((mbfront8r_kernel<384, false>), dim3(8 * nslots), dim3(768), Mf8r<384>::LDS, st, aa);
        ISB_LAUNCHED("mbfront8r", st);
        return ISB_OK;
    }
    if (a.f16) hipLaunchKernelGGL((mbfront8_kernel<384, true>), dim3(NSL * Q), dim3(256), LDSB, st, aa);
    else hipLaunchKernelGGL((mbfront8_kernel<384, false>), dim3(NSL * Q), dim3(256), LDSB, st, aa);
    ISB_LAUNCHED("mbfront8", st);
    return ISB_OK;
}

int mb8_proj_group(int cout) { return cout == 384 ? 2 : 4; }           // Mb8Shape::CBW (the packing of Mb8Block.w2p)

int launch_mb8_chain(const Mb8Args& a, hipStream_t st) {
#ifdef ISB_BUILD_PROBES
    if (a.B < 1 || a.nblocks < 1 || !a.blocks || !a.x || !a.out || !a.dscratch || (a.cin0 != 384 && a.cin0 != 640) ||
        a.dscratch_stride < (size_t)64 * 6 * 640 * 2) {
        set_error("mb8_chain: bad arguments (B=%d nblocks=%d cin0=%d)", a.B, a.nblocks, a.cin0);
        return ISB_ERR_INVALID;
    }
    static DevOnce attr_set;
    if (attr_set.need()) {
        ISB_HIP(hipFuncSetAttribute((const void*)mb8_chain_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, MB8_LDS));
        ISB_HIP(hipFuncSetAttribute((const void*)mb8_chain_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, MB8_LDS));
        attr_set.mark();
    }
    if (a.f16) hipLaunchKernelGGL(mb8_chain_kernel<true>, dim3(a.B), dim3(MB8_NT), MB8_LDS, st, a);
    else hipLaunchKernelGGL(mb8_chain_kernel<false>, dim3(a.B), dim3(MB8_NT), MB8_LDS, st, a);
    ISB_LAUNCHED("mb8_chain", st);
    return ISB_OK;
#else
    (void)a; (void)st;
    set_error("mb8_chain: the per-sample chain of the 8 x 8 stages is a probe (2x slower than the five launches); build with ISB_BUILD_PROBES=1");
    return ISB_ERR_INVALID;
#endif
}

}  // namespace isb
