// SE-gated projections of the 16 x 16 stage with the WEIGHTS STATIONARY IN REGISTERS (round 5; tile variant 157):
// 1344 -> 224 (18 launches per pass), 1152 -> 224, 768 -> 192 on batches.
//
// What bounds gemm1x1_dma_kernel<.., GATE> on these layers is the L2 -> LDS delivery of its operands (DESIGN.md section 3:
// 22 KiB per 32-channel k-step and workgroup, 14 of them WEIGHTS that every one of the 512 workgroups pulls in again; the k
// loop runs at 25 B/clk per CU and the matrix pipe at 20 % over the launch). Here a wave keeps the weights of its 32 output
// channels for ALL of K in registers (K / 4 registers: 336 of a lone wave's 512 for K = 1344; MFMA A operand), a workgroup
// of four waves (one per SIMD, one workgroup per CU) owns a 128-channel slice for its whole life and walks 128-row tiles:
// only the activations stream. They travel global -> registers -> LDS, not by LDS-DMA, because the squeeze-excite gate has
// to meet them on the way: each 16-byte chunk is scaled ONCE per CU, in the registers it arrives in (two v_fma_mix per
// dword: f16(f32(x) * g), the rounding of T16::gate8), where the tile kernel scales every A fragment it reads -- with one
// fragment per MFMA, as here, that would be 16 vector instructions per MFMA. Two 16-KiB step buffers (64 channels per
// step), one barrier per step; the requests run two steps ahead and across tile boundaries.
// Same k order and gate rounding as the tile kernel, same shared epilogue: bit-identical results (tested).
#include "conv_tiles.h"

namespace isb {

namespace {

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N)
template <int I0, int N, class F>
__device__ __attribute__((always_inline)) void wsk_static_for(F&& f) {
    if constexpr (I0 < N) {
        f(std::integral_constant<int, I0>{});
        wsk_static_for<I0 + 1, N>(f);
    }
}

// f16(f32(x.lo) * g_lo) | f16(f32(x.hi) * g_hi) << 16: one rounding of the f32 product per half, as T16<true>::gate2
__device__ __forceinline__ uint32_t wsk_gate2(uint32_t x, float g_lo, float g_hi) {
    uint32_t d = 0;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "+v"(d) : "v"(x), "v"(g_lo));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(x), "v"(g_hi));
    return d;
}

}  // namespace

// NKS = K / 64 (steps of 64 channels). fp16 storage (the default precision); bf16 layers keep the tile kernel.
template <int NKS>
__global__ __launch_bounds__(256, 1) void gemm1x1_wsk_kernel(ConvArgs p) {
    T16<true>::enter();
    constexpr int K = 64 * NKS, NK16 = 4 * NKS;
    constexpr int BM = 128;
    constexpr int STEP_BYTES = BM * 128;                    // one step: two [128 rows][64 B] images (channels 0-31 / 32-63 of the step)
    constexpr int STAGE_OFF = 2 * STEP_BYTES;               // the shared epilogue's staging area: 128 x (128 x 2 + 16) bytes
    unsigned char* const lds = conv_lds_dyn;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // workgroup -> (128-channel slice, tile sequence): the slices of one tile sequence sit on ONE XCD (ids 8 apart share an XCD), so
    // the activation rows they both stream are fetched into that L2 once
    const int nsl = p.grid_n, MS = p.grid_m;                // slices, tile sequences (launcher)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int slice = slot % nsl, mseq = (slot / nsl) * 8 + xcd;
    const int n0 = slice * 128 + wave * 32;
    const int ntiles = p.M / BM;
    if (mseq >= ntiles) return;
    const int ohw = p.OH * p.OW;

    // the wave's weights, for the whole kernel: fragment s = W[n0 + (lane & 31)][16 s + 8 h .. + 8]
    uint4 wfr[NK16];
    {
        const uint16_t* wrow = p.w + (size_t)min(n0 + r, p.Cout - 1) * K + 8 * h;
#pragma unroll
        for (int s = 0; s < NK16; ++s) wfr[s] = *reinterpret_cast<const uint4*>(wrow + 16 * s);
    }
    // staging of a step: thread = (row tid / 8 + 32 i, 16-byte chunk kc = tid & 7 of the step's 128 bytes), i = 0..3
    const int kc = tid & 7, srow = tid >> 3;
    const uint32_t st_off = (uint32_t)((kc >> 2) * (BM * 64) + swz(srow, kc & 3));        // (rows 32 i further: + 2048 i, the swizzle repeats every 16 rows)
    const unsigned char* const a_base = reinterpret_cast<const unsigned char*>(p.in) + (size_t)srow * (K * 2) + kc * 16;
    int a_sw[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) a_sw[ks] = swz(r, 2 * ks + h);

    struct Req {                                            // one step in flight: four activation chunks and the gates of their eight channels
        uint4 a[4];
        float4 g[2];
    };
    auto request = [&](Req& q, int tile, int step) __attribute__((always_inline)) {
        const unsigned char* src = a_base + (size_t)tile * BM * (K * 2) + step * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) q.a[i] = *reinterpret_cast<const uint4*>(src + (size_t)i * 32 * (K * 2));
        const float* gs = p.gate + (size_t)((tile * BM) / ohw) * K + step * 64 + kc * 8;
        q.g[0] = *reinterpret_cast<const float4*>(gs);
        q.g[1] = *reinterpret_cast<const float4*>(gs + 4);
    };
    auto deposit = [&](const Req& q, int buf) __attribute__((always_inline)) {             // gate in registers, then the step image
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint4 v;
            v.x = wsk_gate2(q.a[i].x, q.g[0].x, q.g[0].y);
            v.y = wsk_gate2(q.a[i].y, q.g[0].z, q.g[0].w);
            v.z = wsk_gate2(q.a[i].z, q.g[1].x, q.g[1].y);
            v.w = wsk_gate2(q.a[i].w, q.g[1].z, q.g[1].w);
            *reinterpret_cast<uint4*>(lds + buf * STEP_BYTES + st_off + i * 2048) = v;
        }
    };

    const bool live = n0 < p.Cout;                          // (224 outputs = 128 + 96: the second slice's fourth wave only stages and meets the barriers)
    Req q0, q1;
    int tile = mseq;
    request(q0, tile, 0);
    request(q1, tile, 1);
    deposit(q0, 0);
    __syncthreads();
    int par = 0;                                            // LDS buffer of the tile's step 0
    // one tile. On entry buffer par holds step 0, `qi` is in flight with step 1 and `qf` is free; the sets alternate per step. Returns
    // with the next tile's step 0 in LDS and its step 1 in flight -- in qi again if NKS is even, in qf if it is odd.
    auto run_tile = [&](Req& qf, Req& qi) __attribute__((always_inline)) {
        const int next = tile + MS;
        const bool more = next < ntiles;
        f32x16 acc[4][1];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][0][e] = 0.f;
        auto step_body = [&](auto sc, Req& qin, Req& qout) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value;
            const int buf = (par + s) & 1;
            // request step s + 2 (into the set whose data went to LDS during the previous step)
            if (s + 2 < NKS) request(qout, tile, s + 2);
            else if (more) request(qout, next, s + 2 - NKS);
            if (live) {
                // fragment reads one k16 step ahead of the MFMAs (second register set; sched_group_barrier keeps the order): a lone
                // wave has nobody to hide an LDS round trip behind
                uint4 af[2][4];
                auto rd = [&](int ks4, uint4 (&f)[4]) __attribute__((always_inline)) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        f[i] = *reinterpret_cast<const uint4*>(lds + buf * STEP_BYTES + (ks4 >> 1) * (BM * 64) + a_sw[ks4 & 1] + i * 2048);
                };
                rd(0, af[0]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks4 = 0; ks4 < 4; ++ks4) {
                    if (ks4 + 1 < 4) rd(ks4 + 1, af[(ks4 + 1) & 1]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][0] = T16<true>::mfma32(wfr[4 * s + ks4], af[ks4 & 1][i], acc[i][0]);
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // step s + 1: gated and written to the other buffer (its readers finished before the last barrier)
            if (s + 1 < NKS || more) deposit(qin, buf ^ 1);
            __syncthreads();
        };
        wsk_static_for<0, NKS / 2>([&](auto pc) __attribute__((always_inline)) {
            constexpr int s = 2 * decltype(pc)::value;
            step_body(std::integral_constant<int, s>{}, qi, qf);
            step_body(std::integral_constant<int, s + 1>{}, qf, qi);
        });
        if constexpr (NKS % 2 == 1) {
            step_body(std::integral_constant<int, NKS - 1>{}, qi, qf);
            par ^= 1;
        }
        conv_epilogue<4, 1, 1, 4, false, true>(p, acc, lds + STAGE_OFF, tile * BM, slice * 128, 0, wave, r, h, tid);
        tile = next;
    };
    while (tile < ntiles) {
        run_tile(q0, q1);
        if constexpr (NKS % 2 == 1) {
            if (tile < ntiles) run_tile(q1, q0);            // odd step count: the two register sets have traded roles
        }
    }
}

constexpr int WSK_LDS = 2 * 128 * 128 + 128 * (128 * 2 + 16);

// tile variant 157 (launch_conv_igemm): the gated fp16 projections with K = 1344 / 1152 / 768
int launch_conv_wsk(const ConvArgs& a, ConvArgs& aa, hipStream_t st) {
    const int ohw = a.OH * a.OW;
    if (!a.gate || !a.f16 || a.KH != 1 || a.stride != 1 || a.pad != 0 || a.splits > 1 || a.out_f32 || a.out_ld || a.act || a.act_after_res ||
        (a.Cin != 1344 && a.Cin != 1152 && a.Cin != 768) || a.Cout % 32 != 0 || a.M % 128 != 0 || (ohw % 128 != 0 && 128 % ohw != 0) ||
        ohw < 128 || (size_t)a.M * a.Cin * 2 >= 0xffffffffull) {
        set_error("conv_igemm: variant 157 is the gated fp16 1x1 projection with 1344 / 1152 / 768 input channels on maps of >= 128 pixels, M %% 128 == 0");
        return ISB_ERR_INVALID;
    }
    const int nsl = cdiv(a.Cout, 128);
    int n_cu = 0, dev = 0;
    ISB_HIP(hipGetDevice(&dev));
    ISB_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    const int ntiles = a.M / 128;
    // one workgroup per CU: tile sequences in groups of 8 (one per XCD), nsl slices each
    int groups = std::max(1, n_cu / (8 * nsl));
    groups = std::min(groups, cdiv(ntiles, 8));
    aa.grid_n = nsl;
    aa.grid_m = groups * 8;
    const dim3 g(groups * 8 * nsl);
#define ISB_WSK(NKS_)                                                                                                        \
    do {                                                                                                                     \
        static DevOnce attr_set;                                                                                             \
        if (attr_set.need()) {                                                                                               \
            ISB_HIP(hipFuncSetAttribute((const void*)gemm1x1_wsk_kernel<NKS_>, hipFuncAttributeMaxDynamicSharedMemorySize, WSK_LDS)); \
            attr_set.mark();                                                                                                 \
        }                                                                                                                    \
        hipLaunchKernelGGL((gemm1x1_wsk_kernel<NKS_>), g, dim3(256), WSK_LDS, st, aa);                                       \
    } while (0)
    if (a.Cin == 1344) ISB_WSK(21);
    else if (a.Cin == 1152) ISB_WSK(18);
    else ISB_WSK(12);
#undef ISB_WSK
    return ISB_OK;
}

}  // namespace isb
