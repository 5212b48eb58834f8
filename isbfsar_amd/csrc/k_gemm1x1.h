// The lean 1x1 stride-1 GEMM kernel of the convolution family (un-gated: conv_gemm1x1.hip; SE-gated / split-K with the folded
// squeeze-excite FC2: conv_gemm1x1_gate.hip).
#pragma once
#include "conv_tiles.h"

namespace isb {

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
    T16<F16>::enter();
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
    conv_epilogue<TM, TN, WGM, WGN, true, F16>(p, acc, lds, m0, n0, wm, wn, r, h, tid, bias_off);
    if constexpr (STAMPS) {
        const uint64_t t_end = __builtin_amdgcn_s_memtime();
        const int wgid = blockIdx.x + gridDim.x * blockIdx.y;
        if (tid == 0 && wgid < 64) {
            uint64_t* o = reinterpret_cast<uint64_t*>(p.part) + (size_t)wgid * 8;
            o[0] = st_loop0 - st_t0; o[1] = st_wait; o[2] = st_bar; o[3] = st_loop1 - st_loop0; o[4] = t_end - st_loop1; o[5] = (uint64_t)nkt;
        }
    }
}

}  // namespace isb
