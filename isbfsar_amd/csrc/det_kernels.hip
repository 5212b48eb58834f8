// YOLOv4 person detector (reference `yolo.engine`, modules/hpe/hpe.py:42,51-60): the pieces around the convolutions.
//   det_preprocess : u8 BGR frame -> 256 x 256 area resize (cv2 INTER_AREA, hpe.py:51) -> RGB, / 255 (hpe.py:53-56), f32 NHWC
//   det_stem       : conv 3x3 stride 1 pad 1, 3 -> 32, folded BN, Mish, in f32 on the f32 image -> bf16 NHWC
//   concat         : channel concatenation of two NHWC tensors, the second optionally 2x nearest-upsampled (PANet routes)
//   spp            : cat[maxpool13, maxpool9, maxpool5, x] (stride 1, same padding)
//   yolo_decode    : the YOLO layer of the public implementation in inference mode (boxes + class confidences)
// The 109 other convolutions run on the conv_igemm kernel family (conv_dispatch.hip and the conv_*.hip families) with Mish / LeakyReLU epilogues.
#include "isb_common.h"
#include "kernels.h"

namespace isb {

__device__ __forceinline__ float bf2f_d(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t f2bf_d(float x) { return __builtin_bit_cast(uint16_t, (__bf16)x); }
// x * tanh(softplus(x)) with n = e^x: tanh(ln(1 + n)) = n (n + 2) / (n (n + 2) + 2)  (conv_common.h mish_fast)
__device__ __forceinline__ float mish_stem(float x) {
    const float n = __builtin_amdgcn_exp2f(1.4426950408889634f * fminf(x, 20.0f));
    const float w = n * (n + 2.0f);
    return x * w * __builtin_amdgcn_rcpf(w + 2.0f);
}

// ------------------------------------------------------------------------------------------
// area resize: output pixel (oy, ox) = area-weighted mean of the source rectangle [oy sy, (oy + 1) sy) x [ox sx, (ox + 1) sx),
// separable: rows first (ascending y), then columns (ascending x), float32 sums with float32(overlap / scale) weights;
// rounded to the nearest integer (ties to even) like cv2's uint8 output, then RGB order and / 255.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void det_preprocess_kernel(const uint8_t* frames, int B, int FH, int FW, float* out) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)B * 256 * 256) return;
    const int ox = (int)(idx & 255), oy = (int)((idx >> 8) & 255), b = (int)(idx >> 16);
    const double sx = (double)FW / 256.0, sy = (double)FH / 256.0;
    const double xlo = ox * sx, xhi = (ox + 1) * sx, ylo = oy * sy, yhi = (oy + 1) * sy;
    const int x0 = (int)floor(xlo), x1 = min((int)ceil(xhi), FW), y0 = (int)floor(ylo), y1 = min((int)ceil(yhi), FH);
    const uint8_t* f = frames + (size_t)b * FH * FW * 3;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int x = x0; x < x1; ++x) {
        const float wx = (float)(fmax(0.0, fmin(xhi, (double)(x + 1)) - fmax(xlo, (double)x)) / sx);
        float col[3] = {0.f, 0.f, 0.f};
        for (int y = y0; y < y1; ++y) {
            const float wy = (float)(fmax(0.0, fmin(yhi, (double)(y + 1)) - fmax(ylo, (double)y)) / sy);
            const uint8_t* px = f + ((size_t)y * FW + x) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) col[c] = __fadd_rn(col[c], __fmul_rn(wy, (float)px[c]));
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = __fadd_rn(acc[c], __fmul_rn(wx, col[c]));
    }
    float* o = out + idx * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = fminf(fmaxf(rintf(acc[2 - c]), 0.f), 255.f);        // BGR -> RGB
        o[c] = __fdiv_rn(v, 255.0f);
    }
}

int launch_det_preprocess(const uint8_t* frames, int B, int FH, int FW, float* out, hipStream_t st) {
    hipLaunchKernelGGL(det_preprocess_kernel, dim3((unsigned)cdivz((size_t)B * 65536, 256)), dim3(256), 0, st, frames, B, FH, FW, out);
    ISB_LAUNCHED("det_preprocess", st);
    return ISB_OK;
}

// ------------------------------------------------------------------------------------------
// first convolution: 3 -> 32, 3x3, stride 1, pad 1, bias (folded BN), Mish. thread = one output pixel, all 32 channels.
// ------------------------------------------------------------------------------------------
// (round 5, as the pose backbone's stem_kernel: 570 us per 256 frames at the vector ALU's issue rate before) channel PAIRS on v_pk_fma_f32
// with pair-major scalar weights -- the bits of the scalar chain --, unconditional tap loads from clamped addresses with the select on the
// value, 32-bit indexing on a (pixels / 256, B) grid, 1-KiB pixel rows stored through LDS.
__global__ __launch_bounds__(256) void det_stem_kernel(StemArgs p) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const unsigned pix = blockIdx.x * 256u + threadIdx.x;
    const unsigned npix = (unsigned)(p.H * p.W);
    if (pix - (threadIdx.x & 63) >= npix) return;                  // (whole waves only: a wave's lanes store for one another)
    const unsigned pixc = min(pix, npix - 1u);
    const int b = blockIdx.y, oy = (int)(pixc / (unsigned)p.W), ox = (int)(pixc - (unsigned)oy * (unsigned)p.W);
    const size_t idx = (size_t)b * npix + pix;
    float x[27];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = oy - 1 + ky, ix = ox - 1 + kx;
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const float* src = p.in + ((size_t)(b * p.H + min(max(iy, 0), p.H - 1)) * p.W + min(max(ix, 0), p.W - 1)) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = src[c];
                x[(ky * 3 + kx) * 3 + c] = ok ? v : 0.f;
            }
        }
    const f32x2_t* w2 = reinterpret_cast<const f32x2_t*>(p.wt);
    const f32x2_t* b2 = reinterpret_cast<const f32x2_t*>(p.bias);
    uint32_t o[16];
#pragma unroll
    for (int c2 = 0; c2 < 16; ++c2) {                  // one channel pair at a time: its 27 weight pairs fit the scalar registers
        f32x2_t acc = b2[c2];
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            const f32x2_t xk = {x[k], x[k]};
            acc = __builtin_elementwise_fma(xk, w2[c2 * 27 + k], acc);
        }
        // Mish as every other layer computes it (one v_exp + one v_rcp; with the exact library functions -- expf, log1pf, tanhf
        // per output -- this launch took 570 us per 64 frames, 12 % of the detector)
        o[c2] = (uint32_t)f2bf_d(mish_stem(acc.x)) | ((uint32_t)f2bf_d(mish_stem(acc.y)) << 16);
    }
    __shared__ __attribute__((aligned(16))) unsigned char tile[256 * 64];
    const int t = threadIdx.x, sw = (t >> 2) & 3;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        *reinterpret_cast<uint4*>(tile + t * 64 + ((c ^ sw) << 4)) = make_uint4(o[4 * c], o[4 * c + 1], o[4 * c + 2], o[4 * c + 3]);
    const int lane = t & 63, w0 = t & ~63, cc = lane & 3;
    uint16_t* const base = p.out + (idx - lane) * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pp = (lane >> 2) + 16 * i;
        const uint4 v = *reinterpret_cast<const uint4*>(tile + (w0 + pp) * 64 + ((cc ^ ((pp >> 2) & 3)) << 4));
        if (pix - lane + pp < npix) *reinterpret_cast<uint4*>(base + pp * 32 + cc * 8) = v;
    }
}

int launch_det_stem(const StemArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(det_stem_kernel, dim3((unsigned)cdiv(a.H * a.W, 256), (unsigned)a.B), dim3(256), 0, st, a);
    ISB_LAUNCHED("det_stem", st);
    return ISB_OK;
}

// ------------------------------------------------------------------------------------------
// out[b, y, x, 0:Ca] = a[b, y, x, :];  out[b, y, x, Ca:Ca+Cb] = b[b, y >> up, x >> up, :]   (bf16, 16-byte pieces)
// a / bsrc may be null: that side of `out` was written in place by its producer (ConvArgs.out_ld), only the other one is copied
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void concat_kernel(const uint16_t* a, const uint16_t* bsrc, uint16_t* out, int B, int H, int W, int Ca,
                                                     int Cb, int up) {
    const int p0 = a ? 0 : (Ca >> 3), p1 = bsrc ? ((Ca + Cb) >> 3) : (Ca >> 3);   // pieces of a pixel this launch copies
    const int pc = p1 - p0;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)B * H * W * pc) return;
    const int piece = p0 + (int)(idx % pc);
    const size_t pix = idx / pc;
    const int c = piece * 8;
    uint4 v;
    if (c < Ca) {
        v = *reinterpret_cast<const uint4*>(a + pix * Ca + c);
    } else {
        const int x = (int)(pix % W), y = (int)((pix / W) % H), bb = (int)(pix / ((size_t)W * H));
        const int hb = H >> up, wb = W >> up;
        v = *reinterpret_cast<const uint4*>(bsrc + (((size_t)bb * hb + (y >> up)) * wb + (x >> up)) * Cb + (c - Ca));
    }
    *reinterpret_cast<uint4*>(out + pix * (Ca + Cb) + c) = v;
}

int launch_concat(const uint16_t* a, const uint16_t* b, uint16_t* out, int B, int H, int W, int Ca, int Cb, int up, hipStream_t st) {
    if (Ca % 8 != 0 || Cb % 8 != 0 || (up && ((H | W) & 1))) {
        set_error("concat: channel counts must be multiples of 8 (Ca=%d Cb=%d) and an upsampled map even-sized", Ca, Cb);
        return ISB_ERR_INVALID;
    }
    if (!a && !b) return ISB_OK;                               // both sides written in place
    const size_t total = (size_t)B * H * W * (((a ? Ca : 0) + (b ? Cb : 0)) >> 3);
    hipLaunchKernelGGL(concat_kernel, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, st, a, b, out, B, H, W, Ca, Cb, up);
    ISB_LAUNCHED("concat", st);
    return ISB_OK;
}

// ------------------------------------------------------------------------------------------
// SPP: out[b, y, x, :] = [maxpool13(x), maxpool9(x), maxpool5(x), x]  (stride 1, out-of-map taps ignored = -inf padding)
// thread = (pixel, 8-channel piece); the three windows are nested, one sweep of the 13 x 13 neighbourhood feeds all.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spp_kernel(const uint16_t* in, uint16_t* out, int B, int H, int W, int C) {
    const int pc = C >> 3;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)B * H * W * pc) return;
    const int piece = (int)(idx % pc);
    const size_t pix = idx / pc;
    const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((size_t)W * H));
    float m5[8], m9[8], m13[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m5[e] = m9[e] = m13[e] = -3.0e38f;
    for (int dy = -6; dy <= 6; ++dy) {
        const int yy = y + dy;
        if ((unsigned)yy >= (unsigned)H) continue;
        for (int dx = -6; dx <= 6; ++dx) {
            const int xx = x + dx;
            if ((unsigned)xx >= (unsigned)W) continue;
            const uint4 v = *reinterpret_cast<const uint4*>(in + (((size_t)b * H + yy) * W + xx) * C + piece * 8);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            const int r = max(abs(dy), abs(dx));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = __uint_as_float(w[e] << 16), hi = __uint_as_float(w[e] & 0xffff0000u);
                m13[2 * e] = fmaxf(m13[2 * e], lo); m13[2 * e + 1] = fmaxf(m13[2 * e + 1], hi);
                if (r <= 4) { m9[2 * e] = fmaxf(m9[2 * e], lo); m9[2 * e + 1] = fmaxf(m9[2 * e + 1], hi); }
                if (r <= 2) { m5[2 * e] = fmaxf(m5[2 * e], lo); m5[2 * e + 1] = fmaxf(m5[2 * e + 1], hi); }
            }
        }
    }
    auto pack = [](const float (&m)[8]) {
        return make_uint4((__float_as_uint(m[0]) >> 16) | (__float_as_uint(m[1]) & 0xffff0000u), (__float_as_uint(m[2]) >> 16) | (__float_as_uint(m[3]) & 0xffff0000u),
                          (__float_as_uint(m[4]) >> 16) | (__float_as_uint(m[5]) & 0xffff0000u), (__float_as_uint(m[6]) >> 16) | (__float_as_uint(m[7]) & 0xffff0000u));
    };
    uint16_t* o = out + pix * (4 * C) + piece * 8;
    *reinterpret_cast<uint4*>(o) = pack(m13);
    *reinterpret_cast<uint4*>(o + C) = pack(m9);
    *reinterpret_cast<uint4*>(o + 2 * C) = pack(m5);
    *reinterpret_cast<uint4*>(o + 3 * C) = *reinterpret_cast<const uint4*>(in + pix * C + piece * 8);
}

int launch_spp(const uint16_t* in, uint16_t* out, int B, int H, int W, int C, hipStream_t st) {
    if (C % 8 != 0) {
        set_error("spp: C=%d must be a multiple of 8", C);
        return ISB_ERR_INVALID;
    }
    const size_t total = (size_t)B * H * W * (C >> 3);
    hipLaunchKernelGGL(spp_kernel, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, st, in, out, B, H, W, C);
    ISB_LAUNCHED("spp", st);
    return ISB_OK;
}

// ------------------------------------------------------------------------------------------
// YOLO layer, inference (yolo_forward_dynamic of the public implementation): map f32 [B, H, W, ldm] whose channels
// a * 85 + {0,1: txy, 2,3: twh, 4: objectness, 5..84: classes} belong to anchor a of this scale.
//   bxy = sigmoid(txy) * sxy - (sxy - 1) / 2 + grid;  bwh = exp(twh) * anchor / stride;  all / grid size
//   boxes (x1, y1, x2, y2) = (bx - bw / 2, by - bh / 2, x1 + bw, y1 + bh);  confs = sigmoid(cls) * sigmoid(obj)
// box index inside the scale = a * H * W + y * W + x. thread = (b, a, cell).
// ------------------------------------------------------------------------------------------
// thread = (box row, piece): piece 0 writes the box, pieces 1..20 four class confidences each (one thread per row walked 85
// strided channels and 84 exponentials: 116 us for the 32 x 32 scale at 64 frames)
__global__ __launch_bounds__(256) void yolo_decode_kernel(const float* map, int B, int H, int W, int ldm, float aw0, float ah0, float aw1,
                                                          float ah1, float aw2, float ah2, float sxy, float* boxes, float* confs,
                                                          int n_boxes, int box_off) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int piece = (int)(gid % 21);
    const size_t idx = gid / 21;
    if (idx >= (size_t)B * 3 * H * W) return;
    const int cell = (int)(idx % (H * W)), a = (int)((idx / (H * W)) % 3), b = (int)(idx / ((size_t)3 * H * W));
    const int y = cell / W, x = cell - y * W;
    const float* t = map + ((size_t)b * H * W + cell) * ldm + a * 85;
    auto sig = [](float v) { return 1.0f / (1.0f + expf(-v)); };
    const size_t row = (size_t)b * n_boxes + box_off + (size_t)a * H * W + cell;
    if (piece == 0) {
        const float aw = a == 0 ? aw0 : (a == 1 ? aw1 : aw2), ah = a == 0 ? ah0 : (a == 1 ? ah1 : ah2);
        const float off = 0.5f * (sxy - 1.0f);
        const float bx = (sig(t[0]) * sxy - off + (float)x) / (float)W;
        const float by = (sig(t[1]) * sxy - off + (float)y) / (float)H;
        const float bw = expf(t[2]) * aw / (float)W;
        const float bh = expf(t[3]) * ah / (float)H;
        const float x1 = bx - bw * 0.5f, y1 = by - bh * 0.5f;
        *reinterpret_cast<float4*>(boxes + row * 4) = make_float4(x1, y1, x1 + bw, y1 + bh);
        return;
    }
    const float det = sig(t[4]);
    const int k = (piece - 1) * 4;
    *reinterpret_cast<float4*>(confs + row * 80 + k) = make_float4(sig(t[5 + k]) * det, sig(t[6 + k]) * det, sig(t[7 + k]) * det, sig(t[8 + k]) * det);
}

int launch_yolo_decode(const float* map, int B, int H, int W, int ldm, const float* anchors_wh, float sxy, float* boxes, float* confs,
                       int n_boxes, int box_off, hipStream_t st) {
    const size_t total = (size_t)B * 3 * H * W * 21;
    hipLaunchKernelGGL(yolo_decode_kernel, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, st, map, B, H, W, ldm, anchors_wh[0], anchors_wh[1],
                       anchors_wh[2], anchors_wh[3], anchors_wh[4], anchors_wh[5], sxy, boxes, confs, n_boxes, box_off);
    ISB_LAUNCHED("yolo_decode", st);
    return ISB_OK;
}

}  // namespace isb
