// Pose-estimation stages around the backbone (all HBM/latency bound, no MFMA):
//   crop_params : bbox -> (new_K, R, H)            reference modules/hpe/utils/misc.py:243-296, hpe.py:85,96
//   warp        : homography gather, nearest/trunc modules/hpe/setup/6_create_image_transformation_onnx.py:23-56, hpe.py:97-100
//   hpe_post    : soft-argmax decode, FOV gate, absolute reconstruction, un-rotate, 32->122 joint
//                 expansion + selection            hpe.py:109-169, misc.py:141-220
// The reference does the geometry in float64 numpy on the host with two float32 islands (K and its
// inverse, new_K.astype(float32) and its inverse); the same dtypes are kept here, on the device, so
// no host round trip is needed between detector box and crop.
#include "isb_common.h"
#include "kernels.h"

namespace isb {

// ------------------------------------------------------------------------------------------
// crop_params: one thread per frame, float64
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void mat3_mul(const double* a, const double* b, double* c) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) c[i * 3 + j] = a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j] + a[i * 3 + 2] * b[2 * 3 + j];
}

__device__ __forceinline__ void mat3_inv(const double* m, double* o) {
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    const double id = 1.0 / det;
    o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

__global__ void crop_params_kernel(CropParamArgs p) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= p.B) return;
    const int box = p.n_aug > 0 ? b / p.n_aug : b, aug = p.n_aug > 0 ? b - box * p.n_aug : 0;
    const bool nobox = p.bbox[4 * box] < 0;             // select_person found nobody: keep the math finite
    const double x1 = nobox ? 0 : p.bbox[4 * box + 0], x2 = nobox ? 1 : p.bbox[4 * box + 1];
    const double y1 = nobox ? 0 : p.bbox[4 * box + 2], y2 = nobox ? 1 : p.bbox[4 * box + 3];
    const double* K = p.K;
    // numpy inverts the float32 K in float32 (one correctly rounded division per entry)
    const float fx = (float)K[0], fy = (float)K[4], cx = (float)K[2], cy = (float)K[5];
    const double i00 = (double)__fdiv_rn(1.0f, fx), i02 = (double)(-__fdiv_rn(cx, fx));
    const double i11 = (double)__fdiv_rn(1.0f, fy), i12 = (double)(-__fdiv_rn(cy, fy));
    const double mx = (x1 + x2) / 2, my = (y1 + y2) / 2;
    const double px[5] = {mx, mx, x2, mx, x1};
    const double py[5] = {my, y1, my, y2, my};
    double cam[5][2];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        cam[i][0] = px[i] * i00 + py[i] * 0.0 + i02;
        cam[i][1] = px[i] * 0.0 + py[i] * i11 + i12;
    }
    // look-at rotation towards the box centre, up = (0,-1,0)   (misc.py:223-236)
    double z[3] = {cam[0][0], cam[0][1], 1.0};
    const double zn = sqrt(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
    z[0] /= zn; z[1] /= zn; z[2] /= zn;
    double x[3] = {z[1] * 0.0 - z[2] * (-1.0), z[2] * 0.0 - z[0] * 0.0, z[0] * (-1.0) - z[1] * 0.0};
    double xn = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    if (xn == 0.0) { x[0] = z[2]; x[1] = 0.0; x[2] = -z[0]; xn = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]); }
    x[0] /= xn; x[1] /= xn; x[2] /= xn;
    const double y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
    const double R0[9] = {x[0], x[1], x[2], y[0], y[1], y[2], z[0], z[1], z[2]};
    double KR[9];
    mat3_mul(K, R0, KR);
    double pr[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double sx = cam[i + 1][0], sy = cam[i + 1][1];
        const double u = sx * KR[0] + sy * KR[1] + KR[2];
        const double v = sx * KR[3] + sy * KR[4] + KR[5];
        const double w = sx * KR[6] + sy * KR[7] + KR[8];
        pr[i][0] = u / w; pr[i][1] = v / w;
    }
    const double vs = sqrt((pr[0][0] - pr[2][0]) * (pr[0][0] - pr[2][0]) + (pr[0][1] - pr[2][1]) * (pr[0][1] - pr[2][1]));
    const double hs = sqrt((pr[1][0] - pr[3][0]) * (pr[1][0] - pr[3][0]) + (pr[1][1] - pr[3][1]) * (pr[1][1] - pr[3][1]));
    const double scale = 256.0 / fmax(vs, hs);
    double nK[9] = {K[0] * scale, K[1] * scale, 128.0, K[3] * scale, K[4] * scale, 128.0, 0.0, 0.0, 1.0};
    double R[9];
    if (p.n_aug > 0) {                                  // hpe.py:88-93
        const double s = p.aug_scale[aug];
        nK[0] *= s; nK[1] *= s; nK[3] *= s; nK[4] *= s;
        mat3_mul(p.aug_rotflip + 9 * aug, R0, R);
    } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = R0[i];
    }
    double M[9], Mi[9], H[9];
    mat3_mul(nK, R, M);
    mat3_inv(M, Mi);
    mat3_mul(K, Mi, H);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        p.H[9 * b + i] = (float)H[i];
        p.newK[9 * b + i] = nK[i];
        p.R[9 * b + i] = R[i];
    }
}

int launch_crop_params(const CropParamArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(crop_params_kernel, dim3(cdiv(a.B, 64)), dim3(64), 0, st, a);
    ISB_LAUNCHED("crop_params", st);
    return ISB_OK;
}

// ------------------------------------------------------------------------------------------
// warp: thread per output pixel; float32 arithmetic in the reference's op order, no contraction
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void warp_kernel(WarpArgs p) {
    const int b = blockIdx.y;
    const int fb = p.n_aug > 1 ? b / p.n_aug : b;        // source frame
    const int pix = blockIdx.x * 256 + threadIdx.x;      // 0..65535
    const int y = pix >> 8, x = pix & 255;
    // per workgroup, once: the eight homography quotients (the same for every pixel of a sample) and the 256 values an 8-bit channel can
    // take, each by the reference's own operations (H[i] / H[8] in f32; int / 255.0 in f64, cast to f32) -- the kernel ran AT the vector
    // ALU's issue rate with eight IEEE f32 divisions and three f64 divisions per pixel (valu_active_share 1.0, 94 us per 256 frames)
    __shared__ __attribute__((aligned(16))) float tq[8];
    __shared__ float u8f[256];
    {
        const float* H = p.H + 9 * b;
        if (threadIdx.x < 8) tq[threadIdx.x] = __fdiv_rn(H[threadIdx.x], H[8]);
        u8f[threadIdx.x] = (float)((double)threadIdx.x / 255.0);        // hpe.py:100
    }
    __syncthreads();
    const float4 tqa = *reinterpret_cast<const float4*>(tq), tqb = *reinterpret_cast<const float4*>(tq + 4);
    const float t0 = tqa.x, t1 = tqa.y, t2 = tqa.z, t3 = tqa.w, t4 = tqb.x, t5 = tqb.y, t6 = tqb.z, t7 = tqb.w;
    const float xf = (float)x, yf = (float)y;
    const float k = __fadd_rn(__fadd_rn(__fmul_rn(t6, xf), __fmul_rn(t7, yf)), 1.0f);
    const float xs = __fdiv_rn(__fadd_rn(__fadd_rn(__fmul_rn(t0, xf), __fmul_rn(t1, yf)), t2), k);
    const float ys = __fdiv_rn(__fadd_rn(__fadd_rn(__fmul_rn(t3, xf), __fmul_rn(t4, yf)), t5), k);
    // torch .int(): truncation toward zero; keep huge/NaN values out of range
    const bool finite = fabsf(xs) < 1.0e9f && fabsf(ys) < 1.0e9f;
    const int xi = finite ? (int)xs : -1, yi = finite ? (int)ys : -1;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
    if (xi >= 0 && xi < p.FW && yi >= 0 && yi < p.FH) {
        const uint8_t* s;
        bool in = true;
        if (p.roi) {
            const RoiDesc rd = p.roi[fb];
            const int xr = xi - rd.x0, yr = yi - rd.y0;
            in = xr >= 0 && xr < rd.w && yr >= 0 && yr < rd.h;
            s = p.frames + rd.off + ((size_t)(in ? yr : 0) * rd.w + (in ? xr : 0)) * 3;
        } else {
            s = p.frames + (((size_t)fb * p.FH + yi) * p.FW + xi) * 3;
        }
        if (in) {
        o0 = u8f[s[0]];
        o1 = u8f[s[1]];
        o2 = u8f[s[2]];
        }
    }
    float* d = p.crops + ((size_t)b * 65536 + pix) * 3;
    d[0] = o0; d[1] = o1; d[2] = o2;
}

// grid (pieces of 2048 16-byte chunks, B): chunk id -> (row, 16-byte piece of the row). Reads over PCIe pay a microsecond of
// latency each, so a thread requests all EIGHT of its chunks before it stores the first (one load in flight per thread moved
// 20 GB/s; the link does more than twice that).
__global__ __launch_bounds__(256) void roi_gather_kernel(const uint8_t* frames, const RoiDesc* roi, uint8_t* dst, int FH, int FW) {
    const int b = blockIdx.y;
    const RoiDesc rd = roi[b];
    const int cpr = (rd.w * 3) >> 4;                         // 16-byte pieces per row (w % 16 == 0)
    const int total = cpr * rd.h;
    const int first = blockIdx.x * 2048 + threadIdx.x;
    if (first >= total) return;
    const uint8_t* src = frames + ((size_t)b * FH + rd.y0) * FW * 3 + (size_t)rd.x0 * 3;
    uint8_t* out = dst + rd.off;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int id = first + 256 * k;
        const int idc = min(id, total - 1);
        const int row = idc / cpr, pc = idc - row * cpr;
        v[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src + (size_t)row * FW * 3 + pc * 16));
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int id = first + 256 * k;
        if (id < total) *reinterpret_cast<u32x4*>(out + (size_t)id * 16) = v[k];
    }
}

int launch_roi_gather(const uint8_t* frames_mapped, const RoiDesc* roi, uint8_t* dst, int B, int FH, int FW, hipStream_t st) {
    if ((FW * 3) % 16 != 0 || FW % 16 != 0) {
        set_error("roi_gather: frame rows must be multiples of 16 pixels (FW=%d)", FW);
        return ISB_ERR_INVALID;
    }
    const int max_chunks = (FW * 3 / 16) * FH;
    hipLaunchKernelGGL(roi_gather_kernel, dim3(cdiv(max_chunks, 2048), B), dim3(256), 0, st, frames_mapped, roi, dst, FH, FW);
    ISB_LAUNCHED("roi_gather", st);
    return ISB_OK;
}

int launch_warp(const WarpArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(warp_kernel, dim3(256, a.B), dim3(256), 0, st, a);
    ISB_LAUNCHED("warp", st);
    return ISB_OK;
}

// ------------------------------------------------------------------------------------------
// hpe_post: one wave per sample. lanes 0-31: joint j, 3D heatmap; lanes 32-63: joint j, 2D heatmap
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wsum32(double v) {
#pragma unroll
    for (int s = 16; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
    return v;
}

// symmetric 3x3 eigen-decomposition (cyclic Jacobi) -> pseudo-inverse solve, like lstsq's
// singular-value cut-off (misc.py:174, rcond=None)
__device__ void solve_sym3_pinv(const double A[6], const double rhs[3], double out[3]) {
    double a[3][3] = {{A[0], A[1], A[2]}, {A[1], A[3], A[4]}, {A[2], A[4], A[5]}};
    double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 12; ++sweep) {
        for (int pq = 0; pq < 3; ++pq) {
            const int pI = pq == 2 ? 1 : 0, qI = pq == 0 ? 1 : 2;
            if (fabs(a[pI][qI]) < 1e-300) continue;
            const double theta = (a[qI][qI] - a[pI][pI]) / (2.0 * a[pI][qI]);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < 3; ++k) {
                const double akp = a[k][pI], akq = a[k][qI];
                a[k][pI] = c * akp - s * akq; a[k][qI] = s * akp + c * akq;
            }
            for (int k = 0; k < 3; ++k) {
                const double apk = a[pI][k], aqk = a[qI][k];
                a[pI][k] = c * apk - s * aqk; a[qI][k] = s * apk + c * aqk;
            }
            for (int k = 0; k < 3; ++k) {
                const double vkp = v[k][pI], vkq = v[k][qI];
                v[k][pI] = c * vkp - s * vkq; v[k][qI] = s * vkp + c * vkq;
            }
        }
    }
    const double lmax = fmax(fmax(a[0][0], a[1][1]), a[2][2]);
    const double cut = lmax * (64.0 * 2.220446049250313e-16) * (64.0 * 2.220446049250313e-16);
    out[0] = out[1] = out[2] = 0.0;
    for (int i = 0; i < 3; ++i) {
        const double lam = a[i][i];
        if (lam <= cut) continue;
        const double proj = (v[0][i] * rhs[0] + v[1][i] * rhs[1] + v[2][i] * rhs[2]) / lam;
        out[0] += v[0][i] * proj; out[1] += v[1][i] * proj; out[2] += v[2][i] * proj;
    }
}

// 512 threads per frame: thread = (joint j, part of the heatmap). The soft-argmax sums run as 16 partial sums per
// joint (4 heatmap cells x 8 depths each, in (h, w, d) order) combined in part order -- a fixed order, the same for
// every frame and batch size; the geometry that follows runs on the first wave.
__global__ __launch_bounds__(512) void hpe_post_kernel(PostArgs p) {
    __shared__ double sh2d[32][2];
    __shared__ double pose32[32][3];
    __shared__ float red_mx[2][16][32];
    __shared__ double red_den[2][16][32];
    __shared__ double red_c[5][16][32];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, j = tid & 31, half = lane >> 5, part = tid >> 5;
    const float* lg = p.logits + (size_t)b * 64 * 288;

    float x3[4][8], x2d[4];                 // this thread's cells: hw = 4 * part + c
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int hw = 4 * part + c;
        x2d[c] = lg[hw * 288 + j];
#pragma unroll
        for (int d = 0; d < 8; ++d) x3[c][d] = lg[hw * 288 + 32 + d * 32 + j];
    }
    float m3 = -INFINITY, m2 = -INFINITY;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        m2 = fmaxf(m2, x2d[c]);
#pragma unroll
        for (int d = 0; d < 8; ++d) m3 = fmaxf(m3, x3[c][d]);
    }
    red_mx[0][part][j] = m3; red_mx[1][part][j] = m2;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) { m3 = fmaxf(m3, red_mx[0][q][j]); m2 = fmaxf(m2, red_mx[1][q][j]); }
    // exponentials once; 3D: softmax jointly over (h, w, d), hpe.py:114-129; 2D: over (h, w), hpe.py:131-146
    double den3 = 0, den2 = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        x2d[c] = expf(x2d[c] - m2);
        den2 += (double)x2d[c];
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            x3[c][d] = expf(x3[c][d] - m3);
            den3 += (double)x3[c][d];
        }
    }
    red_den[0][part][j] = den3; red_den[1][part][j] = den2;
    __syncthreads();
    den3 = 0; den2 = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) { den3 += red_den[0][q][j]; den2 += red_den[1][q][j]; }
    const float den3f = (float)den3, den2f = (float)den2;
    double s0 = 0, s1 = 0, s2 = 0, t0 = 0, t1 = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int hw = 4 * part + c, hh = hw >> 3, ww = hw & 7;
        const double r2 = (double)__fdiv_rn(x2d[c], den2f);
        t0 += r2 * ((double)ww / 7.0); t1 += r2 * ((double)hh / 7.0);
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const double r = (double)__fdiv_rn(x3[c][d], den3f);
            s0 += r * ((double)ww / 7.0); s1 += r * ((double)hh / 7.0); s2 += r * ((double)d / 7.0);
        }
    }
    red_c[0][part][j] = s0; red_c[1][part][j] = s1; red_c[2][part][j] = s2; red_c[3][part][j] = t0; red_c[4][part][j] = t1;
    __syncthreads();
    if (tid >= 64) return;                  // the rest is one wave's work: lanes 0-31 carry the 3D joint, 32-63 the 2D one
    double c0 = 0, c1 = 0, c2 = 0;
    if (half == 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) { c0 += red_c[0][q][j]; c1 += red_c[1][q][j]; c2 += red_c[2][q][j]; }
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) { c0 += red_c[3][q][j]; c1 += red_c[4][q][j]; }
        sh2d[j][0] = c0 * 255.0; sh2d[j][1] = c1 * 255.0;
    }
    __syncthreads();
    const double x2 = sh2d[j][0], y2 = sh2d[j][1];
    if (p.dbg && half == 0) {
        double* d = p.dbg + ((size_t)b * 32 + j) * 5;
        d[0] = x2; d[1] = y2; d[2] = c0; d[3] = c1; d[4] = c2;
    }
    // FOV mask (misc.py:212-220) and visibility gate (hpe.py:149-153)
    const bool infov = x2 >= 18.0 && x2 <= 238.0 && y2 >= 18.0 && y2 <= 238.0;
    const unsigned long long bal = __ballot(infov && half == 0);
    const int nvis = __popcll(bal);
    const bool ok = nvis >= 8 && !(p.bbox && p.bbox[4 * b] < 0);
    if (lane == 0) p.valid[b] = ok ? 1 : 0;

    // reconstruct_absolute (misc.py:183-204): inverse of new_K.astype(float32) in float32
    const double* nK = p.newK + 9 * b;
    const float f0 = (float)nK[0], f1 = (float)nK[4], pp0 = (float)nK[2], pp1 = (float)nK[5];
    const double i00 = (double)__fdiv_rn(1.0f, f0), i02 = (double)(-__fdiv_rn(pp0, f0));
    const double i11 = (double)__fdiv_rn(1.0f, f1), i12 = (double)(-__fdiv_rn(pp1, f1));
    const double nx = x2 * i00 + i02, ny = y2 * i11 + i12;
    // rms scales over the 64 interleaved coordinates (misc.py:156-168)
    const double rbx = nx * c2 - c0, rby = ny * c2 - c1;
    const double s2d = sqrt(wsum32(nx * nx + ny * ny) / 64.0);
    const double srb = sqrt(wsum32(rbx * rbx + rby * rby) / 64.0);
    const double xh = nx / s2d, yh = ny / s2d, bx = rbx / srb, by = rby / srb;
    const double wj = (double)((infov ? 1.0f : 0.0f) + 1e-4f);
    const double w2 = wj * wj;
    double A[6], rhs[3], ref[3];
    A[0] = wsum32(w2); A[1] = 0.0; A[2] = wsum32(-w2 * xh);
    A[3] = A[0]; A[4] = wsum32(-w2 * yh); A[5] = wsum32(w2 * (xh * xh + yh * yh));
    rhs[0] = wsum32(w2 * bx); rhs[1] = wsum32(w2 * by); rhs[2] = wsum32(-w2 * (xh * bx + yh * by));
    solve_sym3_pinv(A, rhs, ref);
    ref[0] *= srb; ref[1] *= srb; ref[2] = ref[2] / s2d * srb;
    double ax, ay, az;
    if (infov) { const double dz = c2 + ref[2]; ax = nx * dz; ay = ny * dz; az = dz; }
    else { ax = c0 + ref[0]; ay = c1 + ref[1]; az = c2 + ref[2]; }
    const double* R = p.R + 9 * b;                          // pred3d @ homo_inv, hpe.py:159
    if (half == 0) {
        pose32[j][0] = ax * R[0] + ay * R[3] + az * R[6];
        pose32[j][1] = ax * R[1] + ay * R[4] + az * R[7];
        pose32[j][2] = ax * R[2] + ay * R[5] + az * R[8];
    }
    __syncthreads();
    // joint expansion 32 -> 122 and selection (hpe.py:162-164)
    for (int q = lane; q < p.n_out; q += 64) {
        const int src = p.indices ? p.indices[q] : q;
        double o0 = 0, o1 = 0, o2 = 0;
        for (int jj = 0; jj < 32; ++jj) {
            const double w = (double)p.expand[jj * 122 + src];
            o0 += w * pose32[jj][0]; o1 += w * pose32[jj][1]; o2 += w * pose32[jj][2];
        }
        float* o = p.joints + ((size_t)b * p.n_out + q) * 3;
        o[0] = ok ? (float)o0 : 0.f; o[1] = ok ? (float)o1 : 0.f; o[2] = ok ? (float)o2 : 0.f;
    }
}

int launch_hpe_post(const PostArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(hpe_post_kernel, dim3(a.B), dim3(512), 0, st, a);
    ISB_LAUNCHED("hpe_post", st);
    return ISB_OK;
}

// ------------------------------------------------------------------------------------------
// select_person: detector post-processing reduced to what estimate() consumes (hpe.py:59-79).
// The reference thresholds the per-anchor max class confidence, keeps class 0 (person), runs NMS
// (misc.py:27-107) and then takes the most confident survivor -- which NMS can never remove (it is
// the first box NMS keeps), so the result is the arg-max-confidence person anchor. One WG per frame.
// bbox out = (x1, x2, y1, y2) = int(coord * size) clamped at 0, float32 product like numpy 2;
// (-1,-1,-1,-1) when no person is found (estimate() returns None, hpe.py:72-73).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void select_person_kernel(const float* boxes, const float* confs, int n_anchor, int n_cls, float thresh,
                                                            int width, int height, int32_t* bbox, uint8_t* found) {
    __shared__ float sc[256];
    __shared__ int si[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    float best = -1.f;
    int besti = -1;
    for (int a = tid; a < n_anchor; a += 256) {
        const float* c = confs + ((size_t)b * n_anchor + a) * n_cls;
        float mx = c[0];
        int id = 0;
        for (int k = 1; k < n_cls; ++k)
            if (c[k] > mx) { mx = c[k]; id = k; }          // np.argmax: first maximum
        if (id == 0 && mx > thresh && mx > best) { best = mx; besti = a; }
    }
    sc[tid] = best; si[tid] = besti;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (tid < s) {
            const float o = sc[tid + s];
            const int oi = si[tid + s];
            if (oi >= 0 && (si[tid] < 0 || o > sc[tid] || (o == sc[tid] && oi < si[tid]))) { sc[tid] = o; si[tid] = oi; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const int a = si[0];
        int32_t* o = bbox + 4 * b;
        if (a < 0) {
            o[0] = o[1] = o[2] = o[3] = -1;
            if (found) found[b] = 0;
        } else {
            const float* bx = boxes + ((size_t)b * n_anchor + a) * 4;
            const int x1 = (int)__fmul_rn(bx[0], (float)width), y1 = (int)__fmul_rn(bx[1], (float)height);
            const int x2 = (int)__fmul_rn(bx[2], (float)width), y2 = (int)__fmul_rn(bx[3], (float)height);
            o[0] = x1 > 0 ? x1 : 0; o[1] = x2 > 0 ? x2 : 0; o[2] = y1 > 0 ? y1 : 0; o[3] = y2 > 0 ? y2 : 0;
            if (found) found[b] = 1;
        }
    }
}

int launch_select_person(const float* boxes, const float* confs, int B, int n_anchor, int n_cls, float thresh, int width, int height,
                         int32_t* bbox, uint8_t* found, hipStream_t st) {
    hipLaunchKernelGGL(select_person_kernel, dim3(B), dim3(256), 0, st, boxes, confs, n_anchor, n_cls, thresh, width, height, bbox, found);
    ISB_LAUNCHED("select_person", st);
    return ISB_OK;
}

// ------------------------------------------------------------------------------------------
// pose_windows: root-centre (main.py:103) + flatten (main.py:105) + sliding windows (ar.py:42-50)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pose_windows_kernel(const float* joints, int n_cam, int n_frames, int J, int L, float* windows) {
    const int nw = n_frames - L + 1;
    const size_t total = (size_t)n_cam * nw * L * J * 3;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int d3 = J * 3;
    const int e = (int)(idx % d3);
    size_t rest = idx / d3;
    const int l = (int)(rest % L); rest /= L;
    const int w = (int)(rest % nw);
    const int cam = (int)(rest / nw);
    const float* pose = joints + ((size_t)cam * n_frames + w + l) * d3;
    windows[idx] = pose[e] - pose[e % 3];
}

// pose_distance: the frame's "distance" element (main.py:102): ||pose[0]|| * 2.5 on the camera-frame root joint. The reference
// evaluates it in float64 on the float64 pose the estimator returns (hpe.py:171: the float32 joints widened): same here.
__global__ __launch_bounds__(256) void pose_distance_kernel(const float* joints, int n, int J, float* distance) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* r = joints + (size_t)i * J * 3;
    const double x = (double)r[0], y = (double)r[1], z = (double)r[2];
    distance[i] = (float)(sqrt(x * x + y * y + z * z) * 2.5);
}

int launch_pose_distance(const float* joints, int n, int J, float* distance, hipStream_t st) {
    hipLaunchKernelGGL(pose_distance_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, joints, n, J, distance);
    ISB_LAUNCHED("pose_distance", st);
    return ISB_OK;
}

int launch_pose_windows(const float* joints, int n_cam, int n_frames, int J, int L, float* windows, hipStream_t st) {
    const size_t total = (size_t)n_cam * (n_frames - L + 1) * L * J * 3;
    hipLaunchKernelGGL(pose_windows_kernel, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, st, joints, n_cam, n_frames, J, L, windows);
    ISB_LAUNCHED("pose_windows", st);
    return ISB_OK;
}

}  // namespace isb
