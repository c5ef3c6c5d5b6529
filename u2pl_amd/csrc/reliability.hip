// Reliability split kernels:
//   bilinear up-sampling (align_corners=True, torch-CPU FMA form, bit-exact),
//   teacher pseudo label, per-pixel softmax entropy, EXACT order-statistic
//   selection (4-pass 8-bit radix select on the fp32 bit pattern) with numpy's
//   float32 percentile lerp, threshold masks, legacy-nearest down-sampling and
//   the label_onehot batch-slot-0 quirk packed as per-pixel class bitmasks.
// All kernels are HBM-bound; reads/writes are lane-contiguous along W.
#include "common.h"
#include "u2pl_hip.h"
#include <stdlib.h>

// ---------------------------------------------------------------------------
// a7: out[n][c][oy][ox] (NCHW contiguous) from a strided low-res tensor.
// One thread per output pixel, channel loop inside: stores are coalesced per
// class plane, the low-res source (<= 11 MB) stays L2 resident.
// ---------------------------------------------------------------------------
__global__ void k_bilinear_up(const float* __restrict__ in, long sn, long sc, long sh, long sw,
                              int N, int C, int h, int w, float* __restrict__ out, int H, int W,
                              float sy, float sx) {
    long total = (long)N * H * W;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < total;
         p += (long)gridDim.x * blockDim.x) {
        int ox = (int)(p % W);
        long t = p / W;
        int oy = (int)(t % H);
        int n = (int)(t / H);
        AcCoord cy = ac_coord(oy, sy, h), cx = ac_coord(ox, sx, w);
        const float* b = in + n * sn;
        long o00 = cy.i0 * sh + cx.i0 * sw, o01 = cy.i0 * sh + cx.i1 * sw;
        long o10 = cy.i1 * sh + cx.i0 * sw, o11 = cy.i1 * sh + cx.i1 * sw;
        float* o = out + ((long)n * C * H + oy) * W + ox;
        for (int c = 0; c < C; ++c) {
            const float* bc = b + c * sc;
            float top = __fmaf_rn(cx.l0, bc[o00], __fmul_rn(cx.l1, bc[o01]));
            float bot = __fmaf_rn(cx.l0, bc[o10], __fmul_rn(cx.l1, bc[o11]));
            o[(long)c * H * W] = __fmaf_rn(cy.l0, top, __fmul_rn(cy.l1, bot));
        }
    }
}

U2PL_API int u2pl_bilinear_up_f32(const float* in, long sn, long sc, long sh, long sw, int N, int C,
                                  int h, int w, float* out, int H, int W, hipStream_t stream) {
    if (N <= 0 || C <= 0) return 0;
    long total = (long)N * H * W;
    U2PL_LAUNCH(k_bilinear_up, dim3(grid_for(total, 256)), dim3(256), 0, stream, in, sn, sc, sh,
                       sw, N, C, h, w, out, H, W, ac_scale_host(h, H), ac_scale_host(w, W));
    U2PL_LAUNCH_CHECK();
    return 0;
}

// Backward of a7 (student branches): gather form, deterministic.  One thread
// per low-res element (n,c,iy,ix) sums the contributions of the output pixels
// that reference it, re-deriving the forward indices/lambdas exactly.
__global__ void k_bilinear_up_bwd(const float* __restrict__ gout, int N, int C, int H, int W,
                                  float* __restrict__ gin, long sn, long sc, long sh, long sw, int h,
                                  int w, float sy, float sx) {
    long total = (long)N * C * h * w;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < total;
         p += (long)gridDim.x * blockDim.x) {
        int ix = (int)(p % w);
        long t = p / w;
        int iy = (int)(t % h);
        t /= h;
        int c = (int)(t % C);
        int n = (int)(t / C);
        // generous candidate output range, filtered exactly below
        int oy_lo = sy > 0 ? (int)floorf((iy - 1) / sy) - 1 : 0, oy_hi = sy > 0 ? (int)ceilf((iy + 1) / sy) + 1 : H - 1;
        int ox_lo = sx > 0 ? (int)floorf((ix - 1) / sx) - 1 : 0, ox_hi = sx > 0 ? (int)ceilf((ix + 1) / sx) + 1 : W - 1;
        oy_lo = max(oy_lo, 0); ox_lo = max(ox_lo, 0);
        oy_hi = min(oy_hi, H - 1); ox_hi = min(ox_hi, W - 1);
        const float* g = gout + ((long)n * C + c) * H * W;
        float acc = 0.f;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            AcCoord cy = ac_coord(oy, sy, h);
            if (cy.i0 != iy && cy.i1 != iy) continue;
            float wy = (cy.i0 == iy ? cy.l0 : 0.f) + (cy.i1 == iy ? cy.l1 : 0.f);
            float racc = 0.f;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                AcCoord cx = ac_coord(ox, sx, w);
                if (cx.i0 != ix && cx.i1 != ix) continue;
                float wx = (cx.i0 == ix ? cx.l0 : 0.f) + (cx.i1 == ix ? cx.l1 : 0.f);
                racc += wx * g[(long)oy * W + ox];
            }
            acc += wy * racc;
        }
        gin[n * sn + c * sc + iy * sh + ix * sw] = acc;
    }
}

U2PL_API int u2pl_bilinear_up_bwd_f32(const float* gout, int N, int C, int H, int W, float* gin, long sn,
                                      long sc, long sh, long sw, int h, int w, hipStream_t stream) {
    if (N <= 0 || C <= 0) return 0;
    long total = (long)N * C * h * w;
    U2PL_LAUNCH(k_bilinear_up_bwd, dim3(grid_for(total, 256, 1 << 20)), dim3(256), 0, stream, gout, N, C, H,
                       W, gin, sn, sc, sh, sw, h, w, ac_scale_host(h, H), ac_scale_host(w, W));
    U2PL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// a8: softmax max / argmax of full-res logits (train_semi.py:323-324)
// ---------------------------------------------------------------------------
__global__ void k_pseudo_label(const float* __restrict__ z, int N, int C, long HW,
                               float* __restrict__ conf, long long* __restrict__ label) {
    long total = (long)N * HW;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < total;
         p += (long)gridDim.x * blockDim.x) {
        long n = p / HW, q = p % HW;
        const float* b = z + n * C * HW + q;
        float m = b[0];
        int am = 0;
        for (int c = 1; c < C; ++c) {
            float v = b[(long)c * HW];
            if (v > m) { m = v; am = c; }
        }
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(b[(long)c * HW] - m);
        conf[p] = 1.0f / s;
        label[p] = am;
    }
}

U2PL_API int u2pl_pseudo_label_f32(const float* logits, int N, int C, int H, int W, float* conf,
                                   long long* label, hipStream_t stream) {
    long total = (long)N * H * W;
    if (total <= 0) return 0;
    U2PL_LAUNCH(k_pseudo_label, dim3(grid_for(total, 256)), dim3(256), 0, stream, logits, N, C,
                       (long)H * W, conf, label);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// a11/a12: per-pixel entropy of the teacher logits (train_semi.py:402-403,
// loss_helper.py:35-36).  NaN marks label==ignore pixels so one float stream
// carries value and validity.  The kernel also counts the valid pixels
// (ws[0]) and builds the pass-0 histogram of the radix select (ws[128..]) so
// the selection needs no extra sweep.
//   entropy = log(s) - sum_c e_c*(z_c-m)/s,  e_c = exp(z_c-m), s = sum e_c
// (= -sum p*log(p); the reference's +1e-10 inside the log changes the value by
//  < C*1e-10, far below the Tier-B tolerance 2e-6, and needs one exp per class
//  instead of two exps, a log and a divide).
// ---------------------------------------------------------------------------
#define SEL_H0 128
__device__ __forceinline__ void hist0_flush(unsigned* sh, unsigned cnt, unsigned* __restrict__ ws) {
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += blockDim.x)
        if (sh[i]) atomicAdd(&ws[SEL_H0 + i], sh[i]);
    block_count_flush(cnt, &ws[0]);
}

__global__ void k_entropy(const float* __restrict__ z, const long long* __restrict__ label, int ignore,
                          int N, int C, long HW, float* __restrict__ ent, unsigned* __restrict__ ws) {
    __shared__ unsigned sh[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) sh[i] = 0;
    __syncthreads();
    long total = (long)N * HW;
    unsigned cnt = 0;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < total;
         p += (long)gridDim.x * blockDim.x) {
        long n = p / HW, q = p % HW;
        const float* b = z + n * C * HW + q;
        float m = b[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, b[(long)c * HW]);
        float s = 0.f, t = 0.f;
        for (int c = 0; c < C; ++c) {
            const float d = b[(long)c * HW] - m;
            const float e = expf(d);
            s += e;
            t += e * d;
        }
        float e = logf(s) - t / s;
        const bool valid = label == nullptr || label[p] != (long long)ignore;
        e = valid ? e : __uint_as_float(0x7fc00000u);
        ent[p] = e;
        cnt += valid ? 1u : 0u;
        atomicAdd(&sh[f32_key(e) >> 21], 1u);
    }
    hist0_flush(sh, cnt, ws);
}

U2PL_API int u2pl_entropy_f32(const float* logits, const long long* label, int ignore, int N, int C, int H,
                              int W, float* entropy, unsigned* ws, hipStream_t stream) {
    long total = (long)N * H * W;
    if (total <= 0) return 0;
    U2PL_LAUNCH(k_entropy, dim3(grid_for(total, 256, 512)), dim3(256), 0, stream, logits, label, ignore,
                       N, C, (long)H * W, entropy, ws);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// Fused bilinear up-sampling (bit-exact FMA form, a7) + entropy from the LOW-RES teacher logits:
// the (B,C,H,W) full-resolution logit tensor (90 MB at 769^2) is never materialised.
// CT = compile-time class count (registers hold the C up-sampled logits); CT == 0: generic two-sweep.
template <int CT>
__global__ void k_entropy_up(const float* __restrict__ in, long sn, long sc, long sh_, long sw, int N, int C,
                             int h, int w, int H, int W, float sy, float sx,
                             const long long* __restrict__ label, int ignore, float* __restrict__ ent,
                             unsigned* __restrict__ ws) {
    __shared__ unsigned sh[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) sh[i] = 0;
    __syncthreads();
    const long total = (long)N * H * W;
    unsigned cnt = 0;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(p % W);
        const long t0 = p / W;
        const int oy = (int)(t0 % H), n = (int)(t0 / H);
        const AcCoord cy = ac_coord(oy, sy, h), cx = ac_coord(ox, sx, w);
        const float* b = in + n * sn;
        const long o00 = cy.i0 * sh_ + cx.i0 * sw, o01 = cy.i0 * sh_ + cx.i1 * sw;
        const long o10 = cy.i1 * sh_ + cx.i0 * sw, o11 = cy.i1 * sh_ + cx.i1 * sw;
        auto up = [&](int c) {
            const float* bc = b + c * sc;
            const float top = __fmaf_rn(cx.l0, bc[o00], __fmul_rn(cx.l1, bc[o01]));
            const float bot = __fmaf_rn(cx.l0, bc[o10], __fmul_rn(cx.l1, bc[o11]));
            return __fmaf_rn(cy.l0, top, __fmul_rn(cy.l1, bot));
        };
        float m, s = 0.f, t = 0.f;
        if (CT > 0) {
            float zc[CT > 0 ? CT : 1];
#pragma unroll
            for (int c = 0; c < CT; ++c) zc[c] = up(c);
            m = zc[0];
#pragma unroll
            for (int c = 1; c < CT; ++c) m = fmaxf(m, zc[c]);
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const float d = zc[c] - m, e = expf(d);
                s += e;
                t += e * d;
            }
        } else {
            m = up(0);
            for (int c = 1; c < C; ++c) m = fmaxf(m, up(c));
            for (int c = 0; c < C; ++c) {
                const float d = up(c) - m, e = expf(d);
                s += e;
                t += e * d;
            }
        }
        float e = logf(s) - t / s;
        const bool valid = label == nullptr || label[p] != (long long)ignore;
        e = valid ? e : __uint_as_float(0x7fc00000u);
        ent[p] = e;
        cnt += valid ? 1u : 0u;
        atomicAdd(&sh[f32_key(e) >> 21], 1u);
    }
    hist0_flush(sh, cnt, ws);
}

// Integer up-sampling ratio R = (H-1)/(h-1) (4 for the stride-4 logits): one thread owns one low-res CELL
// and produces its R x R output pixels.  The 4 x CT corner logits are fetched ONCE with CT*4 independent
// loads issued back to back (class loop fully unrolled, values stay in registers for both sweeps), instead
// of 4 gathers per class per output pixel (the generic kernel is load-issue bound: 76 gathers / pixel).
// Arithmetic per output pixel is the identical FMA form (ac_coord per pixel) => same bits.
template <int R, int CT, int RY>
__global__ __launch_bounds__(64, 1) void k_entropy_up_cell(const float* __restrict__ in, long sn, long sc, long sh_, long sw, int N,
                                  int h, int w, int H, int W, float sy, float sx,
                                  const long long* __restrict__ label, int ignore, float* __restrict__ ent,
                                  unsigned* __restrict__ ws) {
    __shared__ unsigned sh[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) sh[i] = 0;
    __syncthreads();
    constexpr int NSUB = R / RY;                       // threads per cell (each owns RY output rows)
    const long nwork = (long)N * h * w * NSUB;
    unsigned cnt = 0;
    for (long qq = blockIdx.x * (long)blockDim.x + threadIdx.x; qq < nwork; qq += (long)gridDim.x * blockDim.x) {
        const int cj = (int)(qq % w);
        long t0 = qq / w;
        const int sub = (int)(t0 % NSUB);
        t0 /= NSUB;
        const int ci = (int)(t0 % h), n = (int)(t0 / h);
        const int oy0 = ci * R + sub * RY, ox0 = cj * R;
        if (oy0 >= H) continue;
        const int ny = min(RY, H - oy0), nx = min(R, W - ox0);
        float ly0[R], ly1[R], lx0[R], lx1[R];
        int y0 = ci, y1 = ci, x0 = cj, x1 = cj;
#pragma unroll
        for (int a = 0; a < R; ++a) {
            const AcCoord cy = ac_coord(min(oy0 + a, H - 1), sy, h), cx = ac_coord(min(ox0 + a, W - 1), sx, w);
            ly0[a] = cy.l0; ly1[a] = cy.l1; lx0[a] = cx.l0; lx1[a] = cx.l1;
            if (a == 0) { y0 = cy.i0; y1 = cy.i1; x0 = cx.i0; x1 = cx.i1; }
        }
        const float* b = in + n * sn;
        const long o00 = y0 * sh_ + x0 * sw, o01 = y0 * sh_ + x1 * sw, o10 = y1 * sh_ + x0 * sw, o11 = y1 * sh_ + x1 * sw;
        float v00[CT], v01[CT], v10[CT], v11[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const float* bc = b + c * sc;
            v00[c] = bc[o00]; v01[c] = bc[o01]; v10[c] = bc[o10]; v11[c] = bc[o11];
        }
#pragma unroll
        for (int a = 0; a < RY; ++a) {
            if (a >= ny) continue;
            float m[R], s[R], t[R];
#pragma unroll
            for (int bb = 0; bb < R; ++bb) { m[bb] = -INFINITY; s[bb] = 0.f; t[bb] = 0.f; }
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int bb = 0; bb < R; ++bb) {
                    const float top = __fmaf_rn(lx0[bb], v00[c], __fmul_rn(lx1[bb], v01[c]));
                    const float bot = __fmaf_rn(lx0[bb], v10[c], __fmul_rn(lx1[bb], v11[c]));
                    m[bb] = fmaxf(m[bb], __fmaf_rn(ly0[a], top, __fmul_rn(ly1[a], bot)));
                }
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int bb = 0; bb < R; ++bb) {
                    const float top = __fmaf_rn(lx0[bb], v00[c], __fmul_rn(lx1[bb], v01[c]));
                    const float bot = __fmaf_rn(lx0[bb], v10[c], __fmul_rn(lx1[bb], v11[c]));
                    const float d = __fmaf_rn(ly0[a], top, __fmul_rn(ly1[a], bot)) - m[bb];
                    const float e = expf(d);
                    s[bb] += e;
                    t[bb] += e * d;
                }
            const long p0 = ((long)n * H + oy0 + a) * W + ox0;
#pragma unroll
            for (int bb = 0; bb < R; ++bb)
                if (bb < nx) {
                    float e = logf(s[bb]) - t[bb] / s[bb];
                    const bool valid = label == nullptr || label[p0 + bb] != (long long)ignore;
                    e = valid ? e : __uint_as_float(0x7fc00000u);
                    ent[p0 + bb] = e;
                    cnt += valid ? 1u : 0u;
                    atomicAdd(&sh[f32_key(e) >> 21], 1u);
                }
        }
    }
    hist0_flush(sh, cnt, ws);
}

// LDS-shared variant: a 256-thread block owns 64 consecutive cells; the 4 x CT corner logits of each cell are
// fetched once by the block (thread t loads corner t/64 of cell t%64: CT loads per thread), parked in LDS
// as [corner][class][cell] (conflict-free), then thread (cell, row a) produces the 4 pixels of output row a.
// 4x the waves of the one-thread-per-cell kernel at the same global-load count.
template <int CT>
__global__ __launch_bounds__(256) void k_entropy_up_cell_lds(const float* __restrict__ in, long sn, long sc, long sh_,
                                                             long sw, int N, int h, int w, int H, int W, float sy,
                                                             float sx, const long long* __restrict__ label, int ignore,
                                                             float* __restrict__ ent, unsigned* __restrict__ ws) {
    constexpr int R = 4;
    __shared__ unsigned sh[2048];
    __shared__ float cv[4][CT][64];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) sh[i] = 0;
    const long ncell = (long)N * h * w;
    const int cell = threadIdx.x & 63, part = threadIdx.x >> 6;
    unsigned cnt = 0;
    for (long base = (long)blockIdx.x * 64; base < ncell; base += (long)gridDim.x * 64) {
        const long q = base + cell;
        const bool live = q < ncell;
        const long qq = live ? q : ncell - 1;
        const int cj = (int)(qq % w);
        const long t0 = qq / w;
        const int ci = (int)(t0 % h), n = (int)(t0 / h);
        const AcCoord cy0 = ac_coord(min(ci * R, H - 1), sy, h), cx0 = ac_coord(min(cj * R, W - 1), sx, w);
        {   // corner `part` of this cell -> LDS
            const int yy = (part & 2) ? cy0.i1 : cy0.i0, xx = (part & 1) ? cx0.i1 : cx0.i0;
            const float* b = in + n * sn + yy * sh_ + xx * sw;
#pragma unroll
            for (int c = 0; c < CT; ++c) cv[part][c][cell] = b[c * sc];
        }
        __syncthreads();
        const int oy = ci * R + part, ox0 = cj * R;
        if (live && oy < H) {
            const AcCoord cy = ac_coord(oy, sy, h);
            float lx0[R], lx1[R];
#pragma unroll
            for (int a = 0; a < R; ++a) {
                const AcCoord cx = ac_coord(min(ox0 + a, W - 1), sx, w);
                lx0[a] = cx.l0; lx1[a] = cx.l1;
            }
            float m[R], s[R], t[R], z[R];
#pragma unroll
            for (int bb = 0; bb < R; ++bb) { m[bb] = -INFINITY; s[bb] = 0.f; t[bb] = 0.f; }
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const float v00 = cv[0][c][cell], v01 = cv[1][c][cell], v10 = cv[2][c][cell], v11 = cv[3][c][cell];
#pragma unroll
                for (int bb = 0; bb < R; ++bb) {
                    const float top = __fmaf_rn(lx0[bb], v00, __fmul_rn(lx1[bb], v01));
                    const float bot = __fmaf_rn(lx0[bb], v10, __fmul_rn(lx1[bb], v11));
                    m[bb] = fmaxf(m[bb], __fmaf_rn(cy.l0, top, __fmul_rn(cy.l1, bot)));
                }
            }
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const float v00 = cv[0][c][cell], v01 = cv[1][c][cell], v10 = cv[2][c][cell], v11 = cv[3][c][cell];
#pragma unroll
                for (int bb = 0; bb < R; ++bb) {
                    const float top = __fmaf_rn(lx0[bb], v00, __fmul_rn(lx1[bb], v01));
                    const float bot = __fmaf_rn(lx0[bb], v10, __fmul_rn(lx1[bb], v11));
                    z[bb] = __fmaf_rn(cy.l0, top, __fmul_rn(cy.l1, bot)) - m[bb];
                    const float e = expf(z[bb]);
                    s[bb] += e;
                    t[bb] += e * z[bb];
                }
            }
            const int nx = min(R, W - ox0);
            const long p0 = ((long)n * H + oy) * W + ox0;
#pragma unroll
            for (int bb = 0; bb < R; ++bb)
                if (bb < nx) {
                    float e = logf(s[bb]) - t[bb] / s[bb];
                    const bool valid = label == nullptr || label[p0 + bb] != (long long)ignore;
                    e = valid ? e : __uint_as_float(0x7fc00000u);
                    ent[p0 + bb] = e;
                    cnt += valid ? 1u : 0u;
                    atomicAdd(&sh[f32_key(e) >> 21], 1u);
                }
        }
        __syncthreads();
    }
    hist0_flush(sh, cnt, ws);
}

U2PL_API int u2pl_entropy_up_f32(const float* in, long sn, long sc, long sh, long sw, int N, int C, int h, int w,
                                 int H, int W, const long long* label, int ignore, float* entropy, unsigned* ws,
                                 hipStream_t stream) {
    long total = (long)N * H * W;
    if (total <= 0) return 0;
    dim3 grid(grid_for(total, 256, 512)), block(256);
    const float sy = ac_scale_host(h, H), sx = ac_scale_host(w, W);
    if (h > 1 && w > 1 && H - 1 == 4 * (h - 1) && W - 1 == 4 * (w - 1) && (C == 19 || C == 21)) {
        static int ry = 0;   // 0/unset: LDS-shared kernel (default); 1|2|4: rows of a cell per thread (register kernel)
        if (!ry) { const char* e = getenv("U2PL_ENTROPY_RY"); ry = e ? atoi(e) : 8; if (ry != 1 && ry != 2 && ry != 4) ry = 8; }
        if (ry == 8) {
            const long ncell8 = (long)N * h * w;
            dim3 lg(grid_for(ncell8, 64, 2048)), lb(256);
            if (C == 19)
                U2PL_LAUNCH(k_entropy_up_cell_lds<19>, lg, lb, 0, stream, in, sn, sc, sh, sw, N, h, w, H, W, sy, sx, label, ignore, entropy, ws);
            else
                U2PL_LAUNCH(k_entropy_up_cell_lds<21>, lg, lb, 0, stream, in, sn, sc, sh, sw, N, h, w, H, W, sy, sx, label, ignore, entropy, ws);
            U2PL_LAUNCH_CHECK();
            return 0;
        }
        const long nwork = (long)N * h * w * (4 / ry);
        dim3 cgrid(grid_for(nwork, 64, 8192)), cblock(64);
#define ENT_CASE(CC, RR)                                                                                       \
    U2PL_LAUNCH((k_entropy_up_cell<4, CC, RR>), cgrid, cblock, 0, stream, in, sn, sc, sh, sw, N, h, w, H, W, sy, \
                       sx, label, ignore, entropy, ws)
        if (C == 19) { if (ry == 1) ENT_CASE(19, 1); else if (ry == 2) ENT_CASE(19, 2); else ENT_CASE(19, 4); }
        else { if (ry == 1) ENT_CASE(21, 1); else if (ry == 2) ENT_CASE(21, 2); else ENT_CASE(21, 4); }
#undef ENT_CASE
        U2PL_LAUNCH_CHECK();
        return 0;
    }
    if (C == 19)
        U2PL_LAUNCH(k_entropy_up<19>, grid, block, 0, stream, in, sn, sc, sh, sw, N, C, h, w, H, W, sy, sx, label, ignore, entropy, ws);
    else if (C == 21)
        U2PL_LAUNCH(k_entropy_up<21>, grid, block, 0, stream, in, sn, sc, sh, sw, N, C, h, w, H, W, sy, sx, label, ignore, entropy, ws);
    else
        U2PL_LAUNCH(k_entropy_up<0>, grid, block, 0, stream, in, sn, sc, sh, sw, N, C, h, w, H, W, sy, sx, label, ignore, entropy, ws);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// a11: target[entropy >= thr] = 255 (NaN entropy = already-ignored pixel);
// counts kept pixels into ws_count (for weight = B*H*W / #kept, loss_helper.py:44)
// ---------------------------------------------------------------------------
__global__ void k_apply_drop(const float* __restrict__ ent, const unsigned* __restrict__ thr_bits,
                             long long* __restrict__ target, int ignore, long n,
                             unsigned* __restrict__ nkept) {
    const float thr = __uint_as_float(*thr_bits);
    unsigned cnt = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        long long t = target[i];
        if (ent[i] >= thr && t != ignore) { t = ignore; target[i] = t; }
        cnt += t != ignore;
    }
    block_count_flush(cnt, nkept);
}

U2PL_API int u2pl_apply_drop_i64(const float* entropy, const unsigned* thr_bits, long long* target,
                                 int ignore, long n, unsigned* nkept, hipStream_t stream) {
    if (n <= 0) return 0;
    U2PL_LAUNCH(k_apply_drop, dim3(grid_for(n, 256, 512)), dim3(256), 0, stream, entropy, thr_bits,
                       target, ignore, n, nkept);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// a12/a13: threshold masks + legacy-nearest down-sampling + label multi-hot
// bits with the label_onehot batch-slot-0 quirk (utils.py:50-59, Q0), for the
// concatenated batch [labeled B | unlabeled B] at (h,w).
//   low_mask/high_mask : float (2B,1,h,w)      lbits : u32 (2B,h,w)
// ---------------------------------------------------------------------------
__global__ void k_reliability_masks(const float* __restrict__ ent, const unsigned* __restrict__ thr_lo_bits,
                                    const unsigned* __restrict__ thr_hi_bits,
                                    const long long* __restrict__ label_l,
                                    const long long* __restrict__ label_u, int ignore, int B, int H, int W,
                                    int h, int w, float ny, float nx, int neg_high,
                                    float* __restrict__ low_mask, float* __restrict__ high_mask,
                                    unsigned* __restrict__ lbits) {
    const float tlo = __uint_as_float(*thr_lo_bits), thi = __uint_as_float(*thr_hi_bits);
    long total = (long)2 * B * h * w;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < total;
         p += (long)gridDim.x * blockDim.x) {
        int x = (int)(p % w);
        long t = p / w;
        int y = (int)(t % h);
        int n = (int)(t / h);
        long src = (long)nearest_src(y, ny, H) * W + nearest_src(x, nx, W);
        const long HW = (long)H * W;
        float lo, hi;
        const long long* lab = n < B ? label_l : label_u;
        int b = n < B ? n : n - B;
        if (n < B) {
            lo = hi = lab[b * HW + src] != ignore ? 1.f : 0.f;
        } else {
            float e = ent[b * HW + src];  // NaN where label_u == ignore -> both false
            lo = e <= tlo ? 1.f : 0.f;
            hi = neg_high ? (e >= thi ? 1.f : 0.f) : 1.f;
        }
        low_mask[p] = lo;
        high_mask[p] = hi;
        unsigned bits = 0;
        if (b == 0 && lab[src] != ignore) {  // slot 0: union over the half-batch, zeroed on own ignore
            for (int bb = 0; bb < B; ++bb) {
                long long l = lab[bb * HW + src];
                bits |= 1u << (l == ignore ? 0 : (int)l);
            }
        }
        lbits[p] = bits;
    }
}

// One launch for the whole tail of the split: unsup target overwrite (loss_helper.py:41-44) on the
// full-res grid AND the low-res masks / class bits (train_semi.py:408-465).  Thresholds are read from
// the select workspace: thr_bits[0] = drop, [1] = low, [2] = high.
__global__ void k_reliability_apply(const float* __restrict__ ent, const unsigned* __restrict__ thr_bits,
                                    const long long* __restrict__ label_l, const long long* __restrict__ label_u,
                                    int ignore, int B, int H, int W, int h, int w, float ny, float nx, int neg_high,
                                    long long* __restrict__ target_u, unsigned* __restrict__ nkept,
                                    float* __restrict__ low_mask, float* __restrict__ high_mask,
                                    unsigned* __restrict__ lbits) {
    const float tdrop = __uint_as_float(thr_bits[0]), tlo = __uint_as_float(thr_bits[1]),
                thi = __uint_as_float(thr_bits[2]);
    const long HW = (long)H * W, nfull = (long)B * HW, nlow = (long)2 * B * h * w;
    unsigned cnt = 0;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < nfull + nlow; p += (long)gridDim.x * blockDim.x) {
        if (p < nfull) {
            long long t = label_u[p];
            if (ent[p] >= tdrop && t != ignore) t = ignore;
            target_u[p] = t;
            cnt += t != ignore;
        } else {
            const long q = p - nfull;
            const int x = (int)(q % w);
            const long t = q / w;
            const int y = (int)(t % h), n = (int)(t / h);
            const long src = (long)nearest_src(y, ny, H) * W + nearest_src(x, nx, W);
            const long long* lab = n < B ? label_l : label_u;
            const int b = n < B ? n : n - B;
            float lo, hi;
            if (n < B) lo = hi = lab[b * HW + src] != ignore ? 1.f : 0.f;
            else {
                const float e = ent[b * HW + src];
                lo = e <= tlo ? 1.f : 0.f;
                hi = neg_high ? (e >= thi ? 1.f : 0.f) : 1.f;
            }
            low_mask[q] = lo;
            high_mask[q] = hi;
            unsigned bits = 0;
            if (b == 0 && lab[src] != ignore)
                for (int bb = 0; bb < B; ++bb) {
                    const long long l = lab[bb * HW + src];
                    bits |= 1u << (l == ignore ? 0 : (int)l);
                }
            lbits[q] = bits;
        }
    }
    block_count_flush(cnt, nkept);
}
U2PL_API int u2pl_reliability_apply(const float* entropy, const unsigned* thr_bits, const long long* label_l,
                                    const long long* label_u, int ignore, int B, int H, int W, int h, int w,
                                    int negative_high_entropy, long long* target_u, unsigned* nkept, float* low_mask,
                                    float* high_mask, unsigned* lbits, hipStream_t stream) {
    const long total = (long)B * H * W + (long)2 * B * h * w;
    if (total <= 0) return 0;
    U2PL_LAUNCH(k_reliability_apply, dim3(grid_for(total, 256, 1024)), dim3(256), 0, stream, entropy, thr_bits, label_l,
                       label_u, ignore, B, H, W, h, w, (float)H / (float)h, (float)W / (float)w, negative_high_entropy,
                       target_u, nkept, low_mask, high_mask, lbits);
    U2PL_LAUNCH_CHECK();
    return 0;
}

U2PL_API int u2pl_reliability_masks(const float* entropy, const unsigned* thr_lo_bits,
                                    const unsigned* thr_hi_bits, const long long* label_l,
                                    const long long* label_u, int ignore, int B, int H, int W, int h, int w,
                                    int negative_high_entropy, float* low_mask, float* high_mask,
                                    unsigned* lbits, hipStream_t stream) {
    long total = (long)2 * B * h * w;
    if (total <= 0) return 0;
    float ny = (float)H / (float)h, nx = (float)W / (float)w;
    U2PL_LAUNCH(k_reliability_masks, dim3(grid_for(total, 256)), dim3(256), 0, stream, entropy,
                       thr_lo_bits, thr_hi_bits, label_l, label_u, ignore, B, H, W, h, w, ny, nx,
                       negative_high_entropy, low_mask, high_mask, lbits);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// bits <-> (N,C,h,w) int64 multi-hot (API parity with compute_contra_memobank_loss inputs)
__global__ void k_pack_bits(const long long* __restrict__ oh, int N, int C, long hw, unsigned* __restrict__ bits) {
    long total = (long)N * hw;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
        long n = p / hw, q = p % hw;
        unsigned b = 0;
        for (int c = 0; c < C; ++c) b |= (oh[(n * C + c) * hw + q] != 0 ? 1u : 0u) << c;
        bits[p] = b;
    }
}
__global__ void k_unpack_bits(const unsigned* __restrict__ bits, int N, int C, long hw, long long* __restrict__ oh) {
    long total = (long)N * C * hw;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
        long q = p % hw;
        long t = p / hw;
        int c = (int)(t % C);
        long n = t / C;
        oh[p] = (bits[n * hw + q] >> c) & 1u;
    }
}
U2PL_API int u2pl_pack_class_bits(const long long* onehot, int N, int C, int h, int w, unsigned* bits,
                                  hipStream_t stream) {
    if (C > 32) return U2PL_EINVAL;
    long total = (long)N * h * w;
    if (total <= 0) return 0;
    U2PL_LAUNCH(k_pack_bits, dim3(grid_for(total, 256)), dim3(256), 0, stream, onehot, N, C, (long)h * w, bits);
    U2PL_LAUNCH_CHECK();
    return 0;
}
U2PL_API int u2pl_unpack_class_bits(const unsigned* bits, int N, int C, int h, int w, long long* onehot,
                                    hipStream_t stream) {
    long total = (long)N * C * h * w;
    if (total <= 0) return 0;
    U2PL_LAUNCH(k_unpack_bits, dim3(grid_for(total, 256)), dim3(256), 0, stream, bits, N, C, (long)h * w, onehot);
    U2PL_LAUNCH_CHECK();
    return 0;
}
