// Reliability split kernels:
//   bilinear up-sampling (align_corners=True, torch-CPU FMA form, bit-exact),
//   teacher pseudo label, per-pixel softmax entropy, EXACT order-statistic
//   selection (4-pass 8-bit radix select on the fp32 bit pattern) with numpy's
//   float32 percentile lerp, threshold masks, legacy-nearest down-sampling and
//   the label_onehot batch-slot-0 quirk packed as per-pixel class bitmasks.
// All kernels are HBM-bound; reads/writes are lane-contiguous along W.
#include "common.h"
#include "u2pl_hip.h"

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
    hipLaunchKernelGGL(k_bilinear_up, dim3(grid_for(total, 256)), dim3(256), 0, stream, in, sn, sc, sh,
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
    hipLaunchKernelGGL(k_bilinear_up_bwd, dim3(grid_for(total, 256, 1 << 20)), dim3(256), 0, stream, gout, N, C, H,
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
    hipLaunchKernelGGL(k_pseudo_label, dim3(grid_for(total, 256)), dim3(256), 0, stream, logits, N, C,
                       (long)H * W, conf, label);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// a11/a12: entropy = -sum p*log(p+1e-10); NaN marks label==ignore pixels so a
// single float stream carries both value and validity; counts valid pixels.
// state[0] += #valid (atomic, integer => deterministic)
// ---------------------------------------------------------------------------
__global__ void k_entropy(const float* __restrict__ z, const long long* __restrict__ label, int ignore,
                          int N, int C, long HW, float* __restrict__ ent, unsigned* __restrict__ nvalid) {
    long total = (long)N * HW;
    unsigned cnt = 0;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < total;
         p += (long)gridDim.x * blockDim.x) {
        long n = p / HW, q = p % HW;
        const float* b = z + n * C * HW + q;
        float m = b[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, b[(long)c * HW]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(b[(long)c * HW] - m);
        float e = 0.f;
        for (int c = 0; c < C; ++c) {
            float pr = expf(b[(long)c * HW] - m) / s;
            e += pr * logf(pr + 1e-10f);
        }
        e = -e;
        bool valid = label == nullptr || label[p] != (long long)ignore;
        ent[p] = valid ? e : __uint_as_float(0x7fc00000u);
        cnt += valid ? 1u : 0u;
    }
    cnt = wave_sum_u(cnt);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(nvalid, cnt);
}

U2PL_API int u2pl_entropy_f32(const float* logits, const long long* label, int ignore, int N, int C, int H,
                              int W, float* entropy, unsigned* nvalid, hipStream_t stream) {
    long total = (long)N * H * W;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_entropy, dim3(grid_for(total, 256)), dim3(256), 0, stream, logits, label, ignore,
                       N, C, (long)H * W, entropy, nvalid);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// Exact selection.  Workspace layout (unsigned words), see u2pl_hip.h:
//   [0]            n_valid (non-NaN count; written by producer or by pass 0)
//   [1]            n_total
//   [8 + s]        prefix key of slot s   (s < U2PL_SEL_MAX_SLOTS)
//   [24 + s]       remaining rank of slot s inside its prefix
//   [40 + s]       float bits of the selected value (after last resolve)
//   [56 + j]       float bits of threshold j (percentile lerp or OHEM thr)
//   [64 + j]       float bits of gamma_j
//   [128 ...]      histograms: [pass 4][slot 16][256]
// ---------------------------------------------------------------------------
#define SEL_PREFIX 8
#define SEL_KREM 24
#define SEL_VAL 40
#define SEL_THR 56
#define SEL_GAMMA 64
#define SEL_HIST 128

__global__ void k_select_hist(const float* __restrict__ v, long n, int pass, int nslots,
                              unsigned* __restrict__ ws) {
    __shared__ unsigned sh[U2PL_SEL_MAX_SLOTS * 256];
    __shared__ unsigned s_prefix[U2PL_SEL_MAX_SLOTS];
    __shared__ int s_active[U2PL_SEL_MAX_SLOTS];
    for (int i = threadIdx.x; i < U2PL_SEL_MAX_SLOTS * 256; i += blockDim.x) sh[i] = 0;
    if (threadIdx.x < U2PL_SEL_MAX_SLOTS) {
        // a slot is active if it is the first slot carrying its prefix (pass 0: slot 0 only)
        int s = threadIdx.x;
        unsigned pf = (s < nslots && pass > 0) ? ws[SEL_PREFIX + s] : 0u;
        int act = s < nslots;
        if (pass == 0) act = (s == 0);
        else
            for (int t = 0; t < s; ++t)
                if (ws[SEL_PREFIX + t] == pf) act = 0;
        s_prefix[s] = pf;
        s_active[s] = act;
    }
    __syncthreads();
    const int shift = 24 - 8 * pass;
    const unsigned himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    unsigned prefix[U2PL_SEL_MAX_SLOTS];
    bool active[U2PL_SEL_MAX_SLOTS];
#pragma unroll
    for (int s = 0; s < U2PL_SEL_MAX_SLOTS; ++s) {
        prefix[s] = s_prefix[s];
        active[s] = s_active[s] != 0;
    }
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned k = f32_key(v[i]);
        unsigned d = (k >> shift) & 0xffu;
#pragma unroll
        for (int s = 0; s < U2PL_SEL_MAX_SLOTS; ++s)
            if (active[s] && (k & himask) == prefix[s]) atomicAdd(&sh[s * 256 + d], 1u);
    }
    __syncthreads();
    unsigned* gh = ws + SEL_HIST + (long)pass * U2PL_SEL_MAX_SLOTS * 256;
    for (int i = threadIdx.x; i < U2PL_SEL_MAX_SLOTS * 256; i += blockDim.x)
        if (sh[i]) atomicAdd(&gh[i], sh[i]);
}

// One block of 256 threads.  pass==0 additionally derives the ranks from n_valid.
// spec_kind[j]: 0 = numpy percentile with q32[j] -> slots (2j, 2j+1)
//               1 = explicit rank min(n_total, kparam[j]) - 1 -> slots 2j and 2j+1
__global__ void k_select_resolve(int pass, int nspec, const int* __restrict__ spec_kind,
                                 const float* __restrict__ q32, const long long* __restrict__ kparam,
                                 unsigned* __restrict__ ws) {
    __shared__ unsigned cum[256];
    __shared__ unsigned s_prefix[U2PL_SEL_MAX_SLOTS], s_krem[U2PL_SEL_MAX_SLOTS];
    __shared__ unsigned s_newprefix[U2PL_SEL_MAX_SLOTS], s_newkrem[U2PL_SEL_MAX_SLOTS];
    const int nslots = 2 * nspec;
    const int tid = threadIdx.x;
    if (pass == 0 && tid < nspec) {
        const unsigned nv = ws[0], nt = ws[1];
        long lo, hi;
        float gamma = 0.f;
        if (spec_kind[tid] == 0) {
            long n = nv;
            float vi = __fmul_rn((float)(n - 1), q32[tid]);  // numpy: (n - 1) * q, float32
            float fl = floorf(vi);
            gamma = __fsub_rn(vi, fl);
            if (n <= 0) { lo = hi = 0; }
            else if (!(vi == vi) || vi >= (float)(n - 1)) { lo = hi = n - 1; }
            else if (vi < 0.f) { lo = hi = 0; }
            else { lo = (long)fl; hi = lo + 1; }
        } else {
            long k = kparam[tid];
            long n = nt;
            lo = hi = (k < n ? k : n) - 1;
            if (lo < 0) lo = hi = 0;
        }
        s_krem[2 * tid] = (unsigned)lo;
        s_krem[2 * tid + 1] = (unsigned)hi;
        s_prefix[2 * tid] = 0;
        s_prefix[2 * tid + 1] = 0;
        ws[SEL_GAMMA + tid] = __float_as_uint(gamma);
    } else if (pass > 0 && tid < nslots) {
        s_prefix[tid] = ws[SEL_PREFIX + tid];
        s_krem[tid] = ws[SEL_KREM + tid];
    }
    __syncthreads();
    const int shift = 24 - 8 * pass;
    const unsigned* gh = ws + SEL_HIST + (long)pass * U2PL_SEL_MAX_SLOTS * 256;
    for (int s = 0; s < nslots; ++s) {
        const unsigned pf = s_prefix[s], k = s_krem[s];
        int l = s;  // histogram owner = first slot with this prefix (must match k_select_hist)
        if (pass == 0) l = 0;
        else
            for (int t = 0; t < s; ++t)
                if (s_prefix[t] == pf) { l = t; break; }
        cum[tid] = gh[l * 256 + tid];
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {  // inclusive Hillis-Steele scan
            unsigned t = tid >= o ? cum[tid - o] : 0;
            __syncthreads();
            cum[tid] += t;
            __syncthreads();
        }
        const unsigned before = tid ? cum[tid - 1] : 0;
        if (k >= before && k < cum[tid]) {
            s_newprefix[s] = pf | ((unsigned)tid << shift);
            s_newkrem[s] = k - before;
        }
        __syncthreads();
    }
    if (tid < nslots) {
        ws[SEL_PREFIX + tid] = s_newprefix[tid];
        ws[SEL_KREM + tid] = s_newkrem[tid];
        if (pass == 3) ws[SEL_VAL + tid] = __float_as_uint(key_f32(s_newprefix[tid]));
    }
}

// thresholds: numpy _lerp in float32 (no FMA):  d=b-a;  t>=.5 ? b-d*(1-t) : a+d*t
// spec kind 1 (OHEM): thr = kth > fthresh ? kth : fthresh, or +inf when
// min_kept > n_valid ("no filtering", loss_helper.py:513-515)
__global__ void k_select_finish(int nspec, const int* __restrict__ spec_kind,
                                const long long* __restrict__ kparam, const float* __restrict__ fparam,
                                unsigned* __restrict__ ws) {
    int j = threadIdx.x;
    if (j >= nspec) return;
    float a = __uint_as_float(ws[SEL_VAL + 2 * j]), b = __uint_as_float(ws[SEL_VAL + 2 * j + 1]);
    float thr;
    if (spec_kind[j] == 0) {
        float t = __uint_as_float(ws[SEL_GAMMA + j]);
        float d = __fsub_rn(b, a);
        thr = (t >= 0.5f) ? __fsub_rn(b, __fmul_rn(d, __fsub_rn(1.0f, t))) : __fadd_rn(a, __fmul_rn(d, t));
        if (ws[0] == 0) thr = __uint_as_float(0x7fc00000u);  // empty selection -> NaN (numpy)
    } else {
        long long nv = ws[0];
        if (kparam[j] > nv) thr = __uint_as_float(0x7f800000u);  // keep every valid pixel
        else thr = a > fparam[j] ? a : fparam[j];
    }
    ws[SEL_THR + j] = __float_as_uint(thr);
}

U2PL_API size_t u2pl_select_workspace_bytes(void) {
    return (size_t)(SEL_HIST + 4 * U2PL_SEL_MAX_SLOTS * 256) * sizeof(unsigned);
}

// Caller contract: ws was zeroed (hipMemsetAsync) before the producer wrote
// ws[0] (n_valid) / ws[1] (n_total); spec arrays live in device memory.
U2PL_API int u2pl_select_f32(const float* values, long n, int nspec, const int* spec_kind,
                             const float* q32, const long long* kparam, const float* fparam,
                             unsigned* ws, hipStream_t stream) {
    if (nspec < 1 || 2 * nspec > U2PL_SEL_MAX_SLOTS) return U2PL_EINVAL;
    const int nslots = 2 * nspec;
    const int grid = grid_for(n, 256, 1024);
    for (int pass = 0; pass < 4; ++pass) {
        hipLaunchKernelGGL(k_select_hist, dim3(grid), dim3(256), 0, stream, values, n, pass, nslots, ws);
        U2PL_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_select_resolve, dim3(1), dim3(256), 0, stream, pass, nspec, spec_kind, q32,
                           kparam, ws);
        U2PL_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_select_finish, dim3(1), dim3(64), 0, stream, nspec, spec_kind, kparam, fparam, ws);
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
    cnt = wave_sum_u(cnt);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(nkept, cnt);
}

U2PL_API int u2pl_apply_drop_i64(const float* entropy, const unsigned* thr_bits, long long* target,
                                 int ignore, long n, unsigned* nkept, hipStream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_apply_drop, dim3(grid_for(n, 256)), dim3(256), 0, stream, entropy, thr_bits,
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

U2PL_API int u2pl_reliability_masks(const float* entropy, const unsigned* thr_lo_bits,
                                    const unsigned* thr_hi_bits, const long long* label_l,
                                    const long long* label_u, int ignore, int B, int H, int W, int h, int w,
                                    int negative_high_entropy, float* low_mask, float* high_mask,
                                    unsigned* lbits, hipStream_t stream) {
    long total = (long)2 * B * h * w;
    if (total <= 0) return 0;
    float ny = (float)H / (float)h, nx = (float)W / (float)w;
    hipLaunchKernelGGL(k_reliability_masks, dim3(grid_for(total, 256)), dim3(256), 0, stream, entropy,
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
    hipLaunchKernelGGL(k_pack_bits, dim3(grid_for(total, 256)), dim3(256), 0, stream, onehot, N, C, (long)h * w, bits);
    U2PL_LAUNCH_CHECK();
    return 0;
}
U2PL_API int u2pl_unpack_class_bits(const unsigned* bits, int N, int C, int h, int w, long long* onehot,
                                    hipStream_t stream) {
    long total = (long)N * C * h * w;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_unpack_bits, dim3(grid_for(total, 256)), dim3(256), 0, stream, bits, N, C, (long)h * w, onehot);
    U2PL_LAUNCH_CHECK();
    return 0;
}
