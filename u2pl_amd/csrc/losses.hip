// Segmentation loss heads on full-resolution NCHW logits (SURVEY 8a rows a10,
// a11): cross entropy with ignore_index (forward partial sums + backward),
// OHEM target-probability extraction and kept-target rewrite.
// Reference: u2pl/utils/loss_helper.py:30-48 (unsup CE), 295-320, 339-360, 502-531.
#include "common.h"
#include "u2pl_hip.h"

std::atomic<unsigned long long> u2pl_kernel_launch_count{0};
U2PL_API size_t u2pl_kernel_launches(void) { return (size_t)u2pl_kernel_launch_count.load(std::memory_order_relaxed); }

// per-pixel log-softmax pick; block partial sums in double -> partial[2*blk+{0,1}]
__global__ void k_ce_fwd(const float* __restrict__ z, const long long* __restrict__ target, int ignore, int N,
                         int C, long HW, double* __restrict__ partial, const float* __restrict__ cw) {
    __shared__ double sh_l[4], sh_c[4];
    long total = (long)N * HW;
    double lsum = 0.0, cnt = 0.0;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
        long long t = target[p];
        if (t == ignore) continue;
        long n = p / HW, q = p % HW;
        const float* b = z + n * C * HW + q;
        float m = b[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, b[(long)c * HW]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(b[(long)c * HW] - m);
        float lse = m + logf(s);
        // nn.CrossEntropyLoss(weight=w, reduction="mean"): sum_i w[t_i] * l_i / sum_i w[t_i]  (loss_helper.py:258-320,451-500)
        const double wt = cw ? (double)cw[t] : 1.0;
        lsum += wt * (double)(lse - b[(long)t * HW]);
        cnt += wt;
    }
    lsum = wave_sum_d(lsum);
    cnt = wave_sum_d(cnt);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh_l[wave] = lsum; sh_c[wave] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = sh_l[0] + sh_l[1] + sh_l[2] + sh_l[3];
        partial[2 * blockIdx.x + 1] = sh_c[0] + sh_c[1] + sh_c[2] + sh_c[3];
    }
}

// out[0] = loss (mean over valid, times total/n_valid when unsup_weight),
// out[1] = per-pixel gradient scale = weight / n_valid, out[2] = n_valid
__global__ void k_ce_finish(const double* __restrict__ partial, int nblk, double total_pixels, int unsup_weight,
                            float* __restrict__ out) {
    __shared__ double sh[2][256];
    double l = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) { l += partial[2 * i]; c += partial[2 * i + 1]; }
    sh[0][threadIdx.x] = l;
    sh[1][threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { sh[0][threadIdx.x] += sh[0][threadIdx.x + o]; sh[1][threadIdx.x] += sh[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double n = sh[1][0];
        double w = unsup_weight ? total_pixels / n : 1.0;  // loss_helper.py:44 (inf when n == 0)
        out[0] = (float)(w * (sh[0][0] / n));               // 0/0 -> NaN like torch
        out[1] = (float)(w / n);
        out[2] = (float)n;
    }
}

#define CE_BLOCKS 2048
U2PL_API size_t u2pl_ce_workspace_bytes(void) { return (size_t)CE_BLOCKS * 2 * sizeof(double); }

U2PL_API int u2pl_ce_fwd_f32(const float* logits, const long long* target, int ignore, int N, int C, int H,
                             int W, int unsup_weight, void* workspace, float* out3, hipStream_t stream) {
    long total = (long)N * H * W;
    if (total <= 0) return U2PL_EINVAL;
    int nblk = grid_for(total, 256, CE_BLOCKS);
    U2PL_LAUNCH(k_ce_fwd, dim3(nblk), dim3(256), 0, stream, logits, target, ignore, N, C, (long)H * W, (double*)workspace,
                       (const float*)nullptr);
    U2PL_LAUNCH_CHECK();
    U2PL_LAUNCH(k_ce_finish, dim3(1), dim3(256), 0, stream, (const double*)workspace, nblk, (double)total, unsup_weight, out3);
    U2PL_LAUNCH_CHECK();
    return 0;
}
// class-weighted variant (use_weight: True, loss_helper.py:265-292,461-488): out3 = {loss, 1 / sum of weights, sum of weights}
U2PL_API int u2pl_ce_fwd_weighted_f32(const float* logits, const long long* target, int ignore, int N, int C, int H,
                                      int W, const float* class_weight, void* workspace, float* out3,
                                      hipStream_t stream) {
    long total = (long)N * H * W;
    if (total <= 0 || !class_weight) return U2PL_EINVAL;
    int nblk = grid_for(total, 256, CE_BLOCKS);
    U2PL_LAUNCH(k_ce_fwd, dim3(nblk), dim3(256), 0, stream, logits, target, ignore, N, C, (long)H * W, (double*)workspace,
                       class_weight);
    U2PL_LAUNCH_CHECK();
    U2PL_LAUNCH(k_ce_finish, dim3(1), dim3(256), 0, stream, (const double*)workspace, nblk, (double)total, 0, out3);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// grad[n][c][q] = (softmax_c - [c==t]) * scale * gout  for valid pixels, else 0
__global__ void k_ce_bwd(const float* __restrict__ z, const long long* __restrict__ target, int ignore, int N,
                         int C, long HW, const float* __restrict__ scale, const float* __restrict__ gout,
                         float gmul, float* __restrict__ grad, const float* __restrict__ cw) {
    const float sc0 = scale[1] * (gout ? *gout : 1.0f) * gmul;
    long total = (long)N * HW;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
        long n = p / HW, q = p % HW;
        const float* b = z + n * C * HW + q;
        float* g = grad + n * C * HW + q;
        long long t = target[p];
        if (t == ignore) {
            for (int c = 0; c < C; ++c) g[(long)c * HW] = 0.f;
            continue;
        }
        const float sc = cw ? sc0 * cw[t] : sc0;
        float m = b[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, b[(long)c * HW]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(b[(long)c * HW] - m);
        const float inv = sc / s;
        for (int c = 0; c < C; ++c) {
            float pr = expf(b[(long)c * HW] - m) * inv;
            g[(long)c * HW] = pr - (c == t ? sc : 0.f);
        }
    }
}
U2PL_API int u2pl_ce_bwd_f32(const float* logits, const long long* target, int ignore, int N, int C, int H,
                             int W, const float* out3_dev, const float* gout_dev, float gmul, float* grad,
                             hipStream_t stream) {
    long total = (long)N * H * W;
    if (total <= 0) return 0;
    U2PL_LAUNCH(k_ce_bwd, dim3(grid_for(total, 256)), dim3(256), 0, stream, logits, target, ignore, N, C,
                       (long)H * W, out3_dev, gout_dev, gmul, grad, (const float*)nullptr);
    U2PL_LAUNCH_CHECK();
    return 0;
}
U2PL_API int u2pl_ce_bwd_weighted_f32(const float* logits, const long long* target, int ignore, int N, int C, int H,
                                      int W, const float* class_weight, const float* out3_dev, const float* gout_dev,
                                      float gmul, float* grad, hipStream_t stream) {
    long total = (long)N * H * W;
    if (total <= 0 || !class_weight) return U2PL_EINVAL;
    U2PL_LAUNCH(k_ce_bwd, dim3(grid_for(total, 256)), dim3(256), 0, stream, logits, target, ignore, N, C,
                       (long)H * W, out3_dev, gout_dev, gmul, grad, class_weight);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// OHEM (loss_helper.py:502-520): mask_prob = softmax(pred)[target] (1.0 on
// ignored pixels); counts valid pixels into ws[0] (select workspace word 0)
__global__ void k_ohem_prob(const float* __restrict__ z, const long long* __restrict__ target, int ignore,
                            int N, int C, long HW, float* __restrict__ mp, unsigned* __restrict__ ws) {
    __shared__ unsigned sh[2048];   // pass-0 histogram of the k-th-smallest selection (select.hip)
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) sh[i] = 0;
    __syncthreads();
    long total = (long)N * HW;
    unsigned cnt = 0;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
        long long t = target[p];
        float v = 1.0f;
        if (t != ignore) {
            long n = p / HW, q = p % HW;
            const float* b = z + n * C * HW + q;
            float m = b[0];
            for (int c = 1; c < C; ++c) m = fmaxf(m, b[(long)c * HW]);
            float s = 0.f;
            for (int c = 0; c < C; ++c) s += expf(b[(long)c * HW] - m);
            v = expf(b[(long)t * HW] - m) / s;
            cnt++;
        }
        mp[p] = v;
        atomicAdd(&sh[f32_key(v) >> 21], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += blockDim.x)
        if (sh[i]) atomicAdd(&ws[128 + i], sh[i]);
    block_count_flush(cnt, &ws[0]);
}
U2PL_API int u2pl_ohem_prob_f32(const float* logits, const long long* target, int ignore, int N, int C, int H,
                                int W, float* mask_prob, unsigned* nvalid, hipStream_t stream) {
    long total = (long)N * H * W;
    if (total <= 0) return 0;
    U2PL_LAUNCH(k_ohem_prob, dim3(grid_for(total, 256, 512)), dim3(256), 0, stream, logits, target, ignore, N,
                       C, (long)H * W, mask_prob, nvalid);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// kept = mask_prob <= thr (thr = +inf keeps everything valid); dropped -> ignore
__global__ void k_ohem_apply(const float* __restrict__ mp, const unsigned* __restrict__ thr_bits,
                             const long long* __restrict__ target, int ignore, long n,
                             long long* __restrict__ kept) {
    const float thr = __uint_as_float(*thr_bits);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        long long t = target[i];
        kept[i] = (t != ignore && mp[i] <= thr) ? t : (long long)ignore;
    }
}
U2PL_API int u2pl_ohem_apply_i64(const float* mask_prob, const unsigned* thr_bits, const long long* target,
                                 int ignore, long n, long long* kept_target, hipStream_t stream) {
    if (n <= 0) return 0;
    U2PL_LAUNCH(k_ohem_apply, dim3(grid_for(n, 256)), dim3(256), 0, stream, mask_prob, thr_bits, target, ignore, n, kept_target);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// validate(): argmax over classes + intersection / union / target histograms on device
// (train_semi.py:620-641, utils.py:568-580).  hist: int64 [3][C] = intersection, area_output, area_target
// (area_union = output + target - intersection); integer atomics => deterministic.
__global__ void k_confusion(const float* __restrict__ z, const long long* __restrict__ target, int ignore, int N,
                            int C, long HW, unsigned long long* __restrict__ hist) {
    extern __shared__ unsigned sh_h[];   // [3][C]
    for (int i = threadIdx.x; i < 3 * C; i += blockDim.x) sh_h[i] = 0;
    __syncthreads();
    const long total = (long)N * HW;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
        const long long t = target[p];
        if (t == ignore) continue;   // output[target == ignore] = ignore (utils.py:574)
        const long n = p / HW, q = p % HW;
        const float* b = z + n * C * HW + q;
        float m = b[0];
        int am = 0;
        for (int c = 1; c < C; ++c) {
            const float v = b[(long)c * HW];
            if (v > m) { m = v; am = c; }
        }
        atomicAdd(&sh_h[C + am], 1u);
        // a label outside [0, C) that is not the ignore value (wrong label map): np.histogram(range=(0, K-1)) drops it
        // from area_target / area_intersection (utils.py:576-579); the prediction still counts in area_output
        if (t >= 0 && t < C) {
            atomicAdd(&sh_h[2 * C + (int)t], 1u);
            if (am == (int)t) atomicAdd(&sh_h[am], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * C; i += blockDim.x)
        if (sh_h[i]) atomicAdd(&hist[i], (unsigned long long)sh_h[i]);
}
U2PL_API int u2pl_confusion_hist_f32(const float* logits, const long long* target, int ignore, int N, int C, int H,
                                     int W, long long* hist3c, hipStream_t stream) {
    const long total = (long)N * H * W;
    if (total <= 0) return 0;
    U2PL_LAUNCH(k_confusion, dim3(grid_for(total, 256, 512)), dim3(256), 3 * C * sizeof(unsigned), stream, logits,
                       target, ignore, N, C, (long)H * W, (unsigned long long*)hist3c);
    U2PL_LAUNCH_CHECK();
    return 0;
}
