// Memory-bound layers of the ResNet-101 / DeepLabv3+ stack on NHWC "rows"
// (8a rows a2-a6): (Sync)BatchNorm statistics /
// apply / backward fused with residual-add, ReLU and Dropout2d scaling,
// ceil-mode max-pool, global average pool, bilinear feature up-sampling,
// channel-slice copies (concat).  All accesses are float4 along the channel
// axis (lane-contiguous), reductions are two-stage and ordered (deterministic).
#include "common.h"
#include "u2pl_hip.h"

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// ---------------------------------------------------------------------------
// Column reduction skeleton: for rows [seg*Mseg, (seg+1)*Mseg) accumulate two
// per-channel sums produced by Op::get(row, col4, &v0, &v1).
//   partial: float [seg][nblk][2][C]   ->   out: double [seg][2][C]
// ---------------------------------------------------------------------------
struct StatOp {  // v0 = x - pivot, v1 = (x - pivot)^2     (BN forward statistics)
    const float* x; long ld; const float* pivot;
    __device__ __forceinline__ void get(long row, int c4, float4& v0, float4& v1) const {
        float4 v = *(const float4*)(x + row * ld + c4 * 4);
        if (pivot) v = f4sub(v, *(const float4*)(pivot + c4 * 4));
        v0 = v; v1 = f4mul(v, v);
    }
};
struct SumOp {  // v0 = x, v1 = 0     (global average pool / bias gradient)
    const float* x; long ld;
    __device__ __forceinline__ void get(long row, int c4, float4& v0, float4& v1) const {
        v0 = *(const float4*)(x + row * ld + c4 * 4); v1 = f4zero();
    }
};
// the ReLU mask of y = relu((x - mean) * invstd * gamma + beta) recomputed from x: the forward's own expression (k_bn_apply, no
// contraction), so the same bits and the same mask as [y > 0] -- without reading y (round 6: BatchNorms without a residual)
__device__ __forceinline__ float4 relu_mask_from_x(float4 g, float4 xv, const float* mean, const float* invstd, const float* gamma,
                                                   const float* beta, int c) {
    const float4 mu = *(const float4*)(mean + c), is = *(const float4*)(invstd + c), ga = *(const float4*)(gamma + c), be = *(const float4*)(beta + c);
    g.x = ((xv.x - mu.x) * is.x * ga.x + be.x) > 0.f ? g.x : 0.f;
    g.y = ((xv.y - mu.y) * is.y * ga.y + be.y) > 0.f ? g.y : 0.f;
    g.z = ((xv.z - mu.z) * is.z * ga.z + be.z) > 0.f ? g.z : 0.f;
    g.w = ((xv.w - mu.w) * is.w * ga.w + be.w) > 0.f ? g.w : 0.f;
    return g;
}
struct BnBwdOp {  // g = dy*[y>0]*drop ; v0 = g, v1 = g * xhat     (BN backward sums)
    const float* dy; long lddy; const float* x; long ldx; const float* y; long ldy;
    const float* mean; const float* invstd; const float* drop; long rows_per_image; int C;
    const float* gamma; const float* relu_beta;         // y == NULL and relu_beta != NULL: the mask from x
    __device__ __forceinline__ void get(long row, int c4, float4& v0, float4& v1) const {
        float4 g = *(const float4*)(dy + row * lddy + c4 * 4);
        float4 xv = *(const float4*)(x + row * ldx + c4 * 4);
        if (y) {
            float4 yy = *(const float4*)(y + row * ldy + c4 * 4);
            g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
            g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
        } else if (relu_beta) {
            g = relu_mask_from_x(g, xv, mean, invstd, gamma, relu_beta, c4 * 4);
        }
        if (drop) g = f4mul(g, *(const float4*)(drop + (row / rows_per_image) * C + c4 * 4));
        float4 xh = f4mul(f4sub(xv, *(const float4*)(mean + c4 * 4)), *(const float4*)(invstd + c4 * 4));
        v0 = g; v1 = f4mul(g, xh);
    }
};

// grid = (row blocks, segments, column slabs of 64 float4): enough blocks in flight to cover HBM latency
// even when the tensor has few rows (layer3/4: 37636 rows x 1024-2048 channels)
template <class Op>
__global__ U2PL_HBM_KERNEL void k_colreduce_partial(Op op, long Mseg, int C, float* __restrict__ partial) {
    __shared__ float4 sh0[256], sh1[256];
    const int C4 = C >> 2;
    const int slab0 = blockIdx.z * 64;
    const int wcols = min(64, C4 - slab0);          // float4 columns handled by this block
    const int tc = wcols, tr = 256 / tc;
    const int tid = threadIdx.x, cl = tid % tc, rg = tid / tc;
    const int nblk = gridDim.x, seg = blockIdx.y;
    const long per = (Mseg + nblk - 1) / nblk;
    const long rb = seg * Mseg + blockIdx.x * per, re = min(seg * Mseg + Mseg, rb + per);
    float* out = partial + ((long)seg * nblk + blockIdx.x) * 2 * C;
    const int c4 = slab0 + cl;
    float4 a0 = f4zero(), a1 = f4zero();
    if (rg < tr) {
        // 4 independent row streams per thread: keeps >= 4 x 16 B loads in flight per lane
        float4 b0 = f4zero(), b1 = f4zero(), c0 = f4zero(), c1 = f4zero(), d0 = f4zero(), d1 = f4zero();
        long r = rb + rg;
        for (; r + 3 * (long)tr < re; r += 4 * (long)tr) {
            float4 v0, v1, w0, w1, x0, x1, y0, y1;
            op.get(r, c4, v0, v1);
            op.get(r + tr, c4, w0, w1);
            op.get(r + 2 * (long)tr, c4, x0, x1);
            op.get(r + 3 * (long)tr, c4, y0, y1);
            a0 = f4add(a0, v0); a1 = f4add(a1, v1);
            b0 = f4add(b0, w0); b1 = f4add(b1, w1);
            c0 = f4add(c0, x0); c1 = f4add(c1, x1);
            d0 = f4add(d0, y0); d1 = f4add(d1, y1);
        }
        for (; r < re; r += tr) {
            float4 v0, v1;
            op.get(r, c4, v0, v1);
            a0 = f4add(a0, v0); a1 = f4add(a1, v1);
        }
        a0 = f4add(f4add(a0, b0), f4add(c0, d0));
        a1 = f4add(f4add(a1, b1), f4add(c1, d1));
    }
    sh0[tid] = a0; sh1[tid] = a1;
    __syncthreads();
    if (rg == 0) {
        for (int k = 1; k < tr; ++k) { a0 = f4add(a0, sh0[k * tc + cl]); a1 = f4add(a1, sh1[k * tc + cl]); }
        *(float4*)(out + c4 * 4) = a0;
        *(float4*)(out + C + c4 * 4) = a1;
    }
}
// ordered (deterministic) second stage: 64 columns x 16 partial-row groups per 1024-thread block; every
// thread keeps 8 independent loads in flight (the kernel is pure latency: <= 512 x 2C floats in total)
__global__ __launch_bounds__(1024) void k_colreduce_final(const float* __restrict__ partial, int nblk, int C,
                                                          double* __restrict__ out) {
    __shared__ double sh[16][64];
    const int seg = blockIdx.y;
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + cl;
    double acc = 0.0;
    if (i < 2 * C) {
        const float* p = partial + (long)seg * nblk * 2 * C + i;
        int b = rg;
        for (; b + 7 * 16 < nblk; b += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(long)(b + u * 16) * 2 * C];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += (double)v[u];
        }
        for (; b < nblk; b += 16) acc += (double)p[(long)b * 2 * C];
    }
    sh[rg][cl] = acc;
    __syncthreads();
    if (rg == 0 && i < 2 * C) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sh[k][cl];
        out[(long)seg * 2 * C + i] = t;
    }
}
static int colreduce_blocks(long Mseg) {
    long b = (Mseg + 63) / 64;
    return (int)(b < 1 ? 1 : (b > 512 ? 512 : b));
}
U2PL_API size_t u2pl_colreduce_workspace_bytes(long Mseg, int nseg, int C) {
    return (size_t)nseg * colreduce_blocks(Mseg) * 2 * C * sizeof(float);
}
template <class Op>
static int run_colreduce(Op op, long Mseg, int nseg, int C, void* ws, double* out, hipStream_t stream) {
    if (C % 4 || Mseg <= 0) return U2PL_EINVAL;
    const int nblk = colreduce_blocks(Mseg);
    U2PL_LAUNCH((k_colreduce_partial<Op>), dim3(nblk, nseg, cdiv(C / 4, 64)), dim3(256), 0, stream, op, Mseg, C, (float*)ws);
    U2PL_LAUNCH_CHECK();
    U2PL_LAUNCH(k_colreduce_final, dim3(cdiv(2 * C, 64), nseg), dim3(1024), 0, stream, (const float*)ws, nblk, C, out);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// ordered finish of externally produced partials (conv epilogue statistics): [nblk][2][C] float -> double [2][C]
U2PL_API int u2pl_colreduce_finish_f32(const float* partial, int nblk, int C, double* sums, hipStream_t stream) {
    U2PL_LAUNCH(k_colreduce_final, dim3(cdiv(2 * C, 64), 1), dim3(1024), 0, stream, partial, nblk, C, sums);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// nn.SyncBatchNorm / nn.BatchNorm2d training statistics (base.py:6-8): local
// shifted sums  S1 = sum(x - pivot), S2 = sum((x - pivot)^2)  -> out double [2][C]
// Few rows (M <= 64: the BatchNorm behind the ASPP image-pooling branch normalises over the N pooled vectors, which
// differ by ~1 %): S2 - S1^2/M cancels ~1e4..1e5 : 1 there, so fp32 squares would leave only 2-3 correct digits of
// the variance.  Two passes in float64 (mean, then centred squares), re-expressed in the shifted-sum protocol.
__global__ void k_bn_stats_small(const float* __restrict__ x, long ld, int M, int C, const float* __restrict__ pivot,
                                 double* __restrict__ sums) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int m = 0; m < M; ++m) s += (double)x[m * ld + c];
    const double mean = s / M;
    double q = 0.0;
    for (int m = 0; m < M; ++m) { const double d = (double)x[m * ld + c] - mean; q += d * d; }
    const double dm = mean - (pivot ? (double)pivot[c] : 0.0);
    sums[c] = M * dm;
    sums[C + c] = q + M * dm * dm;
}
U2PL_API int u2pl_bn_stats_f32(const float* x, long ld, long M, int C, const float* pivot, void* workspace,
                               double* sums, hipStream_t stream) {
    if (M <= 64) {
        U2PL_LAUNCH(k_bn_stats_small, dim3(cdiv(C, 64)), dim3(64), 0, stream, x, ld, (int)M, C, pivot, sums);
        U2PL_LAUNCH_CHECK();
        return 0;
    }
    StatOp op = {x, ld, pivot};
    return run_colreduce(op, M, 1, C, workspace, sums, stream);
}
// per-image channel sums (AdaptiveAvgPool2d numerator base.py:24; conv bias gradient with nseg=1)
U2PL_API int u2pl_colsum_f32(const float* x, long ld, long Mseg, int nseg, int C, void* workspace, double* sums,
                             hipStream_t stream) {
    SumOp op = {x, ld};
    return run_colreduce(op, Mseg, nseg, C, workspace, sums, stream);
}
U2PL_API int u2pl_bn_bwd_sums_f32(const float* dy, long lddy, const float* x, long ldx, const float* y, long ldy,
                                  const float* mean, const float* invstd, const float* drop, long rows_per_image,
                                  long M, int C, void* workspace, double* sums, hipStream_t stream) {
    BnBwdOp op = {dy, lddy, x, ldx, y, ldy, mean, invstd, drop, rows_per_image, C, nullptr, nullptr};
    return run_colreduce(op, M, 1, C, workspace, sums, stream);
}
// the same for y = relu(BN(x)) WITHOUT reading y: the mask is recomputed from x (gamma, beta: the forward's parameters)
U2PL_API int u2pl_bn_bwd_sums_mx_f32(const float* dy, long lddy, const float* x, long ldx, const float* mean, const float* invstd,
                                     const float* gamma, const float* beta, const float* drop, long rows_per_image, long M, int C,
                                     void* workspace, double* sums, hipStream_t stream) {
    if (!gamma || !beta) return U2PL_EINVAL;
    BnBwdOp op = {dy, lddy, x, ldx, nullptr, 0, mean, invstd, drop, rows_per_image, C, gamma, beta};
    return run_colreduce(op, M, 1, C, workspace, sums, stream);
}

// sums (global, already all-reduced over ranks) -> mean, invstd, running stats.
// torch semantics: biased var for normalisation, unbiased var into running_var,
// running = (1-momentum)*running + momentum*batch.
__global__ void k_bn_finalize(const double* __restrict__ sums, double count, const float* __restrict__ pivot, int C,
                              float eps, float momentum, float* __restrict__ mean, float* __restrict__ invstd,
                              float* __restrict__ running_mean, float* __restrict__ running_var) {
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
        const double m1 = sums[c] / count, m2 = sums[C + c] / count;
        const double p = pivot ? (double)pivot[c] : 0.0;
        double var = m2 - m1 * m1;
        if (var < 0.0) var = 0.0;
        const double mu = p + m1;
        mean[c] = (float)mu;
        invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
        }
    }
}
U2PL_API int u2pl_bn_finalize_f32(const double* sums, double count, const float* pivot, int C, float eps,
                                  float momentum, float* mean, float* invstd, float* running_mean,
                                  float* running_var, hipStream_t stream) {
    U2PL_LAUNCH(k_bn_finalize, dim3(cdiv(C, 256)), dim3(256), 0, stream, sums, count, pivot, C, eps, momentum,
                       mean, invstd, running_mean, running_var);
    U2PL_LAUNCH_CHECK();
    return 0;
}
// ---- round 5: the ordered finish of the conv epilogue's partial statistics AND the BatchNorm finalisation in ONE launch
// (single-rank train mode: nothing sits between the two; under a process group the all-reduce does and the two kernels
// above stay).  Same arithmetic in the same order as k_colreduce_final followed by k_bn_finalize: a block owns 32 channels,
// lanes 0..31 of a wave column reduce S1, lanes 32..63 S2, 16 row groups with 8 loads in flight each, ordered LDS combine.
__global__ __launch_bounds__(1024) void k_bn_finish_finalize(const float* __restrict__ partial, int nblk, int C, double count,
                                                             const float* __restrict__ pivot, float eps, float momentum,
                                                             float* __restrict__ mean, float* __restrict__ invstd,
                                                             float* __restrict__ running_mean, float* __restrict__ running_var,
                                                             double* __restrict__ sums_out) {
    __shared__ double sh[16][64];
    __shared__ double tot[64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 32 + (cl & 31);
    const int i = (cl >> 5) * C + c;                 // column of the [2][C] partial rows: S1 of channel c, or S2
    double acc = 0.0;
    if (c < C) {
        const float* p = partial + i;
        int b = rg;
        for (; b + 7 * 16 < nblk; b += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(long)(b + u * 16) * 2 * C];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += (double)v[u];
        }
        for (; b < nblk; b += 16) acc += (double)p[(long)b * 2 * C];
    }
    sh[rg][cl] = acc;
    __syncthreads();
    if (rg == 0) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sh[k][cl];
        tot[cl] = t;
        if (sums_out && c < C) sums_out[i] = t;
    }
    __syncthreads();
    if (threadIdx.x < 32 && c < C) {
        const double m1 = tot[cl] / count, m2 = tot[32 + cl] / count;
        const double pv = pivot ? (double)pivot[c] : 0.0;
        double var = m2 - m1 * m1;
        if (var < 0.0) var = 0.0;
        const double mu = pv + m1;
        mean[c] = (float)mu;
        invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
        }
    }
}
U2PL_API int u2pl_bn_finish_finalize_f32(const float* partial, int nblk, int C, double count, const float* pivot, float eps,
                                         float momentum, float* mean, float* invstd, float* running_mean, float* running_var,
                                         double* sums_out, hipStream_t stream) {
    if (!partial || nblk <= 0 || C <= 0 || !mean || !invstd) return U2PL_EINVAL;
    U2PL_LAUNCH(k_bn_finish_finalize, dim3(cdiv(C, 32)), dim3(1024), 0, stream, partial, nblk, C, count, pivot, eps, momentum, mean,
                invstd, running_mean, running_var, sums_out);
    U2PL_LAUNCH_CHECK();
    return 0;
}
// eval mode: invstd from running_var
__global__ void k_bn_eval_prep(const float* __restrict__ rv, int C, float eps, float* __restrict__ invstd) {
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x)
        invstd[c] = (float)(1.0 / sqrt((double)rv[c] + (double)eps));
}
U2PL_API int u2pl_bn_eval_invstd_f32(const float* running_var, int C, float eps, float* invstd, hipStream_t stream) {
    U2PL_LAUNCH(k_bn_eval_prep, dim3(cdiv(C, 256)), dim3(256), 0, stream, running_var, C, eps, invstd);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// every BatchNorm of a model in ONE launch (an eval-mode pass used to issue one k_bn_eval_prep per layer: 115 launches of a
// few hundred threads per R101 pass).  jobs: device array of {running_var ptr, invstd ptr, first element, C, eps} (32 bytes).
struct EvalPrepJob { const float* rv; float* out; long long begin; int C; float eps; };
__global__ void k_bn_eval_prep_multi(const EvalPrepJob* __restrict__ jobs, int njobs, long total) {
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        int lo = 0, hi = njobs - 1;
        while (lo < hi) {                     // last job whose first element is <= e
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].begin <= e) lo = mid; else hi = mid - 1;
        }
        const EvalPrepJob j = jobs[lo];
        const int c = (int)(e - j.begin);
        if (c < j.C) j.out[c] = (float)(1.0 / sqrt((double)j.rv[c] + (double)j.eps));
    }
}
U2PL_API size_t u2pl_bn_eval_invstd_job_bytes(void) { return sizeof(EvalPrepJob); }
U2PL_API int u2pl_bn_eval_invstd_multi_f32(const void* jobs_dev, int njobs, long total, hipStream_t stream) {
    if (njobs <= 0 || total <= 0) return 0;
    if (!jobs_dev) return U2PL_EINVAL;
    U2PL_LAUNCH(k_bn_eval_prep_multi, dim3(grid_for(total, 256)), dim3(256), 0, stream, (const EvalPrepJob*)jobs_dev, njobs, total);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// y = [relu]( (x - mean)*invstd*gamma + beta [+ res] ) [* drop[n][c]]
__global__ U2PL_HBM_KERNEL void k_bn_apply(const float* __restrict__ x, long ldx, const float* __restrict__ mean,
                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                           const float* __restrict__ beta, const float* __restrict__ res, long ldr, int relu,
                           const float* __restrict__ drop, long rows_per_image, float* __restrict__ y, long ldy,
                           long M, int C, unsigned* __restrict__ y_amax) {
    const int C4 = C >> 2;
    const long total = M * C4;
    unsigned am = 0u;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C4;
        const int c = (int)(i % C4) * 4;
        float4 v = *(const float4*)(x + r * ldx + c);
        const float4 mu = *(const float4*)(mean + c), is = *(const float4*)(invstd + c);
        const float4 ga = *(const float4*)(gamma + c), be = *(const float4*)(beta + c);
        v.x = (v.x - mu.x) * is.x * ga.x + be.x; v.y = (v.y - mu.y) * is.y * ga.y + be.y;
        v.z = (v.z - mu.z) * is.z * ga.z + be.z; v.w = (v.w - mu.w) * is.w * ga.w + be.w;
        if (res) v = f4add(v, *(const float4*)(res + r * ldr + c));
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (drop) v = f4mul(v, *(const float4*)(drop + (r / rows_per_image) * C + c));
        *(float4*)(y + r * ldy + c) = v;
        am = amax_bits4(am, v);
    }
    if (y_amax) amax_wave_publish(am, y_amax);       // (uniform branch; split-fp16: the output is a GEMM operand)
}
U2PL_API int u2pl_bn_apply_f32(const float* x, long ldx, const float* mean, const float* invstd, const float* gamma,
                               const float* beta, const float* res, long ldr, int relu, const float* drop,
                               long rows_per_image, float* y, long ldy, long M, int C, hipStream_t stream) {
    if (C % 4) return U2PL_EINVAL;
    U2PL_LAUNCH(k_bn_apply, dim3(grid_for(M * (C / 4), 256)), dim3(256), 0, stream, x, ldx, mean, invstd, gamma,
                       beta, res, ldr, relu, drop, rows_per_image, y, ldy, M, C, (unsigned*)nullptr);
    U2PL_LAUNCH_CHECK();
    return 0;
}
// the same + max |y| into *y_amax (a device float the caller zeroed: the x_amax of the split-fp16 GEMMs that read y)
U2PL_API int u2pl_bn_apply_amax_f32(const float* x, long ldx, const float* mean, const float* invstd, const float* gamma,
                                    const float* beta, const float* res, long ldr, int relu, const float* drop,
                                    long rows_per_image, float* y, long ldy, long M, int C, float* y_amax, hipStream_t stream) {
    if (C % 4) return U2PL_EINVAL;
    U2PL_LAUNCH(k_bn_apply, dim3(grid_for(M * (C / 4), 256)), dim3(256), 0, stream, x, ldx, mean, invstd, gamma,
                       beta, res, ldr, relu, drop, rows_per_image, y, ldy, M, C, (unsigned*)y_amax);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// g = dy*[y>0]*drop ; dres = g ; dx = gamma*invstd*(g - S0/cnt - xhat*S1/cnt)  (train)
//                                dx = gamma*invstd*g                           (eval: sums == NULL)
__global__ U2PL_HBM_KERNEL void k_bn_bwd_apply(const float* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx,
                               const float* __restrict__ y, long ldy, const float* __restrict__ mean,
                               const float* __restrict__ invstd, const float* __restrict__ gamma,
                               const float* __restrict__ drop, long rows_per_image,
                               const double* __restrict__ sums, double count, float* __restrict__ dx, long lddx,
                               float* __restrict__ dres, long lddr, long M, int C, const double* __restrict__ psums,
                               float* __restrict__ gsink, float* __restrict__ bsink, int accumulate,
                               unsigned* __restrict__ dx_amax, unsigned* __restrict__ dres_amax, const float* __restrict__ relu_beta) {
    unsigned am_x = 0u, am_r = 0u;
    if (psums) {   // (u2pl_bn_bwd_apply_pg_f32) the parameter gradients ride along: k_sums_to_f32's arithmetic, dgamma = S1, dbeta = S0
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < C; i += (long)gridDim.x * blockDim.x) {
            gsink[i] = (accumulate ? gsink[i] : 0.f) + (float)(psums[C + i] * (double)1.0f);
            bsink[i] = (accumulate ? bsink[i] : 0.f) + (float)(psums[i] * (double)1.0f);
        }
    }
    const int C4 = C >> 2;
    const long total = M * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C4;
        const int c = (int)(i % C4) * 4;
        float4 g = *(const float4*)(dy + r * lddy + c);
        if (y) {
            const float4 yy = *(const float4*)(y + r * ldy + c);
            g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
            g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
        } else if (relu_beta) {       // (y = relu(BN(x)) without a residual: the mask from x, see relu_mask_from_x)
            g = relu_mask_from_x(g, *(const float4*)(x + r * ldx + c), mean, invstd, gamma, relu_beta, c);
        }
        if (drop) g = f4mul(g, *(const float4*)(drop + (r / rows_per_image) * C + c));
        if (dres) *(float4*)(dres + r * lddr + c) = g;
        am_r = amax_bits4(am_r, g);
        const float4 is = *(const float4*)(invstd + c), ga = *(const float4*)(gamma + c);
        float4 o;
        if (sums) {
            const float4 xv = *(const float4*)(x + r * ldx + c), mu = *(const float4*)(mean + c);
            const float m0x = (float)(sums[c] / count), m0y = (float)(sums[c + 1] / count);
            const float m0z = (float)(sums[c + 2] / count), m0w = (float)(sums[c + 3] / count);
            const float m1x = (float)(sums[C + c] / count), m1y = (float)(sums[C + c + 1] / count);
            const float m1z = (float)(sums[C + c + 2] / count), m1w = (float)(sums[C + c + 3] / count);
            o.x = ga.x * is.x * (g.x - m0x - (xv.x - mu.x) * is.x * m1x);
            o.y = ga.y * is.y * (g.y - m0y - (xv.y - mu.y) * is.y * m1y);
            o.z = ga.z * is.z * (g.z - m0z - (xv.z - mu.z) * is.z * m1z);
            o.w = ga.w * is.w * (g.w - m0w - (xv.w - mu.w) * is.w * m1w);
        } else {
            o = f4mul(f4mul(ga, is), g);
        }
        *(float4*)(dx + r * lddx + c) = o;
        am_x = amax_bits4(am_x, o);
    }
    if (dx_amax) amax_wave_publish(am_x, dx_amax);
    if (dres_amax) amax_wave_publish(am_r, dres_amax);
}
U2PL_API int u2pl_bn_bwd_apply_f32(const float* dy, long lddy, const float* x, long ldx, const float* y, long ldy,
                                   const float* mean, const float* invstd, const float* gamma, const float* drop,
                                   long rows_per_image, const double* sums, double count, float* dx, long lddx,
                                   float* dres, long lddr, long M, int C, hipStream_t stream) {
    if (C % 4) return U2PL_EINVAL;
    U2PL_LAUNCH(k_bn_bwd_apply, dim3(grid_for(M * (C / 4), 256)), dim3(256), 0, stream, dy, lddy, x, ldx, y, ldy,
                       mean, invstd, gamma, drop, rows_per_image, sums, count, dx, lddx, dres, lddr, M, C, (const double*)nullptr,
                       (float*)nullptr, (float*)nullptr, 0, (unsigned*)nullptr, (unsigned*)nullptr, (const float*)nullptr);
    U2PL_LAUNCH_CHECK();
    return 0;
}
// the same + the layer's parameter gradients written by the same launch (dgamma = psums[C..2C), dbeta = psums[0..C): the LOCAL
// backward sums -- single-rank training passes sums == psums; `accumulate` != 0 adds into gsink / bsink like u2pl_sums_to_f32)
U2PL_API int u2pl_bn_bwd_apply_pg_f32(const float* dy, long lddy, const float* x, long ldx, const float* y, long ldy,
                                      const float* mean, const float* invstd, const float* gamma, const float* drop,
                                      long rows_per_image, const double* sums, double count, float* dx, long lddx,
                                      float* dres, long lddr, long M, int C, const double* psums, float* gsink, float* bsink,
                                      int accumulate, hipStream_t stream) {
    if (C % 4 || !psums || !gsink || !bsink) return U2PL_EINVAL;
    U2PL_LAUNCH(k_bn_bwd_apply, dim3(grid_for(M * (C / 4), 256)), dim3(256), 0, stream, dy, lddy, x, ldx, y, ldy,
                       mean, invstd, gamma, drop, rows_per_image, sums, count, dx, lddx, dres, lddr, M, C, psums, gsink, bsink,
                       accumulate, (unsigned*)nullptr, (unsigned*)nullptr, (const float*)nullptr);
    U2PL_LAUNCH_CHECK();
    return 0;
}
// u2pl_bn_bwd_apply_f32 (psums == NULL) / u2pl_bn_bwd_apply_pg_f32 + max |dx| and max |dres| into the caller-zeroed device floats
// dx_amax / dres_amax (either may be NULL): the dy_amax of the split-fp16 data- and weight-gradient GEMMs that read them
U2PL_API int u2pl_bn_bwd_apply_amax_f32(const float* dy, long lddy, const float* x, long ldx, const float* y, long ldy,
                                        const float* mean, const float* invstd, const float* gamma, const float* drop,
                                        long rows_per_image, const double* sums, double count, float* dx, long lddx,
                                        float* dres, long lddr, long M, int C, const double* psums, float* gsink, float* bsink,
                                        int accumulate, float* dx_amax, float* dres_amax, const float* relu_beta, hipStream_t stream) {
    if (C % 4 || (psums && (!gsink || !bsink)) || (relu_beta && y)) return U2PL_EINVAL;
    U2PL_LAUNCH(k_bn_bwd_apply, dim3(grid_for(M * (C / 4), 256)), dim3(256), 0, stream, dy, lddy, x, ldx, y, ldy,
                       mean, invstd, gamma, drop, rows_per_image, sums, count, dx, lddx, dres, lddr, M, C, psums, gsink, bsink,
                       accumulate, (unsigned*)dx_amax, (unsigned*)dres_amax, relu_beta);
    U2PL_LAUNCH_CHECK();
    return 0;
}
// dgamma = S1 (sum g*xhat), dbeta = S0 (sum g): double -> float (optionally accumulate)
__global__ void k_sums_to_f32(const double* __restrict__ s, int n, float scale, int accumulate, float* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[i] = (accumulate ? out[i] : 0.f) + (float)(s[i] * (double)scale);
}
U2PL_API int u2pl_sums_to_f32(const double* sums, int n, float scale, int accumulate, float* out, hipStream_t stream) {
    U2PL_LAUNCH(k_sums_to_f32, dim3(cdiv(n, 256)), dim3(256), 0, stream, sums, n, scale, accumulate, out);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// nn.MaxPool2d(3, 2, 1, ceil_mode=True)  (resnet.py:189-191); tap index saved
// ---------------------------------------------------------------------------
__global__ void k_maxpool_fwd(const float* __restrict__ x, long ldx, int N, int H, int W, int C, int Ho, int Wo,
                              float* __restrict__ y, long ldy, unsigned char* __restrict__ tap) {
    const int C4 = C >> 2;
    const long total = (long)N * Ho * Wo * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        long p = i / C4;
        const int wo = (int)(p % Wo);
        long t = p / Wo;
        const int ho = (int)(t % Ho), n = (int)(t / Ho);
        float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        uchar4 bt = make_uchar4(0, 0, 0, 0);
        bool first = true;
        for (int r = 0; r < 3; ++r) {
            const int ih = ho * 2 - 1 + r;
            if (ih < 0 || ih >= H) continue;
            for (int s = 0; s < 3; ++s) {
                const int iw = wo * 2 - 1 + s;
                if (iw < 0 || iw >= W) continue;
                const float4 v = *(const float4*)(x + ((long)(n * H + ih) * W + iw) * ldx + c);
                const unsigned char k = (unsigned char)(r * 3 + s);
                if (first || v.x > best.x || v.x != v.x) { best.x = v.x; bt.x = k; }
                if (first || v.y > best.y || v.y != v.y) { best.y = v.y; bt.y = k; }
                if (first || v.z > best.z || v.z != v.z) { best.z = v.z; bt.z = k; }
                if (first || v.w > best.w || v.w != v.w) { best.w = v.w; bt.w = k; }
                first = false;
            }
        }
        *(float4*)(y + p * ldy + c) = best;
        *(uchar4*)(tap + p * C + c) = bt;
    }
}
U2PL_API int u2pl_maxpool3s2_fwd_f32(const float* x, long ldx, int N, int H, int W, int C, int Ho, int Wo, float* y,
                                     long ldy, unsigned char* tap, hipStream_t stream) {
    if (C % 4) return U2PL_EINVAL;
    U2PL_LAUNCH(k_maxpool_fwd, dim3(grid_for((long)N * Ho * Wo * (C / 4), 256)), dim3(256), 0, stream, x, ldx, N, H,
                       W, C, Ho, Wo, y, ldy, tap);
    U2PL_LAUNCH_CHECK();
    return 0;
}
// gather form: every input pixel checks the <= 4 windows that cover it
__global__ void k_maxpool_bwd(const float* __restrict__ dy, long lddy, const unsigned char* __restrict__ tap, int N,
                              int H, int W, int C, int Ho, int Wo, float* __restrict__ dx, long lddx) {
    const int C4 = C >> 2;
    const long total = (long)N * H * W * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        long p = i / C4;
        const int iw = (int)(p % W);
        long t = p / W;
        const int ih = (int)(t % H), n = (int)(t / H);
        float4 acc = f4zero();
        for (int ho = ih / 2; ho <= (ih + 1) / 2; ++ho) {
            if (ho >= Ho) continue;
            const int r = ih - (ho * 2 - 1);
            if (r < 0 || r > 2) continue;
            for (int wo = iw / 2; wo <= (iw + 1) / 2; ++wo) {
                if (wo >= Wo) continue;
                const int s = iw - (wo * 2 - 1);
                if (s < 0 || s > 2) continue;
                const long q = ((long)n * Ho + ho) * Wo + wo;
                const uchar4 k = *(const uchar4*)(tap + q * C + c);
                const float4 g = *(const float4*)(dy + q * lddy + c);
                const unsigned char me = (unsigned char)(r * 3 + s);
                if (k.x == me) acc.x += g.x;
                if (k.y == me) acc.y += g.y;
                if (k.z == me) acc.z += g.z;
                if (k.w == me) acc.w += g.w;
            }
        }
        *(float4*)(dx + p * lddx + c) = acc;
    }
}
U2PL_API int u2pl_maxpool3s2_bwd_f32(const float* dy, long lddy, const unsigned char* tap, int N, int H, int W, int C,
                                     int Ho, int Wo, float* dx, long lddx, hipStream_t stream) {
    U2PL_LAUNCH(k_maxpool_bwd, dim3(grid_for((long)N * H * W * (C / 4), 256)), dim3(256), 0, stream, dy, lddy, tap, N,
                       H, W, C, Ho, Wo, dx, lddx);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// row utilities: strided copy (concat / slice), per-image broadcast of a
// [N][C] vector (bilinear up-sampling of a 1x1 map, base.py:92-94, and the
// backward of the global average pool), scaled
// ---------------------------------------------------------------------------
__global__ void k_copy_rows(const float* __restrict__ src, long lds, float* __restrict__ dst, long ldd, long M, int C,
                            int accumulate) {
    const int C4 = C >> 2;
    const long total = M * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C4;
        const int c = (int)(i % C4) * 4;
        float4 v = *(const float4*)(src + r * lds + c);
        if (accumulate) v = f4add(v, *(const float4*)(dst + r * ldd + c));
        *(float4*)(dst + r * ldd + c) = v;
    }
}
U2PL_API int u2pl_copy_rows_f32(const float* src, long lds, float* dst, long ldd, long M, int C, int accumulate,
                                hipStream_t stream) {
    if (C % 4) return U2PL_EINVAL;
    if (M <= 0) return 0;
    U2PL_LAUNCH(k_copy_rows, dim3(grid_for(M * (C / 4), 256)), dim3(256), 0, stream, src, lds, dst, ldd, M, C, accumulate);
    U2PL_LAUNCH_CHECK();
    return 0;
}
// scalar variant for narrow rows (C = num_classes, ld not 16 B aligned): dst[r][c] = src[r][c], c < C
__global__ void k_copy_cols(const float* __restrict__ src, long lds, float* __restrict__ dst, long ldd, long M, int C) {
    const long total = M * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C;
        const int c = (int)(i % C);
        dst[r * ldd + c] = src[r * lds + c];
    }
}
U2PL_API int u2pl_copy_cols_f32(const float* src, long lds, float* dst, long ldd, long M, int C, hipStream_t stream) {
    if (M <= 0) return 0;
    U2PL_LAUNCH(k_copy_cols, dim3(grid_for(M * C, 256)), dim3(256), 0, stream, src, lds, dst, ldd, M, C);
    U2PL_LAUNCH_CHECK();
    return 0;
}
__global__ void k_broadcast_rows(const float* __restrict__ v, long ldv, float scale, float* __restrict__ dst, long ldd,
                                 long rows_per_image, long M, int C) {
    const int C4 = C >> 2;
    const long total = M * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C4;
        const int c = (int)(i % C4) * 4;
        float4 a = *(const float4*)(v + (r / rows_per_image) * ldv + c);
        a.x *= scale; a.y *= scale; a.z *= scale; a.w *= scale;
        *(float4*)(dst + r * ldd + c) = a;
    }
}
U2PL_API int u2pl_broadcast_rows_f32(const float* v, long ldv, float scale, float* dst, long ldd, long rows_per_image,
                                     long M, int C, hipStream_t stream) {
    if (C % 4) return U2PL_EINVAL;
    U2PL_LAUNCH(k_broadcast_rows, dim3(grid_for(M * (C / 4), 256)), dim3(256), 0, stream, v, ldv, scale, dst, ldd,
                       rows_per_image, M, C);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// bilinear (align_corners=True) up-sampling of NHWC feature maps
// (decoder.py:114-116: 97^2 -> 193^2, 256 channels) forward / backward
// ---------------------------------------------------------------------------
__global__ void k_bilinear_rows_fwd(const float* __restrict__ x, long ldx, int N, int h, int w, int C, int H, int W,
                                    float sy, float sx, float* __restrict__ y, long ldy) {
    const int C4 = C >> 2;
    const long total = (long)N * H * W * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        long p = i / C4;
        const int ox = (int)(p % W);
        long t = p / W;
        const int oy = (int)(t % H), n = (int)(t / H);
        const AcCoord cy = ac_coord(oy, sy, h), cx = ac_coord(ox, sx, w);
        const float* b = x + (long)n * h * w * ldx + c;
        const float4 v00 = *(const float4*)(b + ((long)cy.i0 * w + cx.i0) * ldx), v01 = *(const float4*)(b + ((long)cy.i0 * w + cx.i1) * ldx);
        const float4 v10 = *(const float4*)(b + ((long)cy.i1 * w + cx.i0) * ldx), v11 = *(const float4*)(b + ((long)cy.i1 * w + cx.i1) * ldx);
        float4 o;
#define BL(f) o.f = __fmaf_rn(cy.l0, __fmaf_rn(cx.l0, v00.f, __fmul_rn(cx.l1, v01.f)), __fmul_rn(cy.l1, __fmaf_rn(cx.l0, v10.f, __fmul_rn(cx.l1, v11.f))))
        BL(x); BL(y); BL(z); BL(w);
#undef BL
        *(float4*)(y + p * ldy + c) = o;
    }
}
U2PL_API int u2pl_bilinear_rows_fwd_f32(const float* x, long ldx, int N, int h, int w, int C, int H, int W, float* y,
                                        long ldy, hipStream_t stream) {
    if (C % 4) return U2PL_EINVAL;
    U2PL_LAUNCH(k_bilinear_rows_fwd, dim3(grid_for((long)N * H * W * (C / 4), 256)), dim3(256), 0, stream, x, ldx, N, h,
                       w, C, H, W, ac_scale_host(h, H), ac_scale_host(w, W), y, ldy);
    U2PL_LAUNCH_CHECK();
    return 0;
}
__global__ void k_bilinear_rows_bwd(const float* __restrict__ dy, long lddy, int N, int h, int w, int C, int H, int W,
                                    float sy, float sx, float* __restrict__ dx, long lddx) {
    const int C4 = C >> 2;
    const long total = (long)N * h * w * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        long p = i / C4;
        const int ix = (int)(p % w);
        long t = p / w;
        const int iy = (int)(t % h), n = (int)(t / h);
        int oy_lo = sy > 0 ? (int)floorf((iy - 1) / sy) - 1 : 0, oy_hi = sy > 0 ? (int)ceilf((iy + 1) / sy) + 1 : H - 1;
        int ox_lo = sx > 0 ? (int)floorf((ix - 1) / sx) - 1 : 0, ox_hi = sx > 0 ? (int)ceilf((ix + 1) / sx) + 1 : W - 1;
        oy_lo = max(oy_lo, 0); ox_lo = max(ox_lo, 0);
        oy_hi = min(oy_hi, H - 1); ox_hi = min(ox_hi, W - 1);
        float4 acc = f4zero();
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            const AcCoord cy = ac_coord(oy, sy, h);
            if (cy.i0 != iy && cy.i1 != iy) continue;
            const float wy = (cy.i0 == iy ? cy.l0 : 0.f) + (cy.i1 == iy ? cy.l1 : 0.f);
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                const AcCoord cx = ac_coord(ox, sx, w);
                if (cx.i0 != ix && cx.i1 != ix) continue;
                const float ww = wy * ((cx.i0 == ix ? cx.l0 : 0.f) + (cx.i1 == ix ? cx.l1 : 0.f));
                const float4 g = *(const float4*)(dy + (((long)n * H + oy) * W + ox) * lddy + c);
                acc.x += ww * g.x; acc.y += ww * g.y; acc.z += ww * g.z; acc.w += ww * g.w;
            }
        }
        *(float4*)(dx + p * lddx + c) = acc;
    }
}
U2PL_API int u2pl_bilinear_rows_bwd_f32(const float* dy, long lddy, int N, int h, int w, int C, int H, int W, float* dx,
                                        long lddx, hipStream_t stream) {
    if (C % 4) return U2PL_EINVAL;
    U2PL_LAUNCH(k_bilinear_rows_bwd, dim3(grid_for((long)N * h * w * (C / 4), 256)), dim3(256), 0, stream, dy, lddy, N,
                       h, w, C, H, W, ac_scale_host(h, H), ac_scale_host(w, W), dx, lddx);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// softmax over the channel axis of strided low-res logits (train_semi.py:365)
// ---------------------------------------------------------------------------
__global__ void k_softmax_rows(const float* __restrict__ x, long ldx, float* __restrict__ y, long ldy, long M, int C) {
    for (long r = blockIdx.x * (long)blockDim.x + threadIdx.x; r < M; r += (long)gridDim.x * blockDim.x) {
        const float* b = x + r * ldx;
        float m = b[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, b[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(b[c] - m);
        for (int c = 0; c < C; ++c) y[r * ldy + c] = expf(b[c] - m) / s;
    }
}
U2PL_API int u2pl_softmax_rows_f32(const float* x, long ldx, float* y, long ldy, long M, int C, hipStream_t stream) {
    U2PL_LAUNCH(k_softmax_rows, dim3(grid_for(M, 256)), dim3(256), 0, stream, x, ldx, y, ldy, M, C);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// optimizer / EMA on flat parameter arenas
//   torch.optim.SGD: g += wd*p ; buf = first ? g : mom*buf + g ; p -= lr*buf
//   (lr_helper.py:18-19), per-segment lr (3 param groups, train_semi.py:100-110)
//   EMA: t = d*t + (1-d)*s   (train_semi.py:543-548)
// ---------------------------------------------------------------------------
__global__ void k_sgd(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long n, long b1,
                      long b2, float lr0, float lr1, float lr2, float mom, float wd, int first, float gscale) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float lr = i < b1 ? lr0 : (i < b2 ? lr1 : lr2);
        const float pv = p[i];
        const float gv = __fadd_rn(__fmul_rn(g[i], gscale), __fmul_rn(wd, pv));
        const float bv = first ? gv : __fadd_rn(__fmul_rn(mom, buf[i]), gv);
        buf[i] = bv;
        p[i] = __fsub_rn(pv, __fmul_rn(lr, bv));
    }
}
U2PL_API int u2pl_sgd_step_f32(float* p, const float* g, float* buf, long n, long b1, long b2, float lr0, float lr1,
                               float lr2, float momentum, float weight_decay, int first, float grad_scale,
                               hipStream_t stream) {
    if (n <= 0) return 0;
    U2PL_LAUNCH(k_sgd, dim3(grid_for(n, 256)), dim3(256), 0, stream, p, g, buf, n, b1, b2, lr0, lr1, lr2, momentum,
                       weight_decay, first, grad_scale);
    U2PL_LAUNCH_CHECK();
    return 0;
}
// torch.optim.Adam (lr_helper.py:20-21 `optim.Adam(parms, **kwargs)`; amsgrad = False) on the flat arena, torch's
// single-tensor update order:  g += wd p;  m += (1 - b1)(g - m);  v = b2 v + (1 - b2) g g;
// p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps).   bc1 = 1 - b1^t and bc2s = sqrt(1 - b2^t) come from the host.
__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                       long b1, long b2, float lr0, float lr1, float lr2, float beta1, float beta2, float eps, float wd,
                       float bc1, float bc2s, float gscale) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float lr = i < b1 ? lr0 : (i < b2 ? lr1 : lr2);
        const float pv = p[i];
        const float gv = __fadd_rn(__fmul_rn(g[i], gscale), __fmul_rn(wd, pv));
        const float mv = __fadd_rn(m[i], __fmul_rn(1.0f - beta1, __fsub_rn(gv, m[i])));             // lerp_(grad, 1 - beta1), weight < 0.5
        const float vv = __fadd_rn(__fmul_rn(v[i], beta2), __fmul_rn(1.0f - beta2, __fmul_rn(gv, gv)));
        m[i] = mv;
        v[i] = vv;
        const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(vv), bc2s), eps);
        p[i] = __fsub_rn(pv, __fmul_rn(__fdiv_rn(lr, bc1), __fdiv_rn(mv, denom)));
    }
}
U2PL_API int u2pl_adam_step_f32(float* p, const float* g, float* exp_avg, float* exp_avg_sq, long n, long b1, long b2,
                                float lr0, float lr1, float lr2, float beta1, float beta2, float eps, float weight_decay,
                                float bias_correction1, float bias_correction2_sqrt, float grad_scale, hipStream_t stream) {
    if (n <= 0) return 0;
    if (!(bias_correction1 > 0.f) || !(bias_correction2_sqrt > 0.f)) return U2PL_EINVAL;
    U2PL_LAUNCH(k_adam, dim3(grid_for(n, 256)), dim3(256), 0, stream, p, g, exp_avg, exp_avg_sq, n, b1, b2, lr0, lr1, lr2, beta1,
                beta2, eps, weight_decay, bias_correction1, bias_correction2_sqrt, grad_scale);
    U2PL_LAUNCH_CHECK();
    return 0;
}
__global__ void k_ema(float* __restrict__ t, const float* __restrict__ s, long n, float d, float omd) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        t[i] = __fadd_rn(__fmul_rn(d, t[i]), __fmul_rn(omd, s[i]));
}
U2PL_API int u2pl_ema_update_f32(float* t, const float* s, long n, float decay, float one_minus_decay,
                                 hipStream_t stream) {
    if (n <= 0) return 0;
    U2PL_LAUNCH(k_ema, dim3(grid_for(n, 256)), dim3(256), 0, stream, t, s, n, decay, one_minus_decay);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// 1x1 convolution on a 1x1 map = dense layer on M <= 16 rows (ASPP image-pooling branch, base.py:24-28): the rows
// are the global averages of the batch, which differ from each other by ~1 %, and the BatchNorm that follows
// normalises over just those M values, i.e. it amplifies the relative error of y by |y| / |y_m - y_m'| ~ 50x.
// The products are therefore accumulated in float64 (one wave per output channel, lanes stride K): y is the
// correctly rounded dot product, at a cost of M * K * Cout = 2 MFLOP.
// ---------------------------------------------------------------------------
#define DENSE_MAXM 16
__global__ __launch_bounds__(256) void k_dense_small(const float* __restrict__ x, long ldx, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ y, long ldy,
                                                     int M, int K, int Cout) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (o >= Cout) return;
    double acc[DENSE_MAXM];
#pragma unroll
    for (int m = 0; m < DENSE_MAXM; ++m) acc[m] = 0.0;
    const float* wr = w + (long)o * K;
    for (int k = lane * 4; k < K; k += 256) {
        const float4 wv = *(const float4*)(wr + k);
#pragma unroll
        for (int m = 0; m < DENSE_MAXM; ++m)
            if (m < M) {
                const float4 xv = *(const float4*)(x + m * ldx + k);
                acc[m] += (double)xv.x * (double)wv.x + (double)xv.y * (double)wv.y + (double)xv.z * (double)wv.z +
                          (double)xv.w * (double)wv.w;
            }
    }
#pragma unroll
    for (int m = 0; m < DENSE_MAXM; ++m)
        if (m < M) {
            const double t = wave_sum_d(acc[m]);
            if (lane == 0) y[m * ldy + o] = (float)(t + (bias ? (double)bias[o] : 0.0));
        }
}
U2PL_API int u2pl_dense_small_f32(const float* x, long ldx, const float* w, const float* bias, float* y, long ldy, int M,
                                  int K, int Cout, hipStream_t stream) {
    if (M < 1 || M > DENSE_MAXM || K % 4) return U2PL_EINVAL;
    U2PL_LAUNCH(k_dense_small, dim3(cdiv(Cout, 4)), dim3(256), 0, stream, x, ldx, w, bias, y, ldy, M, K, Cout);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// CutMix (augmentation.py:498-541): dst[i] = box_i ? src[(i+1)%B] : src[i]
// for image (float, C planes), label (int64) and confidence (float)
// boxes: int32 [B][4] = y0,y1,x0,x1 (device)
// ---------------------------------------------------------------------------
__global__ void k_cutmix(const float* __restrict__ img, const long long* __restrict__ lab, const float* __restrict__ conf,
                         const int* __restrict__ boxes, int B, int C, int H, int W, float* __restrict__ oimg,
                         long long* __restrict__ olab, float* __restrict__ oconf) {
    const long HW = (long)H * W, total = (long)B * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / HW);
        const long q = i % HW;
        const int yy = (int)(q / W), xx = (int)(q % W);
        const int* bx = boxes + 4 * b;
        const bool in = yy >= bx[0] && yy < bx[1] && xx >= bx[2] && xx < bx[3];
        const int sb = in ? (b + 1) % B : b;
        for (int c = 0; c < C; ++c) oimg[((long)b * C + c) * HW + q] = img[((long)sb * C + c) * HW + q];
        olab[i] = lab[(long)sb * HW + q];
        oconf[i] = conf[(long)sb * HW + q];
    }
}
U2PL_API int u2pl_cutmix_f32(const float* img, const long long* label, const float* conf, const int* boxes_dev, int B,
                             int C, int H, int W, float* out_img, long long* out_label, float* out_conf,
                             hipStream_t stream) {
    U2PL_LAUNCH(k_cutmix, dim3(grid_for((long)B * H * W, 256)), dim3(256), 0, stream, img, label, conf, boxes_dev, B, C,
                       H, W, out_img, out_label, out_conf);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// The other two strong augmentations of generate_unsup_data (augmentation.py:486-541), same arithmetic as the
// reference (`a * m + b * (1 - m)` in fp32, so signed zeros / non-finite inputs behave identically):
//   mode 1 "cutout":   m = 0 inside box_i, 1 outside;  img*m, conf*m, label = 255 inside (augmentation.py:506-513)
//   mode 2 "classmix": m = [label_i(y,x) in selected_i];  out = x_i*m + x_{(i+1)%B}*(1-m)  (augmentation.py:517-535)
// boxes: int32 [B][4] = y0,y1,x0,x1;  sel: uint64 [B], bit c = class c of image i selected (generate_class_mask)
// ---------------------------------------------------------------------------
__global__ void k_strong_aug(const float* __restrict__ img, const long long* __restrict__ lab, const float* __restrict__ conf,
                             const int* __restrict__ boxes, const unsigned long long* __restrict__ sel, int mode, int B,
                             int C, int H, int W, float* __restrict__ oimg, long long* __restrict__ olab,
                             float* __restrict__ oconf) {
    const long HW = (long)H * W, total = (long)B * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / HW);
        const long q = i % HW;
        const long long l = lab[i];
        if (mode == 1) {
            const int yy = (int)(q / W), xx = (int)(q % W);
            const int* bx = boxes + 4 * b;
            const bool in = yy >= bx[0] && yy < bx[1] && xx >= bx[2] && xx < bx[3];
            const float m = in ? 0.f : 1.f;
            for (int c = 0; c < C; ++c) oimg[((long)b * C + c) * HW + q] = img[((long)b * C + c) * HW + q] * m;
            olab[i] = in ? 255 : l;
            oconf[i] = conf[i] * m;
        } else {
            const int nb = (b + 1) % B;
            const float m = (l >= 0 && l < 64 && ((sel[b] >> l) & 1ull)) ? 1.f : 0.f, m1 = 1.f - m;
            for (int c = 0; c < C; ++c)
                oimg[((long)b * C + c) * HW + q] = img[((long)b * C + c) * HW + q] * m + img[((long)nb * C + c) * HW + q] * m1;
            olab[i] = (long long)((float)l * m + (float)lab[(long)nb * HW + q] * m1);
            oconf[i] = conf[i] * m + conf[(long)nb * HW + q] * m1;
        }
    }
}
U2PL_API int u2pl_strong_aug_f32(const float* img, const long long* label, const float* conf, const int* boxes_dev,
                                 const unsigned long long* sel_dev, int mode, int B, int C, int H, int W, float* out_img,
                                 long long* out_label, float* out_conf, hipStream_t stream) {
    if (mode != 1 && mode != 2) return 1;
    if ((mode == 1 && !boxes_dev) || (mode == 2 && !sel_dev)) return 1;
    U2PL_LAUNCH(k_strong_aug, dim3(grid_for((long)B * H * W, 256)), dim3(256), 0, stream, img, label, conf, boxes_dev,
                       sel_dev, mode, B, C, H, W, out_img, out_label, out_conf);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// torch.unique(pseudo_labels) of generate_class_mask (augmentation.py:488) as a presence bitmask per image:
// bits[b] |= 1 << label for every pixel (labels outside [0,64) are ignored).  bits must be zeroed by the caller.
__global__ void k_label_presence(const long long* __restrict__ lab, long HW, unsigned long long* __restrict__ bits) {
    const int b = blockIdx.y;
    unsigned long long acc = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < HW; i += (long)gridDim.x * blockDim.x) {
        const long long l = lab[(long)b * HW + i];
        if (l >= 0 && l < 64) acc |= 1ull << l;
    }
    for (int o = 32; o > 0; o >>= 1) acc |= __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicOr(bits + b, acc);
}
U2PL_API int u2pl_label_presence_i64(const long long* label, int B, long HW, unsigned long long* bits, hipStream_t stream) {
    long nb = (HW + 255) / 256;
    if (nb > 256) nb = 256;
    U2PL_LAUNCH(k_label_presence, dim3((unsigned)nb, B), dim3(256), 0, stream, label, HW, bits);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// Sliding-window evaluation (reference eval.py:184-224, scale_crop_process): logits of one crop window are
// added into the padded full-image accumulator, a per-pixel window count is kept, and the sum is divided by
// the count at the end.  pred: [C][H][W] planar, count: [H][W], src: [C][hc][wc] planar (one window).
// ---------------------------------------------------------------------------
__global__ void k_window_acc(float* __restrict__ pred, float* __restrict__ count, int C, int H, int W,
                             const float* __restrict__ src, int h0, int w0, int hc, int wc) {
    const long total = (long)C * hc * wc;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % wc);
        const long t = i / wc;
        const int y = (int)(t % hc), c = (int)(t / hc);
        pred[((long)c * H + h0 + y) * W + w0 + x] += src[i];
        if (c == 0) count[(long)(h0 + y) * W + w0 + x] += 1.0f;
    }
}
U2PL_API int u2pl_window_accumulate_f32(float* pred, float* count, int C, int H, int W, const float* src, int h0,
                                        int w0, int hc, int wc, hipStream_t stream) {
    if (h0 < 0 || w0 < 0 || h0 + hc > H || w0 + wc > W) return U2PL_EINVAL;
    const long total = (long)C * hc * wc;
    if (total <= 0) return 0;
    U2PL_LAUNCH(k_window_acc, dim3(grid_for(total, 256)), dim3(256), 0, stream, pred, count, C, H, W, src, h0, w0, hc, wc);
    U2PL_LAUNCH_CHECK();
    return 0;
}
__global__ void k_window_div(float* __restrict__ pred, const float* __restrict__ count, int C, long HW) {
    const long total = (long)C * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        pred[i] = pred[i] / count[i % HW];
}
U2PL_API int u2pl_window_normalize_f32(float* pred, const float* count, int C, int H, int W, hipStream_t stream) {
    const long total = (long)C * H * W;
    if (total <= 0) return 0;
    U2PL_LAUNCH(k_window_div, dim3(grid_for(total, 256)), dim3(256), 0, stream, pred, count, C, (long)H * W);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// Training-time data pipeline on the device (reference u2pl/dataset/augmentation.py:51-266 as
// composed by cityscapes.py:47-77): ToTensor -> Normalize -> RandResize (bilinear align_corners=False for
// the image, legacy nearest for the label) -> RandomHorizontalFlip -> Crop (zero padding of image AND label,
// augmentation.py:241-245) fused into ONE gather per output pixel from the decoded uint8 sample.  The random
// draws stay on the host (python `random`, reference order); per sample
//   params = {rh, rw (resized size), flip, pad_top, pad_left, ho, wo (crop origin in the padded image), 0}.
// img: uint8 [B][H][W][3] (decoder layout), lab: uint8 [B][H][W]; out_img fp32 [B][3][Sh][Sw], out_lab int64.
// ---------------------------------------------------------------------------
__global__ void k_augment(const unsigned char* __restrict__ img, const unsigned char* __restrict__ lab,
                          const int* __restrict__ params, int B, int H, int W, int Sh, int Sw, float m0, float m1,
                          float m2, float s0, float s1, float s2, float* __restrict__ out_img,
                          long long* __restrict__ out_lab) {
    const long total = (long)B * Sh * Sw;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % Sw);
        const long t = i / Sw;
        const int y = (int)(t % Sh), b = (int)(t / Sh);
        const int* p = params + b * 8;
        const int rh = p[0], rw = p[1], flip = p[2];
        const int ry = p[5] + y - p[3];
        int rx = p[6] + x - p[4];
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        long long l = 0;
        if (ry >= 0 && ry < rh && rx >= 0 && rx < rw) {
            if (flip) rx = rw - 1 - rx;
            const unsigned char* ib = img + (long)b * H * W * 3;
            const unsigned char* lb = lab + (long)b * H * W;
            // label: legacy nearest, src = min(floor(dst * float(in/out)), in-1)
            const int ly = nearest_src(ry, (float)H / (float)rh, H), lx = nearest_src(rx, (float)W / (float)rw, W);
            l = lb[(long)ly * W + lx];
            if (rh == H && rw == W) {     // interpolate() with an unchanged size is the identity in torch too
                const unsigned char* q = ib + ((long)ry * W + rx) * 3;
                v0 = __fdiv_rn(__fsub_rn((float)q[0], m0), s0);
                v1 = __fdiv_rn(__fsub_rn((float)q[1], m1), s1);
                v2 = __fdiv_rn(__fsub_rn((float)q[2], m2), s2);
            } else {
                // image: bilinear, align_corners=False: src = scale*(dst+0.5)-0.5 clamped at 0 (ATen area_pixel_compute_source_index)
                const float sy = (float)H / (float)rh, sx = (float)W / (float)rw;
                // fused multiply-add like the AVX2/AVX-512 ATen kernel (one rounding): the fractional part of a source
                // index near 150 otherwise differs by ~1e-5, i.e. ~5e-5 on the interpolated pixel
                float fy = __fmaf_rn(sy, (float)ry + 0.5f, -0.5f), fx = __fmaf_rn(sx, (float)rx + 0.5f, -0.5f);
                fy = fy < 0.f ? 0.f : fy;
                fx = fx < 0.f ? 0.f : fx;
                const int y0 = (int)fy, x0 = (int)fx;
                const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
                const float ly1 = __fsub_rn(fy, (float)y0), lx1 = __fsub_rn(fx, (float)x0);
                const float ly0 = __fsub_rn(1.f, ly1), lx0 = __fsub_rn(1.f, lx1);
                const float mm[3] = {m0, m1, m2}, ss[3] = {s0, s1, s2};
                float r[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float a00 = __fdiv_rn(__fsub_rn((float)ib[((long)y0 * W + x0) * 3 + c], mm[c]), ss[c]);
                    const float a01 = __fdiv_rn(__fsub_rn((float)ib[((long)y0 * W + x1) * 3 + c], mm[c]), ss[c]);
                    const float a10 = __fdiv_rn(__fsub_rn((float)ib[((long)y1 * W + x0) * 3 + c], mm[c]), ss[c]);
                    const float a11 = __fdiv_rn(__fsub_rn((float)ib[((long)y1 * W + x1) * 3 + c], mm[c]), ss[c]);
                    const float top = __fadd_rn(__fmul_rn(lx0, a00), __fmul_rn(lx1, a01));
                    const float bot = __fadd_rn(__fmul_rn(lx0, a10), __fmul_rn(lx1, a11));
                    r[c] = __fadd_rn(__fmul_rn(ly0, top), __fmul_rn(ly1, bot));
                }
                v0 = r[0]; v1 = r[1]; v2 = r[2];
            }
        }
        const long plane = (long)Sh * Sw, o = (long)b * 3 * plane + (long)y * Sw + x;
        out_img[o] = v0;
        out_img[o + plane] = v1;
        out_img[o + 2 * plane] = v2;
        out_lab[i] = l;
    }
}
U2PL_API int u2pl_augment_u8_f32(const unsigned char* img, const unsigned char* lab, const int* params, int B, int H,
                                 int W, int Sh, int Sw, const float* mean3, const float* std3, float* out_img,
                                 long long* out_lab, hipStream_t stream) {
    const long total = (long)B * Sh * Sw;
    if (total <= 0) return 0;
    // mean / std are HOST pointers (three floats each): they travel as kernel arguments
    U2PL_LAUNCH(k_augment, dim3(grid_for(total, 256)), dim3(256), 0, stream, img, lab, params, B, H, W, Sh, Sw,
                       mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], out_img, out_lab);
    U2PL_LAUNCH_CHECK();
    return 0;
}
