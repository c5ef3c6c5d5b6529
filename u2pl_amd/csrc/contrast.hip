// Contrastive path: per-pixel class-rank membership,
// ordered compaction of anchor / prototype / negative-key pixel lists, class
// prototypes, device-resident per-class memory bank (FIFO ring), and the
// pixel-wise InfoNCE loss + gradient with wave64 shuffle reductions.
// Reference: u2pl/utils/loss_helper.py:51-235, u2pl/utils/utils.py:27-47.
#include "common.h"
#include "u2pl_hip.h"

// ---------------------------------------------------------------------------
// Phase 1a: per low-res pixel of the concatenated batch compute three class
// bitmasks (loss_helper.py:103-141):
//   abits  : anchor candidate      (prob_i > thr_p) & label_i & low_mask
//   lbits_o: low-valid membership  label_i & low_mask            (prototype)
//   nbits  : negative key          (prob_i < thr_n) & label_i & high_mask & class_mask_i
// class_mask (unlabeled): rank_i in [low_rank, high_rank); (labeled): rank_i <
// low_rank and label_i == 0.  rank = position in the descending sort; ties are
// broken towards the lower class index (documented deterministic rule).
// prob is addressed through strides so NCHW or NHWC both work.
// ---------------------------------------------------------------------------
#define MAXC 32
__global__ void k_contra_classify(const float* __restrict__ prob, long sn, long sc, long sp,
                                  const unsigned* __restrict__ lbits, const float* __restrict__ low_mask,
                                  const float* __restrict__ high_mask, int N2, int num_labeled, int C,
                                  long hw, float thr_p, float thr_n, int low_rank, int high_rank,
                                  unsigned* __restrict__ abits, unsigned* __restrict__ lowbits,
                                  unsigned* __restrict__ nbits) {
    long total = (long)N2 * hw;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < total;
         p += (long)gridDim.x * blockDim.x) {
        long n = p / hw, q = p % hw;
        const unsigned lb = lbits[p];
        const bool lo = low_mask[p] != 0.f, hi = high_mask[p] != 0.f;
        unsigned a = 0, l = 0, ng = 0;
        if (lb != 0) {   // every output needs label_i == 1 (labeled negatives are structurally empty, Q2)
            const float* b = prob + n * sn + q * sp;
            float pr[MAXC];
#pragma unroll
            for (int j = 0; j < MAXC; ++j) pr[j] = j < C ? b[j * sc] : -1.f;
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                const bool has = (i < C) && ((lb >> i) & 1u);
                if (!has) continue;
                const float pi = pr[i];
                int rank = 0;
#pragma unroll
                for (int j = 0; j < MAXC; ++j) rank += (pr[j] > pi) || (pr[j] == pi && j < i);
                bool cmask = n < num_labeled ? (rank < low_rank && !has) : (rank >= low_rank && rank < high_rank);
                if (has && lo) {
                    l |= 1u << i;
                    if (pi > thr_p) a |= 1u << i;
                }
                if (has && hi && pi < thr_n && cmask) ng |= 1u << i;
            }
        }
        abits[p] = a;
        lowbits[p] = l;
        nbits[p] = ng;
    }
}

U2PL_API int u2pl_contra_classify(const float* prob, long sn, long sc, long sp, const unsigned* lbits,
                                  const float* low_mask, const float* high_mask, int N2, int num_labeled,
                                  int C, int h, int w, float thr_p, float thr_n, int low_rank,
                                  int high_rank, unsigned* abits, unsigned* lowbits, unsigned* nbits,
                                  hipStream_t stream) {
    if (C > MAXC) return U2PL_EINVAL;
    long total = (long)N2 * h * w;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_contra_classify, dim3(grid_for(total, 256)), dim3(256), 0, stream, prob, sn, sc, sp,
                       lbits, low_mask, high_mask, N2, num_labeled, C, (long)h * w, thr_p, thr_n, low_rank,
                       high_rank, abits, lowbits, nbits);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// Phase 1b: ordered compaction.  Lists are in row-major (n,y,x) pixel order,
// exactly the order of torch boolean-mask indexing (loss_helper.py:115-116,142).
// kinds: 0 = anchor, 1 = low-valid, 2 = negative.  Integer-only => exact.
//   pass A: per block (1024 pixels) counts   [nblk][3][C]
//   pass B: exclusive scan over blocks       (one thread per (kind,class))
//   pass C: re-evaluate ballots and write    idx[kind][class][cap]
// ---------------------------------------------------------------------------
#define CP_PIX 1024
__device__ __forceinline__ unsigned long long lanemask_lt() {
    unsigned lane = threadIdx.x & 63;
    return lane ? (~0ull >> (64 - lane)) : 0ull;
}

__global__ void k_compact_count(const unsigned* __restrict__ b0, const unsigned* __restrict__ b1,
                                const unsigned* __restrict__ b2, long P, int C, unsigned* __restrict__ blk) {
    __shared__ unsigned cnt[3 * MAXC];
    for (int i = threadIdx.x; i < 3 * MAXC; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    const long base = (long)blockIdx.x * CP_PIX;
    for (int it = 0; it < CP_PIX / 256; ++it) {
        long p = base + it * 256 + threadIdx.x;
        unsigned v0 = p < P ? b0[p] : 0, v1 = p < P ? b1[p] : 0, v2 = p < P ? b2[p] : 0;
        for (int c = 0; c < C; ++c) {
            unsigned long long m0 = __ballot((v0 >> c) & 1u), m1 = __ballot((v1 >> c) & 1u), m2 = __ballot((v2 >> c) & 1u);
            if ((threadIdx.x & 63) == 0) {
                if (m0) atomicAdd(&cnt[0 * MAXC + c], (unsigned)__popcll(m0));
                if (m1) atomicAdd(&cnt[1 * MAXC + c], (unsigned)__popcll(m1));
                if (m2) atomicAdd(&cnt[2 * MAXC + c], (unsigned)__popcll(m2));
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * MAXC; i += blockDim.x) blk[(long)blockIdx.x * 3 * MAXC + i] = cnt[i];
}

// exclusive scan over blocks: one wave64 per (kind, class) pair, shuffle scan in chunks of 64 blocks
__global__ void k_compact_scan(unsigned* __restrict__ blk, int nblk, unsigned* __restrict__ counts) {
    const int i = blockIdx.x, lane = threadIdx.x;   // grid = 3*MAXC, block = 64
    unsigned carry = 0;
    for (int base = 0; base < nblk; base += 64) {
        const int b = base + lane;
        const unsigned v = b < nblk ? blk[(long)b * 3 * MAXC + i] : 0;
        unsigned x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            unsigned u = __shfl_up(x, o, 64);
            if (lane >= o) x += u;
        }
        if (b < nblk) blk[(long)b * 3 * MAXC + i] = carry + x - v;
        carry += __shfl(x, 63, 64);
    }
    if (lane == 0) counts[i] = carry;
}

__global__ void k_compact_write(const unsigned* __restrict__ b0, const unsigned* __restrict__ b1,
                                const unsigned* __restrict__ b2, long P, int C,
                                const unsigned* __restrict__ blk, int* __restrict__ idx, long cap) {
    // per (wave-iteration, kind, class) counts -> in-block exclusive offsets
    __shared__ unsigned wcnt[16][3 * MAXC];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long base = (long)blockIdx.x * CP_PIX;
    unsigned v[4][3];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        long p = base + it * 256 + threadIdx.x;
        v[it][0] = p < P ? b0[p] : 0;
        v[it][1] = p < P ? b1[p] : 0;
        v[it][2] = p < P ? b2[p] : 0;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            for (int c = 0; c < C; ++c) {
                unsigned long long m = __ballot((v[it][k] >> c) & 1u);
                if (lane == 0) wcnt[it * 4 + wave][k * MAXC + c] = (unsigned)__popcll(m);
            }
    }
    __syncthreads();
    // exclusive scan over the 16 wave-iterations (pixel order: it-major, wave-minor)
    for (int i = threadIdx.x; i < 3 * MAXC; i += blockDim.x) {
        unsigned run = blk[(long)blockIdx.x * 3 * MAXC + i];
        for (int s = 0; s < 16; ++s) {
            unsigned t = wcnt[s][i];
            wcnt[s][i] = run;
            run += t;
        }
    }
    __syncthreads();
    const unsigned long long lt = lanemask_lt();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        long p = base + it * 256 + threadIdx.x;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            for (int c = 0; c < C; ++c) {
                bool on = (v[it][k] >> c) & 1u;
                unsigned long long m = __ballot(on);
                if (on) {
                    unsigned pos = wcnt[it * 4 + wave][k * MAXC + c] + (unsigned)__popcll(m & lt);
                    idx[((long)k * MAXC + c) * cap + pos] = (int)p;
                }
            }
    }
}

U2PL_API size_t u2pl_compact_workspace_bytes(long P) {
    return (size_t)cdiv(P, CP_PIX) * 3 * MAXC * sizeof(unsigned);
}

// idx: int32 [3][32][cap] ; counts: u32 [3][32]
U2PL_API int u2pl_compact_lists(const unsigned* abits, const unsigned* lowbits, const unsigned* nbits, long P,
                                int C, void* workspace, int* idx, long cap, unsigned* counts,
                                hipStream_t stream) {
    if (C > MAXC || P <= 0) return U2PL_EINVAL;
    int nblk = cdiv(P, CP_PIX);
    unsigned* blk = (unsigned*)workspace;
    hipLaunchKernelGGL(k_compact_count, dim3(nblk), dim3(256), 0, stream, abits, lowbits, nbits, P, C, blk);
    U2PL_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_compact_scan, dim3(3 * MAXC), dim3(64), 0, stream, blk, nblk, counts);
    U2PL_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_compact_write, dim3(nblk), dim3(256), 0, stream, abits, lowbits, nbits, P, C, blk, idx, cap);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// Phase 1c: class prototypes = mean of rep_teacher rows over the low-valid list
// (loss_helper.py:119-123).  rows: row r of the (pixel, D) view = base + r*ld.
// Deterministic two-stage sum: chunks of PR_ROWS rows -> partial[C][nchunk][D]
// (float), then an ordered double-precision finish.
// ---------------------------------------------------------------------------
// Streaming formulation: every pixel row of rep_teacher that belongs to at least one class is read ONCE
// (1 KiB coalesced by D/4 lanes) and added into a per-block [C][D] LDS accumulator; thread d owns column
// d of every class, visits the block's pixels in order => deterministic.  Only images {0, B} carry class
// bits (Q0), so ~2*h*w rows are streamed instead of sum_c |low_valid_c| gathered rows.
#define PR_PIX 256
// 4 waves per block; wave w streams pixels w, w+4, ... of the block's range (8 row loads in flight per
// wave), lane l owns channels [4l, 4l+4) (D == 256) or strides over D; per-class accumulators live in
// registers (class loop fully unrolled and predicated), combined across the 4 waves through LDS in a
// fixed order => deterministic.
template <int CT>
__global__ __launch_bounds__(256, 2) void k_proto_stream(const float* __restrict__ rows, long ld, int D, const unsigned* __restrict__ lowbits,
                               long P, float* __restrict__ partial) {
    extern __shared__ float red[];   // [4][CT][D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long p0 = (long)blockIdx.x * PR_PIX;
    float* out = partial + (long)blockIdx.x * CT * D;
    for (int d0 = 0; d0 < D; d0 += 256) {       // D <= 256: one trip
        const int d = d0 + lane * 4;
        float4 acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool act = d < D;       // lanes beyond D only take part in the ballots / broadcasts
        {
            // this wave's 64 pixels are wave + 4*i; lane i holds the class bits of pixel i of that set
            const long pmine = p0 + wave + 4 * lane;
            const unsigned mybits = pmine < P ? lowbits[pmine] : 0u;
            unsigned long long todo = __ballot(mybits != 0);
            while (todo) {          // wave-uniform loop: up to 8 contributing pixels per trip
                int sel[8];
                int nsel = 0;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    sel[u] = 0;
                    if (todo) {
                        sel[u] = __ffsll((long long)todo) - 1;
                        todo &= todo - 1;
                        nsel = u + 1;
                    }
                }
                float4 v[8];
                unsigned bits[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {   // unconditional loads (sel[u] = 0 is a valid pixel of the set)
                    v[u] = act ? *(const float4*)(rows + (p0 + wave + 4 * (long)sel[u]) * ld + d) : make_float4(0.f, 0.f, 0.f, 0.f);
                    bits[u] = u < nsel ? __shfl(mybits, sel[u], 64) : 0u;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int c = 0; c < CT; ++c)
                        if ((bits[u] >> c) & 1u) {
                            acc[c].x += v[u].x; acc[c].y += v[u].y; acc[c].z += v[u].z; acc[c].w += v[u].w;
                        }
            }
            if (act) {
#pragma unroll
                for (int c = 0; c < CT; ++c) *(float4*)(red + ((long)wave * CT + c) * D + d) = acc[c];
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < CT * D / 4; i += blockDim.x) {
            const float4 a = ((float4*)red)[i], b2 = ((float4*)red)[CT * D / 4 + i];
            const float4 c2 = ((float4*)red)[2 * CT * D / 4 + i], e = ((float4*)red)[3 * CT * D / 4 + i];
            float4 r;
            r.x = (a.x + b2.x) + (c2.x + e.x); r.y = (a.y + b2.y) + (c2.y + e.y);
            r.z = (a.z + b2.z) + (c2.z + e.z); r.w = (a.w + b2.w) + (c2.w + e.w);
            ((float4*)out)[i] = r;
        }
        __syncthreads();
    }
}
// grid (C, D/64): 64 channels x 4 partial groups per block, 8 loads in flight per thread
__global__ void k_proto_finish(const float* __restrict__ partial, int D, const unsigned* __restrict__ counts,
                               int nblk, int C, float* __restrict__ proto) {
    __shared__ double sh[4][64];
    const int c = blockIdx.x, cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int d = blockIdx.y * 64 + cl;
    const unsigned n = counts[1 * MAXC + c];
    double acc = 0.0;
    if (d < D) {
        const float* p = partial + (long)c * D + d;
        int b = rg;
        for (; b + 7 * 4 < nblk; b += 8 * 4) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(long)(b + 4 * u) * C * D];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += (double)v[u];
        }
        for (; b < nblk; b += 4) acc += (double)p[(long)b * C * D];
    }
    sh[rg][cl] = acc;
    __syncthreads();
    if (rg == 0 && d < D) {
        const double t = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
        proto[(long)c * D + d] = n ? (float)(t / (double)n) : __uint_as_float(0x7fc00000u);
    }
}

U2PL_API size_t u2pl_proto_workspace_bytes(long P, int C, int D) {
    return (size_t)cdiv(P, PR_PIX) * C * D * sizeof(float);
}
// idx/cap are unused by the streaming formulation (kept in the ABI for list-based callers)
U2PL_API int u2pl_class_prototypes(const float* rows, long ld, int D, const int* idx, long cap,
                                   const unsigned* counts, int C, long P, void* workspace, float* proto,
                                   const unsigned* lowbits, hipStream_t stream) {
    (void)idx; (void)cap;
    const int nblk = cdiv(P, PR_PIX);
    if (D % 4 || D > 256) return U2PL_EINVAL;
    const size_t lds = (size_t)4 * C * D * sizeof(float);
#define PROTO_CASE(CT)                                                                                           \
    case CT: {                                                                                                   \
        static bool set_##CT = false;                                                                            \
        if (!set_##CT) { (void)hipFuncSetAttribute((const void*)k_proto_stream<CT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * CT * 256 * 4)); set_##CT = true; } \
        hipLaunchKernelGGL(k_proto_stream<CT>, dim3(nblk), dim3(256), lds, stream, rows, ld, D, lowbits, P, (float*)workspace); \
    } break;
    switch (C) {
        PROTO_CASE(19) PROTO_CASE(21) PROTO_CASE(32)
        default: return U2PL_EINVAL;
    }
#undef PROTO_CASE
    U2PL_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_proto_finish, dim3(C, cdiv(D, 64)), dim3(256), 0, stream, (const float*)workspace, D, counts, nblk, C, proto);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// Row gather (keys = rep_teacher[negative_mask], loss_helper.py:142) and the
// memory-bank FIFO (utils.py:27-47) as a device ring:  logical row j of the
// queue lives at physical slot (head + j) % cap.
// ---------------------------------------------------------------------------
__global__ void k_gather_rows(const float* __restrict__ rows, long ld, int D, const int* __restrict__ list,
                              long n, float* __restrict__ out) {
    const int D4 = D >> 2;
    long total = n * D4;
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        long r = t / D4;
        int d = (int)(t % D4);
        long src = list ? (long)list[r] : r;
        ((float4*)out)[r * D4 + d] = *(const float4*)(rows + src * ld + 4 * d);
    }
}
U2PL_API int u2pl_gather_rows_f32(const float* rows, long ld, int D, const int* list, long n, float* out,
                                  hipStream_t stream) {
    if (D % 4) return U2PL_EINVAL;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_gather_rows, dim3(grid_for(n * (D / 4), 256)), dim3(256), 0, stream, rows, ld, D, list, n, out);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// append n_new rows (already rank-major concatenated); only the last `cap`
// matter (utils.py:38-41).  tail = (head + len) % cap is passed by the host,
// which owns head/len (they are needed on the host for the RNG bound anyway).
__global__ void k_bank_append(float* __restrict__ bank, long cap, long tail, int D,
                              const float* __restrict__ rows, long ld, const int* __restrict__ list,
                              long n_new, long skip) {
    const int D4 = D >> 2;
    long total = (n_new - skip) * D4;
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        long j = skip + t / D4;
        int d = (int)(t % D4);
        long src = list ? (long)list[j] : j;
        long slot = (tail + j) % cap;
        ((float4*)bank)[slot * D4 + d] = *(const float4*)(rows + src * ld + 4 * d);
    }
}
// all classes in one launch: desc = int64 [C][6] = {bank ptr, cap, tail, rows ptr, list ptr (or 0), n_new}
__global__ void k_bank_append_multi(const long long* __restrict__ desc, int D, long ld) {
    const long long* d = desc + 6 * blockIdx.y;
    float* bank = (float*)d[0];
    const long cap = d[1], tail = d[2];
    const float* rows = (const float*)d[3];
    const int* list = (const int*)d[4];
    const long n_new = d[5];
    const long skip = n_new > cap ? n_new - cap : 0;
    const int D4 = D >> 2;
    const long total = (n_new - skip) * D4;
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long j = skip + t / D4;
        const int dd = (int)(t % D4);
        const long src = list ? (long)list[j] : j;
        const long slot = (tail + j) % cap;
        ((float4*)bank)[slot * D4 + dd] = *(const float4*)(rows + src * ld + 4 * dd);
    }
}
U2PL_API int u2pl_bank_append_multi_f32(const long long* desc_dev, int nclass, int D, long ld, long max_new,
                                        hipStream_t stream) {
    if (D % 4) return U2PL_EINVAL;
    if (nclass <= 0 || max_new <= 0) return 0;
    dim3 grid(grid_for(max_new * (D / 4), 256, 64), nclass);
    hipLaunchKernelGGL(k_bank_append_multi, grid, dim3(256), 0, stream, desc_dev, D, ld);
    U2PL_LAUNCH_CHECK();
    return 0;
}

U2PL_API int u2pl_bank_append_f32(float* bank, long cap, long tail, int D, const float* rows, long ld,
                                  const int* list, long n_new, hipStream_t stream) {
    if (D % 4 || cap <= 0) return U2PL_EINVAL;
    if (n_new <= 0) return 0;
    long skip = n_new > cap ? n_new - cap : 0;
    hipLaunchKernelGGL(k_bank_append, dim3(grid_for((n_new - skip) * (D / 4), 256)), dim3(256), 0, stream, bank,
                       cap, tail, D, rows, ld, list, n_new, skip);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// Phase 2: InfoNCE (loss_helper.py:173-230).  One wave64 per anchor: the lane
// holds D/64 contiguous floats of every 1 KiB feature row (coalesced float4
// when D == 256), dot products and norms are wave shuffle reductions.
//   logits_j = cos(a, f_j) / temp, f_0 = prototype, f_1.. = sampled negatives
//   loss_q   = logsumexp(logits) - logits_0
//   ganchor  = d loss_q / d a  (un-scaled; scaled by 1/(Q*valid_seg)*gout in the scatter)
// cosine_similarity = (a / max(|a|,eps)) . (f / max(|f|,eps)), eps = 1e-8.
// ---------------------------------------------------------------------------
struct NceJob {
    const int* cand;          // anchor candidate pixel list of class slot i
    const long long* idx_a;   // [Q]   torch.randint(len(cand))
    const long long* idx_n;   // [Q*K] torch.randint(len(bank))
    const float* proto;       // [D]
    const float* bank;        // ring base of bank[valid_classes[i]]
    long bank_cap, bank_head;
};

template <int VPL>  // floats per lane = D / 64
__global__ void k_infonce(const NceJob* __restrict__ jobs, const float* __restrict__ rep, long ld, int D,
                          int Q, int K, float inv_temp, float* __restrict__ loss_q,
                          float* __restrict__ ganchor, int* __restrict__ anchor_pix) {
    const int job = blockIdx.y;
    const int q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= Q) return;
    const NceJob J = jobs[job];
    const int pix = J.cand[J.idx_a[q]];
    float a[VPL], ah[VPL], acc[VPL];
    const float* ar = rep + (long)pix * ld + lane * VPL;
    float na = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) { a[i] = ar[i]; na += a[i] * a[i]; acc[i] = 0.f; }
    na = fmaxf(sqrtf(wave_sum(na)), 1e-8f);
#pragma unroll
    for (int i = 0; i < VPL; ++i) ah[i] = a[i] / na;
    // online softmax over the 1+K logits, accumulating sum_j softmax_j * fhat_j.  Features are processed
    // four at a time: four independent 1 KiB row loads in flight per wave and four interleaved shuffle
    // reductions (the kernel is a latency-bound gather otherwise).
    float m = -INFINITY, s = 0.f, l0 = 0.f, cw = 0.f;  // cw = sum_j w_j * cos_j (un-normalised)
    float f0h[VPL];
    // the K sampled bank rows of this anchor: one coalesced index load, then lane broadcasts (no dependent
    // index -> row latency chain inside the loop)
    const bool pre = K <= 64;
    long myrow = 0;
    if (pre && lane < K) myrow = (J.bank_head + J.idx_n[(long)q * K + lane]) % J.bank_cap;
    auto row_ptr = [&](int j) -> const float* {
        if (j == 0) return J.proto + lane * VPL;
        const long r = pre ? __shfl(myrow, j - 1, 64) : (J.bank_head + J.idx_n[(long)q * K + (j - 1)]) % J.bank_cap;
        return J.bank + r * D + lane * VPL;
    };
    for (int j0 = 0; j0 <= K; j0 += 4) {
        float f[4][VPL], nf[4], dot[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* fr = row_ptr(min(j0 + u, K));
#pragma unroll
            for (int i = 0; i < VPL; ++i) f[u][i] = fr[i];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            nf[u] = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i) nf[u] += f[u][i] * f[u][i];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int u = 0; u < 4; ++u) nf[u] += __shfl_xor(nf[u], o, 64);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            nf[u] = fmaxf(sqrtf(nf[u]), 1e-8f);
            dot[u] = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i) { f[u][i] = f[u][i] / nf[u]; dot[u] += ah[i] * f[u][i]; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int u = 0; u < 4; ++u) dot[u] += __shfl_xor(dot[u], o, 64);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (j0 + u > K) continue;
            const float cosv = dot[u];
            const float l = cosv * inv_temp;
            if (j0 + u == 0) {
                l0 = l;
#pragma unroll
                for (int i = 0; i < VPL; ++i) f0h[i] = f[u][i];
            }
            const float mn = fmaxf(m, l);
            const float sc = expf(m - mn), w = expf(l - mn);
            s = s * sc + w;
            cw = cw * sc + w * cosv;
#pragma unroll
            for (int i = 0; i < VPL; ++i) acc[i] = acc[i] * sc + w * f[u][i];
            m = mn;
        }
    }
    const float lse = m + logf(s);
    if (lane == 0) {
        loss_q[(long)job * Q + q] = lse - l0;
        anchor_pix[(long)job * Q + q] = pix;
    }
    // d loss/d cos_j = (softmax_j - [j==0]) * inv_temp ; d cos_j/d a = (fhat_j - cos_j*ahat)/|a|
    const float inv_s = 1.0f / s;
    const float cosbar = cw * inv_s - l0 / inv_temp;  // sum_j (softmax_j-[j==0]) cos_j
    float* g = ganchor + ((long)job * Q + q) * D + lane * VPL;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
        g[i] = inv_temp * ((acc[i] * inv_s - f0h[i]) - cosbar * ah[i]) / na;
}

U2PL_API int u2pl_infonce_f32(const void* jobs_dev, int njobs, const float* rep, long ld, int D, int Q, int K,
                              float temp, float* loss_q, float* ganchor, int* anchor_pix,
                              hipStream_t stream) {
    if (njobs <= 0) return 0;
    dim3 grid(cdiv(Q, 4), njobs), block(256);
    const NceJob* jobs = (const NceJob*)jobs_dev;
    float it = 1.0f / temp;
    switch (D) {
        case 64: hipLaunchKernelGGL(k_infonce<1>, grid, block, 0, stream, jobs, rep, ld, D, Q, K, it, loss_q, ganchor, anchor_pix); break;
        case 128: hipLaunchKernelGGL(k_infonce<2>, grid, block, 0, stream, jobs, rep, ld, D, Q, K, it, loss_q, ganchor, anchor_pix); break;
        case 256: hipLaunchKernelGGL(k_infonce<4>, grid, block, 0, stream, jobs, rep, ld, D, Q, K, it, loss_q, ganchor, anchor_pix); break;
        case 512: hipLaunchKernelGGL(k_infonce<8>, grid, block, 0, stream, jobs, rep, ld, D, Q, K, it, loss_q, ganchor, anchor_pix); break;
        default: return U2PL_EINVAL;
    }
    U2PL_LAUNCH_CHECK();
    return 0;
}
U2PL_API size_t u2pl_infonce_job_bytes(void) { return sizeof(NceJob); }

// loss = (sum_jobs mean_q loss_q) / valid_seg = sum_all loss_q / (Q * valid_seg)   (loss_helper.py:228-233)
// one 1024-thread block, fixed summation order (deterministic), double accumulation
__global__ __launch_bounds__(1024) void k_infonce_reduce(const float* __restrict__ loss_q, int njobs, int Q,
                                                         float inv_valid_seg, float* __restrict__ loss) {
    __shared__ double sh[1024];
    const long n = (long)njobs * Q;
    double acc = 0.0;
    for (long i = threadIdx.x; i < n; i += 1024) acc += (double)loss_q[i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = (float)(sh[0] / (double)Q * (double)inv_valid_seg);
}
U2PL_API int u2pl_infonce_reduce_f32(const float* loss_q, int njobs, int Q, float inv_valid_seg, float* loss,
                                     hipStream_t stream) {
    hipLaunchKernelGGL(k_infonce_reduce, dim3(1), dim3(1024), 0, stream, loss_q, njobs, Q, inv_valid_seg, loss);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// grad_rep[pix] += scale * ganchor  (anchors are sampled with replacement =>
// duplicates: float atomics; everything else in grad_rep stays zero)
__global__ void k_scatter_add_rows(float* __restrict__ dst, long ld, int D, const int* __restrict__ pix,
                                   const float* __restrict__ src, long n, const float* __restrict__ gout,
                                   float scale) {
    const float sc = scale * (gout ? *gout : 1.0f);
    long total = n * D;
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        long r = t / D;
        int d = (int)(t % D);
        atomicAdd(&dst[(long)pix[r] * ld + d], sc * src[t]);
    }
}
U2PL_API int u2pl_scatter_add_rows_f32(float* dst, long ld, int D, const int* pix, const float* src, long n,
                                       const float* gout_dev, float scale, hipStream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_scatter_add_rows, dim3(grid_for(n * D, 256)), dim3(256), 0, stream, dst, ld, D, pix, src, n, gout_dev, scale);
    U2PL_LAUNCH_CHECK();
    return 0;
}
