// Contrastive path: per-pixel class-rank membership,
// ordered compaction of anchor / prototype / negative-key pixel lists, class
// prototypes, device-resident per-class memory bank (FIFO ring), and the
// pixel-wise InfoNCE loss + gradient with wave64 shuffle reductions.
// Reference: u2pl/utils/loss_helper.py:51-235, u2pl/utils/utils.py:27-47.
#include <stdlib.h>
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
// (debug, -DU2PL_P1_DBG builds only: `python -m u2pl_amd.build_ext --variant p1dbg -DU2PL_P1_DBG`) per-block start / end
// times and three marks of the phase-1 kernels: u2pl_debug_phase1_times(buf) arms them, NULL disarms.
// buf: uint32 [3][4096][2] (kernel 0 = classify, 1 = prototype stream, 2 = tail), 100 MHz ticks.
#ifdef U2PL_P1_DBG
__device__ unsigned* g_p1_dbg = nullptr;
U2PL_API int u2pl_debug_phase1_times(unsigned* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_p1_dbg), &buf, sizeof(buf));
}
#define P1_DBG_START(k) unsigned dbg_t0_ = 0; if (g_p1_dbg && threadIdx.x == 0) dbg_t0_ = (unsigned)wall_clock64();
#define P1_DBG_MARK(k, slot, b) if (g_p1_dbg && threadIdx.x == 0) g_p1_dbg[((k) * 4096 + 1024 * (slot) + (b)) * 2] = (unsigned)wall_clock64();
#define P1_DBG_END(k, b) if (g_p1_dbg && threadIdx.x == 0 && (b) < 4096) { g_p1_dbg[((k) * 4096 + (b)) * 2] = dbg_t0_; g_p1_dbg[((k) * 4096 + (b)) * 2 + 1] = (unsigned)wall_clock64(); }
#else
U2PL_API int u2pl_debug_phase1_times(unsigned*) { return U2PL_EINVAL; }     // not an instrumented build
#define P1_DBG_START(k)
#define P1_DBG_MARK(k, slot, b)
#define P1_DBG_END(k, b)
#endif
#define CP_PIX 256          // pixels per block of the classify / compaction kernels (one per thread)
#define PF_MAXBLK 4096      // prototype partial blocks a finish can order
// The classify kernel also counts, per block of CP_PIX pixels, the members of every (kind, class) list:
// blk[(kind*32 + c) * nblk + b]   (kinds: 0 = anchor, 1 = low-valid, 2 = negative)
__global__ __launch_bounds__(CP_PIX) void k_contra_classify(
    const float* __restrict__ prob, long sn, long sc, long sp, const unsigned* __restrict__ lbits,
    const float* __restrict__ low_mask, const float* __restrict__ high_mask, int N2, int num_labeled, int C,
    long hw, float thr_p, float thr_n, int low_rank, int high_rank, unsigned* __restrict__ abits,
    unsigned* __restrict__ lowbits, unsigned* __restrict__ nbits, unsigned* __restrict__ blk, int nblk) {
    __shared__ unsigned cnt[3 * MAXC];
    if (threadIdx.x < 3 * MAXC) cnt[threadIdx.x] = 0;
    __syncthreads();
    const long total = (long)N2 * hw;
    const long p = blockIdx.x * (long)CP_PIX + threadIdx.x;
    if (p < total) {
        const long n = p / hw, q = p % hw;
        const unsigned lb = lbits[p];
        unsigned a = 0, l = 0, ng = 0;
        if (lb != 0) {   // every output needs label_i == 1 (labeled negatives are structurally empty, Q2)
            const bool lo = low_mask[p] != 0.f, hi = high_mask[p] != 0.f;
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
        if (blk) {   // sparse: one LDS atomic per set bit
            for (unsigned x = a; x; x &= x - 1) atomicAdd(&cnt[0 * MAXC + __ffs(x) - 1], 1u);
            for (unsigned x = l; x; x &= x - 1) atomicAdd(&cnt[1 * MAXC + __ffs(x) - 1], 1u);
            for (unsigned x = ng; x; x &= x - 1) atomicAdd(&cnt[2 * MAXC + __ffs(x) - 1], 1u);
        }
    }
    if (!blk) return;
    __syncthreads();
    if (threadIdx.x < 3 * MAXC) blk[(long)threadIdx.x * nblk + blockIdx.x] = cnt[threadIdx.x];
}

// Fast form for the layout the trainer produces (probabilities as contiguous [pixel][C] rows, C odd): a block's 256 rows
// are ONE contiguous span of 256 * C floats, staged into LDS with coalesced 16-byte loads (the strided form issues C
// dword loads per thread whose lanes sit 4C bytes apart: ~5x the memory instructions), and read back with stride C (odd:
// conflict-free).  Blocks whose pixels carry no label bits (images 1 .. B-1 and B+1 .. under quirk Q0) only write zeros.
// Same arithmetic and outputs as k_contra_classify.
__global__ __launch_bounds__(CP_PIX) void k_contra_classify_rows(
    const float* __restrict__ prob, const unsigned* __restrict__ lbits, const float* __restrict__ low_mask,
    const float* __restrict__ high_mask, int N2, int num_labeled, int C, long hw, float thr_p, float thr_n, int low_rank,
    int high_rank, unsigned* __restrict__ abits, unsigned* __restrict__ lowbits, unsigned* __restrict__ nbits,
    unsigned* __restrict__ blk, int nblk) {
    __shared__ __attribute__((aligned(16))) float rows[CP_PIX * MAXC];
    __shared__ unsigned cnt[3 * MAXC];
    __shared__ int any_s;
    P1_DBG_START(0)
    if (threadIdx.x < 3 * MAXC) cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) any_s = 0;
    __syncthreads();
    const long total = (long)N2 * hw;
    const long p0 = blockIdx.x * (long)CP_PIX, p = p0 + threadIdx.x;
    // label bits, masks and the block's probability rows are all requested at once (ONE memory round trip; the rows of a
    // block without label bits -- half of them under quirk Q0 -- are 19 KB of coalesced loads that are then ignored)
    const unsigned lb = p < total ? lbits[p] : 0u;
    const float lmv = p < total ? low_mask[p] : 0.f, hmv = p < total ? high_mask[p] : 0.f;
    {
        const long npx = min((long)CP_PIX, total - p0);
        const int nfl = (int)(npx * C);
        const float* src = prob + p0 * C;             // 16-byte aligned: p0 is a multiple of 256
        for (int i = threadIdx.x * 4; i < nfl; i += CP_PIX * 4) {
            if (i + 3 < nfl) *(float4*)(rows + i) = *(const float4*)(src + i);
            else for (int k = i; k < nfl; ++k) rows[k] = src[k];
        }
    }
    if (__ballot(lb != 0) && (threadIdx.x & 63) == 0) any_s = 1;
    __syncthreads();
    P1_DBG_MARK(0, 1, blockIdx.x)
    if (!any_s) {        // block-uniform
        if (p < total) { abits[p] = 0; lowbits[p] = 0; nbits[p] = 0; }
        if (blk && threadIdx.x < 3 * MAXC) blk[(long)threadIdx.x * nblk + blockIdx.x] = 0;
        P1_DBG_END(0, blockIdx.x)
        return;
    }
    const bool lo = lb != 0 && lmv != 0.f, hi = lb != 0 && hmv != 0.f;
    if (p < total) {
        const long n = p / hw;
        unsigned a = 0, l = 0, ng = 0;
        if (lb != 0) {
            // only the classes in the pixel's label bits (1-2 of C) need a rank: a loop over the SET bits with the row read
            // from LDS, instead of C unrolled rank computations that a wave executes whenever any of its lanes has the bit
            const float* b = rows + threadIdx.x * C;
            for (unsigned x = lb & ((C < 32 ? (1u << C) : 0u) - 1u); x; x &= x - 1u) {
                const int i = __ffs(x) - 1;
                const float pi = b[i];
                int rank = 0;
                for (int j = 0; j < C; ++j) {
                    const float pj = b[j];
                    rank += (pj > pi) || (pj == pi && j < i);
                }
                const bool cmask = n < num_labeled ? false : (rank >= low_rank && rank < high_rank);    // (labeled: rank < low_rank && !has == false)
                if (lo) {
                    l |= 1u << i;
                    if (pi > thr_p) a |= 1u << i;
                }
                if (hi && pi < thr_n && cmask) ng |= 1u << i;
            }
        }
        P1_DBG_MARK(0, 2, blockIdx.x)
        abits[p] = a;
        lowbits[p] = l;
        nbits[p] = ng;
        if (blk) {
            for (unsigned x = a; x; x &= x - 1) atomicAdd(&cnt[0 * MAXC + __ffs(x) - 1], 1u);
            for (unsigned x = l; x; x &= x - 1) atomicAdd(&cnt[1 * MAXC + __ffs(x) - 1], 1u);
            for (unsigned x = ng; x; x &= x - 1) atomicAdd(&cnt[2 * MAXC + __ffs(x) - 1], 1u);
        }
    }
    if (!blk) return;
    __syncthreads();
    P1_DBG_MARK(0, 3, blockIdx.x)
    if (threadIdx.x < 3 * MAXC) blk[(long)threadIdx.x * nblk + blockIdx.x] = cnt[threadIdx.x];
    P1_DBG_END(0, blockIdx.x)
}

U2PL_API int u2pl_contra_classify(const float* prob, long sn, long sc, long sp, const unsigned* lbits,
                                  const float* low_mask, const float* high_mask, int N2, int num_labeled,
                                  int C, int h, int w, float thr_p, float thr_n, int low_rank,
                                  int high_rank, unsigned* abits, unsigned* lowbits, unsigned* nbits,
                                  void* compact_workspace, hipStream_t stream) {
    if (C > MAXC) return U2PL_EINVAL;
    long total = (long)N2 * h * w;
    if (total <= 0) return 0;
    const int nblk = cdiv(total, CP_PIX);
    if (sc == 1 && sp == C && sn == (long)h * w * C && (C & 1) && ((uintptr_t)prob & 15) == 0)
        U2PL_LAUNCH(k_contra_classify_rows, dim3(nblk), dim3(CP_PIX), 0, stream, prob, lbits, low_mask, high_mask, N2,
                           num_labeled, C, (long)h * w, thr_p, thr_n, low_rank, high_rank, abits, lowbits, nbits,
                           (unsigned*)compact_workspace, nblk);
    else
    U2PL_LAUNCH(k_contra_classify, dim3(nblk), dim3(CP_PIX), 0, stream, prob, sn, sc, sp, lbits, low_mask,
                       high_mask, N2, num_labeled, C, (long)h * w, thr_p, thr_n, low_rank, high_rank, abits,
                       lowbits, nbits, (unsigned*)compact_workspace, nblk);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// Phase 1b: ordered compaction.  Lists are in row-major (n,y,x) pixel order,
// exactly the order of torch boolean-mask indexing (loss_helper.py:115-116,142).
// kinds: 0 = anchor, 1 = low-valid (counts only: the prototypes stream the
// bitmask, nobody reads that list), 2 = negative.  Integer-only => exact.
//   pass A: per block (256 pixels) counts [3*32][nblk]  (by the classify kernel, or k_compact_count)
//   pass B: exclusive scan over blocks, one block per (kind, class)
//   pass C: per wave ballots over the classes PRESENT in the wave, write idx[kind][class][cap]
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long lanemask_lt() {
    unsigned lane = threadIdx.x & 63;
    return lane ? (~0ull >> (64 - lane)) : 0ull;
}
__device__ __forceinline__ unsigned wave_or_uniform(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
    return __builtin_amdgcn_readfirstlane(v);
}

__global__ __launch_bounds__(CP_PIX) void k_compact_count(const unsigned* __restrict__ b0, const unsigned* __restrict__ b1,
                                const unsigned* __restrict__ b2, long P, unsigned* __restrict__ blk, int nblk) {
    __shared__ unsigned cnt[3 * MAXC];
    if (threadIdx.x < 3 * MAXC) cnt[threadIdx.x] = 0;
    __syncthreads();
    const long p = blockIdx.x * (long)CP_PIX + threadIdx.x;
    if (p < P) {
        for (unsigned x = b0[p]; x; x &= x - 1) atomicAdd(&cnt[0 * MAXC + __ffs(x) - 1], 1u);
        for (unsigned x = b1[p]; x; x &= x - 1) atomicAdd(&cnt[1 * MAXC + __ffs(x) - 1], 1u);
        for (unsigned x = b2[p]; x; x &= x - 1) atomicAdd(&cnt[2 * MAXC + __ffs(x) - 1], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 3 * MAXC) blk[(long)threadIdx.x * nblk + blockIdx.x] = cnt[threadIdx.x];
}

// exclusive scan over blocks: one 256-thread block per (kind, class) row of blk (contiguous)
__global__ __launch_bounds__(256) void k_compact_scan(unsigned* __restrict__ blk, int nblk, unsigned* __restrict__ counts) {
    __shared__ unsigned wsum[4];
    unsigned* row = blk + (long)blockIdx.x * nblk;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    unsigned carry = 0;
    for (int base = 0; base < nblk; base += 256) {
        const int b = base + t;
        const unsigned v = b < nblk ? row[b] : 0;
        unsigned x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            unsigned u = __shfl_up(x, o, 64);
            if (lane >= o) x += u;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        unsigned wb = 0;
        for (int w2 = 0; w2 < wave; ++w2) wb += wsum[w2];
        const unsigned tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (b < nblk) row[b] = carry + wb + x - v;
        carry += tot;
        __syncthreads();
    }
    if (t == 0) counts[blockIdx.x] = carry;
}

__global__ __launch_bounds__(CP_PIX) void k_compact_write(const unsigned* __restrict__ b0, const unsigned* __restrict__ b2, long P,
                                const unsigned* __restrict__ blk, int nblk, int* __restrict__ idx, long cap) {
    __shared__ unsigned wcnt[4][2 * MAXC];   // kinds {0, 2}: per-wave counts -> in-block exclusive offsets
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long p = blockIdx.x * (long)CP_PIX + threadIdx.x;
    unsigned v[2];
    v[0] = p < P ? b0[p] : 0;
    v[1] = p < P ? b2[p] : 0;
    if (threadIdx.x < 2 * MAXC) {
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) wcnt[w2][threadIdx.x] = 0;
    }
    __syncthreads();
    unsigned pres[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        pres[k] = wave_or_uniform(v[k]);
        for (unsigned x = pres[k]; x; x &= x - 1) {
            const int c = __ffs(x) - 1;
            const unsigned long long m = __ballot((v[k] >> c) & 1u);
            if (lane == 0) wcnt[wave][k * MAXC + c] = (unsigned)__popcll(m);
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * MAXC) {
        const int k = threadIdx.x >> 5, c = threadIdx.x & 31;
        unsigned run = blk[((long)(2 * k) * MAXC + c) * nblk + blockIdx.x];
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) {
            const unsigned t = wcnt[w2][threadIdx.x];
            wcnt[w2][threadIdx.x] = run;
            run += t;
        }
    }
    __syncthreads();
    const unsigned long long lt = lanemask_lt();
#pragma unroll
    for (int k = 0; k < 2; ++k)
        for (unsigned x = pres[k]; x; x &= x - 1) {
            const int c = __ffs(x) - 1;
            const bool on = (v[k] >> c) & 1u;
            const unsigned long long m = __ballot(on);
            if (on) {
                const unsigned pos = wcnt[wave][k * MAXC + c] + (unsigned)__popcll(m & lt);
                idx[((long)(2 * k) * MAXC + c) * cap + pos] = (int)p;
            }
        }
}

// ---------------------------------------------------------------------------
// Phase-1 tail in ONE launch (1024-thread blocks, two roles by block index):
//   blocks [0, nwb): ordered compaction write of 1024 pixels each.  The exclusive offset of a (kind, class) list in front
//     of this block is summed HERE from the classify kernel's raw per-256-pixel counts (one wave per list that is present
//     in the block, ~10 coalesced loads per lane), so the separate scan launch is gone; block i < 96 also publishes the
//     total of count row i (the list lengths the host reads back).
//   blocks [nwb, nwb + C * D/64): the ordered double-precision finish of the class prototypes (k_proto_finish's body; the
//     class's member count is summed from the same raw counts, the finish does not wait for the totals above).
// The two roles are independent; merging them removes two launch boundaries from the chain in front of the step's one
// host synchronisation.
// ---------------------------------------------------------------------------
#define P1_T 1024
__device__ __forceinline__ unsigned p1_row_sum(const unsigned* __restrict__ row, int n, int lane) {   // one wave: sum of row[0..n)
    // 16 independent loads in flight per lane (a plain `acc += row[i]` loop is issued one load per round trip: ten serial
    // L2 latencies for the 583 count rows of a 769^2 step)
    unsigned acc = 0;
    for (int i0 = 0; i0 < n; i0 += 16 * 64) {
        unsigned v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int i = i0 + u * 64 + lane; v[u] = i < n ? row[i] : 0u; }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
    }
    return wave_sum_u(acc);
}
__global__ __launch_bounds__(P1_T) void k_phase1_tail(const unsigned* __restrict__ b0, const unsigned* __restrict__ b2, long P,
                                                      const unsigned* __restrict__ blk, int nblk, int* __restrict__ idx, long cap,
                                                      unsigned* __restrict__ counts, int nwb,
                                                      const float* __restrict__ partial, int D, int npb, int C,
                                                      const unsigned* __restrict__ flags, float* __restrict__ proto) {
    __shared__ unsigned wcnt[16][2 * MAXC];      // write role: per-wave counts -> exclusive offsets
    __shared__ unsigned long long msk[16][2 * MAXC];   // write role: per-wave membership masks of the 2 * MAXC lists
    __shared__ unsigned base_s[2 * MAXC];        // write role: list offset in front of this block
    __shared__ unsigned pres_s[2];
    __shared__ double sh[16][64];                // finish role
    __shared__ int act[PF_MAXBLK];
    __shared__ int wtot[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    P1_DBG_START(2)
    if ((int)blockIdx.x < nwb) {
        // ------------------------------------------------------------ compaction write
        const long p = blockIdx.x * (long)P1_T + t;
        unsigned v[2];
        v[0] = p < P ? b0[p] : 0;
        v[1] = p < P ? b2[p] : 0;
        // membership masks: lane l of wave w owns msk[w][l] = the 64-bit mask of the wave's pixels that are on list
        // l = kind * MAXC + class.  They are built with one LDS atomic per SET bit of a pixel (1-2 per kind) instead of one
        // ballot per present class: a block of the dense labeled image has all C classes in every wave, and two loops of
        // 2 C ballot rounds (count, then write) were 5 of the block's 9 us -- the tail of the launch.
        msk[wave][lane] = 0ull;
        if (t < 2 * MAXC) base_s[t] = 0;
        if (t < 2) pres_s[t] = 0;
        __syncthreads();
        P1_DBG_MARK(2, 1, blockIdx.x)
#pragma unroll
        for (int k = 0; k < 2; ++k)
            for (unsigned x = v[k]; x; x &= x - 1) atomicOr(&msk[wave][k * MAXC + __ffs(x) - 1], 1ull << lane);
        __syncthreads();
        {
            const unsigned long long m = msk[wave][lane];
            wcnt[wave][lane] = (unsigned)__popcll(m);
            const unsigned long long pm = __ballot(m != 0ull);      // bit k * MAXC + c: list present in this wave
            if (lane == 0) {
                if ((unsigned)pm) atomicOr(&pres_s[0], (unsigned)pm);
                if ((unsigned)(pm >> 32)) atomicOr(&pres_s[1], (unsigned)(pm >> 32));
            }
        }
        __syncthreads();
        {   // offsets in front of this block: list r of the block's present lists is summed by wave (r mod 16).  A wave's
            // lists (<= 4 of the <= 2 * MAXC present ones) are summed TOGETHER: all their loads are in flight at once (one
            // memory round trip; one list after the other was three for the block with the most lists -- the launch's tail)
            const int first = blockIdx.x * (P1_T / CP_PIX);          // count rows are per CP_PIX pixels
            int r = 0, nm = 0, sm[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 2; ++k)
                for (unsigned x = pres_s[k]; x; x &= x - 1, ++r) {
                    if ((r & 15) != wave) continue;
                    const int sidx = k * MAXC + __ffs(x) - 1;
                    if (nm == 0) sm[0] = sidx; else if (nm == 1) sm[1] = sidx; else if (nm == 2) sm[2] = sidx; else sm[3] = sidx;
                    ++nm;
                }
            if (nm > 0) {
                const unsigned* rw[4];
                int nn[4];
                unsigned a[4] = {0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    rw[q] = blk + ((long)(2 * (sm[q] / MAXC)) * MAXC + sm[q] % MAXC) * nblk;
                    nn[q] = q < nm ? first : 0;
                }
                for (int i0 = 0; i0 < first; i0 += 8 * 64) {
                    unsigned v[4][8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = i0 + u * 64 + lane;
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q][u] = i < nn[q] ? rw[q][i] : 0u;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
#pragma unroll
                        for (int q = 0; q < 4; ++q) a[q] += v[q][u];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned sum = wave_sum_u(a[q]);
                    if (lane == 0 && q < nm) base_s[sm[q]] = sum;
                }
            }
        }
        // list lengths (all three kinds) for the host / the InfoNCE jobs: summed by the LAST write blocks (under quirk Q0 the
        // trailing images carry no lists, so these blocks have no offsets to sum; the first blocks are the dense ones)
        for (int r = (nwb - 1 - (int)blockIdx.x) * 16 + wave; r < 3 * MAXC; r += nwb * 16) {
            const unsigned sum = p1_row_sum(blk + (long)r * nblk, nblk, lane);
            if (lane == 0) counts[r] = sum;
        }
        __syncthreads();
        P1_DBG_MARK(2, 2, blockIdx.x)
        if (t < 2 * MAXC) {
            unsigned run = base_s[t];
#pragma unroll
            for (int w2 = 0; w2 < 16; ++w2) {
                const unsigned tmp = wcnt[w2][t];
                wcnt[w2][t] = run;
                run += tmp;
            }
        }
        __syncthreads();
        const unsigned long long lt = lanemask_lt();
#pragma unroll
        for (int k = 0; k < 2; ++k)
            for (unsigned x = v[k]; x; x &= x - 1) {       // this pixel's own lists: rank within the wave from the list's mask
                const int c = __ffs(x) - 1;
                const unsigned pos = wcnt[wave][k * MAXC + c] + (unsigned)__popcll(msk[wave][k * MAXC + c] & lt);
                idx[((long)(2 * k) * MAXC + c) * cap + pos] = (int)p;
            }
        P1_DBG_END(2, blockIdx.x)
        return;
    }
    // ---------------------------------------------------------------- prototype finish (ordered, double precision)
    const int fb = blockIdx.x - nwb;
    const int ndc = (D + 63) / 64;
    const int c = fb / ndc, cl = lane, rg = wave;
    const int d = (fb % ndc) * 64 + cl;
    const unsigned n = p1_row_sum(blk + ((long)1 * MAXC + c) * nblk, nblk, lane);     // members of class c (every wave sums it; its
                                                                                     // loads travel with the flag loads below)
    int basei = 0;
    for (int bb0 = 0; bb0 < npb; bb0 += P1_T) {
        const int bb = bb0 + t;
        const bool on = bb < npb && flags[bb] != 0;
        const unsigned long long m = __ballot(on);
        if (cl == 0) wtot[rg] = __popcll(m);
        __syncthreads();
        int off = basei;
        for (int w2 = 0; w2 < rg; ++w2) off += wtot[w2];
        if (on) act[off + __popcll(m & (cl ? (~0ull >> (64 - cl)) : 0ull))] = bb;
        int tot = 0;
        for (int w2 = 0; w2 < 16; ++w2) tot += wtot[w2];
        basei += tot;
        __syncthreads();
    }
    P1_DBG_MARK(2, 1, blockIdx.x)
    const int n_act = basei;
    double acc = 0.0;
    if (d < D) {
        const float* pp = partial + (long)c * D + d;
        int i = rg;
        for (; i + 15 * 16 < n_act; i += 16 * 16) {      // 16 independent loads per thread in flight (256 flagged blocks = one batch)
            float vv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) vv[u] = pp[(long)act[i + 16 * u] * C * D];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += (double)vv[u];
        }
        for (; i + 3 * 16 < n_act; i += 4 * 16) {
            float vv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) vv[u] = pp[(long)act[i + 16 * u] * C * D];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += (double)vv[u];
        }
        for (; i < n_act; i += 16) acc += (double)pp[(long)act[i] * C * D];
    }
    sh[rg][cl] = acc;
    __syncthreads();
    P1_DBG_MARK(2, 2, blockIdx.x)
    if (rg == 0 && d < D) {
        double tsum = 0.0;
#pragma unroll
        for (int gq = 0; gq < 16; ++gq) tsum += sh[gq][cl];
        proto[(long)c * D + d] = n ? (float)(tsum / (double)n) : __uint_as_float(0x7fc00000u);
    }
    P1_DBG_END(2, blockIdx.x)
}

U2PL_API size_t u2pl_compact_workspace_bytes(long P) {
    return (size_t)cdiv(P, CP_PIX) * 3 * MAXC * sizeof(unsigned);
}

// idx: int32 [3][32][cap] (plane 1 is not written) ; counts: u32 [3][32]
// counted != 0: `workspace` already holds the per-block counts (written by u2pl_contra_classify)
U2PL_API int u2pl_compact_lists(const unsigned* abits, const unsigned* lowbits, const unsigned* nbits, long P,
                                int C, void* workspace, int* idx, long cap, unsigned* counts, int counted,
                                hipStream_t stream) {
    if (C > MAXC || P <= 0) return U2PL_EINVAL;
    int nblk = cdiv(P, CP_PIX);
    unsigned* blk = (unsigned*)workspace;
    if (!counted) {
        U2PL_LAUNCH(k_compact_count, dim3(nblk), dim3(CP_PIX), 0, stream, abits, lowbits, nbits, P, blk, nblk);
        U2PL_LAUNCH_CHECK();
    }
    U2PL_LAUNCH(k_compact_scan, dim3(3 * MAXC), dim3(256), 0, stream, blk, nblk, counts);
    U2PL_LAUNCH_CHECK();
    U2PL_LAUNCH(k_compact_write, dim3(nblk), dim3(CP_PIX), 0, stream, abits, nbits, P, blk, nblk, idx, cap);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// Phase 1c: class prototypes = mean of rep_teacher rows over the low-valid set
// (loss_helper.py:119-123).  rows: row r of the (pixel, D) view = base + r*ld.
// Streaming formulation: every pixel row that belongs to at least one class is read ONCE (1 KiB
// coalesced, lane l owns channels [4l, 4l+4)) and added into per-class REGISTER accumulators; the class
// bits of the row are wave-uniform scalars, so only the classes that are set cost any VALU work.
// Balanced assignment: the launch has a FIXED number of waves NW (512 blocks x 4: two resident blocks per CU, one
// round) and wave W streams pixels W, W + NW, W + 2 NW, ... -- every wave gets the same share of every image, so the
// dense labeled image 0, the 20 %-dense unlabeled image B and the empty images in between (quirk Q0) no longer make
// heavy and idle blocks (contiguous 128-pixel blocks: 582 busy blocks of 4 row rounds over 512 slots = two rounds
// of blocks, 28 us; now ~22 rows per wave).  A wave fetches the bits of all its <= 128 pixels with one batch of loads,
// the block pools its members (see proto_stream_body), every wave requests all of its rows in one batch, and the
// waves of a block are combined through LDS in a fixed order => deterministic.  Blocks without members only write flag 0.
// Then an ordered double-precision finish over the flagged blocks.
// ---------------------------------------------------------------------------
#define PR_BLOCKS 256          // x PR_WPB waves: one 8-wave block per CU (the register file admits two waves per SIMD either way;
#define PR_WPB 8               // eight waves per block halve the number -- and the bytes -- of the block partials the finish reads)
// the streaming body: block `pblk` of `npblk` prototype blocks; bits0 / bits1 = class bits of this lane's two pixels
// (pixels W + NW * lane and W + NW * (64 + lane) of wave W = pblk * PR_WPB + wave, NW = npblk * PR_WPB)
#define PR_SLOTS 32            // member rows a wave keeps in flight (one round of loads for up to 8 * 32 members per block)
template <int CT>
__device__ __forceinline__ void proto_stream_body(const float* __restrict__ rows, long ld, int D, unsigned bits0, unsigned bits1,
                                                  float* __restrict__ partial, unsigned* __restrict__ flags,
                                                  unsigned rows_bytes, float* red, int pblk, int npblk) {
    // The block POOLS its member pixels before streaming: the members of its 8 * 128 pixel slots are compacted, in slot
    // order, into an LDS list and wave w takes entries w, w + 8, w + 16, ...  A wave's own 128 slots hold 22 +- 5 members
    // at 769^2 (the slowest of a block's eight waves had ~30 % more rows than the mean and the block waits for it);
    // pooled, every wave of a block streams the same number of rows (+- 1).  All of a wave's rows (<= PR_SLOTS, else
    // further rounds) are requested in ONE batch: one memory round trip instead of three to five dependent batches of 8.
    __shared__ int wcnt_s[2 * PR_WPB];
    __shared__ unsigned short lst_slot[PR_WPB * 128];
    __shared__ unsigned lst_bits[PR_WPB * 128];
    // (readfirstlane: the wave index is uniform, which keeps every row base in scalar registers -- a row load is then
    // "scalar base + per-lane offset" and needs no address VGPRs)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long NW = (long)npblk * PR_WPB;                // waves of the launch (host: P <= 128 * NW)
    const unsigned long long m0 = __ballot(bits0 != 0), m1 = __ballot(bits1 != 0);
    if (lane == 0) { wcnt_s[2 * wave] = __popcll(m0); wcnt_s[2 * wave + 1] = __popcll(m1); }
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int i = 0; i < 2 * PR_WPB; ++i) {
        const int cw = wcnt_s[i];
        if (i < 2 * wave) base += cw;
        total += cw;
    }
    total = __builtin_amdgcn_readfirstlane(total);
    if (!total) {
        if (threadIdx.x == 0) flags[pblk] = 0;
        return;
    }
    {
        const unsigned long long lt = lanemask_lt();
        if (bits0) { const int pos = base + __popcll(m0 & lt); lst_slot[pos] = (unsigned short)(wave * 128 + lane); lst_bits[pos] = bits0; }
        if (bits1) { const int pos = base + __popcll(m0) + __popcll(m1 & lt); lst_slot[pos] = (unsigned short)(wave * 128 + 64 + lane); lst_bits[pos] = bits1; }
    }
    __syncthreads();
    P1_DBG_MARK(1, 1, pblk)
    const int d = lane * 4;
    const bool act = d < D;       // D <= 256
    const unsigned doff_b = act ? d * 4 : 0;           // byte offset of the lane's channels
    const long ldb = ld * 4;
    const __amdgpu_buffer_rsrc_t rrows = make_rsrc(rows, rows_bytes);
    float4 acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e0 = 0; e0 < total; e0 += PR_WPB * PR_SLOTS) {
        // lane u < PR_SLOTS holds entry e0 + wave + 8 u of the block's list; its fields reach the scalar unit by readlane
        const int e = e0 + wave + PR_WPB * lane;
        const bool mine = lane < PR_SLOTS && e < total;
        const unsigned my_slot = mine ? lst_slot[e] : 0u, my_bits = mine ? lst_bits[e] : 0u;
        const int n_mine = __popcll(__ballot(mine));      // uniform
        float4 V[PR_SLOTS];
#pragma unroll
        for (int u = 0; u < PR_SLOTS; ++u)
            if (u < n_mine) {
                const unsigned sl = __builtin_amdgcn_readlane(my_slot, u);
                const long pix = (long)pblk * PR_WPB + (sl >> 7) + NW * (long)(sl & 127u);
                const int rowoff = __builtin_amdgcn_readfirstlane((int)(pix * ldb));     // P * ld * 4 < 2^31 (host check)
                V[u] = buf_load4s(rrows, (int)doff_b, rowoff);
            }
#pragma unroll
        for (int u = 0; u < PR_SLOTS; ++u)
            if (u < n_mine) {
                const unsigned B = __builtin_amdgcn_readlane(my_bits, u);
#pragma unroll
                for (int c = 0; c < CT; ++c)
                    if ((B >> c) & 1u) {   // scalar branch: the asm keeps it from being if-converted
                        asm volatile("");
                        acc[c].x += V[u].x; acc[c].y += V[u].y; acc[c].z += V[u].z; acc[c].w += V[u].w;
                    }
            }
    }
    P1_DBG_MARK(1, 2, pblk)
    if (act && wave >= 4) {
#pragma unroll
        for (int c = 0; c < CT; ++c) *(float4*)(red + ((long)(wave - 4) * CT + c) * D + d) = acc[c];
    }
    __syncthreads();
    if (act && wave < 4) {      // fixed order: (wave w) + (wave w + 4), then the four-way combine below
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const float4 o = *(const float4*)(red + ((long)wave * CT + c) * D + d);
            acc[c].x += o.x; acc[c].y += o.y; acc[c].z += o.z; acc[c].w += o.w;
        }
    }
    __syncthreads();
    if (act && wave < 4) {
#pragma unroll
        for (int c = 0; c < CT; ++c) *(float4*)(red + ((long)wave * CT + c) * D + d) = acc[c];
    }
    __syncthreads();
    P1_DBG_MARK(1, 3, pblk)
    float* out = partial + (long)pblk * CT * D;
    for (int i = threadIdx.x; i < CT * D / 4; i += blockDim.x) {
        const float4 a = ((float4*)red)[i], b2 = ((float4*)red)[CT * D / 4 + i];
        const float4 c2 = ((float4*)red)[2 * CT * D / 4 + i], e = ((float4*)red)[3 * CT * D / 4 + i];
        float4 r;
        r.x = (a.x + b2.x) + (c2.x + e.x); r.y = (a.y + b2.y) + (c2.y + e.y);
        r.z = (a.z + b2.z) + (c2.z + e.z); r.w = (a.w + b2.w) + (c2.w + e.w);
        ((float4*)out)[i] = r;
    }
    if (threadIdx.x == 0) flags[pblk] = 1;
}
template <int CT>
__global__ __launch_bounds__(64 * PR_WPB) void k_proto_stream(const float* __restrict__ rows, long ld, int D,
                                                         const unsigned* __restrict__ lowbits, long P,
                                                         float* __restrict__ partial, unsigned* __restrict__ flags,
                                                         unsigned rows_bytes) {
    extern __shared__ float red[];   // [4][CT][D]: waves 4..7 hand their sums to waves 0..3, whose four sums are then combined
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long NW = (long)gridDim.x * PR_WPB, W = (long)blockIdx.x * PR_WPB + wave;
    const long pa = W + NW * lane, pb = W + NW * (64 + lane);   // lane j holds the class bits of pixels j and 64 + j
    P1_DBG_START(1)
    const unsigned bits0 = pa < P ? lowbits[pa] : 0u;
    const unsigned bits1 = pb < P ? lowbits[pb] : 0u;
    proto_stream_body<CT>(rows, ld, D, bits0, bits1, partial, flags, rows_bytes, red, (int)blockIdx.x, (int)gridDim.x);
    P1_DBG_END(1, blockIdx.x)
}

// grid (C, D/64), 1024 threads: 64 channels x 16 row groups; the flagged blocks are first compacted (in
// ascending order) into an LDS list so that the partial loads are independent (8 in flight per thread)
__global__ __launch_bounds__(1024) void k_proto_finish(const float* __restrict__ partial, int D,
                                                       const unsigned* __restrict__ counts, int nblk, int C,
                                                       const unsigned* __restrict__ flags, float* __restrict__ proto) {
    __shared__ double sh[16][64];
    __shared__ int act[PF_MAXBLK];
    __shared__ int wtot[16];
    const int c = blockIdx.x, cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int d = blockIdx.y * 64 + cl;
    int base = 0;
    for (int b0 = 0; b0 < nblk; b0 += 1024) {
        const int b = b0 + threadIdx.x;
        const bool on = b < nblk && flags[b] != 0;
        const unsigned long long m = __ballot(on);
        if (cl == 0) wtot[rg] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w2 = 0; w2 < rg; ++w2) off += wtot[w2];
        if (on) act[off + __popcll(m & (cl ? (~0ull >> (64 - cl)) : 0ull))] = b;
        int tot = 0;
        for (int w2 = 0; w2 < 16; ++w2) tot += wtot[w2];
        base += tot;
        __syncthreads();
    }
    const int n_act = base;
    const unsigned n = counts[1 * MAXC + c];
    double acc = 0.0;
    if (d < D) {
        const float* p = partial + (long)c * D + d;
        int i = rg;
        for (; i + 7 * 16 < n_act; i += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(long)act[i + 16 * u] * C * D];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += (double)v[u];
        }
        for (; i < n_act; i += 16) acc += (double)p[(long)act[i] * C * D];
    }
    sh[rg][cl] = acc;
    __syncthreads();
    if (rg == 0 && d < D) {
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += sh[g][cl];
        proto[(long)c * D + d] = n ? (float)(t / (double)n) : __uint_as_float(0x7fc00000u);
    }
}

static int proto_blocks(long P) {   // fixed one-round grid; grows only so that a wave never owns more than 128 pixels
    const long need = (P + 128 * PR_WPB - 1) / (128 * PR_WPB);
    return (int)(need > PR_BLOCKS ? need : PR_BLOCKS);
}
U2PL_API size_t u2pl_proto_workspace_bytes(long P, int C, int D) {
    const size_t nblk = proto_blocks(P);
    return nblk * C * D * sizeof(float) + nblk * sizeof(unsigned);
}
// idx/cap are unused by the streaming formulation (kept in the ABI for list-based callers)
U2PL_API int u2pl_class_prototypes(const float* rows, long ld, int D, const int* idx, long cap,
                                   const unsigned* counts, int C, long P, void* workspace, float* proto,
                                   const unsigned* lowbits, hipStream_t stream) {
    (void)idx; (void)cap;
    const int nblk = proto_blocks(P);
    const long rb = ((P - 1) * ld + D) * 4;       // byte extent of the row view (raw-buffer descriptor)
    if (D % 4 || D > 256 || nblk > PF_MAXBLK || P <= 0 || rb >= (1L << 31)) return U2PL_EINVAL;
    const size_t lds = (size_t)4 * C * D * sizeof(float);
    float* partial = (float*)workspace;
    unsigned* flags = (unsigned*)(partial + (size_t)nblk * C * D);
#define PROTO_CASE(CT)                                                                                           \
    case CT: {                                                                                                   \
        static bool set_##CT = false;                                                                            \
        if (!set_##CT) { (void)hipFuncSetAttribute((const void*)k_proto_stream<CT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * CT * 256 * 4)); set_##CT = true; } \
        U2PL_LAUNCH(k_proto_stream<CT>, dim3(nblk), dim3(64 * PR_WPB), lds, stream, rows, ld, D, lowbits, P, partial, flags, (unsigned)rb); \
    } break;
    switch (C) {
        PROTO_CASE(19) PROTO_CASE(21) PROTO_CASE(32)
        default: return U2PL_EINVAL;
    }
#undef PROTO_CASE
    U2PL_LAUNCH_CHECK();
    U2PL_LAUNCH(k_proto_finish, dim3(C, cdiv(D, 64)), dim3(1024), 0, stream, partial, D, counts, nblk, C, flags, proto);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// Phase 1 of compute_contra_memobank_loss (loss_helper.py:80-154) as THREE launches: classify (+ per-block list counts),
// prototype streaming, and the merged tail (ordered compaction write with in-block offsets + list lengths || ordered
// prototype finish).  Same outputs as u2pl_contra_classify + u2pl_compact_lists + u2pl_class_prototypes (five launches).
U2PL_API size_t u2pl_contra_phase1_workspace_bytes(long P, int C, int D) {
    return ((u2pl_compact_workspace_bytes(P) + 255) & ~(size_t)255) + u2pl_proto_workspace_bytes(P, C, D);
}
U2PL_API int u2pl_contra_phase1(const float* prob, long sn, long sc, long sp, const unsigned* lbits, const float* low_mask,
                                const float* high_mask, int N2, int num_labeled, int C, int h, int w, float thr_p,
                                float thr_n, int low_rank, int high_rank, const float* rows, long ld, int D,
                                unsigned* abits, unsigned* lowbits, unsigned* nbits, int* idx, long cap, unsigned* counts,
                                float* proto, void* workspace, hipStream_t stream) {
    const long P = (long)N2 * h * w;
    if (C > MAXC || P <= 0) return U2PL_EINVAL;
    const int npb = proto_blocks(P);
    const long rb = ((P - 1) * ld + D) * 4;
    if (D % 4 || D > 256 || npb > PF_MAXBLK || rb >= (1L << 31) || !(C == 19 || C == 21 || C == 32)) return U2PL_EINVAL;
    const unsigned* blk = (const unsigned*)workspace;
    const int nblk = cdiv(P, CP_PIX);
    float* partial = (float*)((char*)workspace + ((u2pl_compact_workspace_bytes(P) + 255) & ~(size_t)255));
    unsigned* flags = (unsigned*)(partial + (size_t)npb * C * D);
    // (classify and the prototype stream are independent -- the low-valid bits are label bits & low mask -- but running them
    // as two roles of ONE launch was measured SLOWER: 34.1 us vs 19.2 + 10.4 us; the classify blocks inherit the prototype
    // role's register / LDS footprint and one-block-per-CU residency)
    int rc = u2pl_contra_classify(prob, sn, sc, sp, lbits, low_mask, high_mask, N2, num_labeled, C, h, w, thr_p, thr_n,
                                  low_rank, high_rank, abits, lowbits, nbits, workspace, stream);
    if (rc) return rc;
    const size_t lds = (size_t)4 * C * D * sizeof(float);
#define P1_PROTO(CT)                                                                                             \
    case CT: {                                                                                                   \
        static bool set_##CT = false;                                                                            \
        if (!set_##CT) { (void)hipFuncSetAttribute((const void*)k_proto_stream<CT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * CT * 256 * 4)); set_##CT = true; } \
        U2PL_LAUNCH(k_proto_stream<CT>, dim3(npb), dim3(64 * PR_WPB), lds, stream, rows, ld, D, lowbits, P, partial, flags, (unsigned)rb); \
    } break;
    switch (C) {
        P1_PROTO(19) P1_PROTO(21) P1_PROTO(32)
        default: return U2PL_EINVAL;
    }
#undef P1_PROTO
    U2PL_LAUNCH_CHECK();
    const int nwb = cdiv(P, P1_T);
    U2PL_LAUNCH(k_phase1_tail, dim3(nwb + C * cdiv(D, 64)), dim3(P1_T), 0, stream, abits, nbits, P, blk, nblk, idx, cap,
                       counts, nwb, partial, D, npb, C, flags, proto);
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
    U2PL_LAUNCH(k_gather_rows, dim3(grid_for(n * (D / 4), 256)), dim3(256), 0, stream, rows, ld, D, list, n, out);
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
    const long skip = n_new > cap ? n_new - cap : 0;       // more new rows than slots: only the last `cap` are kept
    const unsigned D4 = (unsigned)D >> 2;
    const unsigned total = (unsigned)((n_new - skip) * D4);   // < 2^31 (host check)
    const unsigned base = (unsigned)((tail + skip) % cap);  // slot of the first kept row; block-uniform (one scalar division)
    // index load -> row load -> store is a chain of dependent round trips: every thread runs the chain for FOUR float4 at
    // once (all index loads, then all row loads, then the stores), in 32-bit arithmetic without branches between the
    // loads, and the grid covers the largest class in one pass of at most one resident round of blocks (3100 blocks of
    // one float4 per thread were 1.5 rounds at 769^2: two chains)
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned t0 = blockIdx.x * blockDim.x + threadIdx.x; t0 < total; t0 += 4 * stride) {
        unsigned jj[4], slot[4], dd[4];
        bool on[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned t = t0 + u * stride;
            on[u] = t < total;
            const unsigned tt = on[u] ? t : 0u;
            jj[u] = tt / D4;                           // kept row index, < cap
            dd[u] = tt - jj[u] * D4;
            const unsigned sl = base + jj[u];          // < 2 cap
            slot[u] = sl >= (unsigned)cap ? sl - (unsigned)cap : sl;
        }
        // (descriptor pointers: explicit global-memory loads / stores -- see common.h ldg; flat accesses are each waited for
        // with vmcnt(0) lgkmcnt(0) and the four chains would run one after the other)
        long srcr[4];
        if (list) {
            int li[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) li[u] = ldg(list + skip + jj[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) srcr[u] = li[u];
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) srcr[u] = skip + (long)jj[u];
        }
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ldg((const u32x4*)(rows + srcr[u] * ld + 4 * dd[u]));
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (on[u]) *(U2PL_GLOBAL u32x4*)((u32x4*)bank + (long)slot[u] * D4 + dd[u]) = v[u];
    }
}
U2PL_API int u2pl_bank_append_multi_f32(const long long* desc_dev, int nclass, int D, long ld, long max_new,
                                        hipStream_t stream) {
    if (D % 4) return U2PL_EINVAL;
    if (nclass <= 0 || max_new <= 0) return 0;
    if (max_new * (D / 4) >= (1L << 31)) return U2PL_EINVAL;      // the kernel indexes a class's float4s with 32 bits
    dim3 grid(grid_for((max_new * (D / 4) + 3) / 4, 256, 2048), nclass);
    U2PL_LAUNCH(k_bank_append_multi, grid, dim3(256), 0, stream, desc_dev, D, ld);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// The memory bank as a device-resident object driven from this header alone (SURVEY 8b `u2pl_bank_t`): the ring
// bookkeeping of dequeue_and_enqueue (utils.py:27-47) lives in a small DEVICE state array, so an enqueue needs neither the
// list lengths nor the ring heads on the host and can be issued before the step's host synchronisation.
//   state: int64 [C][5] = {row offset of the class's ring inside `storage`, cap, head, len, ptr}
//   u2pl_bank_init     caps (host) -> offsets = prefix sums, head = len = ptr = 0
//   u2pl_bank_enqueue  class c appends counts_dev[c] rows (rows[idx[c * idx_stride + j]], j < counts_dev[c]; idx NULL:
//                      row j of a class-major block starting at row_start_dev[c]) at its tail, only the last `cap` of
//                      them if there are more (utils.py:38-41), then the state advances: len' = min(len + n, cap),
//                      head' = (tail + n - len') mod cap, ptr' = cap once full, else (ptr + n) mod cap (utils.py:36-45)
// A host that wants the lengths (the reference samples torch.randint(len) on the CPU) copies the state back, or mirrors
// the same arithmetic from the counts it reads anyway (u2pl_amd.hipops.DeviceMemoryBank does the latter).
// ---------------------------------------------------------------------------
struct BankCaps { long long cap[MAXC]; };
__global__ void k_bank_init(long long* __restrict__ state, int C, BankCaps caps) {
    if (threadIdx.x == 0) {
        long long off = 0;
        for (int c = 0; c < C; ++c) {
            state[5 * c + 0] = off; state[5 * c + 1] = caps.cap[c]; state[5 * c + 2] = 0; state[5 * c + 3] = 0; state[5 * c + 4] = 0;
            off += caps.cap[c];
        }
    }
}
U2PL_API size_t u2pl_bank_state_bytes(int C) { return (size_t)C * 5 * sizeof(long long); }
U2PL_API int u2pl_bank_init(long long* state, int C, const long long* caps_host, hipStream_t stream) {
    if (C <= 0 || C > MAXC || !state || !caps_host) return U2PL_EINVAL;
    BankCaps caps = {};
    for (int c = 0; c < C; ++c) {
        if (caps_host[c] <= 0) return U2PL_EINVAL;
        caps.cap[c] = caps_host[c];
    }
    U2PL_LAUNCH(k_bank_init, dim3(1), dim3(64), 0, stream, state, C, caps);
    U2PL_LAUNCH_CHECK();
    return 0;
}
__global__ void k_bank_enqueue(const long long* __restrict__ state, float* __restrict__ storage, int D,
                               const float* __restrict__ rows, long ld, const int* __restrict__ idx, long idx_stride,
                               const long long* __restrict__ row_start, const unsigned* __restrict__ counts) {
    const int c = blockIdx.y;
    const long n_new = counts[c];
    if (n_new <= 0) return;
    const long off = state[5 * c + 0], cap = state[5 * c + 1], head = state[5 * c + 2], len = state[5 * c + 3];
    const long tail = (head + len) % cap;
    const long skip = n_new > cap ? n_new - cap : 0;
    const int D4 = D >> 2;
    const long total = (n_new - skip) * D4;
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long j = skip + t / D4;
        const int dd = (int)(t % D4);
        const long src = idx ? (long)idx[(long)c * idx_stride + j] : (row_start ? row_start[c] : 0) + j;
        long slot = tail + j;
        slot = slot >= cap ? slot % cap : slot;
        ((float4*)storage)[(off + slot) * D4 + dd] = *(const float4*)(rows + src * ld + 4 * dd);
    }
}
__global__ void k_bank_advance(long long* __restrict__ state, const unsigned* __restrict__ counts, int C) {
    const int c = threadIdx.x;
    if (c >= C) return;
    const long n = counts[c];
    const long cap = state[5 * c + 1], head = state[5 * c + 2], len = state[5 * c + 3], ptr = state[5 * c + 4];
    const long tail = (head + len) % cap;
    const long nl = len + n < cap ? len + n : cap;
    const long new_tail = (tail + n) % cap;
    state[5 * c + 3] = nl;
    state[5 * c + 2] = ((new_tail - nl) % cap + cap) % cap;
    state[5 * c + 4] = nl >= cap ? cap : (ptr + n) % cap;
}
U2PL_API int u2pl_bank_enqueue_f32(long long* state, float* storage, int D, const float* rows, long ld, const int* idx,
                                   long idx_stride, const long long* row_start_dev, const unsigned* counts_dev, int C,
                                   hipStream_t stream) {
    if (D % 4 || C <= 0 || C > MAXC || !state || !storage || !rows || !counts_dev) return U2PL_EINVAL;
    // the list lengths are on the device only: a fixed grid per class, grid-stride over the rows (a step adds ~10^3 keys of
    // 64 float4 per class: one pass of 256 blocks; blocks of a class without new keys leave after one load)
    U2PL_LAUNCH(k_bank_enqueue, dim3(256, C), dim3(256), 0, stream, state, storage, D, rows, ld, idx, idx_stride, row_start_dev,
                counts_dev);
    U2PL_LAUNCH_CHECK();
    U2PL_LAUNCH(k_bank_advance, dim3(1), dim3(64), 0, stream, state, counts_dev, C);
    U2PL_LAUNCH_CHECK();
    return 0;
}

U2PL_API int u2pl_bank_append_f32(float* bank, long cap, long tail, int D, const float* rows, long ld,
                                  const int* list, long n_new, hipStream_t stream) {
    if (D % 4 || cap <= 0) return U2PL_EINVAL;
    if (n_new <= 0) return 0;
    long skip = n_new > cap ? n_new - cap : 0;
    U2PL_LAUNCH(k_bank_append, dim3(grid_for((n_new - skip) * (D / 4), 256)), dim3(256), 0, stream, bank,
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

// PRE: K <= 64 -- the K sampled row indices of an anchor sit in the lanes of one register (no index loads in the loop)
// ONLINE: running-maximum softmax for small temperatures.  The default form shifts every logit by the bound 1/temp, so
// its terms are exp(l - 1/temp) with l in [-1/temp, 1/temp]: below temp ~ 0.023 every term can underflow fp32 and the
// loss would be log(0), where the reference's max-shifted F.cross_entropy (loss_helper.py:228-230) stays finite.  The
// host selects ONLINE when 2/temp > NCE_FIXED_SHIFT_MAX (the stock temp = 0.5 gives 4): the maximum of the logits seen
// so far is carried across the four-row batches and the accumulators are rescaled when it grows (one more exp and
// three more multiplies per batch).
#define NCE_FIXED_SHIFT_MAX 80.0f
// Work folded into the InfoNCE launch (u2pl_infonce_fused_f32; all pointers NULL: plain u2pl_infonce_f32):
//   zero_*: the rows of the persistent row-sparse gradient buffer that the PREVIOUS step's backward wrote are cleared
//     here, one row per wave at the top of the kernel (their consumer finished a step ago) -- instead of a launch of
//     u2pl_zero_rows_f32 in front of the ordered scatter;
//   partial / ticket / loss: the loss reduction (u2pl_infonce_reduce_f32) in the same launch: every block publishes the
//     sum of its four anchors (write-through store), takes a ticket (32 shards, then one top counter: ~40 same-address
//     atomics in a row instead of 1216), and the block that completes the top counter adds the 1216 partials in a fixed
//     order (bit-reproducible) and writes the loss.  The counters are left at zero for the next launch.
struct NceExtra {
    float* zero_dst; long zero_ld; const int* zero_pix; long zero_n;
    float* partial; unsigned* ticket; float* loss; double scale;
};
__device__ __forceinline__ void nce_ld4(const float* p0, const float* p1, const float* p2, const float* p3, float (&v)[4]) {
    // four independent loads past the L1 (sc1), waited for once (chains of atomic loads are issued two per round trip)
    asm volatile("global_load_dword %0, %4, off sc1\n\tglobal_load_dword %1, %5, off sc1\n\t"
                 "global_load_dword %2, %6, off sc1\n\tglobal_load_dword %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}
template <int VPL, bool PRE, bool ONLINE = false>  // floats per lane = D / 64
__global__ void k_infonce(const NceJob* __restrict__ jobs, const float* __restrict__ rep, long ld, int D,
                          int Q, int K, float inv_temp, float* __restrict__ loss_q,
                          float* __restrict__ ganchor, int* __restrict__ anchor_pix, int* __restrict__ head,
                          int* __restrict__ next, const int* __restrict__ seg_len, NceExtra X) {
    const int job = blockIdx.y;
    const int q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    // (the index of the stale row this wave clears is requested now and used at the very end: off the critical path)
    const long zw = ((long)job * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int zpix = -1;
    if (X.zero_dst && zw < X.zero_n) zpix = ldg(X.zero_pix + zw);
    if (q >= Q) return;      // (never taken on the fused path: the host requires Q % 4 == 0 there)
    const NceJob J = jobs[job];
    // Two dependent address chains start here: anchor = rep[cand[idx_a[q]]] (three loads deep) and the sampled bank rows
    // bank[slot(idx_n[q][j])] (two deep).  Both index loads are issued back to back, and the first four feature rows are
    // requested (load4 below) BEFORE the anchor row is consumed, so the chains overlap instead of queueing.
    const long long ia = ldg(J.idx_a + q);
    long long in_lane = 0;
    if (PRE && lane < K) in_lane = ldg(J.idx_n + (long)q * K + lane);
    const int pix = ldg(J.cand + ia);
    float a[VPL], ah[VPL], acc[VPL], na;   // na = 1 / |a|
    const float* ar = rep + (long)pix * ld + lane * VPL;
#pragma unroll
    for (int i = 0; i < VPL; ++i) { a[i] = ar[i]; acc[i] = 0.f; }
    // softmax over the 1+K logits, accumulating sum_j softmax_j * fhat_j.  The logits are cosines / temp, i.e. bounded by
    // 1 / temp in magnitude: shifting by that bound keeps every exponent in [-2/temp, 0] (temp = 0.5: e^-4 .. 1), so no
    // running maximum, no rescaling of the accumulators and ONE exp per row (the online-max form needed two exps and a
    // dependent rescale chain across the rows).  Rows are processed four at a time: four independent 1 KiB row loads in
    // flight per wave and eight interleaved all-lane sums.
    float s = 0.f, l0 = 0.f, cw = 0.f;  // cw = sum_j w_j * cos_j (un-normalised)
    float shift = ONLINE ? -INFINITY : inv_temp;        // fixed form: >= every logit; ONLINE: running maximum
    float f0h[VPL];
    // the K sampled bank rows of this anchor: one coalesced index load, then lane broadcasts (no dependent
    // index -> row latency chain inside the loop)
    long myrow = 0;
    // logical row -> physical slot of the ring: head + idx < 2 * cap (idx < length <= cap), so one conditional subtract
    // replaces the 64-bit modulo (a ~40-instruction software division per row)
    auto slot = [&](long long idx) -> long { const long r = J.bank_head + (long)idx; return r >= J.bank_cap ? r - J.bank_cap : r; };
    if (PRE && lane < K) myrow = slot(in_lane);
    auto row_ptr = [&](int j) -> const float* {
        long r;
        if constexpr (PRE) {   // j is wave-uniform: two v_readlane instead of an LDS-crossbar shuffle, the row base stays scalar
            const int src = max(j - 1, 0);
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)myrow, src);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long)myrow >> 32), src);
            r = (long)(((unsigned long)hi << 32) | lo);
        }
        else r = slot(ldg(J.idx_n + (long)q * K + max(j - 1, 0)));
        return j == 0 ? J.proto + lane * VPL : J.bank + r * D + lane * VPL;
    };
    auto load4 = [&](float (&f)[4][VPL], int j0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* fr = row_ptr(min(j0 + u, K));   // proto / bank pointers come out of the job descriptor: ldg
#pragma unroll
            for (int i = 0; i < VPL; ++i) f[u][i] = ldg(fr + i);
        }
    };
    auto process4 = [&](float (&f)[4][VPL], int j0) {
        float nf[4], dot[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            nf[u] = 0.f;
            dot[u] = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i) { nf[u] = fmaf(f[u][i], f[u][i], nf[u]); dot[u] = fmaf(ah[i], f[u][i], dot[u]); }   // (the library is built with -ffp-contract=off: fused multiply-adds are spelled out where they are wanted)
        }
        // eight wave totals in ONE transposed reduction (19 instructions; eight separate 6-step DPP reductions + readlanes
        // were 64): each step halves the number of live values instead of the number of live lanes.  gfx950's
        // v_permlane32_swap / v_permlane16_swap exchange half-waves / odd-even rows between two registers, so "add the two
        // halves of X and of Y" is one swap + one add and leaves X's sums in one half, Y's in the other.  After the
        // two swap levels row u (lanes 16u .. 16u+15) holds the partial sums of row u's |f|^2 in one register and of its
        // dot product in another; a select + row_ror:8 puts |f|^2 into the row's lanes 0..7 and the dot product into 8..15
        // of ONE register, three more DPP adds (half-mirror, quad xor 1, quad xor 2) finish both.  The per-row scalar
        // chain (rsqrt, exp) then runs once for the four rows: row u's values are read from lane 16 u.
        float vn, vd;
        {
            auto sw32 = [](float x, float y) {
                const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
                return __uint_as_float(r[0]) + __uint_as_float(r[1]);
            };
            auto sw16 = [](float x, float y) {
                const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
                return __uint_as_float(r[0]) + __uint_as_float(r[1]);
            };
            const float a0 = sw32(nf[0], nf[2]), a1 = sw32(nf[1], nf[3]), a2 = sw32(dot[0], dot[2]), a3 = sw32(dot[1], dot[3]);
            const float b0 = sw16(a0, a1), b1 = sw16(a2, a3);           // row u: partial |f_u|^2 / partial a.f_u
            const bool hi8 = (lane & 8) != 0;
            const float keep = hi8 ? b1 : b0, send = hi8 ? b0 : b1;
            float c = keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x128, 0xf, 0xf, false));
            c = dpp_add<0x141>(c);     // row_half_mirror: i <-> 7 - i inside each group of eight
            c = dpp_add<0xB1>(c);      // quad_perm [1,0,3,2]
            c = dpp_add<0x4E>(c);      // quad_perm [2,3,0,1]
            vn = c;                    // (valid in lanes 16u .. 16u+7; the other lanes compute along and are never read)
            vd = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c), 0x128, 0xf, 0xf, false));
        }
        // cos = (ahat . f) / |f|, fhat = f / |f|; softmax weight w = exp(l - shift)
        // 1 / max(|f|, 1e-8) as ONE v_rsq_f32 (1 ulp) instead of an IEEE sqrt + IEEE division (~25 instructions): the
        // kernel is VALU-issue bound (measured: spelling the FMAs out took it from 53.8 to 49.8 us), and the parity
        // tolerance of this float path (1e-4 on the loss, 1e-5 on the gradient) is four orders above 1 ulp
        const float vinv = __frsqrt_rn(fmaxf(vn, 1e-16f));
        const float vcos = vd * vinv;
        const float vl = vcos * inv_temp;
        if constexpr (ONLINE) {
            // rows past the end are clamped duplicates of row K (a real logit): including them in the maximum is harmless
            const float m4 = fmaxf(fmaxf(lane_get(vl, 0), lane_get(vl, 16)), fmaxf(lane_get(vl, 32), lane_get(vl, 48)));
            const float mnew = fmaxf(shift, m4);
            const float resc = __expf(shift - mnew);      // first batch: exp(-inf) = 0 on all-zero accumulators
            shift = mnew;
            s *= resc;
            cw *= resc;
#pragma unroll
            for (int i = 0; i < VPL; ++i) acc[i] *= resc;
        }
        const float vw = __expf(vl - shift);          // v_exp_f32 on an argument in [-2/temp, 0]
        const float vwn = vw * vinv;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (j0 + u > K) continue;
            const float w = lane_get(vw, 16 * u), cosv = lane_get(vcos, 16 * u), wn = lane_get(vwn, 16 * u);
            if (j0 + u == 0) {
                l0 = lane_get(vl, 16 * u);
                const float inv = lane_get(vinv, 16 * u);
#pragma unroll
                for (int i = 0; i < VPL; ++i) f0h[i] = f[u][i] * inv;
            }
            s += w;
            cw = fmaf(w, cosv, cw);
#pragma unroll
            for (int i = 0; i < VPL; ++i) acc[i] = fmaf(wn, f[u][i], acc[i]);   // acc accumulates w * fhat
        }
    };
    // two batches of four rows in flight: the next batch is requested before the current one is reduced.  The requests
    // are UNCONDITIONAL (row indices are clamped to K, a batch past the end re-reads row K from the cache): a load inside
    // an `if` leaves the number of younger loads in flight unknown, the compiler then waits with vmcnt(0) before every
    // reduction -- i.e. also for the batch just requested -- and the two batches stop overlapping (54 us instead of ~40).
    float fa[4][VPL], fb[4][VPL];
    load4(fa, 0);
    {   // the anchor's norm: consumed only now, with the first feature rows already in flight
        float na2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) na2 = fmaf(a[i], a[i], na2);
        na = __frsqrt_rn(fmaxf(wave_sum_sgpr(na2), 1e-16f));     // 1 / max(|a|, 1e-8)
#pragma unroll
        for (int i = 0; i < VPL; ++i) ah[i] = a[i] * na;
    }
    for (int j0 = 0; j0 <= K; j0 += 8) {
        load4(fb, j0 + 4);
        process4(fa, j0);
        if (j0 + 4 > K) break;
        load4(fa, j0 + 8);
        process4(fb, j0 + 4);
    }
    const float lse = shift + logf(s);
    if (X.partial) {
        __shared__ float sl[4];
        __shared__ unsigned s_last;
        if (lane == 0) sl[threadIdx.x >> 6] = lse - l0;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned gb = (unsigned)job * gridDim.x + blockIdx.x, nb = gridDim.x * gridDim.y;
            __hip_atomic_store(X.partial + gb, (sl[0] + sl[1]) + (sl[2] + sl[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // written through before the ticket
            const unsigned sh = gb & 31u, nsh = (nb - sh + 31u) / 32u, nshards = nb < 32u ? nb : 32u;
            unsigned last = 0;
            if (atomicAdd(X.ticket + 1 + sh, 1u) == nsh - 1u) {
                __hip_atomic_store(X.ticket + 1 + sh, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (atomicAdd(X.ticket, 1u) == nshards - 1u) {
                    __hip_atomic_store(X.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    last = 1;
                }
            }
            s_last = last;
        }
        __syncthreads();
        if (s_last) {      // every partial has been written through: add them in a fixed order
            __shared__ double sred[256];
            const int nb = gridDim.x * gridDim.y;
            double acc = 0.0;
            for (int i0 = 0; i0 < nb; i0 += 4 * 256) {
                float v[4];
                const int i = i0 + threadIdx.x;
                nce_ld4(X.partial + min(i, nb - 1), X.partial + min(i + 256, nb - 1), X.partial + min(i + 512, nb - 1),
                        X.partial + min(i + 768, nb - 1), v);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc += i + 256 * u < nb ? (double)v[u] : 0.0;
            }
            sred[threadIdx.x] = acc;
            __syncthreads();
            for (int o = 128; o > 0; o >>= 1) {
                if ((int)threadIdx.x < o) sred[threadIdx.x] += sred[threadIdx.x + o];
                __syncthreads();
            }
            if (threadIdx.x == 0) *X.loss = (float)(sred[0] * X.scale);
        }
    }
    if (lane == 0) {
        const int e = job * Q + q;
        loss_q[e] = lse - l0;
        anchor_pix[e] = pix;
        // Anchors are drawn with replacement: the entries of ONE job that sampled the same candidate were grouped on the
        // host (it drew the indices), seg_len[e] > 0 marks the first entry of such a group.  A pixel can also sit in
        // several class lists, so the group leaders (at most one per job) are chained through an integer exchange; the
        // backward pass sums group by group in ascending entry order: no floating-point atomics, reproducible bits.
        if (head && seg_len[e] > 0) next[e] = atomicExch(head + pix, e);
    }
    // d loss/d cos_j = (softmax_j - [j==0]) * inv_temp ; d cos_j/d a = (fhat_j - cos_j*ahat)/|a|
    const float inv_s = 1.0f / s;
    const float cosbar = cw * inv_s - l0 / inv_temp;  // sum_j (softmax_j-[j==0]) cos_j
    float* g = ganchor + ((long)job * Q + q) * D + lane * VPL;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
        g[i] = inv_temp * ((acc[i] * inv_s - f0h[i]) - cosbar * ah[i]) * na;
    if (X.zero_dst) {      // stale gradient rows of the previous step: one 1 KiB row per wave (D <= 256: 16 bytes per lane)
        const long nw = (long)gridDim.x * gridDim.y * (blockDim.x >> 6);
        if (zpix >= 0 && lane * 4 < D) *(float4*)(X.zero_dst + (long)zpix * X.zero_ld + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        for (long r = zw + nw; r < X.zero_n; r += nw)       // (more stale rows than waves: only when the previous step had more jobs)
            if (lane * 4 < D) *(float4*)(X.zero_dst + (long)X.zero_pix[r] * X.zero_ld + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

static int nce_launch(const void* jobs_dev, int njobs, const float* rep, long ld, int D, int Q, int K, float temp,
                      float* loss_q, float* ganchor, int* anchor_pix, int* head, int* next, const int* seg_len, NceExtra X,
                      hipStream_t stream);
U2PL_API int u2pl_infonce_f32(const void* jobs_dev, int njobs, const float* rep, long ld, int D, int Q, int K,
                              float temp, float* loss_q, float* ganchor, int* anchor_pix, int* head, int* next,
                              const int* seg_len, hipStream_t stream) {
    NceExtra X = {};
    return nce_launch(jobs_dev, njobs, rep, ld, D, Q, K, temp, loss_q, ganchor, anchor_pix, head, next, seg_len, X, stream);
}
// u2pl_infonce_f32 + u2pl_infonce_reduce_f32 + (optionally) u2pl_zero_rows_f32 of the previous step's rows in ONE launch.
// workspace: u2pl_infonce_fused_workspace_bytes(njobs, Q) bytes, zeroed ONCE by the caller and reused (the kernel leaves
// its ticket counters at zero).  zero_dst == NULL or zero_n == 0: nothing to clear.  Q % 4 != 0 -> U2PL_EINVAL (use the
// separate entry points).  *loss = sum of all per-anchor losses * inv_valid_seg / Q (loss_helper.py:228-233).
U2PL_API size_t u2pl_infonce_fused_workspace_bytes(int njobs, int Q) { return ((size_t)njobs * (Q / 4 + 1) + 64) * sizeof(float); }
U2PL_API int u2pl_infonce_fused_f32(const void* jobs_dev, int njobs, const float* rep, long ld, int D, int Q, int K,
                                    float temp, float* loss_q, float* ganchor, int* anchor_pix, int* head, int* next,
                                    const int* seg_len, float* zero_dst, long zero_ld, const int* zero_pix, long zero_n,
                                    void* workspace, float inv_valid_seg, float* loss, hipStream_t stream) {
    if (Q % 4 || !workspace || !loss || D > 256) return U2PL_EINVAL;
    NceExtra X = {};
    if (zero_dst && zero_pix && zero_n > 0) { X.zero_dst = zero_dst; X.zero_ld = zero_ld; X.zero_pix = zero_pix; X.zero_n = zero_n; }
    X.ticket = (unsigned*)workspace;                  // [0] top counter, [1..32] shard counters
    X.partial = (float*)workspace + 64;
    X.loss = loss;
    X.scale = (double)inv_valid_seg / (double)Q;
    return nce_launch(jobs_dev, njobs, rep, ld, D, Q, K, temp, loss_q, ganchor, anchor_pix, head, next, seg_len, X, stream);
}
static int nce_launch(const void* jobs_dev, int njobs, const float* rep, long ld, int D, int Q, int K, float temp,
                      float* loss_q, float* ganchor, int* anchor_pix, int* head, int* next, const int* seg_len, NceExtra X,
                      hipStream_t stream) {
    if (head && (!next || !seg_len)) return U2PL_EINVAL;
    if (njobs <= 0) return 0;
    dim3 grid(cdiv(Q, 4), njobs), block(256);
    const NceJob* jobs = (const NceJob*)jobs_dev;
    const float it = 1.0f / temp;
    if (!(temp > 0.f)) return U2PL_EINVAL;
    const bool online = 2.0f * it > NCE_FIXED_SHIFT_MAX;      // see k_infonce: the fixed shift would underflow
#define NCE_ARGS grid, block, 0, stream, jobs, rep, ld, D, Q, K, it, loss_q, ganchor, anchor_pix, head, next, seg_len, X
#define NCE_LAUNCH(V)                                                                    \
    if (online) {                                                                        \
        if (K <= 64) U2PL_LAUNCH((k_infonce<V, true, true>), NCE_ARGS);           \
        else U2PL_LAUNCH((k_infonce<V, false, true>), NCE_ARGS);                  \
    } else if (K <= 64) U2PL_LAUNCH((k_infonce<V, true>), NCE_ARGS);              \
    else U2PL_LAUNCH((k_infonce<V, false>), NCE_ARGS);
    switch (D) {
        case 64: NCE_LAUNCH(1) break;
        case 128: NCE_LAUNCH(2) break;
        case 256: NCE_LAUNCH(4) break;
        case 512: NCE_LAUNCH(8) break;
        default: return U2PL_EINVAL;
    }
#undef NCE_LAUNCH
#undef NCE_ARGS
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
    U2PL_LAUNCH(k_infonce_reduce, dim3(1), dim3(1024), 0, stream, loss_q, njobs, Q, inv_valid_seg, loss);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// Row-sparse, ordered gradient of the anchors (replaces zero-filling the dense (P, D) gradient -- 152 MB at 769^2 --
// and the float atomicAdd scatter).  dst is a PERSISTENT all-zero buffer.  Host-side grouping (the host drew the anchor
// indices): order[] lists every job's entries sorted by (candidate, entry), seg_pos[e] / seg_len[e] give the group of
// leader e inside order[] (seg_len = 0 for non-leaders).  The wave of the leader that heads a pixel's chain
// (head[pix] == e) collects the chain of leaders (<= one per job), sorts it ascending and adds the rows group by
// group, member by member -- i.e. in ascending entry order -- then writes dst[pix] = scale * gout * sum with a plain
// store and re-arms head[pix] = -1.  u2pl_zero_rows_f32 clears the written rows before the buffer is reused.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_scatter_rows_ordered(float* __restrict__ dst, long ld, int D,
                                                              const int* __restrict__ pix, const int* __restrict__ next,
                                                              int* __restrict__ head, const int* __restrict__ order,
                                                              const int* __restrict__ seg_pos, const int* __restrict__ seg_len,
                                                              const float* __restrict__ src, int n,
                                                              const float* __restrict__ gout, float scale) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= n) return;
    // Everything that is indexed by the entry itself is requested at once, then everything that hangs off those values:
    // three dependent round trips on the common path (a leader that is alone in its chain) instead of seven
    // (seg_len -> pix -> head -> next -> seg_pos -> order -> row).
    const int len_e = seg_len[e], p = pix[e], nxt = next[e], pos_e = seg_pos[e];
    if (len_e <= 0) return;                        // not a group leader (its next[] / seg_pos[] are not meaningful: nothing below uses them)
    const int hd = head[p], id0 = order[pos_e];
    const bool act = lane * 4 < D;
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act) v0 = *(const float4*)(src + (long)id0 * D + lane * 4);     // first member of this leader's own group (used on the common path)
    if (hd != e) return;                           // not the chain head: the head's wave does the work
    const float sc = scale * (gout ? *gout : 1.0f);
    if (nxt < 0) {                                 // the only leader on this pixel: its group in entry order
        float4 acc = v0;
        for (int k = 1; k < len_e; ++k) {
            const int id = order[pos_e + k];
            if (act) {
                const float4 v = *(const float4*)(src + (long)id * D + lane * 4);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        // (0 + v0 == v0 bit for bit unless v0 is -0: the general path below starts from +0, so add it explicitly)
        if (act) *(float4*)(dst + (long)p * ld + lane * 4) = make_float4(sc * (0.f + acc.x), sc * (0.f + acc.y), sc * (0.f + acc.z), sc * (0.f + acc.w));
        if (lane == 0) head[p] = -1;
        return;
    }
    int myid = 0x7fffffff, cnt = 0;
    for (int cur = e; cur >= 0 && cnt < 64; cur = next[cur]) {   // <= one leader per job (MAXC = 32 jobs)
        if (lane == cnt) myid = cur;
        ++cnt;
    }
    int rank = 0;
    for (int j = 0; j < cnt; ++j) rank += __shfl(myid, j, 64) < myid;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < cnt; ++r) {
        const unsigned long long m = __ballot(lane < cnt && rank == r);
        const int L = __shfl(myid, __ffsll((long long)m) - 1, 64);
        const int pos = seg_pos[L], len = seg_len[L];
        for (int k = 0; k < len; ++k) {            // independent row loads: they pipeline
            const int id = order[pos + k];
            if (act) {
                const float4 v = *(const float4*)(src + (long)id * D + lane * 4);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
    }
    if (act) *(float4*)(dst + (long)p * ld + lane * 4) = make_float4(sc * acc.x, sc * acc.y, sc * acc.z, sc * acc.w);
    if (lane == 0) head[p] = -1;
}
U2PL_API int u2pl_scatter_rows_ordered_f32(float* dst, long ld, int D, const int* pix, const int* next, int* head,
                                           const int* order, const int* seg_pos, const int* seg_len, const float* src,
                                           long n, const float* gout_dev, float scale, hipStream_t stream) {
    if (n <= 0) return 0;
    if (D % 4 || D > 256) return U2PL_EINVAL;
    U2PL_LAUNCH(k_scatter_rows_ordered, dim3(cdiv(n, 4)), dim3(256), 0, stream, dst, ld, D, pix, next, head, order,
                       seg_pos, seg_len, src, (int)n, gout_dev, scale);
    U2PL_LAUNCH_CHECK();
    return 0;
}
__global__ void k_zero_rows(float* __restrict__ dst, long ld, int D, const int* __restrict__ pix, long n) {
    const int D4 = D >> 2;
    const long total = n * D4;
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x)
        *(float4*)(dst + (long)pix[t / D4] * ld + 4 * (t % D4)) = make_float4(0.f, 0.f, 0.f, 0.f);
}
U2PL_API int u2pl_zero_rows_f32(float* dst, long ld, int D, const int* pix, long n, hipStream_t stream) {
    if (n <= 0) return 0;
    if (D % 4) return U2PL_EINVAL;
    U2PL_LAUNCH(k_zero_rows, dim3(grid_for(n * (D / 4), 256)), dim3(256), 0, stream, dst, ld, D, pix, n);
    U2PL_LAUNCH_CHECK();
    return 0;
}
