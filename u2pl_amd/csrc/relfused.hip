// Fused reliability split: ONE persistent launch for
//   bilinear(align_corners=True) up-sampling of the teacher logits + per-pixel entropy   (train_semi.py:371-374,402)
//   exact np.percentile thresholds at drop_percent / alpha_t / 100-alpha_t               (loss_helper.py:38-40,
//                                                                                          train_semi.py:405-415)
//   unsup target overwrite, low / high entropy masks, legacy-nearest down-sampling and
//   the label_onehot batch-slot-0 class bits                                             (loss_helper.py:41-43,
//                                                                                          train_semi.py:408-465, utils.py:50-59)
// instead of the five launches of the un-fused path (entropy_up, select pass 1, pass 2, finish, apply), which sat on
// the ~4 us launch floor and on three latency-bound resolve prologues.
//
// Execution model: gridDim.x = G <= #CUs blocks of 1024 threads, all co-resident (one per CU), separated by ONE
// device-wide barrier (atomic counters + agent-scope fences; a second one only for degenerate distributions).  Every
// block owns a contiguous range of 4x4 "cells" of the full-resolution grid; the entropies and labels of its pixels never
// leave registers between the phases.
//   A  entropy (bit-exact FMA bilinear of the 4 corner logits staged in LDS), block histogram over 2048 MONOTONE bins of
//      the entropy value (log-linear below 2^-6, linear above: spreads both the near-zero entropies of a trained model
//      and the near-ln(C) entropies of an untrained one)
//   P  publish: the block counting-sorts its entropies by bin in LDS and writes (write-through 16-byte stores) the SORTED
//      values and the exclusive prefix of its histogram (the offset of every bin inside the sorted run); its bin counts
//      go to the global totals with fire-and-forget atomics                                              [barrier]
//   C  every block scans the totals, derives the six ranks (numpy virtual index (n-1)*q in float32) and finds the bin of
//      each rank (<= 6 distinct bins, ~600 values each at 769^2)
//   G  gather: the members of those bins are read straight out of the 256 blocks' sorted runs (two prefix words per
//      block and bin give offset and count) into LDS -- nobody has to be asked for them, so there is no second barrier
//   D  exact order statistics in LDS (radix select on the order-preserving key), thresholds lerped like numpy (float32,
//      no FMA) and applied to the block's own register-resident pixels; the labeled-half masks and the class bits are
//      label-only and are written at the start.
// Round 2 needed the bin of a rank BEFORE a block could say which of its values were candidates: ranks -> candidate
// emission -> second barrier -> selection (two barriers, 71 us).  Publishing every block's values sorted by bin makes
// the candidates addressable by everyone after the first barrier: 8 KB + <= 36 KB published per block instead of
// 8 KB, one barrier (and its two fences) and the emission phase less.
// Degenerate distributions (more than RF_CAP candidates in the selected bins: all-equal entropies, heavy ties) keep the
// round-2 route behind the same first barrier: candidate emission into one compact global list, a second barrier, and
// the out-of-LDS selection.  Every launch advances the barrier counters by two slots whichever route it takes.
// Exactness: the bins partition the values monotonically, so "bin of the rank, then rank inside the bin" is the
// exact order statistic; integer-only bookkeeping; no floating-point atomics anywhere.
#include "common.h"
#include "u2pl_hip.h"

#define RF_T 1024
#define RF_CELLS 256            // cells per block iteration (thread = cell x output row of the cell)
#define RF_NIT 2                // iterations per block: G * RF_NIT * RF_CELLS cells at most
#define RF_BINS 2048
#define RF_CAP 12288            // candidate keys selected in LDS; longer lists are swept from global memory
#define RF_MAXSLOT 6
// workspace words
#define RFW_BAR 0
#define RFW_NKEPT 2
#define RFW_THR 16              // [3] thresholds (float bits), [6] selected values at 24..29
#define RFW_VAL 24
#define RFW_TOT 8192            // [2][8][2048] bin totals: double-buffered by launch parity, one copy per XCD (block b adds
                                // to copy b % 8: a bin's fire-and-forget atomics then serialise in chains of G/8 instead of G)
#define RFW_SLAB (8192 + 2 * 8 * 2048)   // [G][RF_SLABW]: exclusive prefix of block b's histogram (word 2048 = its #valid)
#define RF_SLABW 2064
#define RF_PXMAX ((RF_NIT * RF_CELLS + 64) * 16)   // pixels a block can own = capacity of its sorted run
#define RFW_BIG 5               // launches on this workspace that needed the second barrier (statistics)
#define RFW_LAUNCH 4

struct RfArgs {
    const float* in; long sn, sc, sh, sw;
    int B, h, w, H, W;
    float sy, sx, ny, nx;
    int hm, wm;                  // size of the low-res masks (= student prediction map)
    const long long* label_u; const long long* label_l;
    int ignore, nspec, neg_high;
    unsigned epoch;              // launch index on this workspace (host counter): barrier targets and totals parity
    int fences;                  // agent-scope fence pair around the first barrier (see rf_grid_arrive_wait)
    float q32[3];
    float bin_scale;
    float* ent; long long* target_u; float* low_mask; float* high_mask; unsigned* lbits;
    unsigned* ws; float* cand;
};

U2PL_API size_t u2pl_reliability_fused_workspace_bytes(int G) { return (size_t)(RFW_SLAB + (size_t)G * RF_SLABW) * sizeof(unsigned); }
// floats of scratch: the G sorted runs, then n_px words for the compact candidate list of the degenerate route
U2PL_API size_t u2pl_reliability_fused_cand_floats(long n_px, int G) { return (size_t)G * RF_PXMAX + (size_t)n_px; }

// exp(x) for x <= 0 (the max-shifted logits): 2^(x * log2 e) on v_exp_f32 with the rounding error of the product carried in
// a first-order correction -- ~1.5 ulp, 6 VALU instructions instead of the 13 of the library expf (no range checks, no
// ldexp: the argument never overflows and flushing below 2^-126 is harmless for a softmax term)
__device__ __forceinline__ float rf_exp_neg(float x) {
    const float t = __fmul_rn(x, 1.44269502162933349609375f);
    float r = __fmaf_rn(x, 1.44269502162933349609375f, -t);
    r = __fmaf_rn(x, 1.92596299e-8f, r);
    const float e = __builtin_amdgcn_exp2f(t);
    return __fmaf_rn(e, __fmul_rn(r, 0.693147180559945f), e);
}

__device__ __forceinline__ int rf_bin(float e, float scale) {
    // monotone non-decreasing in e (e is not NaN); bin 0: e < 2^-22 (incl. the tiny negatives of rounding)
    if (!(e >= 2.384185791015625e-07f)) return 0;
    if (e < 0.015625f) return 1 + (int)((__float_as_uint(e) - 0x34800000u) >> 17);     // 16 octaves x 64 bins -> 1..1024
    const int b = (int)(__fmul_rn(__fsub_rn(e, 0.015625f), scale));
    return 1025 + (b < 1022 ? b : 1022);
}

// Key range of histogram bin b >= 1 (order-preserving keys of the non-negative floats of the bin): [*lo, *lo + 2^*nb).
// Log-linear bins are exact ranges of bit patterns; the linear bins get conservative edges (2e-6 of slack on entropies
// <= ln C: far below a bin's width, far above the rounding of rf_bin's subtract-multiply).  A key outside the returned
// range is detected by the caller, which then takes the generic selection.
__device__ __forceinline__ void rf_bin_key_range(unsigned b, float scale, unsigned& lo, unsigned& nb) {
    if (b <= 1024u) {
        lo = 0xB4800000u + ((b - 1u) << 17);
        nb = 17;
        return;
    }
    const float inv = 1.0f / scale;
    const float j = (float)(b - 1025u);
    const float e_lo = fmaxf(0.015625f + j * inv - 2e-6f, 0.0f);
    const float e_hi = b >= 2047u ? 8.0f : 0.015625f + (j + 1.0f) * inv + 2e-6f;
    lo = f32_key(e_lo);
    const unsigned range = f32_key(e_hi) - lo;
    nb = range ? 32u - (unsigned)__clz(range) : 1u;
}

// Device-wide barrier of a persistent launch whose blocks are all co-resident (gridDim.x <= #CUs, one block per CU).
// The spin is bounded (~seconds): if some block could not become resident the kernel gives up loudly (error word set,
// results undefined) instead of hanging the GPU.
#define RFW_ERR 3
#define RFW_BAR8 64             // 8 arrival counters, 16 words (one 64-byte line) apart
// the arrive + wait part, executed by the first 8 lanes of wave 0 (the caller brackets it with block barriers).
// fences == false: every cross-block byte of the usual route is written with write-through (sc1) stores / agent-scope
// atomics that the writer has DRAINED (s_waitcnt vmcnt(0)) before arriving, and read with sc1 loads (never served from a
// CU's L1) -- the "sc1 stores + drained counter, sc1 loads" hand-off of MI355X_MICROARCH.md -- so the L2 write-back and
// the L1 invalidate of the fence pair (~2 x 1.7 us) buy nothing.  U2PL_RF_FENCES=1 restores them.
__device__ __forceinline__ void rf_grid_arrive_wait(unsigned* ws, unsigned phase, int G, bool fences) {
    if (threadIdx.x < 8) {
        if (threadIdx.x == 0) {
            if (fences) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(ws + RFW_BAR8 + 16 * (blockIdx.x & 7), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // lane k polls counter k: it expects phase * (number of blocks with index % 8 == k)
        const unsigned want = phase * (unsigned)((G - (int)threadIdx.x + 7) / 8);    // phase = 2 * epoch + {1, 2}
        unsigned spins = 0;
        while ((int)(__hip_atomic_load(ws + RFW_BAR8 + 16 * threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {   // wrap-safe
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 19)) { __hip_atomic_store(ws + RFW_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        if (fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
}
__device__ __forceinline__ void rf_grid_sync(unsigned* ws, unsigned phase, int G) {
    // Arrival: thread 0 releases at agent scope (write-back of the XCD's L2: the block's stores were acknowledged at
    // the __syncthreads()), then bumps one of 8 counters (same-address atomics serialise at ~88 / us: one shared counter
    // costs ~3 us per barrier in arrivals alone); lanes 0..7 poll the 8 counters RELAXED (an acquire per poll would
    // invalidate the L2 on every iteration: measured 13-17 us per barrier) and thread 0 acquires once at the end.
    __syncthreads();
    rf_grid_arrive_wait(ws, phase, G, true);
    __syncthreads();
}
#define RFW_CLK 32              // [20] wall-clock stamps of block 0 (100 MHz ticks)
// timing stamps of block 0: kept in LDS and written out once at the end of the kernel (a global store per stamp would
// put a store round trip in front of the next block barrier: the stamps themselves cost ~1.5 us each that way)
__shared__ unsigned rf_clk[32];
__device__ __forceinline__ void rf_stamp(unsigned* ws, int k) {
    if (blockIdx.x == 0 && threadIdx.x == 0) rf_clk[k] = (unsigned)wall_clock64();
}
// (Raw buffer loads with the sc1 cache-policy bit were tried for the bulk cross-block reads -- they pipeline better
// than a chain of atomic loads -- but they returned STALE histogram words on every second launch (the buffer the
// previous launch of the same parity had read): only the atomic loads below are coherent across the XCDs' L2s.)
__device__ __forceinline__ void rf_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned rf_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Wide loads / stores with the cache policy of the relaxed agent-scope atomics above (sc1: loads are not served from
// this CU's L1, stores are written through), issued TOGETHER and waited for once.  (The compiler schedules chains of
// __hip_atomic_load two at a time with a full wait in between: a thread that needs 8 words would pay 4 round trips.)
typedef unsigned int rf_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void rf_ld1x4(const unsigned* p0, const unsigned* p1, const unsigned* p2, const unsigned* p3,
                                         unsigned& a, unsigned& b, unsigned& c, unsigned& d) {
    asm volatile("global_load_dword %0, %4, off sc1\n\tglobal_load_dword %1, %5, off sc1\n\t"
                 "global_load_dword %2, %6, off sc1\n\tglobal_load_dword %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}
__device__ __forceinline__ void rf_ld1x8(const unsigned* const (&p)[8], unsigned (&v)[8]) {
    asm volatile("global_load_dword %0, %8, off sc1\n\tglobal_load_dword %1, %9, off sc1\n\t"
                 "global_load_dword %2, %10, off sc1\n\tglobal_load_dword %3, %11, off sc1\n\t"
                 "global_load_dword %4, %12, off sc1\n\tglobal_load_dword %5, %13, off sc1\n\t"
                 "global_load_dword %6, %14, off sc1\n\tglobal_load_dword %7, %15, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]) : "memory");
}
typedef unsigned int rf_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void rf_ld2x8(const unsigned* base, long stride, rf_u2 (&v)[8]) {      // base + k * stride, k = 0..7 (8-byte aligned)
    const unsigned *p0 = base, *p1 = base + stride, *p2 = base + 2 * stride, *p3 = base + 3 * stride, *p4 = base + 4 * stride,
                   *p5 = base + 5 * stride, *p6 = base + 6 * stride, *p7 = base + 7 * stride;
    asm volatile("global_load_dwordx2 %0, %8, off sc1\n\tglobal_load_dwordx2 %1, %9, off sc1\n\t"
                 "global_load_dwordx2 %2, %10, off sc1\n\tglobal_load_dwordx2 %3, %11, off sc1\n\t"
                 "global_load_dwordx2 %4, %12, off sc1\n\tglobal_load_dwordx2 %5, %13, off sc1\n\t"
                 "global_load_dwordx2 %6, %14, off sc1\n\tglobal_load_dwordx2 %7, %15, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7) : "memory");
}
__device__ __forceinline__ void rf_ld1x12(const unsigned* const (&p)[12], unsigned (&v)[12]) {
    asm volatile("global_load_dword %0, %12, off sc1\n\tglobal_load_dword %1, %13, off sc1\n\t"
                 "global_load_dword %2, %14, off sc1\n\tglobal_load_dword %3, %15, off sc1\n\t"
                 "global_load_dword %4, %16, off sc1\n\tglobal_load_dword %5, %17, off sc1\n\t"
                 "global_load_dword %6, %18, off sc1\n\tglobal_load_dword %7, %19, off sc1\n\t"
                 "global_load_dword %8, %20, off sc1\n\tglobal_load_dword %9, %21, off sc1\n\t"
                 "global_load_dword %10, %22, off sc1\n\tglobal_load_dword %11, %23, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]),
                   "=&v"(v[8]), "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11])
                 : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]), "v"(p[8]), "v"(p[9]),
                   "v"(p[10]), "v"(p[11]) : "memory");
}
__device__ __forceinline__ void rf_st16(void* p, rf_u4 v) {      // (drained by rf_drain_stores() before the barrier)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");   // (s_nop: the data registers of a wide store must not be rewritten in the next slot)
}
__device__ __forceinline__ void rf_st8(void* p, unsigned lo, unsigned hi) {
    rf_u2 v; v[0] = lo; v[1] = hi;
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void rf_drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
struct RfShared {
    unsigned hist[RF_BINS];
    unsigned wsum[16];
    unsigned rank[RF_MAXSLOT], sbin[RF_MAXSLOT], srin[RF_MAXSLOT], skey[RF_MAXSLOT], scnt[RF_MAXSLOT];
    unsigned whist[RF_MAXSLOT][256];     // one 8-bit radix histogram per selecting wave
    float gamma[3];
    int nd; unsigned dbin[RF_MAXSLOT], doff[RF_MAXSLOT], dcnt[RF_MAXSLOT], dblk[RF_MAXSLOT], dbase[RF_MAXSLOT];
    int sdl[RF_MAXSLOT];                 // slot -> index of its candidate list
    unsigned red[2][16];
    unsigned sel_digit, sel_k;
    short invy[1024], invx[1024];
    float thr[3];
    int twin[RF_MAXSLOT];                // slot -> first slot with the same order statistic
    unsigned ncand;
    unsigned ptot[32];                   // gather: totals of the 64-pair groups (prefix over blocks of the per-list counts)
    // selection fused with the gather: key range of every list (bin edges), first-digit results, the collected sub-buckets
    unsigned lmn[RF_MAXSLOT], lnb[RF_MAXSLOT], lsh[RF_MAXSLOT];     // list: smallest key, key bits, first-digit shift
    unsigned sd0[RF_MAXSLOT], sk1[RF_MAXSLOT], scount[RF_MAXSLOT];  // slot: first digit, rank inside it, collected keys
    int bad;
};

// exclusive scan of the 2048-bin histogram: thread t owns bins 2t, 2t+1; returns the exclusive prefix of bin 2t
__device__ __forceinline__ unsigned rf_scan2048(RfShared& S, unsigned h0, unsigned h1) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned v = h0 + h1, x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = __shfl_up(x, o, 64);
        if (lane >= o) x += u;
    }
    if (lane == 63) S.wsum[wave] = x;
    __syncthreads();
    unsigned base = 0;
    for (int w2 = 0; w2 < wave; ++w2) base += S.wsum[w2];
    return base + x - v;
}

// k-th smallest (0-based) of n keys (LDS array or global list), exact.  All threads must call; result in S.sel_k.
template <bool INLDS>
__device__ unsigned rf_select(RfShared& S, const unsigned* lkeys, const float* gvals, unsigned n, unsigned k) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    auto key = [&](unsigned i) -> unsigned { return INLDS ? lkeys[i] : f32_key(__uint_as_float(rf_ld((const unsigned*)gvals + i))); };
    unsigned mn = 0xffffffffu, mx = 0u;
    for (unsigned i = t; i < n; i += RF_T) { const unsigned v = key(i); mn = min(mn, v); mx = max(mx, v); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = min(mn, (unsigned)__shfl_xor(mn, o, 64)); mx = max(mx, (unsigned)__shfl_xor(mx, o, 64)); }
    if (lane == 0) { S.red[0][wave] = mn; S.red[1][wave] = mx; }
    __syncthreads();
    mn = S.red[0][0]; mx = S.red[1][0];
    for (int w2 = 1; w2 < 16; ++w2) { mn = min(mn, S.red[0][w2]); mx = max(mx, S.red[1][w2]); }
    __syncthreads();
    const unsigned range = mx - mn;
    const int nbits = range ? 32 - __clz(range) : 0;
    const int passes = (nbits + 10) / 11;
    unsigned prefix = 0;          // high digits of (key - mn) found so far
    for (int p = 0; p < passes; ++p) {
        const int shift = 11 * (passes - 1 - p);
        S.hist[2 * t] = 0; S.hist[2 * t + 1] = 0;
        __syncthreads();
        for (unsigned i = t; i < n; i += RF_T) {
            const unsigned v = key(i) - mn;
            // (v >> shift) >> 11 avoids an undefined 32-bit shift when shift + 11 == 33
            if (((v >> shift) >> 11) == prefix) atomicAdd(&S.hist[(v >> shift) & 2047u], 1u);
        }
        __syncthreads();
        const unsigned h0 = S.hist[2 * t], h1 = S.hist[2 * t + 1];
        const unsigned ex = rf_scan2048(S, h0, h1);
        if (k >= ex && k < ex + h0) { S.sel_digit = 2 * t; S.sel_k = k - ex; }
        else if (k >= ex + h0 && k < ex + h0 + h1) { S.sel_digit = 2 * t + 1; S.sel_k = k - ex - h0; }
        __syncthreads();
        prefix = (prefix << 11) | S.sel_digit;
        k = S.sel_k;
        __syncthreads();
    }
    return mn + prefix;
}

// k-th smallest (0-based) of the n keys at lkeys[0..n) (LDS), computed by ONE wave with a private 256-bin histogram:
// no block barriers, so the (<= 6) order statistics are selected concurrently by different waves.
__device__ unsigned rf_wave_select(unsigned* __restrict__ hist, const unsigned* __restrict__ lkeys, unsigned n, unsigned k) {
    const int lane = threadIdx.x & 63;
    unsigned mn = 0xffffffffu, mx = 0u;
    for (unsigned i0 = 0; i0 < n; i0 += 256) {          // 4 independent LDS reads in flight per lane
        unsigned v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const unsigned i = i0 + 64 * j + lane; v[j] = i < n ? lkeys[i] : lkeys[0]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { mn = min(mn, v[j]); mx = max(mx, v[j]); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = min(mn, (unsigned)__shfl_xor(mn, o, 64)); mx = max(mx, (unsigned)__shfl_xor(mx, o, 64)); }
    const unsigned range = mx - mn;
    const int nbits = range ? 32 - __clz(range) : 0;
    const int passes = (nbits + 7) / 8;
    unsigned prefix = 0;
    for (int p = 0; p < passes; ++p) {
        const int shift = 8 * (passes - 1 - p);
#pragma unroll
        for (int j = 0; j < 4; ++j) hist[lane * 4 + j] = 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (unsigned i0 = 0; i0 < n; i0 += 256) {
            unsigned v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const unsigned i = i0 + 64 * j + lane; v[j] = i < n ? lkeys[i] - mn : 0xffffffffu; }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i0 + 64 * j + lane < n && ((v[j] >> shift) >> 8) == prefix) atomicAdd(&hist[(v[j] >> shift) & 255u], 1u);   // no-return ds_add
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        unsigned h[4], loc = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { h[j] = hist[lane * 4 + j]; loc += h[j]; }
        unsigned x = loc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned u = __shfl_up(x, o, 64);
            if (lane >= o) x += u;
        }
        const unsigned ex = x - loc;
        const bool mine = k >= ex && k < x;
        unsigned digit = 0, kk = 0;
        if (mine) {
            unsigned run = ex;
            bool done = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!done && k < run + h[j]) { digit = lane * 4 + j; kk = k - run; done = true; }
                run += h[j];
            }
        }
        const unsigned long long m = __ballot(mine);
        const int src = __ffsll((long long)m) - 1;
        digit = __shfl(digit, src, 64);
        k = __shfl(kk, src, 64);
        prefix = (prefix << 8) | digit;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    return mn + prefix;
}

// k-th smallest (0-based) of the n <= 64 * RF_KW keys at list[0..n) (LDS) by ONE wave, keys held in REGISTERS (RF_KW per
// lane), 8-bit radix passes over a wave-private 256-bin LDS histogram.  No block barriers and no LDS re-reads of the keys:
// the six order statistics of a launch are selected concurrently by six waves (the 2-waves-per-statistic block form below
// spends its time in ~10 block barriers).  LDS operations of one wave execute in order, so clear -> atomics -> read back
// needs no explicit waits beyond the one before the values are used.
#define RF_KW 8
__device__ unsigned rf_wave_select_reg(unsigned* __restrict__ hist, const unsigned* __restrict__ list, unsigned n, unsigned k) {
    const int lane = threadIdx.x & 63;
    unsigned key[RF_KW];
    unsigned mn = 0xffffffffu, mx = 0u;
#pragma unroll
    for (int j = 0; j < RF_KW; ++j) {
        const unsigned i = (unsigned)lane + 64u * j;
        key[j] = 0xffffffffu;
        if (64u * j < n) {                       // wave-uniform
            key[j] = i < n ? list[i] : 0xffffffffu;
            if (i < n) { mn = min(mn, key[j]); mx = max(mx, key[j]); }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = min(mn, (unsigned)__shfl_xor(mn, o, 64)); mx = max(mx, (unsigned)__shfl_xor(mx, o, 64)); }
    const unsigned range = mx - mn;
    const int nbits = range ? 32 - __clz(range) : 0;
    const int passes = (nbits + 7) / 8;
    unsigned prefix = 0;
    for (int p = 0; p < passes; ++p) {
        const int shift = 8 * (passes - 1 - p);
        *(uint4*)(hist + lane * 4) = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int j = 0; j < RF_KW; ++j) {
            if (64u * j < n) {                   // wave-uniform
                const unsigned v = key[j] - mn;
                if ((unsigned)lane + 64u * j < n && ((v >> shift) >> 8) == prefix) atomicAdd(&hist[(v >> shift) & 255u], 1u);   // no-return ds_add
            }
        }
        const uint4 hq = *(const uint4*)(hist + lane * 4);
        const unsigned h[4] = {hq.x, hq.y, hq.z, hq.w};
        const unsigned loc = (h[0] + h[1]) + (h[2] + h[3]);
        unsigned x = loc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned u = __shfl_up(x, o, 64);
            if (lane >= o) x += u;
        }
        const unsigned ex = x - loc;
        const bool mine = k >= ex && k < x;
        unsigned digit = 0, kk = 0;
        if (mine) {
            unsigned run = ex;
            bool done = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!done && k < run + h[j]) { digit = lane * 4 + j; kk = k - run; done = true; }
                run += h[j];
            }
        }
        const unsigned long long m = __ballot(mine);
        const int src = __ffsll((long long)m) - 1;
        digit = __shfl(digit, src, 64);
        k = __shfl(kk, src, 64);
        prefix = (prefix << 8) | digit;
    }
    return mn + prefix;
}

// Block-wide selection of all (<= 6) order statistics at once: slot s is served by waves 2s, 2s+1 (128 lanes), each
// lane keeps its <= RF_KREG keys of the slot's candidate list in REGISTERS, so a radix pass is pure VALU + no-return
// LDS atomics (the one-wave LDS-streaming version spent ~10 us per call in read -> atomic round trips).  Two block
// barriers per 8-bit pass; every wave of a slot scans the slot's 256-bin histogram redundantly.  Requires n <= 128 * RF_KREG.
#define RF_KREG 32
__device__ void rf_block_select(RfShared& S, const unsigned* __restrict__ lkeys, int nslot, bool active) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int slot = wave >> 1, sub = (wave & 1) * 64 + lane;          // lane index 0..127 inside the slot's group
    bool work = active && slot < nslot;
    if (work)
        for (int u = 0; u < slot; ++u) work = work && !(S.sbin[u] == S.sbin[slot] && S.srin[u] == S.srin[slot]);   // first twin only
    unsigned n = 0, k = 0;
    const unsigned* list = lkeys;
    if (work) { const int d = S.sdl[slot]; n = S.dcnt[d]; k = S.srin[slot]; list = lkeys + S.doff[d]; }
    unsigned key[RF_KREG];
    unsigned mn = 0xffffffffu, mx = 0u;
#pragma unroll
    for (int j = 0; j < RF_KREG; ++j) {
        const unsigned i = sub + 128u * j;
        key[j] = 0xffffffffu;
        if (128u * j < n) {                      // wave-uniform: a typical list (~600 keys) fills 5 of the 32 register slots
            key[j] = i < n ? list[i] : 0xffffffffu;
            if (i < n) { mn = min(mn, key[j]); mx = max(mx, key[j]); }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = min(mn, (unsigned)__shfl_xor(mn, o, 64)); mx = max(mx, (unsigned)__shfl_xor(mx, o, 64)); }
    if (lane == 0) { S.red[0][wave] = mn; S.red[1][wave] = mx; }
    __syncthreads();
    mn = min(S.red[0][wave & ~1], S.red[0][wave | 1]);
    mx = max(S.red[1][wave & ~1], S.red[1][wave | 1]);
    const unsigned range = work ? mx - mn : 0u;
    const int nbits = range ? 32 - __clz(range) : 0;
    const int passes = (nbits + 7) / 8;
    // all slots run the same number of block barriers: the block-wide maximum of `passes`
    if (lane == 0) S.wsum[wave] = (unsigned)passes;
    __syncthreads();
    int maxp = 0;
    for (int w2 = 0; w2 < 16; ++w2) maxp = max(maxp, (int)S.wsum[w2]);
    if (blockIdx.x == 0 && t == 0) { rf_clk[18] = (unsigned)wall_clock64(); rf_clk[28] = (unsigned)maxp; }
    unsigned* hist = S.whist[slot < RF_MAXSLOT ? slot : 0];
    unsigned prefix = 0;
    for (int p = 0; p < maxp; ++p) {
        const int pp = p - (maxp - passes);              // this slot's pass index (negative: idle rounds first)
        const bool on = work && pp >= 0;
        const int shift = on ? 8 * (passes - 1 - pp) : 0;
        if (on) { hist[sub] = 0; hist[sub + 128] = 0; }
        __syncthreads();
        if (on) {
#pragma unroll
            for (int j = 0; j < RF_KREG; ++j) {
                if (128u * j < n) {              // wave-uniform
                    const unsigned v = key[j] - mn;
                    if (sub + 128u * j < n && ((v >> shift) >> 8) == prefix) atomicAdd(&hist[(v >> shift) & 255u], 1u);
                }
            }
        }
        __syncthreads();
        if (on) {
            unsigned h[4], loc = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) { h[j] = hist[lane * 4 + j]; loc += h[j]; }
            unsigned x = loc;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned u = __shfl_up(x, o, 64);
                if (lane >= o) x += u;
            }
            const unsigned ex = x - loc;
            const bool mine = k >= ex && k < x;
            unsigned digit = 0, kk = 0;
            if (mine) {
                unsigned run = ex;
                bool done = false;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (!done && k < run + h[j]) { digit = lane * 4 + j; kk = k - run; done = true; }
                    run += h[j];
                }
            }
            const unsigned long long m = __ballot(mine);
            const int src = __ffsll((long long)m) - 1;
            digit = __shfl(digit, src, 64);
            k = __shfl(kk, src, 64);
            prefix = (prefix << 8) | digit;
        }
        __syncthreads();      // histogram fully read before the next pass clears it
        if (blockIdx.x == 0 && t == 0 && p < 4) rf_clk[19 + p] = (unsigned)wall_clock64();
    }
    if (work && (wave & 1) == 0 && lane == 0) S.skey[slot] = mn + prefix;
}

template <int CT>
__global__ __launch_bounds__(RF_T, 1) void k_reliability_fused(RfArgs A) {
    extern __shared__ __attribute__((aligned(16))) float dyn[];   // phase A: corner logits [4][CT][RF_CELLS]; P: sorted run; G/D: offsets + candidate keys
    __shared__ RfShared S;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int G = gridDim.x, b = blockIdx.x;
    unsigned* ws = A.ws;
    unsigned* tot = ws + RFW_TOT + ((A.epoch & 1u) * 8u + (unsigned)(blockIdx.x & 7)) * RF_BINS;    // my XCD's copy (this parity)
    const long HW = (long)A.H * A.W;
    rf_stamp(ws, 0);
    unsigned dbg_t0 = 0, dbg_arr = 0, dbg_rel = 0;      // (debug: start / barrier arrival / release time of every block)
    if (t == 0) dbg_t0 = (unsigned)wall_clock64();
    if (t > 0 && t < 32) rf_clk[t] = 0;
    if (b == 0 && t == 7) rf_st(ws + RFW_NKEPT, 0u);
    S.hist[2 * t] = 0; S.hist[2 * t + 1] = 0;
    S.invy[t] = -1; S.invx[t] = -1;
    __syncthreads();
    if (t < A.hm) S.invy[nearest_src(t, A.ny, A.H)] = (short)t;
    if (t < A.wm) S.invx[nearest_src(t, A.nx, A.W)] = (short)t;
    // ---------------------------------------------------------------- phase A: entropies of my cells
    const long ncell = (long)A.B * A.h * A.w;
    const long per = (ncell + G - 1) / G;
    const long c0 = (long)b * per, c1 = min(ncell, c0 + per);
    const int nmain = (int)(per / RF_CELLS);                        // full 256-cell iterations at 4 px / thread (<= 2)
    const int rem = (int)(per - (long)nmain * RF_CELLS);
    const bool tail1 = rem > 0 && rem <= 64;                          // remainder at 1 px / thread (16 threads per cell)
    const int nit4 = nmain + ((rem > 0 && !tail1) ? 1 : 0);           // host guarantees nit4 <= RF_NIT
    float er[RF_NIT][4], er1 = __uint_as_float(0x7fc00000u);
    unsigned labr[RF_NIT], lab1 = 0;   // label bytes of the item's pixels
    int pbase[RF_NIT], pb1 = -1;       // (n << 20 | oy << 10 | ox0) of the item's first pixel, -1: no item
    // (Shifting by the cell's largest corner logit instead of the per-pixel maximum would save the max pass -- the
    // entropy is shift invariant -- but log(s) and t/s then cancel at magnitude |shift - max|: measured 4e-6 instead of
    // < 1e-6 on the entropy.  Parity first: the exact maximum is kept.)
    auto cell_geom = [&](long q, int& n, int& ci, int& cj) {
        cj = (int)(q % A.w);
        const long t0 = q / A.w;
        ci = (int)(t0 % A.h); n = (int)(t0 / A.h);
    };
    // The corner logits of my cells, PIXEL-MAJOR in LDS.  A cell (n, ci, cj) IS low-resolution pixel q = (n h + ci) w + cj
    // (H - 1 = 4 (h - 1): the up-sampling grid has one low-resolution pixel per 4x4 cell corner), its four corners are the
    // pixels q, q + dx, q + dy w, q + dy w + dx (dx / dy = 0 on the last column / row: align_corners clamping), so my
    // cells [c0, c1) need the ONE contiguous pixel span [c0, c1 + w + 1).  For channels-last logits that span is one
    // contiguous run of floats: it is staged with coalesced 16-byte loads (~2 per thread).  (Round 2 staged corner by
    // corner: 19 dword loads per thread whose lanes sit 76 bytes apart -- ~20 x the cache-line requests for the same
    // bytes, ~10 of phase A's 21 us.)  Reading it back with stride CT (odd) is conflict-free.
    float* T = dyn;
    const long pix_lo = c0, pix_hi = min(ncell, c1 + A.w + 1);
    int tshift = 0;                                                   // T[tshift + (pixel - pix_lo) * CT + c]
    if (A.sc == 1 && A.sw == CT && A.sh == (long)A.w * CT && A.sn == (long)A.h * A.w * CT) {
        const float* first = A.in + pix_lo * CT;
        const float* lo16 = (const float*)((uintptr_t)first & ~(uintptr_t)15);
        tshift = (int)(first - lo16);
        const int nfl = tshift + (int)((pix_hi - pix_lo) * CT);
        for (int k = 4 * t; k < nfl; k += 4 * RF_T) {
            if (k + 3 < nfl) *(float4*)(T + k) = *(const float4*)(lo16 + k);
            else for (int kk = k; kk < nfl; ++kk) T[kk] = lo16[kk];
        }
    } else {      // any other layout: the same tile, filled element by element
        const int nel = (int)((pix_hi - pix_lo) * CT);
        for (int k = t; k < nel; k += RF_T) {
            const long px = pix_lo + k / CT;
            const int c = k % CT;
            int n, ci, cj;
            cell_geom(px, n, ci, cj);
            T[k] = A.in[n * A.sn + ci * A.sh + cj * A.sw + c * A.sc];
        }
    }
    // the labels of ALL my pixels are requested now too: one exposed memory latency for the whole phase
    long long lraw[RF_NIT][4], l1raw = (long long)A.ignore;
#pragma unroll
    for (int it = 0; it < RF_NIT; ++it) {
        pbase[it] = -1;
        labr[it] = 0;
#pragma unroll
        for (int a = 0; a < 4; ++a) { er[it][a] = __uint_as_float(0x7fc00000u); lraw[it][a] = (long long)A.ignore; }
        if (it >= nit4) continue;                                     // block-uniform
        const int cl = t & (RF_CELLS - 1), part = t >> 8;
        const long q = c0 + (long)it * RF_CELLS + cl;
        if (q < c1) {
            int n, ci, cj;
            cell_geom(q, n, ci, cj);
            const int oy = ci * 4 + part, ox0 = cj * 4;
            if (oy < A.H) {
                pbase[it] = (n << 20) | (oy << 10) | ox0;
                const long p0 = ((long)n * A.H + oy) * A.W + ox0;
                const int nx = min(4, A.W - ox0);
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    if (a < nx) lraw[it][a] = A.label_u[p0 + a];
            }
        }
    }
    if (tail1) {
        const int cl = t >> 4, a = t & 3, row = (t >> 2) & 3;
        const long q = c0 + (long)nmain * RF_CELLS + cl;
        if (cl < rem && q < c1) {
            int n, ci, cj;
            cell_geom(q, n, ci, cj);
            const int oy = ci * 4 + row, ox = cj * 4 + a;
            if (oy < A.H && ox < A.W) {
                pb1 = (n << 20) | (oy << 10) | ox;
                l1raw = A.label_u[((long)n * A.H + oy) * A.W + ox];
            }
        }
    }
    __syncthreads();
    rf_stamp(ws, 18);
    unsigned vmask[RF_NIT];
#pragma unroll
    for (int it = 0; it < RF_NIT; ++it) {
        vmask[it] = 0;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            labr[it] |= ((unsigned)lraw[it][a] & 255u) << (8 * a);
            vmask[it] |= (lraw[it][a] != (long long)A.ignore ? 1u : 0u) << a;
        }
    }
    const bool valid1 = l1raw != (long long)A.ignore;
    lab1 = (unsigned)l1raw & 255u;
    // LDS offsets of the four corners of low-resolution cell q
    auto corners = [&](long q, int ci, int cj, int& o00, int& o01, int& o10, int& o11) {
        const int dx = cj < A.w - 1 ? CT : 0, dy = ci < A.h - 1 ? A.w * CT : 0;
        o00 = tshift + (int)(q - pix_lo) * CT;
        o01 = o00 + dx; o10 = o00 + dy; o11 = o10 + dx;
    };
    // CT <= 19: the entropies are COMPUTED column-wise and OWNED row-wise.  Thread (cell, part) owns the four pixels of
    // output row `part` of its cell (labels, sort keys, 16-byte stores along x), but evaluating the bilinear form for a
    // row costs 4 horizontal + 2 vertical operations per pixel and class, while a COLUMN shares its horizontal
    // interpolation between the four rows: 1 + 2.  So thread (cell, col) evaluates column `col` of its cell (two rows at a
    // time, the up-sampled logits of the pair kept in registers between the max pass and the exp pass) and hands the four
    // entropies to the row owners through a 17-word-per-cell LDS exchange (odd stride: conflict-free both ways).  Same
    // operations on the same operands as the row form -- bit-identical entropies -- at 5 instead of 7 VALU operations per
    // pixel and class in the first pass.
    constexpr bool XPOSE = CT <= 19;
    float* X = dyn + (((size_t)(per + A.w + 1) * CT + 8 + 3) & ~(size_t)3);     // [RF_NIT][RF_CELLS * 17] behind the tile
    if constexpr (XPOSE) {
#pragma unroll
        for (int it = 0; it < RF_NIT; ++it) {
            if (it >= nit4) continue;                                 // block-uniform
            const int cl = t & (RF_CELLS - 1), col = t >> 8;
            const long q = c0 + (long)it * RF_CELLS + cl;
            if (q < c1) {
                int n, ci, cj;
                cell_geom(q, n, ci, cj);
                int o00, o01, o10, o11;
                corners(q, ci, cj, o00, o01, o10, o11);
                const AcCoord cx = ac_coord(min(cj * 4 + col, A.W - 1), A.sx, A.w);    // (columns / rows past the edge: clamped,
                float ly0[4], ly1[4];                                                  //  computed along, never read)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const AcCoord cy = ac_coord(min(ci * 4 + r, A.H - 1), A.sy, A.h);
                    ly0[r] = cy.l0; ly1[r] = cy.l1;
                }
                float* xo = X + (size_t)it * (RF_CELLS * 17) + cl * 17 + col;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    float z0[CT], z1[CT], m0 = -INFINITY, m1 = -INFINITY;
#pragma unroll
                    for (int c = 0; c < CT; ++c) {
                        const float v00 = T[o00 + c], v01 = T[o01 + c], v10 = T[o10 + c], v11 = T[o11 + c];
                        const float tp = __fmaf_rn(cx.l0, v00, __fmul_rn(cx.l1, v01));
                        const float bt = __fmaf_rn(cx.l0, v10, __fmul_rn(cx.l1, v11));
                        z0[c] = __fmaf_rn(ly0[2 * hf], tp, __fmul_rn(ly1[2 * hf], bt));
                        z1[c] = __fmaf_rn(ly0[2 * hf + 1], tp, __fmul_rn(ly1[2 * hf + 1], bt));
                        m0 = fmaxf(m0, z0[c]);
                        m1 = fmaxf(m1, z1[c]);
                    }
                    float s0 = 0.f, s1 = 0.f, u0 = 0.f, u1 = 0.f;
#pragma unroll
                    for (int c = 0; c < CT; ++c) {
                        const float d0 = z0[c] - m0, d1 = z1[c] - m1;
                        const float e0 = rf_exp_neg(d0), e1 = rf_exp_neg(d1);
                        s0 += e0; u0 += e0 * d0;
                        s1 += e1; u1 += e1 * d1;
                    }
                    xo[(2 * hf) * 4] = logf(s0) - u0 / s0;
                    xo[(2 * hf + 1) * 4] = logf(s1) - u1 / s1;
                }
            }
        }
        __syncthreads();
        rf_stamp(ws, 19);
    }
#pragma unroll
    for (int it = 0; it < RF_NIT; ++it) {
        if (it >= nit4) continue;                                     // block-uniform
        if (pbase[it] >= 0) {
            const int cl = t & (RF_CELLS - 1);
            const int ox0 = pbase[it] & 1023;
            float eraw[4];
            if constexpr (XPOSE) {
                const float* xi = X + (size_t)it * (RF_CELLS * 17) + cl * 17 + (t >> 8) * 4;
#pragma unroll
                for (int a = 0; a < 4; ++a) eraw[a] = xi[a];
            } else {      // more classes: the 2 x CT register copy would spill -> row form, the bilinear form evaluated in both passes
                const long q = c0 + (long)it * RF_CELLS + cl;
                const int oy = (pbase[it] >> 10) & 1023;
                int o00, o01, o10, o11;
                corners(q, oy >> 2, ox0 >> 2, o00, o01, o10, o11);
                const AcCoord cy = ac_coord(oy, A.sy, A.h);
                float lx0[4], lx1[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const AcCoord cx = ac_coord(min(ox0 + a, A.W - 1), A.sx, A.w);
                    lx0[a] = cx.l0; lx1[a] = cx.l1;
                }
                float s[4], tt[4], m[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) { m[a] = -INFINITY; s[a] = 0.f; tt[a] = 0.f; }
#pragma unroll 4
                for (int c = 0; c < CT; ++c) {
                    const float v00 = T[o00 + c], v01 = T[o01 + c], v10 = T[o10 + c], v11 = T[o11 + c];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const float top = __fmaf_rn(lx0[a], v00, __fmul_rn(lx1[a], v01));
                        const float bot = __fmaf_rn(lx0[a], v10, __fmul_rn(lx1[a], v11));
                        m[a] = fmaxf(m[a], __fmaf_rn(cy.l0, top, __fmul_rn(cy.l1, bot)));
                    }
                }
#pragma unroll 4
                for (int c = 0; c < CT; ++c) {
                    const float v00 = T[o00 + c], v01 = T[o01 + c], v10 = T[o10 + c], v11 = T[o11 + c];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const float top = __fmaf_rn(lx0[a], v00, __fmul_rn(lx1[a], v01));
                        const float bot = __fmaf_rn(lx0[a], v10, __fmul_rn(lx1[a], v11));
                        const float z = __fmaf_rn(cy.l0, top, __fmul_rn(cy.l1, bot)) - m[a];
                        const float e = rf_exp_neg(z);
                        s[a] += e;
                        tt[a] += e * z;
                    }
                }
#pragma unroll
                for (int a = 0; a < 4; ++a) eraw[a] = logf(s[a]) - tt[a] / s[a];
            }
            const int nx = min(4, A.W - ox0);
#pragma unroll
            for (int a = 0; a < 4; ++a)
                if (a < nx) {
                    const bool valid = (vmask[it] >> a) & 1u;
                    const float e = valid ? eraw[a] : __uint_as_float(0x7fc00000u);
                    er[it][a] = e;
                    if (valid) atomicAdd(&S.hist[rf_bin(e, A.bin_scale)], 1u);
                }
        }
    }
    if (tail1 && pb1 >= 0) {    // the <= 64 remaining cells: one pixel per thread so that the tail costs ~1/16 of a full iteration
        const int cl = t >> 4;
        const long q = c0 + (long)nmain * RF_CELLS + cl;
        const int oy = (pb1 >> 10) & 1023, ox = pb1 & 1023;
        int o00, o01, o10, o11;
        corners(q, oy >> 2, ox >> 2, o00, o01, o10, o11);
        const AcCoord cy = ac_coord(oy, A.sy, A.h), cx = ac_coord(ox, A.sx, A.w);
        float m = -INFINITY, sm = 0.f, tt = 0.f;
#pragma unroll 4
        for (int c = 0; c < CT; ++c) {
            const float top = __fmaf_rn(cx.l0, T[o00 + c], __fmul_rn(cx.l1, T[o01 + c]));
            const float bot = __fmaf_rn(cx.l0, T[o10 + c], __fmul_rn(cx.l1, T[o11 + c]));
            m = fmaxf(m, __fmaf_rn(cy.l0, top, __fmul_rn(cy.l1, bot)));
        }
#pragma unroll 4
        for (int c = 0; c < CT; ++c) {
            const float top = __fmaf_rn(cx.l0, T[o00 + c], __fmul_rn(cx.l1, T[o01 + c]));
            const float bot = __fmaf_rn(cx.l0, T[o10 + c], __fmul_rn(cx.l1, T[o11 + c]));
            const float z = __fmaf_rn(cy.l0, top, __fmul_rn(cy.l1, bot)) - m;
            const float e = rf_exp_neg(z);
            sm += e;
            tt += e * z;
        }
        float e = logf(sm) - tt / sm;
        e = valid1 ? e : __uint_as_float(0x7fc00000u);
        er1 = e;
        if (valid1) atomicAdd(&S.hist[rf_bin(e, A.bin_scale)], 1u);
    }
    __syncthreads();
    rf_stamp(ws, 10);
    {   // ------------------------------------------------------------ P: publish my values sorted by bin + the bin offsets
        float* srt = dyn;                                    // (the corner logits are no longer needed)
        const unsigned h0 = S.hist[2 * t], h1 = S.hist[2 * t + 1];
        const unsigned ex = rf_scan2048(S, h0, h1);
        unsigned* slab = ws + RFW_SLAB + (size_t)b * RF_SLABW;
        rf_st8(slab + 2 * t, ex, ex + h0);                   // exclusive prefix = offset of bins 2t, 2t+1 in my sorted run
        if (t == RF_T - 1) { rf_st8(slab + RF_BINS, ex + h0 + h1, 0u); S.red[0][0] = ex + h0 + h1; }
        // bin counts -> the totals (fire-and-forget atomics spread over up to 2048 addresses: no same-address serialisation)
        if (h0) __hip_atomic_fetch_add(tot + 2 * t, h0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (h1) __hip_atomic_fetch_add(tot + 2 * t + 1, h1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        S.hist[2 * t] = ex; S.hist[2 * t + 1] = ex + h0;     // the histogram becomes the cursors of the counting sort
        rf_stamp(ws, 11);
        __syncthreads();
        auto place = [&](float e) {
            if (e == e) srt[atomicAdd(&S.hist[rf_bin(e, A.bin_scale)], 1u)] = e;
        };
#pragma unroll
        for (int it = 0; it < RF_NIT; ++it)
#pragma unroll
            for (int a = 0; a < 4; ++a) place(er[it][a]);
        place(er1);
        __syncthreads();
        rf_stamp(ws, 12);
        const unsigned nmine = S.red[0][0];
        float* run = A.cand + (size_t)b * RF_PXMAX;
        for (unsigned i = 4u * t; i < nmine; i += 4u * RF_T) rf_st16(run + i, *(const rf_u4*)(srt + i));   // (tail quad: <= 3 unused words)
        rf_stamp(ws, 13);
        rf_drain_stores();
    }
    rf_stamp(ws, 1);
    __syncthreads();                         // every wave has drained its published stores
    if (t == 0) dbg_arr = (unsigned)wall_clock64();
    if (wave == 0) {
        rf_grid_arrive_wait(ws, 2u * A.epoch + 1u, G, A.fences != 0);
    } else if (A.nspec > 1) {
        // ------------------------------------------------------------ label-only outputs (labeled-half masks, class bits of
        // both halves) in the SHADOW of the device-wide barrier: nothing depends on them and they depend on nothing, so
        // waves 1..15 gather them while wave 0 waits for the other blocks (they used to open the kernel: ~2 us of exposed
        // gather latency in front of phase A)
        const long lowplane = (long)A.hm * A.wm, nlow = (long)2 * A.B * lowplane;
        for (long q = (long)b * (RF_T - 64) + (t - 64); q < nlow; q += (long)G * (RF_T - 64)) {
            const int x = (int)(q % A.wm);
            const long t1 = q / A.wm;
            const int y = (int)(t1 % A.hm), n = (int)(t1 / A.hm);
            const long src = (long)nearest_src(y, A.ny, A.H) * A.W + nearest_src(x, A.nx, A.W);
            const long long* lab = n < A.B ? A.label_l : A.label_u;
            const int bi = n < A.B ? n : n - A.B;
            if (n < A.B) {
                const float v = lab[bi * HW + src] != A.ignore ? 1.f : 0.f;
                A.low_mask[q] = v;
                A.high_mask[q] = v;
            }
            unsigned bits = 0;
            if (bi == 0 && lab[src] != A.ignore)
                for (int bb = 0; bb < A.B; ++bb) {
                    const long long l = lab[bb * HW + src];
                    bits |= 1u << (l == A.ignore ? 0 : (int)l);
                }
            A.lbits[q] = bits;
        }
    }
    __syncthreads();
    if (t == 0) dbg_rel = (unsigned)wall_clock64();
    rf_stamp(ws, 2);
    // ---------------------------------------------------------------- phase C: totals -> ranks -> bins -> list table
    unsigned h0 = 0, h1 = 0;
    {   // bins 2t, 2t+1 of the eight per-XCD copies
        const unsigned* tp = ws + RFW_TOT + (A.epoch & 1u) * 8u * RF_BINS + 2 * t;
        rf_u2 q[8];
        rf_ld2x8(tp, RF_BINS, q);                            // ONE round trip for the 16 words
#pragma unroll
        for (int k = 0; k < 8; ++k) { h0 += q[k][0]; h1 += q[k][1]; }
    }
    if (b < 8) {   // the OTHER parity's totals belong to the previous launch, which has completed: clear them for the next one
        rf_st8(ws + RFW_TOT + ((1 - (A.epoch & 1u)) * 8u + (unsigned)b) * RF_BINS + 2 * t, 0u, 0u);     // (completes by the end of the kernel)
    }
    const unsigned ex = rf_scan2048(S, h0, h1);           // (one block barrier inside; S.wsum = the 16 wave totals)
    unsigned nvalid = 0;
#pragma unroll
    for (int w2 = 0; w2 < 16; ++w2) nvalid += S.wsum[w2];
    rf_stamp(ws, 14);
    const int nslot = 2 * A.nspec;
    // every thread derives the (<= 6) ranks itself (numpy's virtual index (n-1)*q in float32; 32-bit integer arithmetic:
    // n < 2^31) and claims those that fall into its two bins: no serial section
    float gam[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int sp = 0; sp < 3; ++sp) {
        if (sp >= A.nspec) continue;
        const int n = (int)nvalid;
        const float vi = __fmul_rn((float)(n - 1), A.q32[sp]);
        const float fl = floorf(vi);
        int lo, hi;
        if (n <= 0) lo = hi = 0;
        else if (!(vi == vi) || vi >= (float)(n - 1)) lo = hi = n - 1;
        else if (vi < 0.f) lo = hi = 0;
        else { lo = (int)fl; hi = lo + 1; }
        gam[sp] = __fsub_rn(vi, fl);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const unsigned k = (unsigned)(u ? hi : lo);
            const int slot = 2 * sp + u;
            if (k >= ex && k < ex + h0) { S.sbin[slot] = 2 * t; S.srin[slot] = k - ex; S.scnt[slot] = h0; }
            else if (k >= ex + h0 && k < ex + h0 + h1) { S.sbin[slot] = 2 * t + 1; S.srin[slot] = k - ex - h0; S.scnt[slot] = h1; }
        }
    }
    __syncthreads();
    if (wave == 0) {
        // the list table (distinct selected bins, their sizes and offsets in the flat candidate array), one LANE per slot:
        // ~60 instructions of one wave instead of a serial chain of LDS round trips (or 16 waves doing it redundantly)
        const int sl = lane;
        const bool on = nvalid != 0 && sl < nslot;
        const unsigned mybin = on ? S.sbin[sl] : 0xffffff00u + (unsigned)sl, mycnt = on ? S.scnt[sl] : 0u, myrin = on ? S.srin[sl] : 0u;
        int first = sl, twin = sl;
#pragma unroll
        for (int u = RF_MAXSLOT - 1; u >= 0; --u) {
            const unsigned bu = (unsigned)__builtin_amdgcn_readlane((int)mybin, u), ru = (unsigned)__builtin_amdgcn_readlane((int)myrin, u);
            if (u < sl && bu == mybin) first = u;
            if (u < sl && bu == mybin && ru == myrin) twin = u;
        }
        const bool leader = on && first == sl;
        const unsigned long long lm = __ballot(leader);
        const int dlist = __popcll(lm & ((1ull << first) - 1ull));
        unsigned off = 0, total = 0;
#pragma unroll
        for (int u = 0; u < RF_MAXSLOT; ++u) {
            const unsigned cu = (unsigned)__builtin_amdgcn_readlane((int)mycnt, u);
            if ((lm >> u) & 1ull) { total += cu; if (u < first) off += cu; }
        }
        if (sl < RF_MAXSLOT) { S.sdl[sl] = on ? dlist : 0; S.twin[sl] = twin; }
        if (leader) {
            S.dbin[dlist] = mybin; S.doff[dlist] = off; S.dcnt[dlist] = mycnt; S.dblk[dlist] = 0; S.dbase[dlist] = 0;
            unsigned klo = 0, knb = 32;
            if (mybin >= 1u) rf_bin_key_range(mybin, A.bin_scale, klo, knb);
            S.lmn[dlist] = klo; S.lnb[dlist] = knb; S.lsh[dlist] = knb > 9u ? knb - 9u : 0u;
        }
        if (sl < RF_MAXSLOT) S.scount[sl] = 0;
        if (lane == 0) {
            S.nd = __popcll(lm); S.ncand = total;
            // bin 0 (negative / denormal-small entropies: no bounded key range) or > 27 key bits: generic selection
            S.bad = 0;
        }
        if (leader && (mybin == 0u || S.lnb[dlist] > 27u)) S.bad = 1;
    }
    __syncthreads();
    const int nd = S.nd;
    const unsigned ncand = S.ncand;
    rf_stamp(ws, 15);
    const bool big = ncand > (unsigned)RF_CAP;      // the same totals everywhere: every block takes the same route
    unsigned* lkeys = (unsigned*)dyn;
    bool done = false;               // the order statistics were selected by the form fused with the gather
    if (!big) {
        // ------------------------------------------------------------ G: the members of the selected bins, straight out of
        // the blocks' sorted runs.  Two prefix words per (list, block) give offset and count; a block-wide exclusive scan
        // of the counts (wave scans + 24 group totals) gives every block's position in the flat candidate list; then
        // member i finds its block by binary search in that prefix: all loads independent, eight in flight per thread.
        unsigned* offL = (unsigned*)dyn;                  // [RF_MAXSLOT][256] offset of bin X in block sb's run
        unsigned* pfxL = offL + RF_MAXSLOT * 256;         // [RF_MAXSLOT][256] exclusive prefix over blocks of the counts
        lkeys = pfxL + RF_MAXSLOT * 256;
        unsigned* hist0 = lkeys + RF_CAP;                 // [RF_MAXSLOT][512] first-digit histogram of every list
        unsigned* small = hist0 + RF_MAXSLOT * 512;       // [RF_MAXSLOT][512] keys of the sub-bucket that holds a slot's rank
        unsigned* mark = small + RF_MAXSLOT * 512;        // [RF_MAXSLOT][512] slots that want the keys of a (list, first digit)
#pragma unroll
        for (int i = 0; i < 3; ++i) { hist0[t + i * RF_T] = 0u; mark[t + i * RF_T] = 0u; }
        const unsigned* slab0 = ws + RFW_SLAB;
        const int np = nd * G;
        const int pa = t, pb = t + RF_T;                   // nd * G <= 6 * 256: at most two pairs per thread
        const bool va = pa < np, vb = pb < np;
        const int da = va ? pa / G : 0, sa = va ? pa % G : 0, db = vb ? pb / G : 0, sb2 = vb ? pb % G : 0;
        unsigned a0, a1, b0, b1;
        {
            const unsigned* qa = slab0 + (size_t)sa * RF_SLABW + (nd ? S.dbin[da] : 0u);
            const unsigned* qb = slab0 + (size_t)sb2 * RF_SLABW + (nd ? S.dbin[db] : 0u);
            rf_ld1x4(qa, qa + 1, qb, qb + 1, a0, a1, b0, b1);
        }
        const unsigned ca = va ? a1 - a0 : 0u, cb = vb ? b1 - b0 : 0u;
        // inclusive wave scans of the two counts (a list's 256 pairs = 4 consecutive waves when G = 256; 2 when G = 128)
        unsigned xa = ca, xb = cb;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned ua = __shfl_up(xa, o, 64), ub = __shfl_up(xb, o, 64);
            if (lane >= o) { xa += ua; xb += ub; }
        }
        if (lane == 63) { S.ptot[wave] = xa; S.ptot[16 + wave] = xb; }     // group g = pair index / 64
        rf_stamp(ws, 16);
        __syncthreads();
        {
            const int gpl = G / 64;                        // 64-pair groups per list
            unsigned ea = xa - ca, eb = xb - cb;
            const int ga = pa / 64, gb = pb / 64;
            for (int q = (ga / gpl) * gpl; q < ga; ++q) ea += S.ptot[q];
            for (int q = (gb / gpl) * gpl; q < gb; ++q) eb += S.ptot[q];
            if (va) { offL[da * 256 + sa] = a0; pfxL[da * 256 + sa] = ea; }
            if (vb) { offL[db * 256 + sb2] = b0; pfxL[db * 256 + sb2] = eb; }
        }
        __syncthreads();
        rf_stamp(ws, 17);
        // member i of the flat candidate list = i-th element of the concatenated per-block segments.  Thread t takes the K
        // CONSECUTIVE members [t K, t K + K) (K = ceil(ncand / 1024) <= 12): one binary search for the first of them, the
        // others by walking the (list, block) segments forward -- every instruction here is executed by 16 waves, a search
        // per member cost 4 us.  All (<= 12) loads of a thread are independent and go out in ONE waited batch; the keys
        // stay in registers for the selection, which starts right here: every key adds its first digit (top 9 bits of its
        // offset inside the bin's key range) to its list's histogram.
        const unsigned* ad[12];
        unsigned val[12], kd = 0;                          // kd: list index of key j, 3 bits each
        const unsigned K = (ncand + RF_T - 1) / RF_T;
        const unsigned i0 = (unsigned)t * K;
        {
            int d = 0, sb = 0;
            unsigned r = 0, dn = 0, seg0 = 0, seg1 = 0;    // r: index inside list d (dn members); block sb holds [seg0, seg1)
            auto seg_end = [&](int dd, int bb, unsigned n_) -> unsigned { return bb + 1 < G ? pfxL[dd * 256 + bb + 1] : n_; };
            if (i0 < ncand) {
                for (int dd = 1; dd < nd; ++dd) d = i0 >= S.doff[dd] ? dd : d;
                r = i0 - S.doff[d]; dn = S.dcnt[d];
                int lo = 0, hi = G;                        // largest block sb with prefix[sb] <= r (it has a member: r < prefix[sb + 1])
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (pfxL[d * 256 + mid] <= r) lo = mid; else hi = mid;
                }
                sb = lo; seg0 = pfxL[d * 256 + sb]; seg1 = seg_end(d, sb, dn);
            }
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                ad[j] = (const unsigned*)A.cand;
                if ((unsigned)j < K && i0 + j < ncand) {   // (K is block-uniform)
                    ad[j] = (const unsigned*)A.cand + (size_t)sb * RF_PXMAX + offL[d * 256 + sb] + (r - seg0);
                    kd |= (unsigned)d << (3 * j);
                    ++r;
                    if (r == dn && d + 1 < nd) { ++d; r = 0; dn = S.dcnt[d]; sb = 0; seg0 = 0; seg1 = seg_end(d, 0, dn); }
                    while (r >= seg1 && sb + 1 < G) { ++sb; seg0 = seg1; seg1 = seg_end(d, sb, dn); }
                }
            }
        }
        rf_ld1x12(ad, val);
        const bool fast = S.bad == 0;                      // block-uniform (set with the list table)
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const unsigned i = i0 + j;
            val[j] = f32_key(__uint_as_float(val[j]));
            if ((unsigned)j < K && i < ncand) {
                lkeys[i] = val[j];                         // (the generic selection forms read the keys from LDS)
                if (fast) {
                    const int d = (kd >> (3 * j)) & 7;
                    const unsigned rel = val[j] - S.lmn[d];
                    if (rel >> S.lnb[d]) S.bad = 1;        // outside the bin's key range (never observed): generic selection
                    else atomicAdd(&hist0[d * 512 + (rel >> S.lsh[d])], 1u);
                }
            }
        }
        rf_stamp(ws, 3);
        rf_stamp(ws, 4);
        __syncthreads();
        rf_stamp(ws, 8);
        if (fast && S.bad == 0) {
            // ---------------------------------------------------------------- D (fused): first digit of every order statistic
            // from its list's histogram (one wave per slot), then the keys of that sub-bucket (n / 512 of the list on
            // average) are collected and the statistic is selected among them by one wave: 3 block barriers, no key re-reads
            if (wave < nslot && nvalid && S.twin[wave] == wave) {
                const int d = S.sdl[wave];
                unsigned k = S.srin[wave];
                const uint4 ha = *(const uint4*)(hist0 + d * 512 + lane * 8), hb = *(const uint4*)(hist0 + d * 512 + lane * 8 + 4);
                const unsigned h[8] = {ha.x, ha.y, ha.z, ha.w, hb.x, hb.y, hb.z, hb.w};
                unsigned loc = 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) loc += h[q];
                unsigned x = loc;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const unsigned u = __shfl_up(x, o, 64);
                    if (lane >= o) x += u;
                }
                const unsigned exl = x - loc;
                if (k >= exl && k < x) {                   // exactly one lane
                    unsigned run = exl, dig = 0, kk = 0;
                    bool got = false;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        if (!got && k < run + h[q]) { dig = lane * 8 + q; kk = k - run; got = true; }
                        run += h[q];
                    }
                    S.sd0[wave] = dig; S.sk1[wave] = kk;
                    atomicOr(&mark[d * 512 + dig], 1u << wave);      // which slots want the keys of this (list, digit)
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                if ((unsigned)j < K && i0 + j < ncand) {
                    const int d = (kd >> (3 * j)) & 7;
                    const unsigned dig = (val[j] - S.lmn[d]) >> S.lsh[d];
                    for (unsigned m = mark[d * 512 + dig]; m; m &= m - 1u) {      // (one key in 512 gets past this read)
                        const int sl = __ffs(m) - 1;
                        const unsigned pos = atomicAdd(&S.scount[sl], 1u);
                        if (pos < 512u) small[sl * 512 + pos] = val[j];
                    }
                }
            }
            __syncthreads();
            bool over = false;
            for (int sl = 0; sl < nslot; ++sl) over = over || (S.twin[sl] == sl && S.scount[sl] > 512u);
            if (!over || !nvalid) {
                if (wave < nslot && nvalid && S.twin[wave] == wave) {
                    const unsigned key = rf_wave_select_reg(S.whist[wave], small + wave * 512, S.scount[wave], S.sk1[wave]);
                    if (lane == 0) S.skey[wave] = key;
                }
                done = true;
            }
            // (a sub-bucket of more than 512 keys -- heavy ties: the generic selection below)
        }
    } else {
    // ---------------------------------------------------------------- degenerate route (more candidates than LDS holds):
    // my offset inside each list = what the blocks before me put there: bin counts from the published prefixes
    for (int i = t; i < nd * G; i += RF_T) {
        const int d = i / G, sb = i % G;                      // G >= 128: a wave stays inside one list
        const unsigned* q = ws + RFW_SLAB + (size_t)sb * RF_SLABW + S.dbin[d];
        unsigned v = sb < b ? rf_ld(q + 1) - rf_ld(q) : 0u;
        v = wave_sum_u(v);
        if (lane == 0 && v) atomicAdd(&S.dbase[d], v);
    }
    __syncthreads();
    float* cbig = A.cand + (size_t)G * RF_PXMAX;              // the compact candidate list of this route
    auto list_of = [&](float e) -> int {          // candidate list (distinct selected bin) of a valid entropy, or -1
        if (!(e == e)) return -1;
        const unsigned bn = (unsigned)rf_bin(e, A.bin_scale);
        int r = -1;
        for (int d = 0; d < nd; ++d) r = S.dbin[d] == bn ? d : r;
        return r;
    };
    auto emit = [&](float e) {
        const int d = list_of(e);
        if (d >= 0) rf_st((unsigned*)cbig + S.doff[d] + S.dbase[d] + atomicAdd(&S.dblk[d], 1u), __float_as_uint(e));
    };
#pragma unroll
    for (int it = 0; it < RF_NIT; ++it)
#pragma unroll
        for (int a = 0; a < 4; ++a) emit(er[it][a]);
    emit(er1);
    rf_stamp(ws, 3);
    rf_grid_sync(ws, 2u * A.epoch + 2u, G);
    rf_stamp(ws, 4);
    }
    // ---------------------------------------------------------------- phase D: exact selection in LDS, thresholds
    if (!big) {
        if (!done) {
            // generic forms on the keys in LDS (lists whose rank sub-bucket overflowed, keys outside a bin's nominal range)
            __syncthreads();
            bool small_l = true, fits = true;
            for (int d = 0; d < nd; ++d) { small_l = small_l && S.dcnt[d] <= 64u * RF_KW; fits = fits && S.dcnt[d] <= 128u * RF_KREG; }
            if (small_l) {
                if (wave < nslot && nvalid && S.twin[wave] == wave) {
                    const int d = S.sdl[wave];
                    const unsigned key = rf_wave_select_reg(S.whist[wave], lkeys + S.doff[d], S.dcnt[d], S.srin[wave]);
                    if (lane == 0) S.skey[wave] = key;
                }
            } else if (fits) {
                rf_block_select(S, lkeys, nslot, nvalid != 0);
            } else if (wave < nslot && nvalid) {          // a long list: one wave per slot streaming it from LDS
                int same = -1;
                for (int u = 0; u < wave; ++u)
                    if (S.sbin[u] == S.sbin[wave] && S.srin[u] == S.srin[wave]) { same = u; break; }
                if (same < 0) {
                    const int d = S.sdl[wave];
                    const unsigned kk = rf_wave_select(S.whist[wave], lkeys + S.doff[d], S.dcnt[d], S.srin[wave]);
                    if (lane == 0) S.skey[wave] = kk;
                }
            }
        }
        __syncthreads();
        rf_stamp(ws, 9);
    } else {
        const float* cbig = A.cand + (size_t)G * RF_PXMAX;
        for (int d = 0; d < nd; ++d) {
            const unsigned n = S.dcnt[d];
            const bool inlds = n <= RF_CAP;
            if (inlds)
                for (unsigned i = t; i < n; i += RF_T) lkeys[i] = f32_key(__uint_as_float(rf_ld((const unsigned*)cbig + S.doff[d] + i)));
            __syncthreads();
            for (int s = 0; s < nslot; ++s) {
                if (S.sbin[s] != S.dbin[d]) continue;       // block-uniform
                int same = -1;
                for (int u = 0; u < s; ++u)
                    if (S.sbin[u] == S.sbin[s] && S.srin[u] == S.srin[s]) { same = u; break; }
                unsigned kk;
                if (same >= 0) kk = S.skey[same];
                else kk = inlds ? rf_select<true>(S, lkeys, nullptr, n, S.srin[s]) : rf_select<false>(S, nullptr, cbig + S.doff[d], n, S.srin[s]);
                __syncthreads();
                if (t == 0) S.skey[s] = kk;
                __syncthreads();
            }
        }
    }
    // thresholds: every thread lerps them itself like numpy (float32, no FMA) from the selected keys (a statistic shared by
    // two slots was computed by the first of them)
    float thr3[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int sp = 0; sp < 3; ++sp) {
        if (sp >= A.nspec) continue;
        const unsigned ka = S.skey[S.twin[2 * sp]], kb = S.skey[S.twin[2 * sp + 1]];
        const float a = key_f32(ka), bb = key_f32(kb);
        const float g = gam[sp];
        const float dd = __fsub_rn(bb, a);
        float thr = (g >= 0.5f) ? __fsub_rn(bb, __fmul_rn(dd, __fsub_rn(1.0f, g))) : __fadd_rn(a, __fmul_rn(dd, g));
        if (nvalid == 0) thr = __uint_as_float(0x7fc00000u);
        thr3[sp] = thr;
        if (b == 0 && t == sp) {
            ws[RFW_THR + sp] = __float_as_uint(thr);
            ws[RFW_VAL + 2 * sp] = __float_as_uint(a);
            ws[RFW_VAL + 2 * sp + 1] = __float_as_uint(bb);
        }
    }
    // every launch advances the barrier counters by two slots: the usual route used one (added after this block's last
    // barrier, i.e. when every block has passed it: nobody can be released early by it)
    if (t == 0 && !big)
        __hip_atomic_fetch_add(ws + RFW_BAR8 + 16 * (blockIdx.x & 7), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (b == 0 && t == 0) {
        rf_st(ws + RFW_LAUNCH, rf_ld(ws + RFW_LAUNCH) + 1u);
        if (big) rf_st(ws + RFW_BIG, rf_ld(ws + RFW_BIG) + 1u);
    }
    rf_stamp(ws, 5);
    const float tdrop = thr3[0], tlo = thr3[1], thi = thr3[2];
    {   // #kept first (registers only): its block reduction must not sit behind the completion of the stores below
        const unsigned ign8 = (unsigned)A.ignore & 255u;      // (label bytes: the ignore value's low byte marks ignored pixels)
        unsigned kept = 0;
#pragma unroll
        for (int it = 0; it < RF_NIT; ++it) {
            if (pbase[it] < 0) continue;
            const int nx = min(4, A.W - (pbase[it] & 1023));
#pragma unroll
            for (int a = 0; a < 4; ++a)
                if (a < nx) kept += (((labr[it] >> (8 * a)) & 255u) != ign8 && !(er[it][a] >= tdrop)) ? 1u : 0u;
        }
        if (pb1 >= 0) kept += (lab1 != ign8 && !(er1 >= tdrop)) ? 1u : 0u;
        kept = wave_sum_u(kept);
        if (lane == 0) S.wsum[wave] = kept;
        __syncthreads();
        if (t == 0) {
            unsigned tk = 0;
            for (int w2 = 0; w2 < 16; ++w2) tk += S.wsum[w2];
            if (tk) atomicAdd(ws + RFW_NKEPT, tk);
        }
    }
    // (32-bit index arithmetic hoisted per 4-pixel item, and ONE 16-byte store for an item's four entropies / two for its
    // four targets: the phase is store-issue bound -- 22 narrow stores per thread took 4 us)
    const unsigned ign8 = (unsigned)A.ignore & 255u;
    struct __attribute__((packed, aligned(4))) F4 { float v[4]; };
    struct __attribute__((packed, aligned(8))) L2 { long long v[2]; };
    auto target_of = [&](float e, unsigned lb) -> long long { return (lb != ign8 && !(e >= tdrop)) ? (long long)lb : (long long)A.ignore; };
    auto masks = [&](float e, int lq) {
        if (lq >= 0) {
            A.low_mask[lq] = e <= tlo ? 1.f : 0.f;
            A.high_mask[lq] = A.neg_high ? (e >= thi ? 1.f : 0.f) : 1.f;
        }
    };
#pragma unroll
    for (int it = 0; it < RF_NIT; ++it) {
        if (pbase[it] < 0) continue;
        const int n = pbase[it] >> 20, oy = (pbase[it] >> 10) & 1023, ox0 = pbase[it] & 1023;
        const int nx = min(4, A.W - ox0);
        const unsigned p0 = ((unsigned)n * A.H + oy) * A.W + ox0;
        const int ly = A.nspec > 1 ? (int)S.invy[oy] : -1;
        const int lrow = ly >= 0 ? ((A.B + n) * A.hm + ly) * A.wm : -1;
        if (nx == 4) {
            F4 ev; L2 t0, t1;
#pragma unroll
            for (int a = 0; a < 4; ++a) ev.v[a] = er[it][a];
            t0.v[0] = target_of(er[it][0], labr[it] & 255u); t0.v[1] = target_of(er[it][1], (labr[it] >> 8) & 255u);
            t1.v[0] = target_of(er[it][2], (labr[it] >> 16) & 255u); t1.v[1] = target_of(er[it][3], (labr[it] >> 24) & 255u);
            *(F4*)(A.ent + p0) = ev;
            *(L2*)(A.target_u + p0) = t0;
            *(L2*)(A.target_u + p0 + 2) = t1;
        } else {
#pragma unroll
            for (int a = 0; a < 4; ++a)
                if (a < nx) { A.ent[p0 + a] = er[it][a]; A.target_u[p0 + a] = target_of(er[it][a], (labr[it] >> (8 * a)) & 255u); }
        }
        if (lrow >= 0) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
                if (a < nx) { const int lx = (int)S.invx[ox0 + a]; masks(er[it][a], lx >= 0 ? lrow + lx : -1); }
        }
    }
    if (pb1 >= 0) {
        const int n = pb1 >> 20, oy = (pb1 >> 10) & 1023, ox = pb1 & 1023;
        const unsigned p = ((unsigned)n * A.H + oy) * A.W + ox;
        A.ent[p] = er1;
        A.target_u[p] = target_of(er1, lab1);
        const int ly = A.nspec > 1 ? (int)S.invy[oy] : -1, lx = ly >= 0 ? (int)S.invx[ox] : -1;
        masks(er1, lx >= 0 ? ((A.B + n) * A.hm + ly) * A.wm + lx : -1);
    }
    rf_stamp(ws, 6);
    rf_stamp(ws, 7);
    if (t == 0) {
        ws[6144 + b] = dbg_t0; ws[5120 + b] = dbg_arr; ws[5632 + b] = dbg_rel; ws[6656 + b] = (unsigned)wall_clock64();
        if (b == 0)
            for (int k = 0; k < 32; ++k) {
                ws[RFW_CLK + k] = rf_clk[k];
                if (k < 28 && rf_clk[k]) ws[7168 + k] += rf_clk[k] - rf_clk[0];      // (sums over launches: tools/bench_split.py divides by word 7168 + 31)
            }
        if (b == 0) ws[7168 + 31] += 1u;
    }
}

// logits_low: strided (B, C, h, w) view of the TRAIN-mode teacher logits of the unlabeled half; H-1 == 4(h-1), W-1 == 4(w-1).
// q32[nspec]: percentiles / 100 in float32 (host values): [0] drop, [1] alpha_t, [2] 100 - alpha_t; nspec = 1 (no contrastive
// branch: only target_u is written) or 3.  workspace: u2pl_reliability_fused_workspace_bytes(G) bytes, ZEROED by the caller;
// (zeroed ONCE; `epoch` = 0, 1, 2, ... counts the launches on this workspace -- barrier counters only ever grow and the
// bin totals are double-buffered by launch parity, so one buffer per stream is reused step after step with no reset);
// cand: u2pl_reliability_fused_cand_floats(B*H*W, G) floats of scratch (the blocks' sorted runs + the compact list of the
// degenerate route).  Returns U2PL_EINVAL when the shape does not fit the fused kernel (caller falls back to
// u2pl_entropy_up_f32 + u2pl_select_f32 + u2pl_reliability_apply).  Thresholds land in workspace words 16..18; word 4
// counts the launches on the workspace, word 5 those that needed the second barrier.  flags bit 0: agent-scope fence pair
// around the first barrier (not needed: see rf_grid_arrive_wait).
U2PL_API int u2pl_reliability_fused(const float* logits_low, long sn, long sc, long sh, long sw, int B, int C, int h,
                                    int w, int H, int W, const long long* label_u, const long long* label_l,
                                    int ignore, int nspec, const float* q32_host, int negative_high_entropy, int hm,
                                    int wm, float* entropy, long long* target_u, float* low_mask, float* high_mask,
                                    unsigned* lbits, unsigned* workspace, float* cand, int G, unsigned epoch, int flags,
                                    hipStream_t stream) {
    if (!(C == 19 || C == 21) || (nspec != 1 && nspec != 3)) return U2PL_EINVAL;
    if (h < 2 || w < 2 || H - 1 != 4 * (h - 1) || W - 1 != 4 * (w - 1) || H > 1024 || W > 1024) return U2PL_EINVAL;
    if (hm > H || wm > W || hm > 1024 || wm > 1024 || ignore < 0 || ignore > 255) return U2PL_EINVAL;
    if (G < 128 || G > 256 || (G & (G - 1))) return U2PL_EINVAL;     // phase B keeps one bin per wave: 1024 * G / 2048 >= 64
    {   // per-block cells: at most RF_NIT iterations at 4 px / thread (+ a <= 64-cell remainder at 1 px / thread)
        const long per = ((long)B * h * w + G - 1) / G, rem = per % RF_CELLS;
        if (per / RF_CELLS + ((rem > 64) ? 1 : 0) > RF_NIT) return U2PL_EINVAL;
    }
    RfArgs A;
    A.in = logits_low; A.sn = sn; A.sc = sc; A.sh = sh; A.sw = sw;
    A.B = B; A.h = h; A.w = w; A.H = H; A.W = W;
    A.sy = ac_scale_host(h, H); A.sx = ac_scale_host(w, W);
    A.ny = (float)H / (float)hm; A.nx = (float)W / (float)wm;
    A.hm = hm; A.wm = wm;
    A.label_u = label_u; A.label_l = label_l; A.ignore = ignore; A.nspec = nspec; A.neg_high = negative_high_entropy;
    for (int j = 0; j < 3; ++j) A.q32[j] = j < nspec ? q32_host[j] : 0.f;
    A.bin_scale = 1022.0f / (logf((float)C) + 0.02f - 0.015625f);
    A.ent = entropy; A.target_u = target_u; A.low_mask = low_mask; A.high_mask = high_mask; A.lbits = lbits;
    A.ws = workspace; A.cand = cand; A.epoch = epoch;
    A.fences = flags & 1;
    // dynamic LDS: phase A's corner logits | the sorted run (RF_PXMAX floats) | block offsets / prefixes + candidate keys
    // phase A tile: the contiguous low-resolution pixel span [c0, c1 + w + 1) of a block, pixel-major
    // (+ the column -> row exchange of the entropies behind it: RF_NIT x 256 cells x 17 words)
    const size_t lds_a = ((((size_t)(((long)B * h * w + G - 1) / G + w + 1) * C + 8 + 3) & ~(size_t)3) + (size_t)RF_NIT * RF_CELLS * 17) * sizeof(float);
    const size_t lds_g = ((size_t)2 * RF_MAXSLOT * 256 + RF_CAP + 3 * RF_MAXSLOT * 512) * sizeof(unsigned);
    size_t lds = lds_a > lds_g ? lds_a : lds_g;
    if ((size_t)RF_PXMAX * sizeof(float) > lds) lds = (size_t)RF_PXMAX * sizeof(float);
    static size_t attr = 0;
    if (lds > attr) {
        (void)hipFuncSetAttribute((const void*)k_reliability_fused<19>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)k_reliability_fused<21>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = lds;
    }
    if (C == 19) U2PL_LAUNCH(k_reliability_fused<19>, dim3(G), dim3(RF_T), lds, stream, A);
    else U2PL_LAUNCH(k_reliability_fused<21>, dim3(G), dim3(RF_T), lds, stream, A);
    U2PL_LAUNCH_CHECK();
    return 0;
}
