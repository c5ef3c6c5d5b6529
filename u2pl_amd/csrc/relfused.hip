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
// Execution model: gridDim.x = G <= #CUs blocks of 1024 threads, all co-resident (one per CU), separated by three
// device-wide barriers (atomic counter + agent-scope fences).  Every block owns a contiguous range of 4x4 "cells"
// of the full-resolution grid; the entropies and labels of its pixels never leave registers between the phases.
//   A  entropy (bit-exact FMA bilinear of the 4 corner logits staged in LDS) -> global `ent`, block histogram over
//      2048 MONOTONE bins of the entropy value (log-linear below 2^-6, linear above: spreads both the near-zero
//      entropies of a trained model and the near-ln(C) entropies of an untrained one) -> per-block slab
//   B  block b sums bins [b*2048/G, ...) over the G slabs -> totals                                  [barrier 1,2]
//   C  every block scans the totals, derives the six ranks (numpy virtual index (n-1)*q in float32), finds the bin of
//      each rank, and appends ITS entropies that fall into one of those (<= 6) bins to that bin's candidate list
//   D  every block loads the candidate lists (typically ~10^3 values) into LDS and selects the exact order
//      statistics there (radix select on the order-preserving key, 11 bits per pass over the list's key range),
//      lerps the thresholds like numpy (float32, no FMA), and applies them to its own register-resident pixels;
//      the labeled-half masks and the class bits are label-only and are spread over all threads at the end.  [barrier 3]
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
#define RFW_TOT 1024            // [2][2048] totals, double-buffered by launch parity (3072..4095 unused)
#define RFW_SLAB 8192           // [G][2048]

struct RfArgs {
    const float* in; long sn, sc, sh, sw;
    int B, h, w, H, W;
    float sy, sx, ny, nx;
    int hm, wm;                  // size of the low-res masks (= student prediction map)
    const long long* label_u; const long long* label_l;
    int ignore, nspec, neg_high;
    unsigned epoch;              // launch index on this workspace (host counter): barrier targets and totals parity
    float q32[3];
    float bin_scale;
    float* ent; long long* target_u; float* low_mask; float* high_mask; unsigned* lbits;
    unsigned* ws; float* cand;
};

U2PL_API size_t u2pl_reliability_fused_workspace_bytes(int G) { return (size_t)(RFW_SLAB + (size_t)G * RF_BINS) * sizeof(unsigned); }

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

// Device-wide barrier of a persistent launch whose blocks are all co-resident (gridDim.x <= #CUs, one block per CU).
// The spin is bounded (~seconds): if some block could not become resident the kernel gives up loudly (error word set,
// results undefined) instead of hanging the GPU.
#define RFW_ERR 3
#define RFW_BAR8 64             // 8 arrival counters, 16 words (one 64-byte line) apart
__device__ __forceinline__ void rf_grid_sync(unsigned* ws, unsigned phase, int G) {
    // Arrival: thread 0 releases at agent scope (write-back of the XCD's L2: the block's stores were acknowledged at
    // the __syncthreads()), then bumps one of 8 counters (same-address atomics serialise at ~88 / us: one shared counter
    // costs ~3 us per barrier in arrivals alone); lanes 0..7 poll the 8 counters RELAXED (an acquire per poll would
    // invalidate the L2 on every iteration: measured 13-17 us per barrier) and thread 0 acquires once at the end.
    // sc1 stores / atomic loads WITHOUT these fences were measured to be insufficient on ordinary (coarse-grained)
    // device memory: intermittently stale histogram / candidate words from another XCD.
    __syncthreads();
    if (threadIdx.x < 8) {
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(ws + RFW_BAR8 + 16 * (blockIdx.x & 7), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // lane k polls counter k: it expects phase * (number of blocks with index % 8 == k)
        const unsigned want = phase * (unsigned)((G - (int)threadIdx.x + 7) / 8);    // phase = 2 * epoch + {1, 2}
        unsigned spins = 0;
        while ((int)(__hip_atomic_load(ws + RFW_BAR8 + 16 * threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {   // wrap-safe
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 19)) { __hip_atomic_store(ws + RFW_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}
#define RFW_CLK 32              // [10] wall-clock stamps of block 0 (100 MHz ticks)
__device__ __forceinline__ void rf_stamp(unsigned* ws, int k) {
    if (blockIdx.x == 0 && threadIdx.x == 0) ws[RFW_CLK + k] = (unsigned)wall_clock64();
}
// (Raw buffer loads with the sc1 cache-policy bit were tried for the bulk cross-block reads -- they pipeline better
// than a chain of atomic loads -- but they returned STALE histogram words on every second launch (the buffer the
// previous launch of the same parity had read): only the atomic loads below are coherent across the XCDs' L2s.)
__device__ __forceinline__ void rf_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned rf_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

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
        key[j] = i < n ? list[i] : 0xffffffffu;
        if (i < n) { mn = min(mn, key[j]); mx = max(mx, key[j]); }
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
                const unsigned v = key[j] - mn;
                if (sub + 128u * j < n && ((v >> shift) >> 8) == prefix) atomicAdd(&hist[(v >> shift) & 255u], 1u);
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
    }
    if (work && (wave & 1) == 0 && lane == 0) S.skey[slot] = mn + prefix;
}

template <int CT>
__global__ __launch_bounds__(RF_T, 1) void k_reliability_fused(RfArgs A) {
    extern __shared__ float dyn[];                  // phase A: corner logits [4][CT][RF_CELLS]; phase D: candidate keys
    __shared__ RfShared S;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int G = gridDim.x, b = blockIdx.x;
    unsigned* ws = A.ws;
    unsigned* tot = ws + RFW_TOT + (A.epoch & 1u) * RF_BINS;
    const long HW = (long)A.H * A.W;
    rf_stamp(ws, 0);
    if (b == 0 && t == 7) rf_st(ws + RFW_NKEPT, 0u);
    S.hist[2 * t] = 0; S.hist[2 * t + 1] = 0;
    S.invy[t] = -1; S.invx[t] = -1;
    __syncthreads();
    if (t < A.hm) S.invy[nearest_src(t, A.ny, A.H)] = (short)t;
    if (t < A.wm) S.invx[nearest_src(t, A.nx, A.W)] = (short)t;
    // ---------------------------------------------------------------- label-only outputs first (nothing depends on them
    // and they depend on nothing: their gather latency hides under phase A instead of sitting on the kernel's tail)
    if (A.nspec > 1) {
        const long lowplane = (long)A.hm * A.wm, nlow = (long)2 * A.B * lowplane;
        for (long q = (long)b * RF_T + t; q < nlow; q += (long)G * RF_T) {
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
    float (*cv)[CT][RF_CELLS] = (float (*)[CT][RF_CELLS])dyn;
    // (Shifting by the cell's largest corner logit instead of the per-pixel maximum would save the max pass -- the
    // entropy is shift invariant -- but log(s) and t/s then cancel at magnitude |shift - max|: measured 4e-6 instead of
    // < 1e-6 on the entropy.  Parity first: the exact maximum is kept.)
    auto cell_geom = [&](long q, int& n, int& ci, int& cj) {
        cj = (int)(q % A.w);
        const long t0 = q / A.w;
        ci = (int)(t0 % A.h); n = (int)(t0 / A.h);
    };
    auto stage = [&](long q, bool live, int part, int slot) {       // corner `part` of cell q -> cv[part][:][slot]
        if (!live) return;
        int n, ci, cj;
        cell_geom(q, n, ci, cj);
        const AcCoord cy0 = ac_coord(min(ci * 4, A.H - 1), A.sy, A.h), cx0 = ac_coord(min(cj * 4, A.W - 1), A.sx, A.w);
        const int yy = (part & 2) ? cy0.i1 : cy0.i0, xx = (part & 1) ? cx0.i1 : cx0.i0;
        const float* src = A.in + n * A.sn + yy * A.sh + xx * A.sw;
#pragma unroll 7
        for (int c = 0; c < CT; ++c) cv[part][c][slot] = src[c * A.sc];
    };
#pragma unroll
    for (int it = 0; it < RF_NIT; ++it) {
        pbase[it] = -1;
        labr[it] = 0;
#pragma unroll
        for (int a = 0; a < 4; ++a) er[it][a] = __uint_as_float(0x7fc00000u);
        if (it >= nit4) continue;                                     // block-uniform
        const int cl = t & (RF_CELLS - 1), part = t >> 8;
        const long q = c0 + (long)it * RF_CELLS + cl;
        const bool live = q < c1;
        stage(q, live, part, cl);
        __syncthreads();
        int n = 0, ci = 0, cj = 0;
        if (live) cell_geom(q, n, ci, cj);
        const int oy = ci * 4 + part, ox0 = cj * 4;
        if (live && oy < A.H) {
            const AcCoord cy = ac_coord(oy, A.sy, A.h);
            float lx0[4], lx1[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const AcCoord cx = ac_coord(min(ox0 + a, A.W - 1), A.sx, A.w);
                lx0[a] = cx.l0; lx1[a] = cx.l1;
            }
            // two pixels at a time: their CT up-sampled logits stay in registers between the max pass and the exp pass
            // (the bilinear form is evaluated once per value instead of twice)
            float s[4], tt[4];
            if constexpr (CT <= 19) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                float z0[CT], z1[CT], m0 = -INFINITY, m1 = -INFINITY;
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    const float v00 = cv[0][c][cl], v01 = cv[1][c][cl], v10 = cv[2][c][cl], v11 = cv[3][c][cl];
                    const float t0 = __fmaf_rn(lx0[2 * hf], v00, __fmul_rn(lx1[2 * hf], v01));
                    const float b0 = __fmaf_rn(lx0[2 * hf], v10, __fmul_rn(lx1[2 * hf], v11));
                    const float t1 = __fmaf_rn(lx0[2 * hf + 1], v00, __fmul_rn(lx1[2 * hf + 1], v01));
                    const float b1 = __fmaf_rn(lx0[2 * hf + 1], v10, __fmul_rn(lx1[2 * hf + 1], v11));
                    z0[c] = __fmaf_rn(cy.l0, t0, __fmul_rn(cy.l1, b0));
                    z1[c] = __fmaf_rn(cy.l0, t1, __fmul_rn(cy.l1, b1));
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
                s[2 * hf] = s0; s[2 * hf + 1] = s1; tt[2 * hf] = u0; tt[2 * hf + 1] = u1;
            }
            } else {      // more classes: the 2 x CT register copy would spill -> evaluate the bilinear form in both passes
                float m[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) { m[a] = -INFINITY; s[a] = 0.f; tt[a] = 0.f; }
#pragma unroll 4
                for (int c = 0; c < CT; ++c) {
                    const float v00 = cv[0][c][cl], v01 = cv[1][c][cl], v10 = cv[2][c][cl], v11 = cv[3][c][cl];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const float top = __fmaf_rn(lx0[a], v00, __fmul_rn(lx1[a], v01));
                        const float bot = __fmaf_rn(lx0[a], v10, __fmul_rn(lx1[a], v11));
                        m[a] = fmaxf(m[a], __fmaf_rn(cy.l0, top, __fmul_rn(cy.l1, bot)));
                    }
                }
#pragma unroll 4
                for (int c = 0; c < CT; ++c) {
                    const float v00 = cv[0][c][cl], v01 = cv[1][c][cl], v10 = cv[2][c][cl], v11 = cv[3][c][cl];
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
            }
            const int nx = min(4, A.W - ox0);
            const long p0 = ((long)n * A.H + oy) * A.W + ox0;
            pbase[it] = (n << 20) | (oy << 10) | ox0;
#pragma unroll
            for (int a = 0; a < 4; ++a)
                if (a < nx) {
                    const long long l = A.label_u[p0 + a];
                    const bool valid = l != (long long)A.ignore;
                    labr[it] |= ((unsigned)l & 255u) << (8 * a);
                    float e = logf(s[a]) - tt[a] / s[a];
                    e = valid ? e : __uint_as_float(0x7fc00000u);
                    er[it][a] = e;
                    if (valid) atomicAdd(&S.hist[rf_bin(e, A.bin_scale)], 1u);
                } else {
                    labr[it] |= ((unsigned)A.ignore & 255u) << (8 * a);
                }
        }
        __syncthreads();
    }
    if (tail1) {    // the <= 64 remaining cells: one pixel per thread so that the tail costs ~1/16 of a full iteration
        {
            const int cl = t & 63, part = t >> 6;
            if (t < 256) stage(c0 + (long)nmain * RF_CELLS + cl, cl < rem && c0 + (long)nmain * RF_CELLS + cl < c1, part, cl);
        }
        __syncthreads();
        const int cl = t >> 4, a = t & 3, row = (t >> 2) & 3;
        const long q = c0 + (long)nmain * RF_CELLS + cl;
        if (cl < rem && q < c1) {
            int n, ci, cj;
            cell_geom(q, n, ci, cj);
            const int oy = ci * 4 + row, ox = cj * 4 + a;
            if (oy < A.H && ox < A.W) {
                const AcCoord cy = ac_coord(oy, A.sy, A.h), cx = ac_coord(ox, A.sx, A.w);
                float m = -INFINITY, sm = 0.f, tt = 0.f;
#pragma unroll 4
                for (int c = 0; c < CT; ++c) {
                    const float top = __fmaf_rn(cx.l0, cv[0][c][cl], __fmul_rn(cx.l1, cv[1][c][cl]));
                    const float bot = __fmaf_rn(cx.l0, cv[2][c][cl], __fmul_rn(cx.l1, cv[3][c][cl]));
                    m = fmaxf(m, __fmaf_rn(cy.l0, top, __fmul_rn(cy.l1, bot)));
                }
#pragma unroll 4
                for (int c = 0; c < CT; ++c) {
                    const float top = __fmaf_rn(cx.l0, cv[0][c][cl], __fmul_rn(cx.l1, cv[1][c][cl]));
                    const float bot = __fmaf_rn(cx.l0, cv[2][c][cl], __fmul_rn(cx.l1, cv[3][c][cl]));
                    const float z = __fmaf_rn(cy.l0, top, __fmul_rn(cy.l1, bot)) - m;
                    const float e = rf_exp_neg(z);
                    sm += e;
                    tt += e * z;
                }
                const long p0 = ((long)n * A.H + oy) * A.W + ox;
                const long long l = A.label_u[p0];
                const bool valid = l != (long long)A.ignore;
                lab1 = (unsigned)l & 255u;
                pb1 = (n << 20) | (oy << 10) | ox;
                float e = logf(sm) - tt / sm;
                e = valid ? e : __uint_as_float(0x7fc00000u);
                er1 = e;
                if (valid) atomicAdd(&S.hist[rf_bin(e, A.bin_scale)], 1u);
            }
        }
        __syncthreads();
    }
    {   // my histogram -> slab (read back column-wise in phase C) and -> the totals (fire-and-forget atomics, spread
        // over up to 2048 addresses, so no same-address serialisation; saves a barrier + a reduction phase)
        unsigned* slab = ws + RFW_SLAB + (size_t)b * RF_BINS;
        const unsigned v0 = S.hist[2 * t], v1 = S.hist[2 * t + 1];
        rf_st(slab + 2 * t, v0);
        rf_st(slab + 2 * t + 1, v1);
        if (v0) __hip_atomic_fetch_add(tot + 2 * t, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v1) __hip_atomic_fetch_add(tot + 2 * t + 1, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    rf_stamp(ws, 1);
    rf_grid_sync(ws, 2u * A.epoch + 1u, G);
    rf_stamp(ws, 2);
    // ---------------------------------------------------------------- phase C: ranks -> bins -> candidates
    const unsigned h0 = rf_ld(tot + 2 * t), h1 = rf_ld(tot + 2 * t + 1);
    if (b == 1 || G == 1) {   // the OTHER parity's totals belong to the previous launch, which has completed: clear them for the next one
        rf_st(ws + RFW_TOT + (1 - (A.epoch & 1u)) * RF_BINS + 2 * t, 0u);
        rf_st(ws + RFW_TOT + (1 - (A.epoch & 1u)) * RF_BINS + 2 * t + 1, 0u);
    }
    const unsigned ex = rf_scan2048(S, h0, h1);
    if (t == RF_T - 1) S.red[0][0] = ex + h0 + h1;      // n_valid
    __syncthreads();
    const unsigned nvalid = S.red[0][0];
    if (t < A.nspec) {
        const long n = nvalid;
        const float vi = __fmul_rn((float)(n - 1), A.q32[t]);
        const float fl = floorf(vi);
        long lo, hi;
        if (n <= 0) lo = hi = 0;
        else if (!(vi == vi) || vi >= (float)(n - 1)) lo = hi = n - 1;
        else if (vi < 0.f) lo = hi = 0;
        else { lo = (long)fl; hi = lo + 1; }
        S.rank[2 * t] = (unsigned)lo; S.rank[2 * t + 1] = (unsigned)hi;
        S.gamma[t] = __fsub_rn(vi, fl);
    }
    __syncthreads();
    const int nslot = 2 * A.nspec;
    for (int s = 0; s < nslot; ++s) {
        const unsigned k = S.rank[s];
        if (k >= ex && k < ex + h0) { S.sbin[s] = 2 * t; S.srin[s] = k - ex; S.scnt[s] = h0; }
        else if (k >= ex + h0 && k < ex + h0 + h1) { S.sbin[s] = 2 * t + 1; S.srin[s] = k - ex - h0; S.scnt[s] = h1; }
    }
    __syncthreads();
    if (t == 0) {
        int nd = 0;
        unsigned off = 0;
        if (nvalid)
            for (int s = 0; s < nslot; ++s) {
                int at = -1;
                for (int d = 0; d < nd; ++d) at = S.dbin[d] == S.sbin[s] ? d : at;
                if (at < 0) { at = nd; S.dbin[nd] = S.sbin[s]; S.doff[nd] = off; S.dcnt[nd] = S.scnt[s]; off += S.dcnt[nd]; S.dblk[nd] = 0; S.dbase[nd] = 0; ++nd; }
                S.sdl[s] = at;
            }
        S.nd = nd;
    }
    __syncthreads();
    const int nd = S.nd;
    // my offset inside each list = what the blocks before me put there: column `bin` of the slabs (no atomics)
    for (int i = t; i < nd * G; i += RF_T) {
        const int d = i / G, sb = i % G;                      // G >= 128: a wave stays inside one list
        unsigned v = sb < b ? rf_ld(ws + RFW_SLAB + (size_t)sb * RF_BINS + S.dbin[d]) : 0u;
        v = wave_sum_u(v);
        if (lane == 0 && v) atomicAdd(&S.dbase[d], v);
    }
    __syncthreads();
    auto list_of = [&](float e) -> int {          // candidate list (distinct selected bin) of a valid entropy, or -1
        if (!(e == e)) return -1;
        const unsigned bn = (unsigned)rf_bin(e, A.bin_scale);
        int r = -1;
        for (int d = 0; d < nd; ++d) r = S.dbin[d] == bn ? d : r;
        return r;
    };
    auto emit = [&](float e) {
        const int d = list_of(e);
        if (d >= 0) rf_st((unsigned*)A.cand + S.doff[d] + S.dbase[d] + atomicAdd(&S.dblk[d], 1u), __float_as_uint(e));
    };
#pragma unroll
    for (int it = 0; it < RF_NIT; ++it)
#pragma unroll
        for (int a = 0; a < 4; ++a) emit(er[it][a]);
    emit(er1);
    rf_stamp(ws, 3);
    rf_grid_sync(ws, 2u * A.epoch + 2u, G);
    rf_stamp(ws, 4);
    // ---------------------------------------------------------------- phase D: exact selection in LDS, thresholds
    unsigned* lkeys = (unsigned*)dyn;
    const unsigned ncand = nd ? S.doff[nd - 1] + S.dcnt[nd - 1] : 0u;
    if (ncand <= RF_CAP) {
        // the usual case (~10^3 candidates per list): all lists into LDS, one wave per order statistic, no block barriers
        for (unsigned i = t; i < ncand; i += RF_T) lkeys[i] = f32_key(__uint_as_float(rf_ld((const unsigned*)A.cand + i)));
        __syncthreads();
        rf_stamp(ws, 8);
        bool fits = true;
        for (int d = 0; d < nd; ++d) fits = fits && S.dcnt[d] <= 128u * RF_KREG;
        if (fits) {
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
        __syncthreads();
        rf_stamp(ws, 9);
        if (t < nslot && nvalid)
            for (int u = 0; u < t; ++u)
                if (S.sbin[u] == S.sbin[t] && S.srin[u] == S.srin[t]) { S.skey[t] = S.skey[u]; break; }   // first twin computed it
        __syncthreads();
    } else {
        for (int d = 0; d < nd; ++d) {
            const unsigned n = S.dcnt[d];
            const bool inlds = n <= RF_CAP;
            if (inlds)
                for (unsigned i = t; i < n; i += RF_T) lkeys[i] = f32_key(__uint_as_float(rf_ld((const unsigned*)A.cand + S.doff[d] + i)));
            __syncthreads();
            for (int s = 0; s < nslot; ++s) {
                if (S.sbin[s] != S.dbin[d]) continue;       // block-uniform
                int same = -1;
                for (int u = 0; u < s; ++u)
                    if (S.sbin[u] == S.sbin[s] && S.srin[u] == S.srin[s]) { same = u; break; }
                unsigned kk;
                if (same >= 0) kk = S.skey[same];
                else kk = inlds ? rf_select<true>(S, lkeys, nullptr, n, S.srin[s]) : rf_select<false>(S, nullptr, A.cand + S.doff[d], n, S.srin[s]);
                __syncthreads();
                if (t == 0) S.skey[s] = kk;
                __syncthreads();
            }
        }
    }
    if (t < A.nspec) {
        const float a = key_f32(S.skey[2 * t]), bb = key_f32(S.skey[2 * t + 1]);
        const float g = S.gamma[t];
        const float dd = __fsub_rn(bb, a);
        float thr = (g >= 0.5f) ? __fsub_rn(bb, __fmul_rn(dd, __fsub_rn(1.0f, g))) : __fadd_rn(a, __fmul_rn(dd, g));
        if (nvalid == 0) thr = __uint_as_float(0x7fc00000u);
        S.thr[t] = thr;
        if (b == 0) {
            ws[RFW_THR + t] = __float_as_uint(thr);
            ws[RFW_VAL + 2 * t] = __float_as_uint(a);
            ws[RFW_VAL + 2 * t + 1] = __float_as_uint(bb);
        }
    }
    __syncthreads();
    rf_stamp(ws, 5);
    const float tdrop = S.thr[0];
    const float tlo = A.nspec > 1 ? S.thr[1] : 0.f, thi = A.nspec > 2 ? S.thr[2] : 0.f;
    unsigned kept = 0;
    auto apply = [&](int pb, int a, float e, unsigned lb) {
        const int n = pb >> 20, oy = (pb >> 10) & 1023, ox = (pb & 1023) + a;
        const long p = ((long)n * A.H + oy) * A.W + ox;
        long long l = (long long)lb;
        if (l == (long long)(A.ignore & 255)) l = A.ignore;
        if (e >= tdrop && l != A.ignore) l = A.ignore;
        A.ent[p] = e;            // (stored here, not in phase A: keeps the L2 clean for the barriers' write-backs)
        A.target_u[p] = l;
        kept += l != A.ignore;
        if (A.nspec > 1) {
            const int ly = S.invy[oy], lx = S.invx[ox];
            if (ly >= 0 && lx >= 0) {
                const long q = ((long)(A.B + n) * A.hm + ly) * A.wm + lx;
                A.low_mask[q] = e <= tlo ? 1.f : 0.f;
                A.high_mask[q] = A.neg_high ? (e >= thi ? 1.f : 0.f) : 1.f;
            }
        }
    };
#pragma unroll
    for (int it = 0; it < RF_NIT; ++it) {
        if (pbase[it] < 0) continue;
        const int nx = min(4, A.W - (pbase[it] & 1023));
#pragma unroll
        for (int a = 0; a < 4; ++a)
            if (a < nx) apply(pbase[it], a, er[it][a], (labr[it] >> (8 * a)) & 255u);
    }
    if (pb1 >= 0) apply(pb1, 0, er1, lab1);
    rf_stamp(ws, 6);
    kept = wave_sum_u(kept);
    if (lane == 0) S.wsum[wave] = kept;
    __syncthreads();
    if (t == 0) {
        unsigned tot = 0;
        for (int w2 = 0; w2 < 16; ++w2) tot += S.wsum[w2];
        if (tot) atomicAdd(ws + RFW_NKEPT, tot);
    }
    rf_stamp(ws, 7);
}

// logits_low: strided (B, C, h, w) view of the TRAIN-mode teacher logits of the unlabeled half; H-1 == 4(h-1), W-1 == 4(w-1).
// q32[nspec]: percentiles / 100 in float32 (host values): [0] drop, [1] alpha_t, [2] 100 - alpha_t; nspec = 1 (no contrastive
// branch: only target_u is written) or 3.  workspace: u2pl_reliability_fused_workspace_bytes(G) bytes, ZEROED by the caller;
// (zeroed ONCE; `epoch` = 0, 1, 2, ... counts the launches on this workspace -- barrier counters only ever grow and the
// bin totals are double-buffered by launch parity, so one buffer per stream is reused step after step with no reset);
// cand: B*H*W floats of scratch.  Returns U2PL_EINVAL when the shape does not fit the fused kernel (caller falls back to
// u2pl_entropy_up_f32 + u2pl_select_f32 + u2pl_reliability_apply).  Thresholds land in workspace words 16..18.
U2PL_API int u2pl_reliability_fused(const float* logits_low, long sn, long sc, long sh, long sw, int B, int C, int h,
                                    int w, int H, int W, const long long* label_u, const long long* label_l,
                                    int ignore, int nspec, const float* q32_host, int negative_high_entropy, int hm,
                                    int wm, float* entropy, long long* target_u, float* low_mask, float* high_mask,
                                    unsigned* lbits, unsigned* workspace, float* cand, int G, unsigned epoch, hipStream_t stream) {
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
    const size_t lds_a = (size_t)4 * (21 + 1) * RF_CELLS * sizeof(float), lds = lds_a > (size_t)RF_CAP * 4 ? lds_a : (size_t)RF_CAP * 4;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)k_reliability_fused<19>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)k_reliability_fused<21>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    if (C == 19) hipLaunchKernelGGL(k_reliability_fused<19>, dim3(G), dim3(RF_T), lds, stream, A);
    else hipLaunchKernelGGL(k_reliability_fused<21>, dim3(G), dim3(RF_T), lds, stream, A);
    U2PL_LAUNCH_CHECK();
    return 0;
}
