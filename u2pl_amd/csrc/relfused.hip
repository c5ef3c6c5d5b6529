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
#define RFW_CAND 8              // [6] append counters
#define RFW_THR 16              // [3] thresholds (float bits), [6] selected values at 24..29
#define RFW_VAL 24
#define RFW_TOT 1024            // [2048] totals
#define RFW_SLAB 4096           // [G][2048]

struct RfArgs {
    const float* in; long sn, sc, sh, sw;
    int B, h, w, H, W;
    float sy, sx, ny, nx;
    int hm, wm;                  // size of the low-res masks (= student prediction map)
    const long long* label_u; const long long* label_l;
    int ignore, nspec, neg_high;
    float q32[3];
    float bin_scale;
    float* ent; long long* target_u; float* low_mask; float* high_mask; unsigned* lbits;
    unsigned* ws; float* cand;
};

U2PL_API size_t u2pl_reliability_fused_workspace_bytes(int G) { return (size_t)(RFW_SLAB + (size_t)G * RF_BINS) * sizeof(unsigned); }

__device__ __forceinline__ int rf_bin(float e, float scale) {
    // monotone non-decreasing in e (e is not NaN); bin 0: e < 2^-22 (incl. the tiny negatives of rounding)
    if (!(e >= 2.384185791015625e-07f)) return 0;
    if (e < 0.015625f) return 1 + (int)((__float_as_uint(e) - 0x34800000u) >> 17);     // 16 octaves x 64 bins -> 1..1024
    const int b = (int)(__fmul_rn(__fsub_rn(e, 0.015625f), scale));
    return 1025 + (b < 1022 ? b : 1022);
}

__device__ __forceinline__ void rf_grid_sync(unsigned* ctr, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
}
__device__ __forceinline__ void rf_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned rf_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct RfShared {
    unsigned hist[RF_BINS];
    unsigned wsum[16];
    unsigned rank[RF_MAXSLOT], sbin[RF_MAXSLOT], srin[RF_MAXSLOT], skey[RF_MAXSLOT];
    float gamma[3];
    int nd; unsigned dbin[RF_MAXSLOT], doff[RF_MAXSLOT], dcnt[RF_MAXSLOT], dblk[RF_MAXSLOT], dbase[RF_MAXSLOT];
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

template <int CT>
__global__ __launch_bounds__(RF_T, 1) void k_reliability_fused(RfArgs A) {
    extern __shared__ float dyn[];                  // phase A: corner logits [4][CT][RF_CELLS]; phase D: candidate keys
    __shared__ RfShared S;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int G = gridDim.x, b = blockIdx.x;
    unsigned* ws = A.ws;
    const long HW = (long)A.H * A.W;
    // ---------------------------------------------------------------- phase A
    S.hist[2 * t] = 0; S.hist[2 * t + 1] = 0;
    S.invy[t] = -1; S.invx[t] = -1;
    __syncthreads();
    if (t < A.hm) S.invy[nearest_src(t, A.ny, A.H)] = (short)t;
    if (t < A.wm) S.invx[nearest_src(t, A.nx, A.W)] = (short)t;
    const long ncell = (long)A.B * A.h * A.w;
    const long per = (ncell + G - 1) / G;
    const long c0 = (long)b * per, c1 = min(ncell, c0 + per);
    const int cl = t & (RF_CELLS - 1), part = t >> 8;
    float er[RF_NIT][4];
    unsigned labr[RF_NIT];          // 4 label bytes of the item's pixels
    long pbase[RF_NIT];             // flat index of the item's first pixel, -1: no item
    float (*cv)[CT][RF_CELLS] = (float (*)[CT][RF_CELLS])dyn;
#pragma unroll
    for (int it = 0; it < RF_NIT; ++it) {
        const long q = c0 + (long)it * RF_CELLS + cl;
        const bool live = q < c1;
        const long qq = live ? q : (ncell - 1);
        const int cj = (int)(qq % A.w);
        const long t0 = qq / A.w;
        const int ci = (int)(t0 % A.h), n = (int)(t0 / A.h);
        pbase[it] = -1;
        labr[it] = 0;
#pragma unroll
        for (int a = 0; a < 4; ++a) er[it][a] = __uint_as_float(0x7fc00000u);
        const bool any = c0 + (long)it * RF_CELLS < c1;     // block-uniform
        if (!any) continue;
        const AcCoord cy0 = ac_coord(min(ci * 4, A.H - 1), A.sy, A.h), cx0 = ac_coord(min(cj * 4, A.W - 1), A.sx, A.w);
        {
            const int yy = (part & 2) ? cy0.i1 : cy0.i0, xx = (part & 1) ? cx0.i1 : cx0.i0;
            const float* src = A.in + n * A.sn + yy * A.sh + xx * A.sw;
#pragma unroll
            for (int c = 0; c < CT; ++c) cv[part][c][cl] = src[c * A.sc];
        }
        __syncthreads();
        const int oy = ci * 4 + part, ox0 = cj * 4;
        if (live && oy < A.H) {
            const AcCoord cy = ac_coord(oy, A.sy, A.h);
            float lx0[4], lx1[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const AcCoord cx = ac_coord(min(ox0 + a, A.W - 1), A.sx, A.w);
                lx0[a] = cx.l0; lx1[a] = cx.l1;
            }
            float m[4], s[4], tt[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { m[a] = -INFINITY; s[a] = 0.f; tt[a] = 0.f; }
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const float v00 = cv[0][c][cl], v01 = cv[1][c][cl], v10 = cv[2][c][cl], v11 = cv[3][c][cl];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const float top = __fmaf_rn(lx0[a], v00, __fmul_rn(lx1[a], v01));
                    const float bot = __fmaf_rn(lx0[a], v10, __fmul_rn(lx1[a], v11));
                    m[a] = fmaxf(m[a], __fmaf_rn(cy.l0, top, __fmul_rn(cy.l1, bot)));
                }
            }
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const float v00 = cv[0][c][cl], v01 = cv[1][c][cl], v10 = cv[2][c][cl], v11 = cv[3][c][cl];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const float top = __fmaf_rn(lx0[a], v00, __fmul_rn(lx1[a], v01));
                    const float bot = __fmaf_rn(lx0[a], v10, __fmul_rn(lx1[a], v11));
                    const float z = __fmaf_rn(cy.l0, top, __fmul_rn(cy.l1, bot)) - m[a];
                    const float e = expf(z);
                    s[a] += e;
                    tt[a] += e * z;
                }
            }
            const int nx = min(4, A.W - ox0);
            const long p0 = ((long)n * A.H + oy) * A.W + ox0;
            pbase[it] = p0;
#pragma unroll
            for (int a = 0; a < 4; ++a)
                if (a < nx) {
                    const long long l = A.label_u[p0 + a];
                    const bool valid = l != (long long)A.ignore;
                    labr[it] |= ((unsigned)l & 255u) << (8 * a);
                    float e = logf(s[a]) - tt[a] / s[a];
                    e = valid ? e : __uint_as_float(0x7fc00000u);
                    er[it][a] = e;
                    A.ent[p0 + a] = e;
                    if (valid) atomicAdd(&S.hist[rf_bin(e, A.bin_scale)], 1u);
                } else {
                    labr[it] |= ((unsigned)A.ignore & 255u) << (8 * a);
                }
        }
        __syncthreads();
    }
    {
        unsigned* slab = ws + RFW_SLAB + (size_t)b * RF_BINS;
        rf_st(slab + 2 * t, S.hist[2 * t]);
        rf_st(slab + 2 * t + 1, S.hist[2 * t + 1]);
    }
    rf_grid_sync(ws + RFW_BAR, (unsigned)G);
    // ---------------------------------------------------------------- phase B: totals of my bins
    {
        const int bpb = RF_BINS / G;                 // G is a power of two <= 256 -> bpb >= 8
        unsigned acc = 0;
        // thread -> (bin local = t % bpb... ) keep all lanes of a wave on one bin: wave-sum, then one LDS add
        const int per_bin_threads = RF_T / bpb;       // >= 4 ... a multiple of 64 when bpb <= 16
        const int bl = t / per_bin_threads, j = t % per_bin_threads;
        for (int sb = j; sb < G; sb += per_bin_threads) acc += rf_ld(ws + RFW_SLAB + (size_t)sb * RF_BINS + b * bpb + bl);
        S.hist[2 * t] = 0; S.hist[2 * t + 1] = 0;
        __syncthreads();
        acc = wave_sum_u(acc);
        if (lane == 0 && acc) atomicAdd(&S.hist[bl], acc);
        __syncthreads();
        if (t < bpb) rf_st(ws + RFW_TOT + b * bpb + t, S.hist[t]);
    }
    rf_grid_sync(ws + RFW_BAR, 2u * G);
    // ---------------------------------------------------------------- phase C: ranks -> bins -> candidates
    const unsigned h0 = rf_ld(ws + RFW_TOT + 2 * t), h1 = rf_ld(ws + RFW_TOT + 2 * t + 1);
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
        if (k >= ex && k < ex + h0) { S.sbin[s] = 2 * t; S.srin[s] = k - ex; }
        else if (k >= ex + h0 && k < ex + h0 + h1) { S.sbin[s] = 2 * t + 1; S.srin[s] = k - ex - h0; }
    }
    __syncthreads();
    if (t == 0) {
        int nd = 0;
        unsigned off = 0;
        if (nvalid)
            for (int s = 0; s < nslot; ++s) {
                bool seen = false;
                for (int d = 0; d < nd; ++d) seen |= S.dbin[d] == S.sbin[s];
                if (!seen) { S.dbin[nd] = S.sbin[s]; S.doff[nd] = off; S.dcnt[nd] = rf_ld(ws + RFW_TOT + S.sbin[s]); off += S.dcnt[nd]; S.dblk[nd] = 0; ++nd; }
            }
        S.nd = nd;
    }
    __syncthreads();
    const int nd = S.nd;
    unsigned mybin[RF_NIT][4];
#pragma unroll
    for (int it = 0; it < RF_NIT; ++it)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float e = er[it][a];
            mybin[it][a] = 0xffffffffu;
            if (e == e) {
                const unsigned bn = (unsigned)rf_bin(e, A.bin_scale);
                for (int d = 0; d < nd; ++d)
                    if (S.dbin[d] == bn) { mybin[it][a] = d; atomicAdd(&S.dblk[d], 1u); }
            }
        }
    __syncthreads();
    if (t < nd) { S.dbase[t] = S.dblk[t] ? atomicAdd(ws + RFW_CAND + t, S.dblk[t]) : 0u; S.dblk[t] = 0; }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < RF_NIT; ++it)
#pragma unroll
        for (int a = 0; a < 4; ++a)
            if (mybin[it][a] != 0xffffffffu) {
                const unsigned d = mybin[it][a];
                const unsigned pos = S.doff[d] + S.dbase[d] + atomicAdd(&S.dblk[d], 1u);
                rf_st((unsigned*)A.cand + pos, __float_as_uint(er[it][a]));
            }
    rf_grid_sync(ws + RFW_BAR, 3u * G);
    // ---------------------------------------------------------------- phase D: exact selection in LDS, thresholds
    unsigned* lkeys = (unsigned*)dyn;
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
    const float tdrop = S.thr[0];
    const float tlo = A.nspec > 1 ? S.thr[1] : 0.f, thi = A.nspec > 2 ? S.thr[2] : 0.f;
    unsigned kept = 0;
    const long lowplane = (long)A.hm * A.wm;
#pragma unroll
    for (int it = 0; it < RF_NIT; ++it) {
        if (pbase[it] < 0) continue;
        const long p0 = pbase[it];
        const int n = (int)(p0 / HW);
        const long rr = p0 % HW;
        const int oy = (int)(rr / A.W), ox0 = (int)(rr % A.W);
        const int nx = min(4, A.W - ox0);
        const int ly = S.invy[oy];
#pragma unroll
        for (int a = 0; a < 4; ++a)
            if (a < nx) {
                const float e = er[it][a];
                long long l = (long long)((labr[it] >> (8 * a)) & 255u);
                if (l == (long long)(A.ignore & 255)) l = A.ignore;
                if (e >= tdrop && l != A.ignore) l = A.ignore;
                A.target_u[p0 + a] = l;
                kept += l != A.ignore;
                if (A.nspec > 1 && ly >= 0) {
                    const int lx = S.invx[ox0 + a];
                    if (lx >= 0) {
                        const long q = ((long)(A.B + n) * A.hm + ly) * A.wm + lx;
                        A.low_mask[q] = e <= tlo ? 1.f : 0.f;
                        A.high_mask[q] = A.neg_high ? (e >= thi ? 1.f : 0.f) : 1.f;
                    }
                }
            }
    }
    if (A.nspec > 1) {
        // label-only outputs: labeled-half masks and the (quirky) class bits of both halves
        const long nlow = (long)2 * A.B * lowplane;
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
    kept = wave_sum_u(kept);
    if (lane == 0) S.wsum[wave] = kept;
    __syncthreads();
    if (t == 0) {
        unsigned tot = 0;
        for (int w2 = 0; w2 < 16; ++w2) tot += S.wsum[w2];
        if (tot) atomicAdd(ws + RFW_NKEPT, tot);
    }
}

// logits_low: strided (B, C, h, w) view of the TRAIN-mode teacher logits of the unlabeled half; H-1 == 4(h-1), W-1 == 4(w-1).
// q32[nspec]: percentiles / 100 in float32 (host values): [0] drop, [1] alpha_t, [2] 100 - alpha_t; nspec = 1 (no contrastive
// branch: only target_u is written) or 3.  workspace: u2pl_reliability_fused_workspace_bytes(G) bytes, ZEROED by the caller;
// cand: B*H*W floats of scratch.  Returns U2PL_EINVAL when the shape does not fit the fused kernel (caller falls back to
// u2pl_entropy_up_f32 + u2pl_select_f32 + u2pl_reliability_apply).  Thresholds land in workspace words 16..18.
U2PL_API int u2pl_reliability_fused(const float* logits_low, long sn, long sc, long sh, long sw, int B, int C, int h,
                                    int w, int H, int W, const long long* label_u, const long long* label_l,
                                    int ignore, int nspec, const float* q32_host, int negative_high_entropy, int hm,
                                    int wm, float* entropy, long long* target_u, float* low_mask, float* high_mask,
                                    unsigned* lbits, unsigned* workspace, float* cand, int G, hipStream_t stream) {
    if (!(C == 19 || C == 21) || (nspec != 1 && nspec != 3)) return U2PL_EINVAL;
    if (h < 2 || w < 2 || H - 1 != 4 * (h - 1) || W - 1 != 4 * (w - 1) || H > 1024 || W > 1024) return U2PL_EINVAL;
    if (hm > H || wm > W || hm > 1024 || wm > 1024 || ignore < 0 || ignore > 255) return U2PL_EINVAL;
    if (G < 8 || G > 256 || (G & (G - 1))) return U2PL_EINVAL;
    if ((long)B * h * w > (long)G * RF_NIT * RF_CELLS) return U2PL_EINVAL;
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
    A.ws = workspace; A.cand = cand;
    const size_t lds = (size_t)4 * 21 * RF_CELLS * sizeof(float) > (size_t)RF_CAP * 4 ? (size_t)4 * 21 * RF_CELLS * sizeof(float) : (size_t)RF_CAP * 4;
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
