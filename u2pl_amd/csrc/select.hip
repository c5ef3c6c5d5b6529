// Exact order-statistic selection on device (replaces D2H + np.percentile,
// train_semi.py:405-407,412-415, loss_helper.py:38-40, and OHEM's full sort, loss_helper.py:521-526).
//
// 3-pass radix select on the order-preserving fp32 key: 11 + 11 + 10 bits.  The producer of the values
// (entropy / OHEM kernels) builds the pass-0 histogram for free; passes 1 and 2 each make one sweep over
// the 4.7 MB value array (L2/MALL resident).  There are NO separate "resolve" launches: every block
// re-derives the selected prefixes from the global histograms of the earlier passes in its prologue
// (integer-only, deterministic), so the whole selection is: [producer+hist0] -> pass1 -> pass2 -> finish.
//
// Workspace (uint32 words; zeroed by the caller before the producer runs):
//   [0] n_valid (non-NaN)   [1] n_total   [40+s] selected value of slot s (float bits)
//   [56+j] threshold j (float bits)   [64+j] gamma_j (float bits)
//   [128 ..)            hist0[2048]
//   [128+2048 ..)       hist1[8 slots][2048]
//   [128+9*2048 ..)     hist2[8 slots][1024]
#include "common.h"
#include "u2pl_hip.h"

#define SEL_VAL 40
#define SEL_THR 56
#define SEL_GAMMA 64
#define SEL_H0 128
#define SEL_H1 (SEL_H0 + 2048)
#define SEL_H2 (SEL_H1 + U2PL_SEL_MAX_SLOTS * 2048)
#define SEL_WORDS (SEL_H2 + U2PL_SEL_MAX_SLOTS * 1024)
#define MAXS U2PL_SEL_MAX_SLOTS

U2PL_API size_t u2pl_select_workspace_bytes(void) { return (size_t)SEL_WORDS * sizeof(unsigned); }

struct SelState {            // lives in LDS
    unsigned rank[MAXS];     // remaining rank inside the current prefix
    unsigned prefix[MAXS];   // selected key bits so far (high bits)
    int leader[MAXS];        // first slot carrying the same prefix (owner of the histogram)
    unsigned nd[MAXS], nk[MAXS];
    float gamma[MAXS / 2];
    unsigned wsum[MAXS][4];
};

__device__ void set_leaders(SelState& S, int nslots) {
    if (threadIdx.x < nslots) {
        int s = threadIdx.x, l = s;
        for (int u = 0; u < s; ++u)
            if (S.prefix[u] == S.prefix[s]) { l = u; break; }
        S.leader[s] = l;
    }
    __syncthreads();
}

// One level of the chain for ALL slots at once (256 threads).  Every leader's histogram is fetched with
// b128 loads issued back to back (one memory latency per level, not one per leader), scanned with wave64
// shuffle scans + one LDS hop, and every slot locates the bin where the cumulative count crosses its rank
// from the register copy.  Three barriers per level, independent of the number of slots.
template <int PER>   // bins per thread: 8 (2048 bins) or 4 (1024 bins)
__device__ void resolve_level(const unsigned* __restrict__ hbase, int per_leader_stride, int shift, int nslots,
                              SelState& S) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    unsigned h[MAXS][PER];
    bool lead[MAXS];
#pragma unroll
    for (int l = 0; l < MAXS; ++l) {
        lead[l] = l < nslots && S.leader[l] == l;   // uniform (LDS)
        if (lead[l]) {
            const uint4* hp = (const uint4*)(hbase + (long)l * per_leader_stride + t * PER);
#pragma unroll
            for (int i = 0; i < PER / 4; ++i) {
                const uint4 x = hp[i];
                h[l][4 * i] = x.x; h[l][4 * i + 1] = x.y; h[l][4 * i + 2] = x.z; h[l][4 * i + 3] = x.w;
            }
        }
    }
    unsigned loc[MAXS], scan[MAXS];
#pragma unroll
    for (int l = 0; l < MAXS; ++l) {
        loc[l] = scan[l] = 0;
        if (lead[l]) {
#pragma unroll
            for (int i = 0; i < PER; ++i) loc[l] += h[l][i];
            unsigned v = loc[l];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned u = __shfl_up(v, o, 64);
                if (lane >= o) v += u;
            }
            scan[l] = v;
            if (lane == 63) S.wsum[l][wave] = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int l = 0; l < MAXS; ++l) {
        if (!lead[l]) continue;
        unsigned base = 0;
        for (int w2 = 0; w2 < wave; ++w2) base += S.wsum[l][w2];
        const unsigned incl = base + scan[l], excl = incl - loc[l];
        for (int s = l; s < nslots; ++s) {
            if (S.leader[s] != l) continue;
            const unsigned k = S.rank[s];
            if (k >= excl && k < incl) {
                unsigned run = excl;
                bool done = false;
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    if (!done && k < run + h[l][i]) { S.nd[s] = t * PER + i; S.nk[s] = k - run; done = true; }
                    run += h[l][i];
                }
            }
        }
    }
    __syncthreads();
    if (t < nslots) {
        S.prefix[t] |= S.nd[t] << shift;
        S.rank[t] = S.nk[t];
    }
    __syncthreads();
    set_leaders(S, nslots);
}

// ranks from n_valid / n_total (numpy: virtual index (n-1)*q in float32), then the prefixes selected by
// passes 0..upto-1.  upto = 1: after hist0; 2: after hist1; 3: after hist2 (full key).
__device__ void resolve_chain(int upto, int nspec, const int* __restrict__ kind, const float* __restrict__ q32,
                              const long long* __restrict__ kparam, const unsigned* __restrict__ ws, SelState& S) {
    const int nslots = 2 * nspec, t = threadIdx.x;
    if (t < nspec) {
        const unsigned nv = ws[0], nt = ws[1];
        long lo, hi;
        float gamma = 0.f;
        if (kind[t] == 0) {
            const long n = nv;
            const float vi = __fmul_rn((float)(n - 1), q32[t]);
            const float fl = floorf(vi);
            gamma = __fsub_rn(vi, fl);
            if (n <= 0) lo = hi = 0;
            else if (!(vi == vi) || vi >= (float)(n - 1)) lo = hi = n - 1;
            else if (vi < 0.f) lo = hi = 0;
            else { lo = (long)fl; hi = lo + 1; }
        } else {
            const long k = kparam[t], n = nt;
            lo = hi = (k < n ? k : n) - 1;
            if (lo < 0) lo = hi = 0;
        }
        S.rank[2 * t] = (unsigned)lo;
        S.rank[2 * t + 1] = (unsigned)hi;
        S.prefix[2 * t] = S.prefix[2 * t + 1] = 0;
        S.gamma[t] = gamma;
    }
    if (t < MAXS) { S.leader[t] = 0; S.nd[t] = 0; S.nk[t] = 0; }   // pass 0: one shared histogram (owner 0)
    __syncthreads();
    resolve_level<8>(ws + SEL_H0, 0, 21, nslots, S);
    if (upto < 2) return;
    resolve_level<8>(ws + SEL_H1, 2048, 10, nslots, S);
    if (upto < 3) return;
    resolve_level<4>(ws + SEL_H2, 1024, 0, nslots, S);
}

// pass-0 histogram for callers that hand over a plain value array
__global__ void k_sel_hist0(const float* __restrict__ v, long n, unsigned* __restrict__ ws) {
    __shared__ unsigned sh[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) sh[i] = 0;
    __syncthreads();
    const long n4 = ((reinterpret_cast<uintptr_t>(v) & 15) == 0) ? (n >> 2) : 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 x = ((const float4*)v)[i];
        atomicAdd(&sh[f32_key(x.x) >> 21], 1u); atomicAdd(&sh[f32_key(x.y) >> 21], 1u);
        atomicAdd(&sh[f32_key(x.z) >> 21], 1u); atomicAdd(&sh[f32_key(x.w) >> 21], 1u);
    }
    for (long i = 4 * n4 + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        atomicAdd(&sh[f32_key(v[i]) >> 21], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += blockDim.x)
        if (sh[i]) atomicAdd(&ws[SEL_H0 + i], sh[i]);
}

// PASS = 1: digit bits [20:10] of keys whose bits [31:21] match; PASS = 2: bits [9:0] under a 22-bit prefix
template <int PASS>
__global__ void k_sel_pass(const float* __restrict__ v, long n, int nspec, const int* __restrict__ kind,
                           const float* __restrict__ q32, const long long* __restrict__ kparam,
                           unsigned* __restrict__ ws) {
    constexpr int NB = PASS == 1 ? 2048 : 1024;
    constexpr int SHIFT = PASS == 1 ? 21 : 10;          // prefix = key >> SHIFT
    constexpr int GROUP = 4;                            // distinct prefixes histogrammed per sweep
    __shared__ SelState S;
    __shared__ unsigned sh[GROUP * NB];
    resolve_chain(PASS, nspec, kind, q32, kparam, ws, S);
    const int nslots = 2 * nspec;
    // compact list of leader slots
    __shared__ int lead[MAXS];
    __shared__ int nlead;
    if (threadIdx.x == 0) {
        int c = 0;
        for (int s = 0; s < nslots; ++s)
            if (S.leader[s] == s) lead[c++] = s;
        nlead = c;
    }
    __syncthreads();
    unsigned* gh = ws + (PASS == 1 ? SEL_H1 : SEL_H2);
    for (int base = 0; base < nlead; base += GROUP) {
        const int cnt = min(GROUP, nlead - base);
        for (int i = threadIdx.x; i < cnt * NB; i += blockDim.x) sh[i] = 0;
        unsigned pf[GROUP];
#pragma unroll
        for (int j = 0; j < GROUP; ++j) pf[j] = j < cnt ? (S.prefix[lead[base + j]] >> SHIFT) : 0xffffffffu;
        __syncthreads();
        auto put = [&](float f) {
            const unsigned k = f32_key(f);
            const unsigned hi = k >> SHIFT;
            const unsigned d = PASS == 1 ? ((k >> 10) & 2047u) : (k & 1023u);
#pragma unroll
            for (int j = 0; j < GROUP; ++j)
                if (hi == pf[j]) atomicAdd(&sh[j * NB + d], 1u);
        };
        const long n4 = ((reinterpret_cast<uintptr_t>(v) & 15) == 0) ? (n >> 2) : 0;
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
            const float4 x = ((const float4*)v)[i];
            put(x.x); put(x.y); put(x.z); put(x.w);
        }
        for (long i = 4 * n4 + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
            put(v[i]);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt * NB; i += blockDim.x)
            if (sh[i]) atomicAdd(&gh[lead[base + i / NB] * NB + (i % NB)], sh[i]);
        __syncthreads();
    }
}

// thresholds: numpy _lerp in float32 (no FMA): d=b-a; t>=.5 ? b-d*(1-t) : a+d*t.
// kind 1 (OHEM): thr = kth > floor ? kth : floor, or +inf when min_kept > n_valid (loss_helper.py:513-515)
__global__ void k_sel_finish(int nspec, const int* __restrict__ kind, const float* __restrict__ q32,
                             const long long* __restrict__ kparam, const float* __restrict__ fparam,
                             unsigned* __restrict__ ws) {
    __shared__ SelState S;
    resolve_chain(3, nspec, kind, q32, kparam, ws, S);
    const int j = threadIdx.x;
    if (j < 2 * nspec) ws[SEL_VAL + j] = __float_as_uint(key_f32(S.prefix[j]));
    if (j >= nspec) return;
    const float a = key_f32(S.prefix[2 * j]), b = key_f32(S.prefix[2 * j + 1]);
    float thr;
    if (kind[j] == 0) {
        const float t = S.gamma[j];
        const float d = __fsub_rn(b, a);
        thr = (t >= 0.5f) ? __fsub_rn(b, __fmul_rn(d, __fsub_rn(1.0f, t))) : __fadd_rn(a, __fmul_rn(d, t));
        if (ws[0] == 0) thr = __uint_as_float(0x7fc00000u);
    } else {
        const long long nv = ws[0];
        if (kparam[j] > nv) thr = __uint_as_float(0x7f800000u);
        else thr = a > fparam[j] ? a : fparam[j];
    }
    ws[SEL_THR + j] = __float_as_uint(thr);
    ws[SEL_GAMMA + j] = __float_as_uint(S.gamma[j]);
}

// hist0_done != 0: the producer (u2pl_entropy*_f32 / u2pl_ohem_prob_f32) already accumulated hist0 in ws
U2PL_API int u2pl_select_f32(const float* values, long n, int nspec, const int* spec_kind, const float* q32,
                             const long long* kparam, const float* fparam, unsigned* ws, int hist0_done,
                             hipStream_t stream) {
    if (nspec < 1 || 2 * nspec > MAXS) return U2PL_EINVAL;
    const int grid = grid_for(n / 4 + 1, 256, 256);   // one block per CU: fewer histogram flushes
    if (!hist0_done) {
        U2PL_LAUNCH(k_sel_hist0, dim3(grid), dim3(256), 0, stream, values, n, ws);
        U2PL_LAUNCH_CHECK();
    }
    U2PL_LAUNCH(k_sel_pass<1>, dim3(grid), dim3(256), 0, stream, values, n, nspec, spec_kind, q32, kparam, ws);
    U2PL_LAUNCH_CHECK();
    U2PL_LAUNCH(k_sel_pass<2>, dim3(grid), dim3(256), 0, stream, values, n, nspec, spec_kind, q32, kparam, ws);
    U2PL_LAUNCH_CHECK();
    U2PL_LAUNCH(k_sel_finish, dim3(1), dim3(256), 0, stream, nspec, spec_kind, q32, kparam, fparam, ws);
    U2PL_LAUNCH_CHECK();
    return 0;
}
