// Winograd minimal-filtering form of the stride-1 3x3 (dilated, "same") convolutions of the ResNet /
// DeepLabv3+ stack (reference: nn.Conv2d(k=3, padding=d, dilation=d) in u2pl/models/resnet.py:25-41,
// base.py:54-83, decoder.py:60-106,132-138; the reference's cuDNN path picks the same algorithm class for
// fp32).  All arithmetic is fp32; the component products run on the fp32 matrix cores through the batched
// implicit-GEMM kernel of conv.hip (u2pl_gemm_batched_f32).
//
//   F(m x m, 3 x 3), m = 2 or 4, a = m + 2:   Y = A^T [ (G g G^T) . (B^T d B) ] A     (Lavin & Gray)
//
// Dilation d is handled by polyphase decomposition: the outputs with (oy, ox) = (py, px) mod d form an
// undilated pad-1 convolution over the sub-image x[py + d*u, px + d*v]; tiles are laid over the sub-images.
//   tile id t = (((n*d + py)*d + px)*Ty + ty)*Tx + tx        Ty = ceil(ceil(H/d)/m), Tx likewise
//   V  [a*a][tiles][Cin]   transformed input  (GEMM A operand, K = Cin contiguous)
//   U  [a*a][Cout][Cin]    transformed weights (GEMM B operand)
//   Mb [a*a][tiles][Cout]  component products
// The data-gradient of such a convolution is the same convolution with the taps rotated by 180 degrees and
// the channel roles swapped: `transposed` in the weight transform produces U'[a*a][Cin][Cout] for it.
#include "common.h"
#include "u2pl_hip.h"

struct WinoGeom {
    int N, H, W, C, dil, Ty, Tx;
    long tiles;
};

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4mul(float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }
// a + s*b written as separate multiply and add (-ffp-contract=off keeps it that way)
__device__ __forceinline__ float4 f4axpy(float4 a, float s, float4 b) { return f4add(a, f4mul(s, b)); }

// B^T d for one 6-vector / 4-vector of float4 (applied to columns, then to rows)
template <int MT> struct WinoT;
template <> struct WinoT<4> {
    static constexpr int A = 6;
    __device__ static __forceinline__ void bt(const float4 (&d)[6], float4 (&r)[6]) {
        const float4 p = f4axpy(d[4], -4.f, d[2]);   // d4 - 4 d2
        const float4 q = f4axpy(d[3], -4.f, d[1]);   // d3 - 4 d1
        const float4 s = f4sub(d[4], d[2]);          // d4 - d2
        const float4 t = f4mul(2.f, f4sub(d[3], d[1]));
        r[0] = f4add(f4axpy(f4mul(4.f, d[0]), -5.f, d[2]), d[4]);
        r[1] = f4add(p, q);
        r[2] = f4sub(p, q);
        r[3] = f4add(s, t);
        r[4] = f4sub(s, t);
        r[5] = f4add(f4axpy(f4mul(4.f, d[1]), -5.f, d[3]), d[5]);
    }
    // A^T m : 6 -> 4
    __device__ static __forceinline__ void at(const float4 (&m)[6], float4 (&y)[4]) {
        const float4 s12 = f4add(m[1], m[2]), d12 = f4sub(m[1], m[2]);
        const float4 s34 = f4add(m[3], m[4]), d34 = f4sub(m[3], m[4]);
        y[0] = f4add(f4add(m[0], s12), s34);
        y[1] = f4axpy(d12, 2.f, d34);
        y[2] = f4axpy(s12, 4.f, s34);
        y[3] = f4add(f4axpy(d12, 8.f, d34), m[5]);
    }
    // A v : 4 -> 6 (transpose of the output transform: weight-gradient side)
    __device__ static __forceinline__ void av(const float4 (&v)[4], float4 (&r)[6]) {
        const float4 e = f4add(v[0], v[2]), o = f4add(v[1], v[3]);
        const float4 e4 = f4axpy(v[0], 4.f, v[2]), o4 = f4axpy(f4mul(2.f, v[1]), 8.f, v[3]);
        r[0] = v[0];
        r[1] = f4add(e, o);
        r[2] = f4sub(e, o);
        r[3] = f4add(e4, o4);
        r[4] = f4sub(e4, o4);
        r[5] = v[3];
    }
    // G^T v : 6 -> 3 (scalar)
    __device__ static __forceinline__ void gt(const float (&v)[6], float (&r)[3]) {
        const float s12 = v[1] + v[2], s34 = v[3] + v[4];
        r[0] = v[0] * 0.25f - s12 * (1.f / 6.f) + s34 * (1.f / 24.f);
        r[1] = (v[2] - v[1]) * (1.f / 6.f) + (v[3] - v[4]) * (1.f / 12.f);
        r[2] = (s34 - s12) * (1.f / 6.f) + v[5];
    }
    // G g : 3 -> 6 (scalar)
    __device__ static __forceinline__ void gg(const float (&g)[3], float (&r)[6]) {
        const float a = (g[0] + g[2]) * (-1.f / 6.f), b = g[1] * (1.f / 6.f);
        const float c = g[0] * (1.f / 24.f) + g[2] * (1.f / 6.f), e = g[1] * (1.f / 12.f);
        r[0] = g[0] * 0.25f;
        r[1] = a - b;
        r[2] = a + b;
        r[3] = c + e;
        r[4] = c - e;
        r[5] = g[2];
    }
};
template <> struct WinoT<2> {
    static constexpr int A = 4;
    __device__ static __forceinline__ void bt(const float4 (&d)[4], float4 (&r)[4]) {
        r[0] = f4sub(d[0], d[2]);
        r[1] = f4add(d[1], d[2]);
        r[2] = f4sub(d[2], d[1]);
        r[3] = f4sub(d[1], d[3]);
    }
    __device__ static __forceinline__ void at(const float4 (&m)[4], float4 (&y)[2]) {
        y[0] = f4add(f4add(m[0], m[1]), m[2]);
        y[1] = f4sub(f4sub(m[1], m[2]), m[3]);
    }
    __device__ static __forceinline__ void av(const float4 (&v)[2], float4 (&r)[4]) {
        r[0] = v[0];
        r[1] = f4add(v[0], v[1]);
        r[2] = f4sub(v[0], v[1]);
        r[3] = f4mul(-1.f, v[1]);
    }
    __device__ static __forceinline__ void gt(const float (&v)[4], float (&r)[3]) {
        r[0] = v[0] + 0.5f * (v[1] + v[2]);
        r[1] = 0.5f * (v[1] - v[2]);
        r[2] = 0.5f * (v[1] + v[2]) + v[3];
    }
    __device__ static __forceinline__ void gg(const float (&g)[3], float (&r)[4]) {
        r[0] = g[0];
        r[1] = 0.5f * ((g[0] + g[2]) + g[1]);
        r[2] = 0.5f * ((g[0] + g[2]) - g[1]);
        r[3] = g[2];
    }
};

__device__ __forceinline__ void tile_coords(const WinoGeom& g, long t, int& n, int& py, int& px, int& ty, int& tx) {
    tx = (int)(t % g.Tx); t /= g.Tx;
    ty = (int)(t % g.Ty); t /= g.Ty;
    px = (int)(t % g.dil); t /= g.dil;
    py = (int)(t % g.dil);
    n = (int)(t / g.dil);
}

// ---- input transform: one thread per (tile, 4 channels) ----------------------------------------------------
template <int MT>
__global__ __launch_bounds__(256) U2PL_HBM_KERNEL void k_wino_input(const float* __restrict__ x, long ldx, unsigned xbytes, WinoGeom g,
                                                    float* __restrict__ V, unsigned* __restrict__ v_amax) {
    constexpr int A = WinoT<MT>::A;
    const int C4 = g.C >> 2;
    long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= g.tiles * C4) {
        if (!v_amax) return;
        idx = g.tiles * C4 - 1;      // (fused maximum: whole waves reach the publish; the surplus lanes redo the last element)
    }
    const int c4 = (int)(idx % C4);
    const long t = idx / C4;
    int n, py, px, ty, tx;
    tile_coords(g, t, n, py, px, ty, tx);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, xbytes);
    const int ldxb = (int)ldx * 4;
    const int iy0 = py + g.dil * (ty * MT - 1), ix0 = px + g.dil * (tx * MT - 1);
    float4 d[A][A];
#pragma unroll
    for (int i = 0; i < A; ++i) {
        const int iy = iy0 + i * g.dil;
        const bool oky = iy >= 0 && iy < g.H;
#pragma unroll
        for (int j = 0; j < A; ++j) {
            const int ix = ix0 + j * g.dil;
            const bool ok = oky && ix >= 0 && ix < g.W;
            const int off = ((n * g.H + iy) * g.W + ix) * ldxb + c4 * 16;
            d[i][j] = buf_load4(rx, ok ? off : OOB_OFF);
        }
    }
    // columns: t[:, j] = B^T d[:, j]
#pragma unroll
    for (int j = 0; j < A; ++j) {
        float4 col[A], r[A];
#pragma unroll
        for (int i = 0; i < A; ++i) col[i] = d[i][j];
        WinoT<MT>::bt(col, r);
#pragma unroll
        for (int i = 0; i < A; ++i) d[i][j] = r[i];
    }
    // rows: V[i, :] = B^T (t[i, :])^T   (t B == (B^T t^T)^T)
    const long comp_stride = g.tiles * (long)g.C;
    float* out = V + t * g.C + c4 * 4;
    unsigned am = 0u;
#pragma unroll
    for (int i = 0; i < A; ++i) {
        float4 r[A];
        WinoT<MT>::bt(d[i], r);
#pragma unroll
        for (int j = 0; j < A; ++j) {
            *(float4*)(out + (long)(i * A + j) * comp_stride) = r[j];
            am = amax_bits4(am, r[j]);
        }
    }
    if (v_amax) amax_wave_publish(am, v_amax);       // split-fp16: V is the component GEMMs' A operand
}

// ---- weight transform: one thread per (o, c) ------------------------------------------------------------------
// w: [O][3][3][C] (channels_last OIHW).  transposed = 0: U[comp][O][C] = G g G^T
// transposed = 1 (data gradient): taps rotated 180 degrees, U[comp][C][O]
template <int MT>
__device__ __forceinline__ void wino_weight_one(const float* __restrict__ w, int O, int C, int transposed, float* __restrict__ U, long idx) {
    constexpr int A = WinoT<MT>::A;
    // forward: c fastest (coalesced reads and writes); transposed: o fastest (coalesced U' writes, 4x the volume
    // of the reads)
    const int c = transposed ? (int)(idx / O) : (int)(idx % C), o = transposed ? (int)(idx % O) : (int)(idx / C);
    float gk[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int rr = transposed ? 2 - r : r, ss = transposed ? 2 - s : s;
            gk[r][s] = w[(((long)o * 3 + rr) * 3 + ss) * C + c];
        }
    float t[A][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        float col[3] = {gk[0][s], gk[1][s], gk[2][s]}, r[A];
        WinoT<MT>::gg(col, r);
#pragma unroll
        for (int i = 0; i < A; ++i) t[i][s] = r[i];
    }
    const long comp_stride = (long)O * C;
    const long base = transposed ? (long)c * O + o : (long)o * C + c;
#pragma unroll
    for (int i = 0; i < A; ++i) {
        float r[A];
        WinoT<MT>::gg(t[i], r);
#pragma unroll
        for (int j = 0; j < A; ++j) U[(long)(i * A + j) * comp_stride + base] = r[j];
    }
}
template <int MT>
__global__ void k_wino_weight(const float* __restrict__ w, int O, int C, int transposed, float* __restrict__ U) {
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= (long)O * C) return;
    wino_weight_one<MT>(w, O, C, transposed, U, idx);
}
// every filter transform of a model in ONE launch (the transforms of a step's optimizer update: u2pl_amd/nn.py presplit):
// job j covers the (o, c) pairs [begin_j, begin_j+1); same arithmetic per pair as k_wino_weight
struct WinoWeightJob { const float* w; float* U; long begin; int O, C, transposed, mt; };
__global__ void k_wino_weight_multi(const WinoWeightJob* __restrict__ jobs, int njobs, long total) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int lo = 0, hi = njobs - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].begin <= i) lo = mid; else hi = mid - 1;
        }
        const WinoWeightJob j = jobs[lo];
        if (j.mt == 4) wino_weight_one<4>(j.w, j.O, j.C, j.transposed, j.U, i - j.begin);
        else wino_weight_one<2>(j.w, j.O, j.C, j.transposed, j.U, i - j.begin);
    }
}

// ---- output transform: one thread per (tile, 4 output channels) -------------------------------------------
// Optionally produces the following BatchNorm's statistics (pivot-shifted column sums, the two-stage
// column-reduce partial format [nblk][2][O]): a block covers 256 / (O/4) consecutive tiles x all channels.
template <int MT>
__global__ __launch_bounds__(256) U2PL_HBM_KERNEL void k_wino_output(const float* __restrict__ Mb, WinoGeom g, int O,
                                                     const float* __restrict__ bias, float* __restrict__ y, long ldy,
                                                     float* __restrict__ stats, const float* __restrict__ pivot,
                                                     BnEpi epi) {
    constexpr int A = WinoT<MT>::A;
    __shared__ __attribute__((aligned(16))) float red[2][256][4];
    const int O4 = O >> 2;
    const int tpb = 256 / O4 > 0 ? 256 / O4 : 1;      // tiles per block when stats are requested
    long t;
    int o4;
    bool live;
    if (stats) {
        o4 = threadIdx.x % O4;
        const int tl = threadIdx.x / O4;
        t = (long)blockIdx.x * tpb + tl;
        live = tl < tpb && t < g.tiles;
    } else {
        const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
        o4 = (int)(idx % O4);
        t = idx / O4;
        live = t < g.tiles;
    }
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    unsigned am = 0u;
    if (live) {
        int n, py, px, ty, tx;
        tile_coords(g, t, n, py, px, ty, tx);
        const long comp_stride = g.tiles * (long)O;
        const float* in = Mb + t * O + o4 * 4;
        float4 m[A][A];
#pragma unroll
        for (int i = 0; i < A; ++i)
#pragma unroll
            for (int j = 0; j < A; ++j) m[i][j] = *(const float4*)(in + (long)(i * A + j) * comp_stride);
        float4 tmp[MT][A];
#pragma unroll
        for (int j = 0; j < A; ++j) {
            float4 col[A], r[MT];
#pragma unroll
            for (int i = 0; i < A; ++i) col[i] = m[i][j];
            WinoT<MT>::at(col, r);
#pragma unroll
            for (int i = 0; i < MT; ++i) tmp[i][j] = r[i];
        }
        const float* bp = bias + o4 * 4;
        const float* pp = pivot + o4 * 4;
        const float4 bv = bias ? make_float4(bp[0], bp[1], bp[2], bp[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 pv = (stats && pivot) ? make_float4(pp[0], pp[1], pp[2], pp[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        // fused eval-mode BN (+res, ReLU): same operations as k_bn_apply (see BnEpi in common.h); never with stats
        const bool bn_on = epi.mean != nullptr;
        float4 mu4 = bv, is4 = bv, ga4 = bv, be4 = bv;
        if (bn_on) {
            mu4 = *(const float4*)(epi.mean + o4 * 4); is4 = *(const float4*)(epi.invstd + o4 * 4);
            ga4 = *(const float4*)(epi.gamma + o4 * 4); be4 = *(const float4*)(epi.beta + o4 * 4);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float4 r[MT];
            WinoT<MT>::at(tmp[i], r);
            const int oy = py + g.dil * (ty * MT + i);
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const int ox = px + g.dil * (tx * MT + j);
                if (oy < g.H && ox < g.W) {
                    float4 v = f4add(r[j], bv);
                    const long pix = (long)(n * g.H + oy) * g.W + ox;
                    if (bn_on) {
                        v.x = (v.x - mu4.x) * is4.x * ga4.x + be4.x; v.y = (v.y - mu4.y) * is4.y * ga4.y + be4.y;
                        v.z = (v.z - mu4.z) * is4.z * ga4.z + be4.z; v.w = (v.w - mu4.w) * is4.w * ga4.w + be4.w;
                        if (epi.res) v = f4add(v, *(const float4*)(epi.res + pix * epi.ldr + o4 * 4));
                        if (epi.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    }
                    *(float4*)(y + pix * ldy + o4 * 4) = v;
                    am = amax_bits4(am, v);
                    const float4 dv = f4sub(v, pv);
                    s1 = f4add(s1, dv);
                    s2 = f4add(s2, make_float4(dv.x * dv.x, dv.y * dv.y, dv.z * dv.z, dv.w * dv.w));
                }
            }
        }
    }
    if (epi.y_amax) amax_wave_publish(am, epi.y_amax);      // (every thread of the block gets here)
    if (!stats) return;
    *(float4*)red[0][threadIdx.x] = s1;
    *(float4*)red[1][threadIdx.x] = s2;
    __syncthreads();
    // thread c < O sums its channel over the block's tiles in tile order (deterministic)
    for (int c = threadIdx.x; c < O; c += 256) {
        float a1 = 0.f, a2 = 0.f;
        for (int tl = 0; tl < tpb; ++tl) {
            a1 += red[0][tl * O4 + (c >> 2)][c & 3];
            a2 += red[1][tl * O4 + (c >> 2)][c & 3];
        }
        stats[(long)blockIdx.x * 2 * O + c] = a1;
        stats[(long)blockIdx.x * 2 * O + O + c] = a2;
    }
}

// ---- weight-gradient side ----------------------------------------------------------------------------------
// dU[comp] = sum_tiles (A dY_tile A^T)[comp] (x) V[comp]  ;  dg = G^T dU G
// gy transform: one thread per (tile, 4 channels of dY): MT x MT output-gradient tile -> a x a components
template <int MT>
__global__ __launch_bounds__(256) U2PL_HBM_KERNEL void k_wino_gy(const float* __restrict__ gy, long ldg, unsigned gbytes, WinoGeom g,
                                                 float* __restrict__ Mg, unsigned* __restrict__ mg_amax) {
    constexpr int A = WinoT<MT>::A;
    const int C4 = g.C >> 2;
    long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= g.tiles * C4) {
        if (!mg_amax) return;
        idx = g.tiles * C4 - 1;
    }
    const int c4 = (int)(idx % C4);
    const long t = idx / C4;
    int n, py, px, ty, tx;
    tile_coords(g, t, n, py, px, ty, tx);
    const __amdgpu_buffer_rsrc_t rg = make_rsrc(gy, gbytes);
    const int ldb = (int)ldg * 4;
    float4 v[MT][MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int oy = py + g.dil * (ty * MT + i);
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int ox = px + g.dil * (tx * MT + j);
            const bool ok = oy < g.H && ox < g.W;
            const int off = ((n * g.H + oy) * g.W + ox) * ldb + c4 * 16;
            v[i][j] = buf_load4(rg, ok ? off : OOB_OFF);
        }
    }
    float4 tmp[A][MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        float4 col[MT], r[A];
#pragma unroll
        for (int i = 0; i < MT; ++i) col[i] = v[i][j];
        WinoT<MT>::av(col, r);
#pragma unroll
        for (int i = 0; i < A; ++i) tmp[i][j] = r[i];
    }
    const long comp_stride = g.tiles * (long)g.C;
    float* out = Mg + t * g.C + c4 * 4;
    unsigned am = 0u;
#pragma unroll
    for (int i = 0; i < A; ++i) {
        float4 r[A];
        WinoT<MT>::av(tmp[i], r);
#pragma unroll
        for (int j = 0; j < A; ++j) {
            *(float4*)(out + (long)(i * A + j) * comp_stride) = r[j];
            am = amax_bits4(am, r[j]);
        }
    }
    if (mg_amax) amax_wave_publish(am, mg_amax);
}

// part: [nsplit][O][a*a][C] partial slabs of the batched weight-gradient GEMMs -> dw [O][3][3][C]
// (slabs summed in order, then G^T . G); one thread per (o, 4 channels): 16-byte loads, a*a of them in flight
template <int MT>
__global__ __launch_bounds__(256) void k_wino_wgrad_finish(const float* __restrict__ part, int nsplit, int O, int C,
                                                           int accumulate, float* __restrict__ dw) {
    constexpr int A = WinoT<MT>::A;
    const int C4 = C >> 2;
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= (long)O * C4) return;
    const int c4 = (int)(idx % C4), o = (int)(idx / C4);
    const long slab = (long)O * A * A * C;
    const float* p = part + (long)o * A * A * C + c4 * 4;
    float4 u[A][A];
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
        for (int j = 0; j < A; ++j) u[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < nsplit; ++z) {
        float4 v[A][A];
#pragma unroll
        for (int i = 0; i < A; ++i)
#pragma unroll
            for (int j = 0; j < A; ++j) v[i][j] = *(const float4*)(p + (long)z * slab + (long)(i * A + j) * C);
#pragma unroll
        for (int i = 0; i < A; ++i)
#pragma unroll
            for (int j = 0; j < A; ++j) u[i][j] = f4add(u[i][j], v[i][j]);
    }
    // G^T u G, channel by channel (the scalar transform helpers are reused on the 4 components)
    float4 t[3][A];
#pragma unroll
    for (int j = 0; j < A; ++j) {
        float cx[A], cy[A], cz[A], cw[A], rx[3], ry[3], rz[3], rw[3];
#pragma unroll
        for (int i = 0; i < A; ++i) { cx[i] = u[i][j].x; cy[i] = u[i][j].y; cz[i] = u[i][j].z; cw[i] = u[i][j].w; }
        WinoT<MT>::gt(cx, rx); WinoT<MT>::gt(cy, ry); WinoT<MT>::gt(cz, rz); WinoT<MT>::gt(cw, rw);
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i][j] = make_float4(rx[i], ry[i], rz[i], rw[i]);
    }
    const bool vec_ok = (reinterpret_cast<uintptr_t>(dw) & 15) == 0;
#pragma unroll
    for (int r3 = 0; r3 < 3; ++r3) {
        float cx[A], cy[A], cz[A], cw[A], rx[3], ry[3], rz[3], rw[3];
#pragma unroll
        for (int j = 0; j < A; ++j) { cx[j] = t[r3][j].x; cy[j] = t[r3][j].y; cz[j] = t[r3][j].z; cw[j] = t[r3][j].w; }
        WinoT<MT>::gt(cx, rx); WinoT<MT>::gt(cy, ry); WinoT<MT>::gt(cz, rz); WinoT<MT>::gt(cw, rw);
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
            float* d = dw + (((long)o * 3 + r3) * 3 + s3) * C + c4 * 4;
            float4 val = make_float4(rx[s3], ry[s3], rz[s3], rw[s3]);
            if (vec_ok) {
                if (accumulate) val = f4add(*(const float4*)d, val);
                *(float4*)d = val;
            } else {
                d[0] = accumulate ? d[0] + val.x : val.x; d[1] = accumulate ? d[1] + val.y : val.y;
                d[2] = accumulate ? d[2] + val.z : val.z; d[3] = accumulate ? d[3] + val.w : val.w;
            }
        }
    }
}

static int wino_geom(int N, int H, int W, int C, int dil, int mt, WinoGeom& g) {
    if (mt != 2 && mt != 4) return U2PL_EINVAL;
    if (dil < 1 || C % 4) return U2PL_EINVAL;
    g.N = N; g.H = H; g.W = W; g.C = C; g.dil = dil;
    g.Ty = cdiv(cdiv(H, dil), mt);
    g.Tx = cdiv(cdiv(W, dil), mt);
    g.tiles = (long)N * dil * dil * g.Ty * g.Tx;
    return 0;
}

U2PL_API size_t u2pl_wino_tiles(int N, int H, int W, int dil, int mt) {
    WinoGeom g;
    if (wino_geom(N, H, W, 4, dil, mt, g)) return 0;
    return (size_t)g.tiles;
}
// rows of the [nblk][2][O] statistics partials written by u2pl_wino_output_f32
U2PL_API int u2pl_wino_stat_blocks(long tiles, int O) {
    const int tpb = 256 / (O / 4) > 0 ? 256 / (O / 4) : 1;
    return (int)((tiles + tpb - 1) / tpb);
}

// v_amax: NULL, or a caller-zeroed device float that receives max |V| (split-fp16: the x_amax of u2pl_gemm_batched_wsh_f32)
static int run_wino_input(const float* x, long ldx, int N, int H, int W, int C, int dil, int mt, float* V, float* v_amax,
                          hipStream_t stream) {
    WinoGeom g;
    if (wino_geom(N, H, W, C, dil, mt, g)) return U2PL_EINVAL;
    const long xb = (((long)N * H * W - 1) * ldx + C) * 4;
    if (xb >= (1L << 31)) return U2PL_EINVAL;
    const long total = g.tiles * (C / 4);
    const dim3 grid((unsigned)cdiv(total, 256)), block(256);
    if (mt == 4) U2PL_LAUNCH(k_wino_input<4>, grid, block, 0, stream, x, ldx, (unsigned)xb, g, V, (unsigned*)v_amax);
    else U2PL_LAUNCH(k_wino_input<2>, grid, block, 0, stream, x, ldx, (unsigned)xb, g, V, (unsigned*)v_amax);
    U2PL_LAUNCH_CHECK();
    return 0;
}
U2PL_API int u2pl_wino_input_f32(const float* x, long ldx, int N, int H, int W, int C, int dil, int mt, float* V,
                                 hipStream_t stream) {
    return run_wino_input(x, ldx, N, H, W, C, dil, mt, V, nullptr, stream);
}
U2PL_API int u2pl_wino_input_amax_f32(const float* x, long ldx, int N, int H, int W, int C, int dil, int mt, float* V, float* v_amax,
                                      hipStream_t stream) {
    return run_wino_input(x, ldx, N, H, W, C, dil, mt, V, v_amax, stream);
}

// jobs: device array of njobs WinoWeightJob {w, U, begin, O, C, transposed, mt} (begin = prefix sum of O * C), total = sum of O * C
U2PL_API int u2pl_wino_weight_multi_f32(const void* jobs, int njobs, long total, hipStream_t stream) {
    if (njobs <= 0 || total <= 0) return njobs == 0 ? 0 : U2PL_EINVAL;
    U2PL_LAUNCH(k_wino_weight_multi, dim3(grid_for(total, 256, 4096)), dim3(256), 0, stream, (const WinoWeightJob*)jobs, njobs, total);
    U2PL_LAUNCH_CHECK();
    return 0;
}
U2PL_API int u2pl_wino_weight_f32(const float* w, int O, int C, int transposed, int mt, float* U, hipStream_t stream) {
    if (mt != 2 && mt != 4) return U2PL_EINVAL;
    const dim3 grid((unsigned)cdiv((long)O * C, 256)), block(256);
    if (mt == 4) U2PL_LAUNCH(k_wino_weight<4>, grid, block, 0, stream, w, O, C, transposed, U);
    else U2PL_LAUNCH(k_wino_weight<2>, grid, block, 0, stream, w, O, C, transposed, U);
    U2PL_LAUNCH_CHECK();
    return 0;
}

static int run_wino_output(const float* Mb, int N, int H, int W, int O, int dil, int mt, const float* bias, float* y,
                           long ldy, float* stats_partial, const float* pivot, const BnEpi& epi, hipStream_t stream) {
    WinoGeom g;
    if (wino_geom(N, H, W, O, dil, mt, g)) return U2PL_EINVAL;
    if (stats_partial && (O / 4 > 256 || (O & 3))) return U2PL_EINVAL;
    const long total = g.tiles * (O / 4);
    const unsigned nblk = stats_partial ? (unsigned)u2pl_wino_stat_blocks(g.tiles, O) : (unsigned)cdiv(total, 256);
    if (mt == 4) U2PL_LAUNCH(k_wino_output<4>, dim3(nblk), dim3(256), 0, stream, Mb, g, O, bias, y, ldy, stats_partial, pivot, epi);
    else U2PL_LAUNCH(k_wino_output<2>, dim3(nblk), dim3(256), 0, stream, Mb, g, O, bias, y, ldy, stats_partial, pivot, epi);
    U2PL_LAUNCH_CHECK();
    return 0;
}
U2PL_API int u2pl_wino_output_f32(const float* Mb, int N, int H, int W, int O, int dil, int mt, const float* bias,
                                  float* y, long ldy, float* stats_partial, const float* pivot, hipStream_t stream) {
    const BnEpi off = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr};
    return run_wino_output(Mb, N, H, W, O, dil, mt, bias, y, ldy, stats_partial, pivot, off, stream);
}
// output transform fused with the eval-mode BatchNorm (+residual, ReLU) that follows the convolution
// (the Winograd form of u2pl_conv2d_fwd_bnact_f32)
U2PL_API int u2pl_wino_output_bnact_f32(const float* Mb, int N, int H, int W, int O, int dil, int mt, const float* bias,
                                        float* y, long ldy, const float* mean, const float* invstd, const float* gamma,
                                        const float* beta, const float* res, long ldr, int relu, hipStream_t stream) {
    if (!mean || !invstd || !gamma || !beta || (O & 3) || (ldy & 3) || (res && (ldr & 3))) return U2PL_EINVAL;
    const BnEpi epi = {mean, invstd, gamma, beta, res, ldr, relu, nullptr};
    return run_wino_output(Mb, N, H, W, O, dil, mt, bias, y, ldy, nullptr, nullptr, epi, stream);
}
// the same + max |y| into the caller-zeroed amax object y_amax (split-fp16: y is the next convolution's operand)
U2PL_API int u2pl_wino_output_bnact_amax_f32(const float* Mb, int N, int H, int W, int O, int dil, int mt, const float* bias,
                                             float* y, long ldy, const float* mean, const float* invstd, const float* gamma,
                                             const float* beta, const float* res, long ldr, int relu, float* y_amax,
                                             hipStream_t stream) {
    if (!mean || !invstd || !gamma || !beta || (O & 3) || (ldy & 3) || (res && (ldr & 3))) return U2PL_EINVAL;
    const BnEpi epi = {mean, invstd, gamma, beta, res, ldr, relu, (unsigned*)y_amax};
    return run_wino_output(Mb, N, H, W, O, dil, mt, bias, y, ldy, nullptr, nullptr, epi, stream);
}

static int run_wino_gy(const float* gy, long ldg, int N, int H, int W, int O, int dil, int mt, float* Mg, float* mg_amax,
                       hipStream_t stream) {
    WinoGeom g;
    if (wino_geom(N, H, W, O, dil, mt, g)) return U2PL_EINVAL;
    const long gb = (((long)N * H * W - 1) * ldg + O) * 4;
    if (gb >= (1L << 31)) return U2PL_EINVAL;
    const long total = g.tiles * (O / 4);
    const dim3 grid((unsigned)cdiv(total, 256)), block(256);
    if (mt == 4) U2PL_LAUNCH(k_wino_gy<4>, grid, block, 0, stream, gy, ldg, (unsigned)gb, g, Mg, (unsigned*)mg_amax);
    else U2PL_LAUNCH(k_wino_gy<2>, grid, block, 0, stream, gy, ldg, (unsigned)gb, g, Mg, (unsigned*)mg_amax);
    U2PL_LAUNCH_CHECK();
    return 0;
}
U2PL_API int u2pl_wino_gy_f32(const float* gy, long ldg, int N, int H, int W, int O, int dil, int mt, float* Mg,
                              hipStream_t stream) {
    return run_wino_gy(gy, ldg, N, H, W, O, dil, mt, Mg, nullptr, stream);
}
U2PL_API int u2pl_wino_gy_amax_f32(const float* gy, long ldg, int N, int H, int W, int O, int dil, int mt, float* Mg, float* mg_amax,
                                   hipStream_t stream) {
    return run_wino_gy(gy, ldg, N, H, W, O, dil, mt, Mg, mg_amax, stream);
}

U2PL_API int u2pl_wino_wgrad_finish_f32(const float* part, int nsplit, int O, int C, int mt, int accumulate, float* dw,
                                        hipStream_t stream) {
    if (mt != 2 && mt != 4) return U2PL_EINVAL;
    if (C % 4) return U2PL_EINVAL;
    const dim3 grid((unsigned)cdiv((long)O * (C / 4), 256)), block(256);
    if (mt == 4) U2PL_LAUNCH(k_wino_wgrad_finish<4>, grid, block, 0, stream, part, nsplit, O, C, accumulate, dw);
    else U2PL_LAUNCH(k_wino_wgrad_finish<2>, grid, block, 0, stream, part, nsplit, O, C, accumulate, dw);
    U2PL_LAUNCH_CHECK();
    return 0;
}
