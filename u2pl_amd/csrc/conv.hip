// Implicit-GEMM convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32,
// exact f32 == fmaf chain) for the dilated ResNet-101 + DeepLabv3+ stack
// (reference u2pl/models/{resnet,base,decoder}.py run
// through torch/cuDNN).  Activations are NHWC "rows" (pixel-major, channel
// contiguous, leading dim ld so channel slices of concat buffers work in place);
// weights are [Cout][R][S][Cin] (torch OIHW storage in channels_last format).
//
//   conv_igemm : Y[m][co] = sum_{r,s,ci} X[gather(m,r,s)][ci] * W[co][r][s][ci]
//                forward (gather: o*stride - pad + r*dil) and data-gradient
//                (gather: (i + pad - r*dil)/stride with divisibility test, weights
//                pre-transposed to [Cin][R][S][Cout]) share one kernel.
//   conv_wgrad : dW[co][r][s][ci] = sum_m dY[m][co] * X[gather(m,r,s)][ci]
//                split over pixel ranges -> partial slabs -> ordered reduce
//                (deterministic; no float atomics).
//
// Tiling: 256 threads = 4 waves (2x2), each wave owns (32*TM)x(32*TN) outputs as
// TM*TN 32x32 MFMA accumulators; BK = 32; LDS double-buffered, global->register
// prefetch of the next K chunk overlaps the MFMAs of the current one.  K order
// inside an 8-wide group is permuted (lane half h supplies k = 4h+j at step j) so
// each lane fetches its four A / B operands with one ds_read_b128.
#include <stdlib.h>
#include "common.h"
#include "u2pl_hip.h"

#include "conv_geom.h"
#include "wgrad_tr.h"

template <int TM, int TN, int WM = 2, int BF = 0>
__global__ __launch_bounds__(128 * WM, WM) void k_conv_igemm(const float* __restrict__ x, long ldx,
                                                       const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y,
                                                       long ldy, ConvGeom g, unsigned xbytes, unsigned wbytes,
                                                       long m_begin, long m_end, float* __restrict__ stats,
                                                       const float* __restrict__ pivot, long zx, long zw, long zy,
                                                       BnEpi epi) {
    constexpr int BM = 32 * TM * WM, BN = 64 * TN;
    constexpr int NT = 128 * WM, RPP = NT / 8;          // threads, tile rows filled per pass (8 threads x float4 = BK)
    constexpr int RA = BM / RPP, RB = BN / RPP;         // rows per thread per chunk
    // batched use (Winograd components): blockIdx.z selects an independent GEMM, element strides zx/zw/zy
    x += (long)blockIdx.z * zx;
    w += (long)blockIdx.z * zw;
    y += (long)blockIdx.z * zy;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, xbytes), rw = make_rsrc(w, wbytes);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                  // [2][BM][LDP]
    float* Bs = smem + 2 * BM * LDP;   // [2][BN][LDP]
    unsigned short* Ah = (unsigned short*)smem;          // BF: [2][BM][LDPH] bf16 (BF == 3: three such planes, piece-major)
    // BF == 3: ONE LDS buffer (three piece planes of A, three of B: 61 KB for 128 x 128 x 32 -- two buffers would be 120 KB
    // and leave a single 8-wave block per CU; with one buffer two blocks are resident and fill each other's barriers)
    constexpr int NBUF = BF == 3 ? 1 : 2;
    unsigned short* Bh = Ah + (BF == 3 ? 3 : 2) * BM * LDPH;         //     [2][BN][LDPH]
    constexpr long APL = (long)NBUF * BM * LDPH, BPL = (long)NBUF * BN * LDPH;       // plane strides of the split form

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const long M = m_end;  // this launch covers output rows [m_begin, m_end)
    const int K = g.R * g.S * g.Cin;
    const int nk = K / BK;
    const int cpt = g.Cin / BK;  // chunks per tap
    // M tiles fastest: concurrently resident blocks share the same weight tile (L2 reuse)
    const long m0 = m_begin + (long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    const int kq = tid & 7, r0 = tid >> 3;
    int bh[RA], bw[RA];
    int nb[RA];   // pixel index of the image origin (fits int: bytes < 2^31)
    bool mv[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        long m = m0 + r0 + RPP * i;
        mv[i] = m < M;
        const unsigned mm = mv[i] ? (unsigned)m : 0u;   // pixel index < 2^31 (host check): 32-bit divisions
        const unsigned t = mm / (unsigned)g.Wout;
        int wo = (int)(mm - t * (unsigned)g.Wout);
        const unsigned n_ = t / (unsigned)g.Hout;
        int ho = (int)(t - n_ * (unsigned)g.Hout);
        int n = (int)n_;
        bh[i] = ho * g.mul + g.off_h;
        bw[i] = wo * g.mul + g.off_w;
        nb[i] = n * g.Hin * g.Win;
    }
    // two register sets: the global loads of chunk k+2 are issued while chunk k is multiplied (two chunk-times
    // = ~4 us of latency tolerance instead of one), parked in LDS one chunk ahead of their use
    float4 ra0[RA], rb0[RB], ra1[RA], rb1[RB];
    const int ldxb = (int)ldx * 4;
    auto load_chunk = [&](int kc, float4 (&ra)[RA], float4 (&rb)[RB]) {
        const int tap = kc / cpt, c0 = (kc - tap * cpt) * BK;
        const int r = tap / g.S, s = tap - r * g.S;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            int ih, iw;
            const bool okh = gather_coord(bh[i], r, g.step, g.log2div, g.Hin, ih);
            const bool okw = gather_coord(bw[i], s, g.step, g.log2div, g.Win, iw);
            const int off = (nb[i] + ih * g.Win + iw) * ldxb + (c0 + kq * 4) * 4;
            ra[i] = buf_load4(rx, (mv[i] & okh & okw) ? off : OOB_OFF);
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int co = n0 + r0 + RPP * i;
            const int off = (co * K + kc * BK + kq * 4) * 4;
            rb[i] = buf_load4(rw, co < g.Cout ? off : OOB_OFF);
        }
    };
    auto store_chunk = [&](int buf, const float4 (&ra)[RA], const float4 (&rb)[RB]) {
        if (BF == 3) {
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                uint2 p0, p1, p2;
                split3_bf16(ra[i], p0, p1, p2);
                unsigned short* d = Ah + ((long)buf * BM + r0 + RPP * i) * LDPH + kq * 4;
                *(uint2*)d = p0; *(uint2*)(d + APL) = p1; *(uint2*)(d + 2 * APL) = p2;
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                uint2 p0, p1, p2;
                split3_bf16(rb[i], p0, p1, p2);
                unsigned short* d = Bh + ((long)buf * BN + r0 + RPP * i) * LDPH + kq * 4;
                *(uint2*)d = p0; *(uint2*)(d + BPL) = p1; *(uint2*)(d + 2 * BPL) = p2;
            }
            return;
        }
        if (BF) {
#pragma unroll
            for (int i = 0; i < RA; ++i) *(uint2*)(Ah + ((long)buf * BM + r0 + RPP * i) * LDPH + kq * 4) = pack4_bf16(ra[i]);
#pragma unroll
            for (int i = 0; i < RB; ++i) *(uint2*)(Bh + ((long)buf * BN + r0 + RPP * i) * LDPH + kq * 4) = pack4_bf16(rb[i]);
            return;
        }
#pragma unroll
        for (int i = 0; i < RA; ++i) *(float4*)(As + ((long)buf * BM + r0 + RPP * i) * LDP + kq * 4) = ra[i];
#pragma unroll
        for (int i = 0; i < RB; ++i) *(float4*)(Bs + ((long)buf * BN + r0 + RPP * i) * LDP + kq * 4) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    const int li = lane & 31, lh = lane >> 5;
    auto mma = [&](int buf) {
        if (BF == 3) {   // six piece products per 32x32x16 block, smallest weights first
            const unsigned short* Ab = Ah + ((long)buf * BM + wm * 32 * TM + li) * LDPH + 8 * lh;
            const unsigned short* Bb = Bh + ((long)buf * BN + wn * 32 * TN + li) * LDPH + 8 * lh;
#pragma unroll
            for (int gk = 0; gk < BK / 16; ++gk) {
                bf16x8 a8[3][TM], b8[3][TN];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
#pragma unroll
                    for (int a = 0; a < TM; ++a) a8[p][a] = __builtin_bit_cast(bf16x8, *(const uint4*)(Ab + p * APL + a * 32 * LDPH + gk * 16));
#pragma unroll
                    for (int b = 0; b < TN; ++b) b8[p][b] = __builtin_bit_cast(bf16x8, *(const uint4*)(Bb + p * BPL + b * 32 * LDPH + gk * 16));
                }
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        f32x16 c = acc[a][b];
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[2][a], b8[0][b], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[1][a], b8[1][b], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[0][a], b8[2][b], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[1][a], b8[0][b], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[0][a], b8[1][b], c, 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[0][a], b8[0][b], c, 0, 0, 0);
                    }
            }
            return;
        }
        if (BF) {   // lane (li, lh) supplies row li, k = 16*gk + 8*lh + (0..7) of A and of B: one ds_read_b128 each
            const unsigned short* Ab = Ah + ((long)buf * BM + wm * 32 * TM + li) * LDPH + 8 * lh;
            const unsigned short* Bb = Bh + ((long)buf * BN + wn * 32 * TN + li) * LDPH + 8 * lh;
#pragma unroll
            for (int gk = 0; gk < BK / 16; ++gk) {
                bf16x8 a8[TM], b8[TN];
#pragma unroll
                for (int a = 0; a < TM; ++a) a8[a] = __builtin_bit_cast(bf16x8, *(const uint4*)(Ab + a * 32 * LDPH + gk * 16));
#pragma unroll
                for (int b = 0; b < TN; ++b) b8[b] = __builtin_bit_cast(bf16x8, *(const uint4*)(Bb + b * 32 * LDPH + gk * 16));
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[a], b8[b], acc[a][b], 0, 0, 0);
            }
            return;
        }
        const float* Ab = As + ((long)buf * BM + wm * 32 * TM + li) * LDP + 4 * lh;
        const float* Bb = Bs + ((long)buf * BN + wn * 32 * TN + li) * LDP + 4 * lh;
#pragma unroll
        for (int gk = 0; gk < BK / 8; ++gk) {
            float4 a4[TM], b4[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) a4[a] = *(const float4*)(Ab + a * 32 * LDP + gk * 8);
#pragma unroll
            for (int b = 0; b < TN; ++b) b4[b] = *(const float4*)(Bb + b * 32 * LDP + gk * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        const float av = j == 0 ? a4[a].x : j == 1 ? a4[a].y : j == 2 ? a4[a].z : a4[a].w;
                        const float bv = j == 0 ? b4[b].x : j == 1 ? b4[b].y : j == 2 ? b4[b].z : b4[b].w;
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a][b], 0, 0, 0);
                    }
        }
    };
    load_chunk(0, ra0, rb0);
    store_chunk(0, ra0, rb0);
    if (nk > 1) load_chunk(1, ra0, rb0);
    if (nk > 2) load_chunk(2, ra1, rb1);
    __syncthreads();
    if constexpr (BF == 3) {
        // split form: a chunk lasts ~1 us here (a third of the fp32-MFMA kernel's), so the two-chunk prefetch distance has
        // to be real: the loads are issued UNCONDITIONALLY (chunk index clamped: past the end the last chunk is re-read and
        // never used) -- a load under `if` makes the compiler wait with vmcnt(0), i.e. also for the loads issued one chunk
        // ago (compiler behaviour (2))
        for (int kc = 0; kc < nk; kc += 2) {
            mma(0);
            __syncthreads();                      // (single buffer: everybody has read chunk kc before it is overwritten)
            store_chunk(0, ra0, rb0);             // (past the last chunk: stale pieces, never read)
            __syncthreads();
            load_chunk(min(kc + 3, nk - 1), ra0, rb0);
            if (kc + 1 >= nk) break;
            mma(0);
            __syncthreads();
            store_chunk(0, ra1, rb1);
            __syncthreads();
            load_chunk(min(kc + 4, nk - 1), ra1, rb1);
        }
    } else
    for (int kc = 0; kc < nk; kc += 2) {
        mma(0);
        if (kc + 1 < nk) store_chunk(1, ra0, rb0);
        __syncthreads();
        if (kc + 3 < nk) load_chunk(kc + 3, ra0, rb0);
        if (kc + 1 < nk) {
            mma(1);
            if (kc + 2 < nk) store_chunk(0, ra1, rb1);
            __syncthreads();
            if (kc + 4 < nk) load_chunk(kc + 4, ra1, rb1);
        }
    }
    // epilogue: C/D layout col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).  Each wave transposes its
    // (32*TM) x (32*TN) block through LDS (free after the K loop) so that rows leave as 16-byte stores:
    // 4x fewer store instructions than one dword per lane, whole 128-byte lines per row segment.
    {
        constexpr int WN = 32 * TN, PW = WN + 4;           // wave tile width, LDS pitch (floats)
        float* ws_ = smem + (long)wave * (32 * TM) * PW;   // 4 waves x 64 x 68 x 4 B = 69.6 KB <= 2*(BM+BN)*LDP*4
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    ws_[(a * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * PW + b * 32 + li] = acc[a][b][e];
        // no barrier needed: a wave only reads back what it wrote (wave-private region; LDS ops of one wave are ordered)
        constexpr int C4 = WN / 4;                 // float4 per row
        constexpr int RPI = 64 / C4;               // rows per wave-instruction
        const int cq = lane % C4, rr = lane / C4;
        const int cbase = n0 + wn * WN + cq * 4;
        float4 bv4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) {
            bv4.x = cbase + 0 < g.Cout ? bias[cbase + 0] : 0.f;
            bv4.y = cbase + 1 < g.Cout ? bias[cbase + 1] : 0.f;
            bv4.z = cbase + 2 < g.Cout ? bias[cbase + 2] : 0.f;
            bv4.w = cbase + 3 < g.Cout ? bias[cbase + 3] : 0.f;
        }
        const bool vec_ok = ((ldy & 3) == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
        // fused eval-mode BN: channel count is a multiple of 4 (host check), so a float4 is entirely in or out of range
        const bool bn_on = epi.mean != nullptr && cbase < g.Cout;
        float4 mu4 = bv4, is4 = bv4, ga4 = bv4, be4 = bv4;
        if (bn_on) {
            mu4 = *(const float4*)(epi.mean + cbase); is4 = *(const float4*)(epi.invstd + cbase);
            ga4 = *(const float4*)(epi.gamma + cbase); be4 = *(const float4*)(epi.beta + cbase);
        }
#pragma unroll
        for (int it = 0; it < 32 * TM / RPI; ++it) {
            const int row = it * RPI + rr;
            const long m = m0 + wm * 32 * TM + row;
            float4 v = *(const float4*)(ws_ + row * PW + cq * 4);
            v.x += bv4.x; v.y += bv4.y; v.z += bv4.z; v.w += bv4.w;
            if (bn_on && m < M) {
                v.x = (v.x - mu4.x) * is4.x * ga4.x + be4.x; v.y = (v.y - mu4.y) * is4.y * ga4.y + be4.y;
                v.z = (v.z - mu4.z) * is4.z * ga4.z + be4.z; v.w = (v.w - mu4.w) * is4.w * ga4.w + be4.w;
                if (epi.res) {
                    const float4 rv = *(const float4*)(epi.res + m * epi.ldr + cbase);
                    v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                }
                if (epi.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            }
            if (m < M) {
                float* dst = y + m * ldy + cbase;
                if (vec_ok && cbase + 3 < g.Cout) *(float4*)dst = v;
                else {
                    if (cbase + 0 < g.Cout) dst[0] = v.x;
                    if (cbase + 1 < g.Cout) dst[1] = v.y;
                    if (cbase + 2 < g.Cout) dst[2] = v.z;
                    if (cbase + 3 < g.Cout) dst[3] = v.w;
                }
            }
        }
        if (stats) __syncthreads();   // the statistics below reuse the front of the same LDS
    }
    // Fused BatchNorm statistics (saves the separate read pass over y): per-tile pivot-shifted column sums
    // S1 = sum(v - p), S2 = sum((v - p)^2) over the tile's valid rows, written in the two-stage column-reduce
    // partial format [tile][2][Cout]; the ordered double-precision finish is k_colreduce_final.
    if (stats) {
        float* red = smem;   // [WM (wm)][2][BN]; the K loop's last barrier has passed, LDS is free
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int cl = wn * 32 * TN + b * 32 + li, co = n0 + cl;
            const bool cv = co < g.Cout;
            const float sh = (cv ? (bias ? bias[co] : 0.f) : 0.f) - (cv && pivot ? pivot[co] : 0.f);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const long m = m0 + wm * 32 * TM + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    const float v = m < M ? acc[a][b][e] + sh : 0.f;
                    s1 += v;
                    s2 += v * v;
                }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lh == 0) {
                red[(wm * 2 + 0) * BN + cl] = s1;
                red[(wm * 2 + 1) * BN + cl] = s2;
            }
        }
        __syncthreads();
        float* out = stats + (long)blockIdx.x * 2 * g.Cout;
        for (int c = tid; c < BN; c += NT) {
            const int co = n0 + c;
            if (co < g.Cout) {
                float a1 = red[0 * BN + c], a2 = red[1 * BN + c];
#pragma unroll
                for (int r = 1; r < WM; ++r) { a1 += red[(2 * r) * BN + c]; a2 += red[(2 * r + 1) * BN + c]; }
                out[co] = a1;
                out[g.Cout + co] = a2;
            }
        }
    }
}

template <int TM, int TN, int WM = 2, int BF = 0>
static int launch_igemm(const float* x, long ldx, const float* w, const float* bias, float* y, long ldy,
                        const ConvGeom& g, long m_begin, long m_end, hipStream_t stream, float* stats = nullptr,
                        const float* pivot = nullptr, int batch = 1, long zx = 0, long zw = 0, long zy = 0,
                        const BnEpi* epi = nullptr) {
    constexpr int BM = 32 * TM * WM, BN = 64 * TN;
    if (m_end <= m_begin) return 0;
    const BnEpi ep = epi ? *epi : BnEpi{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};
    const size_t lds_op = BF == 3 ? (size_t)3 * (BM + BN) * LDPH * sizeof(unsigned short) : (size_t)2 * (BM + BN) * LDP * sizeof(float);
    const size_t lds_epi = (size_t)(4 * WM / 2) * (32 * TM) * (32 * TN + 4) * sizeof(float);     // the epilogue's per-wave transpose regions
    const size_t lds = lds_op > lds_epi ? lds_op : lds_epi;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_conv_igemm<TM, TN, WM, BF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    // byte extents of the gathered tensor and of the weight matrix (raw-buffer descriptors)
    const long xb = (((long)g.N * g.Hin * g.Win - 1) * ldx + g.Cin) * 4;
    const long wb = (long)g.Cout * g.R * g.S * g.Cin * 4;
    if (xb >= (1L << 31) || wb >= (1L << 31)) return U2PL_EINVAL;
    dim3 grid((unsigned)cdiv(m_end - m_begin, BM), (unsigned)cdiv(g.Cout, BN), (unsigned)batch);
    U2PL_LAUNCH((k_conv_igemm<TM, TN, WM, BF>), grid, dim3(128 * WM), lds, stream, x, ldx, w, bias, y, ldy, g, (unsigned)xb,
                       (unsigned)wb, m_begin, m_end, stats, pivot, zx, zw, zy, ep);
    U2PL_LAUNCH_CHECK();
    return 0;
}

static int log2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

// the fp32 products' arithmetic: 1 (default) = split fp32 on the bf16 matrix cores (k_conv_igemm BF == 3), 0 = the fp32
// matrix-core instruction v_mfma_f32_32x32x2_f32.  U2PL_CONV_SPLIT=0|1; u2pl_conv_set_split() for tests / A-B runs.
static int g_conv_split = -1;
static int conv_split() {
    if (g_conv_split < 0) { const char* e = getenv("U2PL_CONV_SPLIT"); g_conv_split = (e && *e) ? (atoi(e) != 0) : 1; }
    return g_conv_split;
}
U2PL_API int u2pl_conv_set_split(int on) { const int old = conv_split(); g_conv_split = on != 0; return old; }
U2PL_API int u2pl_conv_get_split(void) { return conv_split(); }

// Tile-quantisation planner.  All 256 CUs finish a "round" of equal blocks together, so a launch of
// nblk blocks costs ceil(nblk/256) rounds of one block's area.  The body is covered with 128x128
// tiles in whole rounds; the remaining rows (the partial last round that would leave most CUs idle)
// are covered by a second launch with whichever smaller tile (128x128 / 64x128 / 64x64) is cheapest.
#define NUM_CUS 256
static double tail_cost(long rows, int cout, int bm, int bn, double eff) {
    const long nblk = (long)cdiv(rows, bm) * cdiv(cout, bn);
    return (double)cdiv(nblk, NUM_CUS) * bm * bn / eff;
}
// the planner's decisions, shared by the launcher and by the stat-block query
struct IgemmPlan { long m_body; int tail_tm, tail_tn; int nblk_body, nblk_tail; };
static IgemmPlan plan_igemm(const ConvGeom& g, int batch = 1) {
    const long M = (long)g.N * g.Hout * g.Wout;
    IgemmPlan p = {M, 0, 0, 0, 0};
    if (g.Cout <= 64) { p.nblk_body = cdiv(M, 128); return p; }   // single <2,1> launch
    const int nt = cdiv(g.Cout, 128) * batch;
    const long mtiles = cdiv(M, 128);
    long body_tiles = (mtiles * nt / NUM_CUS) * NUM_CUS / nt;      // M tiles covered by whole rounds
    if (body_tiles > mtiles) body_tiles = mtiles;
    p.m_body = body_tiles * 128 > M ? M : body_tiles * 128;
    p.nblk_body = cdiv(p.m_body, 128);
    const long tail = M - p.m_body;
    if (tail <= 0) return p;
    const double c22 = tail_cost(tail, g.Cout * batch, 128, 128, 1.0), c12 = tail_cost(tail, g.Cout * batch, 64, 128, 0.9);
    const double c11 = tail_cost(tail, g.Cout * batch, 64, 64, 0.8);
    if (c22 <= c12 && c22 <= c11) { p.tail_tm = 2; p.tail_tn = 2; }
    else if (c12 <= c11) { p.tail_tm = 1; p.tail_tn = 2; }
    else { p.tail_tm = 1; p.tail_tn = 1; }
    {   // U2PL_IGEMM_TAIL = 0 (one launch of body tiles) | 22 | 12 | 11 | 14 (128x64, 8 waves) | -1 (the planner's choice).
        // Split form: no tail launch by default -- two body blocks are resident per CU there, a partial last round costs
        // less than a second launch of small tiles whose matrix pipe sits at 9-18 % (measured: 86.6 vs 88.6 ms per step for
        // the group); fp32 instruction: the planner's tail (tools/bench_igemm_tail.py: every forced choice within +-2 %).
        const char* e = getenv("U2PL_IGEMM_TAIL");
        if (!(e && *e) && conv_split()) e = "0";
        if (e && *e && atoi(e) >= 0) {
            const int v = atoi(e);
            if (v == 0) { p.m_body = M; p.nblk_body = cdiv(M, 128); p.tail_tm = p.tail_tn = 0; p.nblk_tail = 0; return p; }
            if (v == 22) { p.tail_tm = 2; p.tail_tn = 2; }
            else if (v == 12) { p.tail_tm = 1; p.tail_tn = 2; }
            else if (v == 11) { p.tail_tm = 1; p.tail_tn = 1; }
            else if (v == 14) { p.tail_tm = 4; p.tail_tn = 1; }     // <1, 1, 4>: 128 x 64 tiles on 8 waves
        }
    }
    p.nblk_tail = p.tail_tm == 4 ? cdiv(tail, 128) : cdiv(tail, 64 * p.tail_tm);
    return p;
}
static int run_igemm(const float* x, long ldx, const float* w, const float* bias, float* y, long ldy,
                     const ConvGeom& g, hipStream_t stream, float* stats = nullptr, const float* pivot = nullptr,
                     int batch = 1, long zx = 0, long zw = 0, long zy = 0, int bf = 0, const BnEpi* epi = nullptr) {
    if (g.Cin % BK) return U2PL_EINVAL;
    if (epi && (bf == 1 || stats || batch != 1 || (g.Cout & 3))) return U2PL_EINVAL;
    const long M = (long)g.N * g.Hout * g.Wout;
    const IgemmPlan p = plan_igemm(g, batch);
    if (bf == 0 && conv_split()) bf = 3;
    if (bf == 3) {   // split-fp32 variants of the same tile shapes (all epilogues)
        if (g.Cout <= 64) return launch_igemm<2, 1, 2, 3>(x, ldx, w, bias, y, ldy, g, 0, M, stream, stats, pivot, batch, zx, zw, zy, epi);
        static int waves3 = 0;
        if (!waves3) { const char* e = getenv("U2PL_IGEMM_WAVES"); waves3 = (e && atoi(e) == 4) ? 4 : 8; }
        int rc = waves3 == 8 ? launch_igemm<1, 2, 4, 3>(x, ldx, w, bias, y, ldy, g, 0, p.m_body, stream, stats, pivot, batch, zx, zw, zy, epi)
                             : launch_igemm<2, 2, 2, 3>(x, ldx, w, bias, y, ldy, g, 0, p.m_body, stream, stats, pivot, batch, zx, zw, zy, epi);
        if (rc || p.nblk_tail == 0) return rc;
        float* st = stats ? stats + (long)p.nblk_body * 2 * g.Cout : nullptr;
        if (p.tail_tm == 4) return launch_igemm<1, 1, 4, 3>(x, ldx, w, bias, y, ldy, g, p.m_body, M, stream, st, pivot, batch, zx, zw, zy, epi);
        if (p.tail_tm == 2) return launch_igemm<2, 2, 2, 3>(x, ldx, w, bias, y, ldy, g, p.m_body, M, stream, st, pivot, batch, zx, zw, zy, epi);
        if (p.tail_tn == 2) return launch_igemm<1, 2, 2, 3>(x, ldx, w, bias, y, ldy, g, p.m_body, M, stream, st, pivot, batch, zx, zw, zy, epi);
        return launch_igemm<1, 1, 2, 3>(x, ldx, w, bias, y, ldy, g, p.m_body, M, stream, st, pivot, batch, zx, zw, zy, epi);
    }
    if (bf) {   // bf16-operand variants of the same tile shapes
        if (g.Cout <= 64) return launch_igemm<2, 1, 2, 1>(x, ldx, w, bias, y, ldy, g, 0, M, stream, stats, pivot, batch, zx, zw, zy);
        int rc = launch_igemm<1, 2, 4, 1>(x, ldx, w, bias, y, ldy, g, 0, p.m_body, stream, stats, pivot, batch, zx, zw, zy);
        if (rc || p.nblk_tail == 0) return rc;
        float* st = stats ? stats + (long)p.nblk_body * 2 * g.Cout : nullptr;
        if (p.tail_tm == 2) return launch_igemm<2, 2, 2, 1>(x, ldx, w, bias, y, ldy, g, p.m_body, M, stream, st, pivot, batch, zx, zw, zy);
        if (p.tail_tn == 2) return launch_igemm<1, 2, 2, 1>(x, ldx, w, bias, y, ldy, g, p.m_body, M, stream, st, pivot, batch, zx, zw, zy);
        return launch_igemm<1, 1, 2, 1>(x, ldx, w, bias, y, ldy, g, p.m_body, M, stream, st, pivot, batch, zx, zw, zy);
    }
    if (g.Cout <= 64) return launch_igemm<2, 1>(x, ldx, w, bias, y, ldy, g, 0, M, stream, stats, pivot, batch, zx, zw, zy, epi);
    // 128x128 body tiles are computed by 8 waves (4x2, 32x64 outputs each; 4 waves per SIMD with two resident
    // blocks): +6 % MFMA throughput over 4 waves of 64x64 (98.7 -> 104.6 TFLOP/s on the step's launch mix; more
    // waves cover each other's LDS / barrier stalls).  U2PL_IGEMM_WAVES=4 selects the older shape.
    static int waves = 0;
    if (!waves) { const char* e = getenv("U2PL_IGEMM_WAVES"); waves = (e && atoi(e) == 4) ? 4 : 8; }
    int rc = waves == 8 ? launch_igemm<1, 2, 4>(x, ldx, w, bias, y, ldy, g, 0, p.m_body, stream, stats, pivot, batch, zx, zw, zy, epi)
                        : launch_igemm<2, 2>(x, ldx, w, bias, y, ldy, g, 0, p.m_body, stream, stats, pivot, batch, zx, zw, zy, epi);
    if (rc || p.nblk_tail == 0) return rc;
    float* st = stats ? stats + (long)p.nblk_body * 2 * g.Cout : nullptr;
    if (p.tail_tm == 4) return launch_igemm<1, 1, 4>(x, ldx, w, bias, y, ldy, g, p.m_body, M, stream, st, pivot, batch, zx, zw, zy, epi);
    if (p.tail_tm == 2) return launch_igemm<2, 2>(x, ldx, w, bias, y, ldy, g, p.m_body, M, stream, st, pivot, batch, zx, zw, zy, epi);
    if (p.tail_tn == 2) return launch_igemm<1, 2>(x, ldx, w, bias, y, ldy, g, p.m_body, M, stream, st, pivot, batch, zx, zw, zy, epi);
    return launch_igemm<1, 1>(x, ldx, w, bias, y, ldy, g, p.m_body, M, stream, st, pivot, batch, zx, zw, zy, epi);
}

// batch independent row-major GEMMs  Y_z[M][Nn] = X_z[M][K] * W_z[Nn][K]^T  (the Winograd component products)
U2PL_API int u2pl_gemm_batched_f32(const float* x, long ldx, long zx, const float* w, long zw, float* y, long ldy,
                                   long zy, long M, int K, int Nn, int batch, hipStream_t stream) {
    if (M <= 0 || batch <= 0) return 0;
    if (M >= (1L << 31)) return U2PL_EINVAL;
    ConvGeom g = {1, (int)M, 1, K, (int)M, 1, Nn, 1, 1, 1, 0, 0, 1, 0};
    return run_igemm(x, ldx, w, nullptr, y, ldy, g, stream, nullptr, nullptr, batch, zx, zw, zy);
}

// nn.Conv2d forward: resnet.py:25-41,178-186; base.py:23-83; decoder.py:60-106,132-138
U2PL_API int u2pl_conv2d_fwd_f32(const float* x, long ldx, const float* w, const float* bias, float* y,
                                 long ldy, int N, int Hin, int Win, int Cin, int Hout, int Wout, int Cout,
                                 int R, int S, int stride, int pad, int dil, hipStream_t stream) {
    ConvGeom g = {N, Hin, Win, Cin, Hout, Wout, Cout, R, S, stride, -pad, -pad, dil, 0};
    return run_igemm(x, ldx, w, bias, y, ldy, g, stream);
}

// forward fused with the eval-mode BatchNorm (+residual, ReLU) that follows it (teacher pseudo-label pass, validate(),
// eval.py): conv -> BN -> ReLU of resnet.py:118-138 / base.py / decoder.py in one launch.  res: [M][ldr] rows or NULL.
U2PL_API int u2pl_conv2d_fwd_bnact_f32(const float* x, long ldx, const float* w, const float* bias, float* y,
                                       long ldy, int N, int Hin, int Win, int Cin, int Hout, int Wout, int Cout,
                                       int R, int S, int stride, int pad, int dil, const float* mean,
                                       const float* invstd, const float* gamma, const float* beta, const float* res,
                                       long ldr, int relu, hipStream_t stream) {
    if (!mean || !invstd || !gamma || !beta || (Cout & 3) || (res && (ldr & 3))) return U2PL_EINVAL;
    ConvGeom g = {N, Hin, Win, Cin, Hout, Wout, Cout, R, S, stride, -pad, -pad, dil, 0};
    const BnEpi epi = {mean, invstd, gamma, beta, res, ldr, relu};
    return run_igemm(x, ldx, w, bias, y, ldy, g, stream, nullptr, nullptr, 1, 0, 0, 0, 0, &epi);
}

// forward fused with the BatchNorm statistics of its output (train-mode conv -> BN pairs): also writes the
// per-tile pivot-shifted column sums [nblk][2][Cout] (nblk = u2pl_conv2d_fwd_stat_blocks) to stats_partial
U2PL_API int u2pl_conv2d_fwd_stat_blocks(int N, int Hout, int Wout, int Cout) {
    ConvGeom g = {N, 0, 0, 32, Hout, Wout, Cout, 1, 1, 1, 0, 0, 1, 0};
    const IgemmPlan p = plan_igemm(g);
    return p.nblk_body + p.nblk_tail;
}
U2PL_API int u2pl_conv2d_fwd_bnstats_f32(const float* x, long ldx, const float* w, const float* bias, float* y,
                                         long ldy, int N, int Hin, int Win, int Cin, int Hout, int Wout, int Cout,
                                         int R, int S, int stride, int pad, int dil, const float* pivot,
                                         float* stats_partial, hipStream_t stream) {
    ConvGeom g = {N, Hin, Win, Cin, Hout, Wout, Cout, R, S, stride, -pad, -pad, dil, 0};
    return run_igemm(x, ldx, w, bias, y, ldy, g, stream, stats_partial, pivot);
}

// the same three products with bf16-rounded operands on the bf16 matrix cores (fp32 accumulate / fp32 tensors)
U2PL_API int u2pl_conv2d_fwd_bf16op_f32(const float* x, long ldx, const float* w, const float* bias, float* y,
                                        long ldy, int N, int Hin, int Win, int Cin, int Hout, int Wout, int Cout,
                                        int R, int S, int stride, int pad, int dil, hipStream_t stream) {
    ConvGeom g = {N, Hin, Win, Cin, Hout, Wout, Cout, R, S, stride, -pad, -pad, dil, 0};
    return run_igemm(x, ldx, w, bias, y, ldy, g, stream, nullptr, nullptr, 1, 0, 0, 0, 1);
}
U2PL_API int u2pl_conv2d_fwd_bnstats_bf16op_f32(const float* x, long ldx, const float* w, const float* bias, float* y,
                                                long ldy, int N, int Hin, int Win, int Cin, int Hout, int Wout,
                                                int Cout, int R, int S, int stride, int pad, int dil,
                                                const float* pivot, float* stats_partial, hipStream_t stream) {
    ConvGeom g = {N, Hin, Win, Cin, Hout, Wout, Cout, R, S, stride, -pad, -pad, dil, 0};
    return run_igemm(x, ldx, w, bias, y, ldy, g, stream, stats_partial, pivot, 1, 0, 0, 0, 1);
}
U2PL_API int u2pl_conv2d_dgrad_bf16op_f32(const float* dy, long lddy, const float* wT, float* dx, long lddx, int N,
                                          int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int R, int S,
                                          int stride, int pad, int dil, hipStream_t stream) {
    int l2 = log2_exact(stride);
    if (l2 < 0) return U2PL_EINVAL;
    ConvGeom g = {N, Hout, Wout, Cout, Hin, Win, Cin, R, S, 1, pad, pad, -dil, l2};
    return run_igemm(dy, lddy, wT, nullptr, dx, lddx, g, stream, nullptr, nullptr, 1, 0, 0, 0, 1);
}

// data gradient: dX[n,hi,wi,ci] = sum_{r,s,co} dY[n,(hi+pad-r*dil)/st,(wi+pad-s*dil)/st,co] * W[co][r][s][ci]
// wT = weights transposed to [Cin][R][S][Cout] (u2pl_weight_transpose_f32)
U2PL_API int u2pl_conv2d_dgrad_f32(const float* dy, long lddy, const float* wT, float* dx, long lddx, int N,
                                   int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int R, int S,
                                   int stride, int pad, int dil, hipStream_t stream) {
    int l2 = log2_exact(stride);
    if (l2 < 0) return U2PL_EINVAL;
    // roles swap: the "input" of the gather is dY (Hout x Wout x Cout), the output is dX
    ConvGeom g = {N, Hout, Wout, Cout, Hin, Win, Cin, R, S, 1, pad, pad, -dil, l2};
    return run_igemm(dy, lddy, wT, nullptr, dx, lddx, g, stream);
}

// [A][T][B] -> [B][T][A]   (A = Cout, T = R*S, B = Cin)
__global__ void k_weight_transpose(const float* __restrict__ w, float* __restrict__ wt, int A, int T, int B) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int a0 = blockIdx.y * 32, b0 = blockIdx.x * 32;
    for (int i = threadIdx.y; i < 32; i += 8) {
        int a = a0 + i, b = b0 + threadIdx.x;
        tile[i][threadIdx.x] = (a < A && b < B) ? w[((long)a * T + t) * B + b] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
        int b = b0 + i, a = a0 + threadIdx.x;
        if (a < A && b < B) wt[((long)b * T + t) * A + a] = tile[threadIdx.x][i];
    }
}
U2PL_API int u2pl_weight_transpose_f32(const float* w, float* wt, int Cout, int RS, int Cin, hipStream_t stream) {
    dim3 grid(cdiv(Cin, 32), cdiv(Cout, 32), RS);
    U2PL_LAUNCH(k_weight_transpose, grid, dim3(32, 8), 0, stream, w, wt, Cout, RS, Cin);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// weight gradient.  GEMM view per tap: A^T = dY (pixels x Cout), B = gathered X
// (pixels x Cin), reduction over pixels.  LDS tiles are [k = pixel][i] so the
// MFMA operand fetch (lane l -> row k = 2*kk + (l>>5), column l&31) is a
// conflict-free ds_read_b32.
// ---------------------------------------------------------------------------
// WM = wave rows of the block (2: 4 waves; 4: 8 waves with half the output rows per wave -- the 128x128 tile then keeps
// 16 waves per CU in flight instead of 8, like the forward kernel's 8-wave body)
template <int TM, int TN, int WM = 2>
__global__ __launch_bounds__(128 * WM, 2) void k_conv_wgrad(const float* __restrict__ dy, long lddy,
                                                            const float* __restrict__ x, long ldx,
                                                            float* __restrict__ part, ConvGeom g, int ctiles,
                                                            int chunks_per_split, unsigned dybytes, unsigned xbytes,
                                                            long zdy, long zx) {
    constexpr int BM = 32 * TM * WM, BN = 64 * TN;
    constexpr int QL = 128 * WM / BK;        // threads per pixel row of a chunk (8 | 16), one float4 each per pass
    constexpr int CW = 4 * QL;               // channels covered per pass (32 | 64)
    // batched use (Winograd components as "taps" with an identity gather): tap t reads dy + t*zdy, x + t*zx
    {
        const int tap_b = blockIdx.x / ctiles;
        dy += (long)tap_b * zdy;
        x += (long)tap_b * zx;
    }
    const __amdgpu_buffer_rsrc_t rdy = make_rsrc(dy, dybytes), rx = make_rsrc(x, xbytes);
    constexpr int PA = BM + 4, PB = BN + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // [2][BK][PA]
    float* Bs = smem + 2 * BK * PA;   // [2][BK][PB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const long M = (long)g.N * g.Hout * g.Wout;
    const int co0 = blockIdx.y * BM;
    const int tap = blockIdx.x / ctiles, ci0 = (blockIdx.x - tap * ctiles) * BN;
    const int r = tap / g.S, s = tap - r * g.S;
    const long nchunks = (M + BK - 1) / BK;
    const long c_begin = (long)blockIdx.z * chunks_per_split;
    const long c_end = min(nchunks, c_begin + chunks_per_split);

    const int prow = tid / QL, q = tid % QL;  // pixel row in chunk, float4 lane within CW channels
    constexpr int JA = BM / CW, JB = BN / CW;
    static_assert(BM % CW == 0 && BN % CW == 0, "tile narrower than one load pass");
    float4 ra[JA], rb[JB];
    const int lddyb = (int)lddy * 4, ldxb = (int)ldx * 4;
    // Cout, Cin are multiples of 4 (host pads narrow heads): a float4 is entirely in or out of range
    auto load_chunk = [&](long ch) {
        const long m = ch * BK + prow;
        const bool mv = m < M;
        const unsigned mm = mv ? (unsigned)m : 0u;
        const unsigned t = mm / (unsigned)g.Wout;
        const int wo = (int)(mm - t * (unsigned)g.Wout);
        const unsigned n_ = t / (unsigned)g.Hout;
        const int ho = (int)(t - n_ * (unsigned)g.Hout);
        const int n = (int)n_;
        int ih, iw;
        const bool okh = gather_coord(ho * g.mul + g.off_h, r, g.step, 0, g.Hin, ih);
        const bool okw = gather_coord(wo * g.mul + g.off_w, s, g.step, 0, g.Win, iw);
        const bool okb = mv & okh & okw;
        const int dyo = (int)mm * lddyb;
        const int xo = (n * g.Hin * g.Win + ih * g.Win + iw) * ldxb;
#pragma unroll
        for (int j = 0; j < JA; ++j) {
            const int co = co0 + j * CW + q * 4;
            ra[j] = buf_load4(rdy, (mv & (co < g.Cout)) ? dyo + co * 4 : OOB_OFF);
        }
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            const int ci = ci0 + j * CW + q * 4;
            rb[j] = buf_load4(rx, (okb & (ci < g.Cin)) ? xo + ci * 4 : OOB_OFF);
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int j = 0; j < JA; ++j) *(float4*)(As + ((long)buf * BK + prow) * PA + j * CW + q * 4) = ra[j];
#pragma unroll
        for (int j = 0; j < JB; ++j) *(float4*)(Bs + ((long)buf * BK + prow) * PB + j * CW + q * 4) = rb[j];
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    const int li = lane & 31, lh = lane >> 5;
    if (c_begin < c_end) {
        load_chunk(c_begin);
        store_chunk(0);
        __syncthreads();
        for (long ch = c_begin; ch < c_end; ++ch) {
            const int buf = (int)((ch - c_begin) & 1);
            if (ch + 1 < c_end) load_chunk(ch + 1);
            const float* Ab = As + (long)buf * BK * PA + wm * 32 * TM + li;
            const float* Bb = Bs + (long)buf * BK * PB + wn * 32 * TN + li;
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                float av[TM], bv[TN];
#pragma unroll
                for (int a = 0; a < TM; ++a) av[a] = Ab[(2 * kk + lh) * PA + a * 32];
#pragma unroll
                for (int b = 0; b < TN; ++b) bv[b] = Bb[(2 * kk + lh) * PB + b * 32];
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
            }
            if (ch + 1 < c_end) store_chunk(buf ^ 1);
            __syncthreads();
        }
    }
    // partial slab [split][Cout][R*S*Cin]
    const long wsz = (long)g.Cout * g.R * g.S * g.Cin;
    float* out = part + (long)blockIdx.z * wsz;
    const long rowlen = (long)g.R * g.S * g.Cin;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int ci = ci0 + wn * 32 * TN + b * 32 + li;
            if (ci >= g.Cin) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = co0 + wm * 32 * TM + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (co < g.Cout) out[(long)co * rowlen + (long)tap * g.Cin + ci] = acc[a][b][e];
            }
        }
}

// bf16-operand weight gradient (config 5): the reduction runs over PIXELS while memory is channel-contiguous, so the
// tiles are transposed on their way into LDS: thread (pixel pair pp, channel quad cq) loads the same four channels of
// two consecutive pixels and writes four 32-bit words {bf16(pixel 2pp), bf16(pixel 2pp+1)} into [channel][pixel]
// rows (pitch 36 bf16 = 72 B), from which lane (li, lh) fetches its 8 consecutive pixels of channel li with two
// ds_read_b64.  Same split-K slabs / ordered reduce as the fp32 kernel.
// SP == 3: split fp32 (see k_conv_igemm BF == 3): three piece planes per operand, six piece products, ONE LDS buffer
// (55 KB for 128 x 128: two blocks per CU); zdy / zx: batched use (Winograd components as taps, like k_conv_wgrad).
#define LDPW 36
template <int TM, int TN, int SP = 1>
__global__ __launch_bounds__(256, 2) void k_conv_wgrad_bf16(const float* __restrict__ dy, long lddy,
                                                            const float* __restrict__ x, long ldx,
                                                            float* __restrict__ part, ConvGeom g, int ctiles,
                                                            int chunks_per_split, unsigned dybytes, unsigned xbytes,
                                                            long zdy, long zx) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int NBUF = SP == 3 ? 1 : 2;
    {
        const int tap_b = blockIdx.x / ctiles;
        dy += (long)tap_b * zdy;
        x += (long)tap_b * zx;
    }
    const __amdgpu_buffer_rsrc_t rdy = make_rsrc(dy, dybytes), rx = make_rsrc(x, xbytes);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short* Ah = (unsigned short*)smem;      // [SP][NBUF][BM][LDPW]
    unsigned short* Bh = Ah + SP * NBUF * BM * LDPW;  // [SP][NBUF][BN][LDPW]
    constexpr long APL = (long)NBUF * BM * LDPW, BPL = (long)NBUF * BN * LDPW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const long M = (long)g.N * g.Hout * g.Wout;
    const int co0 = blockIdx.y * BM;
    const int tap = blockIdx.x / ctiles, ci0 = (blockIdx.x - tap * ctiles) * BN;
    const int r = tap / g.S, s = tap - r * g.S;
    const long nchunks = (M + BK - 1) / BK;
    const long c_begin = (long)blockIdx.z * chunks_per_split;
    const long c_end = min(nchunks, c_begin + chunks_per_split);
    const int pp = tid >> 4, cq = tid & 15;          // pixel pair 0..15 of the 32-pixel chunk, channel quad 0..15 (64 ch / pass)
    constexpr int JA = BM / 64, JB = BN / 64;
    float4 ra[JA][2], rb[JB][2];
    const int lddyb = (int)lddy * 4, ldxb = (int)ldx * 4;
    auto load_chunk = [&](long ch) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long m = ch * BK + 2 * pp + h;
            const bool mv = m < M;
            const unsigned mm = mv ? (unsigned)m : 0u;
            const unsigned t = mm / (unsigned)g.Wout;
            const int wo = (int)(mm - t * (unsigned)g.Wout);
            const unsigned n_ = t / (unsigned)g.Hout;
            const int ho = (int)(t - n_ * (unsigned)g.Hout);
            int ih, iw;
            const bool okh = gather_coord(ho * g.mul + g.off_h, r, g.step, 0, g.Hin, ih);
            const bool okw = gather_coord(wo * g.mul + g.off_w, s, g.step, 0, g.Win, iw);
            const bool okb = mv & okh & okw;
            const int dyo = (int)mm * lddyb;
            const int xo = ((int)n_ * g.Hin * g.Win + ih * g.Win + iw) * ldxb;
#pragma unroll
            for (int j = 0; j < JA; ++j) {
                const int co = co0 + j * 64 + cq * 4;
                ra[j][h] = buf_load4(rdy, (mv & (co < g.Cout)) ? dyo + co * 4 : OOB_OFF);
            }
#pragma unroll
            for (int j = 0; j < JB; ++j) {
                const int ci = ci0 + j * 64 + cq * 4;
                rb[j][h] = buf_load4(rx, (okb & (ci < g.Cin)) ? xo + ci * 4 : OOB_OFF);
            }
        }
    };
    auto put1 = [&](unsigned short* d, long pl, float a, float b) {     // one channel row, pixels 2pp and 2pp+1
        const unsigned w0 = SP == 3 ? pack2_bf16_first(a, b) : pack2_bf16(a, b);
        *(unsigned*)d = w0;
        if (SP == 3) {
            const float a1 = a - bf16_lo_f(w0), b1 = b - bf16_hi_f(w0);
            const unsigned w1 = pack2_bf16(a1, b1);
            *(unsigned*)(d + pl) = w1;
            *(unsigned*)(d + 2 * pl) = pack2_bf16(a1 - bf16_lo_f(w1), b1 - bf16_hi_f(w1));
        }
    };
    auto put4 = [&](unsigned short* base, long pl, const float4& p0, const float4& p1) {   // rows c..c+3, pixels 2pp, 2pp+1
        put1(base + 0 * LDPW, pl, p0.x, p1.x);
        put1(base + 1 * LDPW, pl, p0.y, p1.y);
        put1(base + 2 * LDPW, pl, p0.z, p1.z);
        put1(base + 3 * LDPW, pl, p0.w, p1.w);
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int j = 0; j < JA; ++j) put4(Ah + ((long)buf * BM + j * 64 + cq * 4) * LDPW + 2 * pp, APL, ra[j][0], ra[j][1]);
#pragma unroll
        for (int j = 0; j < JB; ++j) put4(Bh + ((long)buf * BN + j * 64 + cq * 4) * LDPW + 2 * pp, BPL, rb[j][0], rb[j][1]);
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    const int li = lane & 31, lh = lane >> 5;
    if (c_begin < c_end) {
        load_chunk(c_begin);
        store_chunk(0);
        __syncthreads();
        auto ld8 = [&](const unsigned short* p) {
            const uint2 lo = *(const uint2*)p, hi = *(const uint2*)(p + 4);
            return __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
        };
        for (long ch = c_begin; ch < c_end; ++ch) {
            const int buf = NBUF == 1 ? 0 : (int)((ch - c_begin) & 1);
            if (ch + 1 < c_end) load_chunk(ch + 1);
            const unsigned short* Ab = Ah + ((long)buf * BM + wm * 32 * TM + li) * LDPW + 8 * lh;
            const unsigned short* Bb = Bh + ((long)buf * BN + wn * 32 * TN + li) * LDPW + 8 * lh;
#pragma unroll
            for (int gk = 0; gk < BK / 16; ++gk) {
                bf16x8 a8[SP][TM], b8[SP][TN];
#pragma unroll
                for (int p = 0; p < SP; ++p) {
#pragma unroll
                    for (int a = 0; a < TM; ++a) a8[p][a] = ld8(Ab + p * APL + a * 32 * LDPW + gk * 16);
#pragma unroll
                    for (int b = 0; b < TN; ++b) b8[p][b] = ld8(Bb + p * BPL + b * 32 * LDPW + gk * 16);
                }
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        f32x16 c = acc[a][b];
                        if (SP == 3) {      // six piece products, smallest weights first
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[SP - 1][a], b8[0][b], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[SP / 2][a], b8[SP / 2][b], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[0][a], b8[SP - 1][b], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[SP / 2][a], b8[0][b], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[0][a], b8[SP / 2][b], c, 0, 0, 0);
                        }
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[0][a], b8[0][b], c, 0, 0, 0);
                    }
            }
            if (NBUF == 1) __syncthreads();            // (single buffer: everybody has read this chunk)
            if (ch + 1 < c_end) store_chunk(NBUF == 1 ? 0 : buf ^ 1);
            __syncthreads();
        }
    }
    const long wsz = (long)g.Cout * g.R * g.S * g.Cin;
    float* out = part + (long)blockIdx.z * wsz;
    const long rowlen = (long)g.R * g.S * g.Cin;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int ci = ci0 + wn * 32 * TN + b * 32 + li;
            if (ci >= g.Cin) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = co0 + wm * 32 * TM + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (co < g.Cout) out[(long)co * rowlen + (long)tap * g.Cin + ci] = acc[a][b][e];
            }
        }
}

// dW = (accumulate ? dW : 0) + sum_z part[z]   (ordered => deterministic)
__global__ void k_wgrad_reduce(const float* __restrict__ part, long wsz, int nsplit, int accumulate,
                               float* __restrict__ dw) {
    // wsz % 4 == 0 (Cin % 4 == 0); 8 independent slab loads in flight, added in slab order
    const long w4 = wsz >> 2;
    const float4* p4 = (const float4*)part;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < w4; i += (long)gridDim.x * blockDim.x) {
        float4 acc = accumulate ? ((const float4*)dw)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        int z = 0;
        for (; z + 8 <= nsplit; z += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p4[(long)(z + u) * w4 + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
        for (; z < nsplit; ++z) {
            const float4 v = p4[(long)z * w4 + i];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        ((float4*)dw)[i] = acc;
    }
}

static void wgrad_plan(const ConvGeom& g, int BM, int BN, int& ctiles, int& nsplit, int& cps) {
    const long M = (long)g.N * g.Hout * g.Wout;
    const long nchunks = (M + BK - 1) / BK;
    ctiles = cdiv(g.Cin, BN);
    const long tiles = (long)cdiv(g.Cout, BM) * ctiles * g.R * g.S;
    long want = (1024 + tiles - 1) / tiles;          // aim for >= 1024 blocks (2 per CU x 2 rounds)
    long maxsplit = (nchunks + 7) / 8;               // >= 8 chunks (256 pixels) per split
    if (want > maxsplit) want = maxsplit;
    if (want < 1) want = 1;
    cps = (int)((nchunks + want - 1) / want);
    nsplit = (int)((nchunks + cps - 1) / cps);
}

U2PL_API size_t u2pl_conv2d_wgrad_workspace_bytes(int N, int Hout, int Wout, int Cin, int Cout, int R, int S) {
    ConvGeom g = {N, 0, 0, Cin, Hout, Wout, Cout, R, S, 1, 0, 0, 1, 0};
    int ct, ns, cps;
    wgrad_plan(g, Cout > 64 ? 128 : 64, Cin > 64 ? 128 : 64, ct, ns, cps);
    if (wgrad_tr_eligible(g)) {      // (the larger of the two kernels' plans: the arithmetic switch may change between query and launch)
        int ct2, ns2, cps2;
        wgrad_tr_plan(g, R * S, ct2, ns2, cps2);
        if (ns2 > ns) ns = ns2;
    }
    return (size_t)ns * Cout * R * S * Cin * sizeof(float);
}

// U2PL_WGRAD_WAVES = 8 (default) | 4: block shape of the 128x128 weight-gradient tile (A/B switch; same results)
static int wgrad_waves() {
    static int v = 0;
    if (!v) {
        const char* e = getenv("U2PL_WGRAD_WAVES");
        v = (e && atoi(e) == 4) ? 4 : 8;
    }
    return v;
}
template <int TM, int TN, int WM = 2>
static int launch_wgrad(const float* dy, long lddy, const float* x, long ldx, float* part, const ConvGeom& g,
                        int ctiles, int nsplit, int cps, hipStream_t stream, long zdy = 0, long zx = 0) {
    constexpr int BM = 32 * TM * WM, BN = 64 * TN;
    const size_t lds = (size_t)2 * BK * (BM + 4 + BN + 4) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_conv_wgrad<TM, TN, WM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const long dyb = (((long)g.N * g.Hout * g.Wout - 1) * lddy + g.Cout) * 4;
    const long xb = (((long)g.N * g.Hin * g.Win - 1) * ldx + g.Cin) * 4;
    if (dyb >= (1L << 31) || xb >= (1L << 31)) return U2PL_EINVAL;
    dim3 grid((unsigned)(ctiles * g.R * g.S), (unsigned)cdiv(g.Cout, BM), (unsigned)nsplit);
    U2PL_LAUNCH((k_conv_wgrad<TM, TN, WM>), grid, dim3(128 * WM), lds, stream, dy, lddy, x, ldx, part, g, ctiles, cps,
                       (unsigned)dyb, (unsigned)xb, zdy, zx);
    U2PL_LAUNCH_CHECK();
    return 0;
}

template <int TM, int TN, int SP = 1>
static int launch_wgrad_bf16(const float* dy, long lddy, const float* x, long ldx, float* part, const ConvGeom& g,
                             int ctiles, int nsplit, int cps, hipStream_t stream, long zdy = 0, long zx = 0) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    const size_t lds = (size_t)(SP == 3 ? 3 : 2) * (BM + BN) * LDPW * sizeof(unsigned short);
    const long dyb = (((long)g.N * g.Hout * g.Wout - 1) * lddy + g.Cout) * 4;
    const long xb = (((long)g.N * g.Hin * g.Win - 1) * ldx + g.Cin) * 4;
    if (dyb >= (1L << 31) || xb >= (1L << 31)) return U2PL_EINVAL;
    dim3 grid((unsigned)(ctiles * g.R * g.S), (unsigned)cdiv(g.Cout, BM), (unsigned)nsplit);
    U2PL_LAUNCH((k_conv_wgrad_bf16<TM, TN, SP>), grid, dim3(256), lds, stream, dy, lddy, x, ldx, part, g, ctiles, cps,
                       (unsigned)dyb, (unsigned)xb, zdy, zx);
    U2PL_LAUNCH_CHECK();
    return 0;
}
// weight gradient of nn.Conv2d (autograd of the reference's loss.backward(), train_semi.py:527)
U2PL_API int u2pl_conv2d_wgrad_f32(const float* dy, long lddy, const float* x, long ldx, float* dw,
                                   void* workspace, int accumulate, int N, int Hin, int Win, int Cin, int Hout,
                                   int Wout, int Cout, int R, int S, int stride, int pad, int dil,
                                   hipStream_t stream) {
    if (Cin % 4 || Cout % 4) return U2PL_EINVAL;
    ConvGeom g = {N, Hin, Win, Cin, Hout, Wout, Cout, R, S, stride, -pad, -pad, dil, 0};
    const int BM = Cout > 64 ? 128 : 64, BN = Cin > 64 ? 128 : 64;
    int ct, ns, cps;
    wgrad_plan(g, BM, BN, ct, ns, cps);
    float* part = (float*)workspace;
    int rc;
    if (conv_split() && wgrad_tr_eligible(g)) {
        wgrad_tr_plan(g, R * S, ct, ns, cps);
        rc = launch_wgrad_tr(dy, lddy, x, ldx, part, g, ct, ns, cps, stream, 0, 0);
    } else if (conv_split()) {
        if (BM == 128 && BN == 128) rc = launch_wgrad_bf16<2, 2, 3>(dy, lddy, x, ldx, part, g, ct, ns, cps, stream);
        else if (BM == 128) rc = launch_wgrad_bf16<2, 1, 3>(dy, lddy, x, ldx, part, g, ct, ns, cps, stream);
        else if (BN == 128) rc = launch_wgrad_bf16<1, 2, 3>(dy, lddy, x, ldx, part, g, ct, ns, cps, stream);
        else rc = launch_wgrad_bf16<1, 1, 3>(dy, lddy, x, ldx, part, g, ct, ns, cps, stream);
    }
    else if (BM == 128 && BN == 128)
        rc = wgrad_waves() == 8 ? launch_wgrad<1, 2, 4>(dy, lddy, x, ldx, part, g, ct, ns, cps, stream)
                                : launch_wgrad<2, 2>(dy, lddy, x, ldx, part, g, ct, ns, cps, stream);
    else if (BM == 128) rc = launch_wgrad<2, 1>(dy, lddy, x, ldx, part, g, ct, ns, cps, stream);
    else if (BN == 128) rc = launch_wgrad<1, 2>(dy, lddy, x, ldx, part, g, ct, ns, cps, stream);
    else rc = launch_wgrad<1, 1>(dy, lddy, x, ldx, part, g, ct, ns, cps, stream);
    if (rc) return rc;
    const long wsz = (long)Cout * R * S * Cin;
    U2PL_LAUNCH(k_wgrad_reduce, dim3(grid_for(wsz / 4, 256)), dim3(256), 0, stream, part, wsz, ns, accumulate, dw);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// split-fp16 weight gradient (round 6; conv_geom.h, wgrad_tr.hip): dy_amax / x_amax = device scalars max |dY|, max |X| (or upper
// bounds within ~2^8).  Only the layers k_wgrad_tr serves (u2pl_wgrad_h_eligible); workspace / slab plan as u2pl_conv2d_wgrad_f32.
U2PL_API int u2pl_wgrad_h_eligible(int Cin, int Cout) {
    ConvGeom g = {1, 1, 1, Cin, 1, 1, Cout, 1, 1, 1, 0, 0, 1, 0};
    return conv_split() && wgrad_tr_eligible(g);
}
U2PL_API int u2pl_conv2d_wgrad_h_f32(const float* dy, long lddy, const float* dy_amax, const float* x, long ldx, const float* x_amax,
                                     float* dw, void* workspace, int accumulate, int N, int Hin, int Win, int Cin, int Hout,
                                     int Wout, int Cout, int R, int S, int stride, int pad, int dil, hipStream_t stream) {
    if (Cin % 4 || Cout % 4 || !dy_amax || !x_amax) return U2PL_EINVAL;
    ConvGeom g = {N, Hin, Win, Cin, Hout, Wout, Cout, R, S, stride, -pad, -pad, dil, 0};
    if (!(conv_split() && wgrad_tr_eligible(g))) return U2PL_EINVAL;
    int ct, ns, cps;
    wgrad_tr_plan(g, R * S, ct, ns, cps);
    float* part = (float*)workspace;
    const int rc = launch_wgrad_tr(dy, lddy, x, ldx, part, g, ct, ns, cps, stream, 0, 0, dy_amax, x_amax);
    if (rc) return rc;
    const long wsz = (long)Cout * R * S * Cin;
    U2PL_LAUNCH(k_wgrad_reduce, dim3(grid_for(wsz / 4, 256)), dim3(256), 0, stream, part, wsz, ns, accumulate, dw);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// weight gradient with bf16-rounded dY / X operands (fp32 accumulate); workspace as u2pl_conv2d_wgrad_workspace_bytes
U2PL_API int u2pl_conv2d_wgrad_bf16op_f32(const float* dy, long lddy, const float* x, long ldx, float* dw,
                                          void* workspace, int accumulate, int N, int Hin, int Win, int Cin, int Hout,
                                          int Wout, int Cout, int R, int S, int stride, int pad, int dil,
                                          hipStream_t stream) {
    if (Cin % 4 || Cout % 4) return U2PL_EINVAL;
    ConvGeom g = {N, Hin, Win, Cin, Hout, Wout, Cout, R, S, stride, -pad, -pad, dil, 0};
    const int BM = Cout > 64 ? 128 : 64, BN = Cin > 64 ? 128 : 64;
    int ct, ns, cps;
    wgrad_plan(g, BM, BN, ct, ns, cps);
    float* part = (float*)workspace;
    int rc;
    if (BM == 128 && BN == 128) rc = launch_wgrad_bf16<2, 2>(dy, lddy, x, ldx, part, g, ct, ns, cps, stream);
    else if (BM == 128) rc = launch_wgrad_bf16<2, 1>(dy, lddy, x, ldx, part, g, ct, ns, cps, stream);
    else if (BN == 128) rc = launch_wgrad_bf16<1, 2>(dy, lddy, x, ldx, part, g, ct, ns, cps, stream);
    else rc = launch_wgrad_bf16<1, 1>(dy, lddy, x, ldx, part, g, ct, ns, cps, stream);
    if (rc) return rc;
    const long wsz = (long)Cout * R * S * Cin;
    U2PL_LAUNCH(k_wgrad_reduce, dim3(grid_for(wsz / 4, 256)), dim3(256), 0, stream, part, wsz, ns, accumulate, dw);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// batch independent weight-gradient GEMMs  P_z[Cout][Cin] = dY_z[M][Cout]^T * X_z[M][Cin]  (Winograd components;
// reduction over the M tile rows, split into slabs like the direct wgrad).  part: [nsplit][Cout][batch][Cin]
static ConvGeom batched_wgrad_geom(long M, int Cin, int Cout, int batch) {
    // identity gather for every "tap": step = 0, offsets 0
    ConvGeom g = {1, (int)M, 1, Cin, (int)M, 1, Cout, batch, 1, 1, 0, 0, 0, 0};
    return g;
}
U2PL_API int u2pl_wgrad_batched_splits(long M, int Cin, int Cout, int batch) {
    ConvGeom g = batched_wgrad_geom(M, Cin, Cout, batch);
    int ct, ns, cps;
    if (conv_split() && wgrad_tr_eligible(g)) wgrad_tr_plan(g, batch, ct, ns, cps);
    else wgrad_plan(g, Cout > 64 ? 128 : 64, Cin > 64 ? 128 : 64, ct, ns, cps);
    return ns;
}
U2PL_API size_t u2pl_wgrad_batched_workspace_bytes(long M, int Cin, int Cout, int batch) {
    return (size_t)u2pl_wgrad_batched_splits(M, Cin, Cout, batch) * Cout * batch * Cin * sizeof(float);
}
U2PL_API int u2pl_wgrad_batched_f32(const float* dy, long lddy, long zdy, const float* x, long ldx, long zx,
                                    float* part, long M, int Cin, int Cout, int batch, hipStream_t stream) {
    if (Cin % 4 || Cout % 4 || M <= 0 || M >= (1L << 31)) return U2PL_EINVAL;
    ConvGeom g = batched_wgrad_geom(M, Cin, Cout, batch);
    const int BM = Cout > 64 ? 128 : 64, BN = Cin > 64 ? 128 : 64;
    int ct, ns, cps;
    if (conv_split() && wgrad_tr_eligible(g)) {
        wgrad_tr_plan(g, batch, ct, ns, cps);
        return launch_wgrad_tr(dy, lddy, x, ldx, part, g, ct, ns, cps, stream, zdy, zx);
    }
    wgrad_plan(g, BM, BN, ct, ns, cps);
    if (conv_split()) {
        if (BM == 128 && BN == 128) return launch_wgrad_bf16<2, 2, 3>(dy, lddy, x, ldx, part, g, ct, ns, cps, stream, zdy, zx);
        if (BM == 128) return launch_wgrad_bf16<2, 1, 3>(dy, lddy, x, ldx, part, g, ct, ns, cps, stream, zdy, zx);
        if (BN == 128) return launch_wgrad_bf16<1, 2, 3>(dy, lddy, x, ldx, part, g, ct, ns, cps, stream, zdy, zx);
        return launch_wgrad_bf16<1, 1, 3>(dy, lddy, x, ldx, part, g, ct, ns, cps, stream, zdy, zx);
    }
    if (BM == 128 && BN == 128)
        return wgrad_waves() == 8 ? launch_wgrad<1, 2, 4>(dy, lddy, x, ldx, part, g, ct, ns, cps, stream, zdy, zx)
                                  : launch_wgrad<2, 2>(dy, lddy, x, ldx, part, g, ct, ns, cps, stream, zdy, zx);
    if (BM == 128) return launch_wgrad<2, 1>(dy, lddy, x, ldx, part, g, ct, ns, cps, stream, zdy, zx);
    if (BN == 128) return launch_wgrad<1, 2>(dy, lddy, x, ldx, part, g, ct, ns, cps, stream, zdy, zx);
    return launch_wgrad<1, 1>(dy, lddy, x, ldx, part, g, ct, ns, cps, stream, zdy, zx);
}

// split-fp16 form of u2pl_wgrad_batched_f32 (same slab plan and workspace; u2pl_wgrad_h_eligible(Cin, Cout) layers only)
U2PL_API int u2pl_wgrad_batched_h_f32(const float* dy, long lddy, long zdy, const float* dy_amax, const float* x, long ldx, long zx,
                                      const float* x_amax, float* part, long M, int Cin, int Cout, int batch, hipStream_t stream) {
    if (Cin % 4 || Cout % 4 || M <= 0 || M >= (1L << 31) || !dy_amax || !x_amax) return U2PL_EINVAL;
    ConvGeom g = batched_wgrad_geom(M, Cin, Cout, batch);
    if (!(conv_split() && wgrad_tr_eligible(g))) return U2PL_EINVAL;
    int ct, ns, cps;
    wgrad_tr_plan(g, batch, ct, ns, cps);
    return launch_wgrad_tr(dy, lddy, x, ldx, part, g, ct, ns, cps, stream, zdy, zx, dy_amax, x_amax);
}

// ---------------------------------------------------------------------------
// im2col for the 3-channel stem conv (resnet.py:178): rows [M][Kp] with
// k = (r*S + s)*Cin + ci, zero padded to Kp (multiple of 32); the conv then runs
// as a 1x1 implicit GEMM over the patch matrix.
// ---------------------------------------------------------------------------
__global__ void k_im2col(const float* __restrict__ x, long ldx, float* __restrict__ col, ConvGeom g, int Kp) {
    const long M = (long)g.N * g.Hout * g.Wout;
    const long total = M * Kp;
    const int K = g.R * g.S * g.Cin;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kp);
        const long m = i / Kp;
        float v = 0.f;
        if (k < K) {
            const int ci = k % g.Cin, tap = k / g.Cin;
            const int r = tap / g.S, s = tap % g.S;
            const int wo = (int)(m % g.Wout);
            const long t = m / g.Wout;
            const int ho = (int)(t % g.Hout), n = (int)(t / g.Hout);
            int ih, iw;
            if (gather_coord(ho * g.mul + g.off_h, r, g.step, 0, g.Hin, ih) &&
                gather_coord(wo * g.mul + g.off_w, s, g.step, 0, g.Win, iw))
                v = x[((long)n * g.Hin * g.Win + (long)ih * g.Win + iw) * ldx + ci];
        }
        col[i] = v;
    }
}
U2PL_API int u2pl_im2col_f32(const float* x, long ldx, float* col, int Kp, int N, int Hin, int Win, int Cin,
                             int Hout, int Wout, int R, int S, int stride, int pad, int dil, hipStream_t stream) {
    ConvGeom g = {N, Hin, Win, Cin, Hout, Wout, 0, R, S, stride, -pad, -pad, dil, 0};
    const long total = (long)N * Hout * Wout * Kp;
    U2PL_LAUNCH(k_im2col, dim3(grid_for(total, 256, 1 << 16)), dim3(256), 0, stream, x, ldx, col, g, Kp);
    U2PL_LAUNCH_CHECK();
    return 0;
}
