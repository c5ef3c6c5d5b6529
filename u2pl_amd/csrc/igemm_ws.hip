// Split-fp32 implicit GEMM with PRE-SPLIT weights and a software-pipelined main loop (round 4).
//
// Same arithmetic as k_conv_igemm<..., BF == 3> (conv.hip): every fp32 operand is the exact sum of three bf16 pieces and
// a product is accumulated in fp32 from six piece products (a2b0, a1b1, a0b2, a1b0, a0b1, a0b0 -- in this order, per
// 16-deep K block, K blocks in ascending order), so the outputs are bit-identical to that kernel's.  What changes is
// WHERE the work happens:
//   * the weight operand (B) is split ONCE per optimizer step by k_weight_split3 into bf16 piece planes laid out
//     chunk-major in the exact image the LDS tile has ([K/32][3 pieces][rows][32 k], the 16-byte k segments of a row
//     XOR-swizzled by (row >> 2) & 3): the main loop copies 16-byte units global -> register -> LDS, no VALU work, every
//     global read a full contiguous tile (reference: the weights of resnet.py:25-41,120-140 change once per step,
//     train_semi.py:526-528);
//   * only the activation operand (A) is split in the loop, and that work (22 VALU + 3 ds_write_b64 per float4) is
//     issued BETWEEN the matrix instructions of the previous chunk: two LDS stages, ONE barrier per chunk, the global
//     loads of chunk k+2 issued one full chunk ahead of their use;
//   * 128 x 256 block tiles on 8 waves (64 x 64 per wave): the A split is amortised over twice the matrix
//     instructions of the 128 x 128 tile, LDS operand reads are 25 % of the matrix pipe's time.
// One block per CU (144 KB of LDS).  Tiles are mapped to blocks XCD-aware: consecutive tiles of one XCD share the A rows
// (N tiles fastest), every XCD works on a contiguous range of tiles.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "conv_geom.h"
#include "u2pl_hip.h"

// compile-time loop: f(integral_constant<int, B>) ... f(integral_constant<int, E - 1>) as straight-line code (the pinned main
// loop must not depend on the loop unroller's size budget: a rolled slot loop sends the register arrays to scratch)
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

#define WS_ROW_B 64                      // bytes of one piece row of a 32-deep chunk (32 bf16)
// rows of a split matrix are padded (zero rows) to the widest tile that reads them: no read ever leaves the allocation
__host__ __device__ static inline int ws_pad_rows(int rows) { return rows <= 128 ? 128 : (rows + 255) & ~255; }

// ---------------------------------------------------------------------------------------------------------------
// weight split: w [rows][K] fp32 (K % 32 == 0)  ->  out [K/32][3][Np][32] bf16, Np = rows padded to 128 / a multiple of 256 (zero rows),
// 16-byte segment s of row n stored at slot s ^ ((n >> 2) & 3).  batch: independent matrices (Winograd components).
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_weight_split3(const float* __restrict__ w, long zw, unsigned short* __restrict__ out, long zo, int rows,
                                int Np, int K) {
    w += (long)blockIdx.y * zw;
    out += (long)blockIdx.y * zo;
    const int nseg = K / 8;                                  // 8-element (16-byte output) segments per row
    const long total = (long)Np * nseg;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int sg = (int)(i % nseg), n = (int)(i / nseg);
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        if (n < rows) {
            const float4* src = (const float4*)(w + (long)n * K + sg * 8);
            v0 = src[0];
            v1 = src[1];
        }
        uint2 a0, a1, a2, b0, b1, b2;
        split3_bf16(v0, a0, a1, a2);
        split3_bf16(v1, b0, b1, b2);
        const int c = sg >> 2, s = sg & 3;
        const long base = (((long)c * 3) * Np + n) * 32 + ((s ^ ((n >> 2) & 3)) << 3);       // in bf16 elements
        const long pl = (long)Np * 32;
        *(uint4*)(out + base) = make_uint4(a0.x, a0.y, b0.x, b0.y);
        *(uint4*)(out + base + pl) = make_uint4(a1.x, a1.y, b1.x, b1.y);
        *(uint4*)(out + base + 2 * pl) = make_uint4(a2.x, a2.y, b2.x, b2.y);
    }
}

// every split of a model in ONE launch (u2pl_amd/nn.py presplit, after the optimizer / EMA update).  Job j covers the 16-byte
// output segments [seg_begin_j, seg_begin_j+1) of `batch` matrices; kind 0: src = [batch][rows][K] row-major; kind 1: src =
// the convolution weight [Cout][RS][Cin] read as its transpose [rows = Cin][K = RS * Cout] (the data gradient's operand: what
// u2pl_weight_transpose_f32 + u2pl_weight_split3_f32 produce, without the intermediate).  Same pieces, same layout.
struct SplitJob { const float* src; unsigned short* out; long seg_begin; int rows, Np, K, kind, RS, batch; };
__global__ void k_weight_split3_multi(const SplitJob* __restrict__ jobs, int njobs, long total) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int lo = 0, hi = njobs - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].seg_begin <= i) lo = mid; else hi = mid - 1;
        }
        const SplitJob j = jobs[lo];
        const int nseg = j.K / 8;
        const long per = (long)j.Np * nseg;
        long r = i - j.seg_begin;
        const int z = (int)(r / per);
        r -= (long)z * per;
        int sg, n;
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        if (j.kind == 0) {
            sg = (int)(r % nseg);
            n = (int)(r / nseg);
            if (n < j.rows) {
                const float4* src = (const float4*)(j.src + ((long)z * j.rows + n) * j.K + sg * 8);
                v0 = src[0];
                v1 = src[1];
            }
        } else {          // n fastest: neighbouring lanes read neighbouring input channels
            n = (int)(r % j.Np);
            sg = (int)(r / j.Np);
            if (n < j.rows) {
                const int Cout = j.K / j.RS;
                float e[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int k = sg * 8 + q, rs = k / Cout, co = k - rs * Cout;
                    e[q] = j.src[((long)co * j.RS + rs) * j.rows + n];
                }
                v0 = make_float4(e[0], e[1], e[2], e[3]);
                v1 = make_float4(e[4], e[5], e[6], e[7]);
            }
        }
        uint2 a0, a1, a2, b0, b1, b2;
        split3_bf16(v0, a0, a1, a2);
        split3_bf16(v1, b0, b1, b2);
        const int c = sg >> 2, s = sg & 3;
        const long pl = (long)j.Np * 32;
        unsigned short* out = j.out + (long)z * (j.K / 32) * 3 * pl;
        const long base = (((long)c * 3) * j.Np + n) * 32 + ((s ^ ((n >> 2) & 3)) << 3);       // in bf16 elements
        *(uint4*)(out + base) = make_uint4(a0.x, a0.y, b0.x, b0.y);
        *(uint4*)(out + base + pl) = make_uint4(a1.x, a1.y, b1.x, b1.y);
        *(uint4*)(out + base + 2 * pl) = make_uint4(a2.x, a2.y, b2.x, b2.y);
    }
}
// jobs: device array of njobs SplitJob (48 bytes each: src, out, seg_begin, rows, Np = u2pl_weight_split3_pad_rows(rows), K,
// kind, RS, batch); total = sum over jobs of batch * Np * K / 8
U2PL_API int u2pl_weight_split3_multi_f32(const void* jobs, int njobs, long total, hipStream_t stream) {
    if (njobs <= 0 || total <= 0) return njobs == 0 ? 0 : U2PL_EINVAL;
    U2PL_LAUNCH(k_weight_split3_multi, dim3(grid_for(total, 256, 4096)), dim3(256), 0, stream, (const SplitJob*)jobs, njobs, total);
    U2PL_LAUNCH_CHECK();
    return 0;
}
U2PL_API int u2pl_weight_split3_pad_rows(int rows) { return ws_pad_rows(rows); }

U2PL_API size_t u2pl_weight_split3_bytes(int rows, int K, int batch) {
    return (size_t)batch * (K / 32) * 3 * ws_pad_rows(rows) * WS_ROW_B;
}
// split planes of `batch` row-major fp32 matrices [rows][K] (element stride zw between them; the split copies are
// u2pl_weight_split3_bytes(rows, K, 1) bytes apart).  K % 32 == 0.
U2PL_API int u2pl_weight_split3_f32(const float* w, long zw, int rows, int K, int batch, void* out, hipStream_t stream) {
    if (rows <= 0 || K <= 0 || (K % 32) || batch <= 0) return U2PL_EINVAL;
    const int Np = ws_pad_rows(rows);
    const long total = (long)Np * (K / 8);
    dim3 grid(grid_for(total, 256, 2048), batch);
    U2PL_LAUNCH(k_weight_split3, grid, dim3(256), 0, stream, w, zw, (unsigned short*)out,
                (long)(u2pl_weight_split3_bytes(rows, K, 1) / 2), rows, Np, K);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// split-fp16 weight planes (round 6; conv_geom.h): out = [K/32][2 pieces][Np][32 k] fp16 in the same row padding and swizzle,
// followed by `batch` uint32 -- the fp32 bit pattern of max |w| of each matrix (the GEMM derives the power-of-two scale from
// it with split2_exp_bits, exactly as the split did).  Two launches per rebuild: the maxima (atomicMax on the bit patterns of
// |w|: order-independent, deterministic), then the pieces.
// ---------------------------------------------------------------------------------------------------------------
__host__ __device__ static inline size_t ws2_plane_bytes(int Np, int K, int batch) { return (size_t)batch * (K / 32) * 2 * Np * WS_ROW_B; }
__global__ void k_weight_amax_clear(const SplitJob* __restrict__ jobs, int njobs) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= njobs) return;
    const SplitJob j = jobs[t];
    unsigned* slot = (unsigned*)((char*)j.out + ws2_plane_bytes(j.Np, j.K, j.batch));
    for (int z = 0; z < ((j.batch + 3) & ~3); ++z) slot[z] = 0u;        // (the whole 16-byte-rounded tail: no uninitialised bytes)
}
// block b covers the contiguous segments [b * WS_AMAX_SPAN, (b + 1) * WS_AMAX_SPAN): a thread keeps a running maximum for the
// matrix it is in and publishes when it crosses into another (rare) and at its end -- per wave one atomic where the lanes agree on
// the matrix.  (One atomic per wave and segment-strided blocks: 590 K same-line atomics per rebuild, 27 ms -- measured.)
#define WS_AMAX_SPAN 8192
__global__ __launch_bounds__(256) void k_weight_absmax_multi(const SplitJob* __restrict__ jobs, int njobs, long total) {
    const long begin = (long)blockIdx.x * WS_AMAX_SPAN, end = begin + WS_AMAX_SPAN < total ? begin + WS_AMAX_SPAN : total;
    unsigned* slot = nullptr;
    unsigned m = 0u;
    int lo = 0;
    SplitJob j = jobs[0];
    bool have = false;
    for (long i = begin + threadIdx.x; i < end; i += 256) {
        if (!have || i < j.seg_begin || i >= j.seg_begin + (long)j.batch * j.Np * (j.K / 8)) {
            lo = 0;
            int hi = njobs - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (jobs[mid].seg_begin <= i) lo = mid; else hi = mid - 1;
            }
            j = jobs[lo];
            have = true;
        }
        const int nseg = j.K / 8;
        const long per = (long)j.Np * nseg;
        long r = i - j.seg_begin;
        const int z = (int)(r / per);
        r -= (long)z * per;
        unsigned* sl = (unsigned*)((char*)j.out + ws2_plane_bytes(j.Np, j.K, j.batch)) + z;
        if (sl != slot) {
            if (slot && m) atomicMax(slot, m);
            slot = sl;
            m = 0u;
        }
        // the maximum does not depend on the element order: read the source linearly (both kinds: rows * K floats per matrix)
        const int n = (int)(r / nseg), sg = (int)(r % nseg);
        if (n < j.rows) {
            const float4* src = (const float4*)(j.src + ((long)z * j.rows + n) * j.K + sg * 8);
            m = amax_bits4(amax_bits4(m, src[0]), src[1]);
        }
    }
    // (all lanes arrive here; lanes that never entered the loop hold slot == nullptr)
    const unsigned long long key = (unsigned long long)slot;
    const unsigned long long first = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)key) |
                                     ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(key >> 32)) << 32);
    if (__all(key == first)) {
        m = wave_max_u(m);
        if ((threadIdx.x & 63) == 0 && slot && m) atomicMax(slot, m);
    } else if (slot && m) {
        atomicMax(slot, m);
    }
}
// (block b covers the contiguous segments [b * WS_AMAX_SPAN, (b + 1) * WS_AMAX_SPAN) like the maxima kernel: the job record is
// looked up -- ~9 dependent L2 round trips of a binary search -- when a thread crosses into another job, not per segment)
__global__ __launch_bounds__(256) void k_weight_split2h_multi(const SplitJob* __restrict__ jobs, int njobs, long total) {
    const long begin = (long)blockIdx.x * WS_AMAX_SPAN, end = begin + WS_AMAX_SPAN < total ? begin + WS_AMAX_SPAN : total;
    SplitJob j = jobs[0];
    bool have = false;
    for (long i = begin + threadIdx.x; i < end; i += 256) {
        if (!have || i < j.seg_begin || i >= j.seg_begin + (long)j.batch * j.Np * (j.K / 8)) {
            int lo = 0, hi = njobs - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (jobs[mid].seg_begin <= i) lo = mid; else hi = mid - 1;
            }
            j = jobs[lo];
            have = true;
        }
        const int nseg = j.K / 8;
        const long per = (long)j.Np * nseg;
        long r = i - j.seg_begin;
        const int z = (int)(r / per);
        r -= (long)z * per;
        int sg, n;
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        if (j.kind == 0) {
            sg = (int)(r % nseg);
            n = (int)(r / nseg);
            if (n < j.rows) {
                const float4* src = (const float4*)(j.src + ((long)z * j.rows + n) * j.K + sg * 8);
                v0 = src[0];
                v1 = src[1];
            }
        } else {
            n = (int)(r % j.Np);
            sg = (int)(r / j.Np);
            if (n < j.rows) {
                const int Cout = j.K / j.RS;
                float e[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int k = sg * 8 + q, rs = k / Cout, co = k - rs * Cout;
                    e[q] = j.src[((long)co * j.RS + rs) * j.rows + n];
                }
                v0 = make_float4(e[0], e[1], e[2], e[3]);
                v1 = make_float4(e[4], e[5], e[6], e[7]);
            }
        }
        const unsigned* amax = (const unsigned*)((const char*)j.out + ws2_plane_bytes(j.Np, j.K, j.batch));
        const float sc = split2_scale(split2_exp_bits(amax[z]));
        uint2 a0, a1, b0, b1;
        split2_f16(v0, sc, a0, a1);
        split2_f16(v1, sc, b0, b1);
        const int c = sg >> 2, s_ = sg & 3;
        const long pl = (long)j.Np * 32;
        unsigned short* out = j.out + (long)z * (j.K / 32) * 2 * pl;
        const long base = (((long)c * 2) * j.Np + n) * 32 + ((s_ ^ ((n >> 2) & 3)) << 3);       // in fp16 elements
        *(uint4*)(out + base) = make_uint4(a0.x, a0.y, b0.x, b0.y);
        *(uint4*)(out + base + pl) = make_uint4(a1.x, a1.y, b1.x, b1.y);
    }
}
U2PL_API size_t u2pl_weight_split2h_bytes(int rows, int K, int batch) {
    return ws2_plane_bytes(ws_pad_rows(rows), K, batch) + (((size_t)batch * 4 + 15) & ~(size_t)15);
}
// jobs: the SplitJob table of u2pl_weight_split3_multi_f32 (same fields; out = a u2pl_weight_split2h_bytes buffer).  Three
// launches: the maxima behind each job's planes are cleared, computed, and consumed by the split.
U2PL_API int u2pl_weight_split2h_multi_f32(const void* jobs, int njobs, long total, hipStream_t stream) {
    if (njobs <= 0 || total <= 0) return njobs == 0 ? 0 : U2PL_EINVAL;
    U2PL_LAUNCH(k_weight_amax_clear, dim3(cdiv(njobs, 256)), dim3(256), 0, stream, (const SplitJob*)jobs, njobs);
    U2PL_LAUNCH_CHECK();
    U2PL_LAUNCH(k_weight_absmax_multi, dim3((unsigned)cdiv(total, WS_AMAX_SPAN)), dim3(256), 0, stream, (const SplitJob*)jobs, njobs, total);
    U2PL_LAUNCH_CHECK();
    U2PL_LAUNCH(k_weight_split2h_multi, dim3((unsigned)cdiv(total, WS_AMAX_SPAN)), dim3(256), 0, stream, (const SplitJob*)jobs, njobs, total);
    U2PL_LAUNCH_CHECK();
    return 0;
}
// one weight (batch matrices [rows][K], zw floats apart): clears the maxima, computes them, writes the planes
U2PL_API int u2pl_weight_split2h_f32(const float* w, long zw, int rows, int K, int batch, void* out, void* job_scratch, hipStream_t stream) {
    if (rows <= 0 || K <= 0 || (K % 32) || batch <= 0 || !job_scratch) return U2PL_EINVAL;
    if (batch > 1 && zw != (long)rows * K) return U2PL_EINVAL;
    const int Np = ws_pad_rows(rows);
    SplitJob j = {w, (unsigned short*)out, 0, rows, Np, K, 0, 1, batch};
    const long total = (long)batch * Np * (K / 8);
    hipError_t e = hipMemcpyAsync(job_scratch, &j, sizeof j, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return (int)e;
    return u2pl_weight_split2h_multi_f32(job_scratch, 1, total, stream);
}
// max |x| of an activation operand [M][C] (row pitch ld floats) -> the amax object `out` (U2PL_AMAX_WORDS floats, common.h; NaN if
// any element is NaN); clear != 0: zeroed here first.  The A operand's scale of the *_wsh_* entry points.
__global__ void k_absmax_rows(const float* __restrict__ x, long ld, long M, int C4, unsigned* __restrict__ out) {
    const long total = M * C4;
    unsigned m = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C4;
        const int c = (int)(i - r * C4);
        m = amax_bits4(m, *(const float4*)(x + r * ld + c * 4));
    }
    amax_wave_publish(m, out);
}
U2PL_API int u2pl_amax_words(void) { return U2PL_AMAX_WORDS; }
U2PL_API int u2pl_absmax_f32(const float* x, long ld, long M, int C, float* out, int clear, hipStream_t stream) {
    if (M < 0 || C <= 0 || (C & 3) || (ld & 3)) return U2PL_EINVAL;
    if (clear) {            // (clear == 0: the caller hands a zeroed slot -- one fill for a whole pool of them)
        hipError_t e = hipMemsetAsync(out, 0, (size_t)U2PL_AMAX_WORDS * 4, stream);
        if (e != hipSuccess) return (int)e;
    }
    if (M == 0) return 0;
    U2PL_LAUNCH(k_absmax_rows, dim3(grid_for(M * (C / 4), 512, 2048)), dim3(512), 0, stream, x, ld, M, C / 4, (unsigned*)out);
    U2PL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// the GEMM.  TM x TN 32x32 accumulators per wave, WM x WN waves per block.
// ---------------------------------------------------------------------------------------------------------------
// PW: pointwise gather (1x1, stride 1, no padding -- the GEMM view: 1x1 convolutions and the Winograd component
// products): the A addresses are a per-row constant plus a wave-uniform chunk offset, no per-chunk address arithmetic.
#ifdef U2PL_WS_STAMPS
// debug build (python -m u2pl_amd.build_ext --variant stamps -DU2PL_WS_STAMPS): s_memtime stamps every 4th slot of the pinned
// main loop, chunks 8..11 of blocks 0 and 100, waves 0 (plan 0) and NW/2 (plan 1): [block][plan][chunk][14] uint64
__device__ unsigned long long* d_ws_stamps;
static unsigned long long* g_ws_stamps = nullptr;
U2PL_API int u2pl_igemm_ws_set_stamp_buffer(void* p) { g_ws_stamps = (unsigned long long*)p; return 0; }
#endif
// the kernel body: block `bid` of a group of G persistent blocks that share the tiles [tile0, tile0 + total_tiles) of one
// tile shape (k_igemm_ws: the whole grid is one group; k_igemm_ws_mix: a wide group and a narrow group in one launch)
// NP: pieces per operand.  3: bf16 pieces, six piece products (the round-3/4 arithmetic); 2: fp16 pieces of the operands scaled
// per tensor by a power of two, three piece products (conv_geom.h "split-fp16", round 6): x_amax -> max |x| of the A operand
// (device scalar, read once), b_amax -> max |w| per matrix of the batch (the tail of the split planes, written by the split).
template <int TM, int TN, int WM, int WN, bool PW, int ABL = 0, int NP = 3>
__device__ __forceinline__ void igemm_ws_body(
    const float* __restrict__ x, long ldx, const float* __restrict__ x_amax, const unsigned short* __restrict__ ws,
    const unsigned* __restrict__ b_amax, const float* __restrict__ bias,
    float* __restrict__ y, long ldy, const ConvGeom& g, unsigned xbytes, unsigned wsbytes, unsigned ybytes, unsigned resbytes, int Np,
    float* __restrict__ stats, const float* __restrict__ pivot, long zx, long zws, long zy, const BnEpi& epi, int mtiles, int ntiles,
    int total_tiles, int tile0, const int bid, const int G) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NT = 64 * WM * WN;
    constexpr int RPP = NT / 8, RA = BM / RPP;              // A: 8 threads x float4 per 32-deep row
    constexpr int UB = NP * BN * 4, RBU = UB / NT;          // B: 16-byte units per chunk, per thread
    static_assert(BM % RPP == 0 && UB % NT == 0, "tile / thread-count mismatch");
    static_assert(NP == 2 || NP == 3, "pieces per operand");
    constexpr int A_ST = NP * BM * WS_ROW_B, B_ST = NP * BN * WS_ROW_B, ST = A_ST + B_ST;    // bytes per stage
    constexpr int NPROD = NP == 3 ? 6 : 3;                  // piece products per 16-deep k block
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;

    // PERSISTENT: block b works through the tiles v = b, b + G, b + 2G, ... (G = gridDim.x <= 256, one block per CU) as ONE
    // stream of (tile, chunk) pairs: the loads of the next tile's first chunks are in flight while the last chunks of
    // this tile are multiplied, and the tile's results leave straight from the accumulator registers (plain dword
    // stores, a 128-byte row segment per half wave) while the next tile's products are already being issued.  With one
    // tile per launch slot the 256 blocks ran their load / multiply / store phases in lockstep: for the K = 256 GEMMs
    // (Winograd components, conv3, conv1's data gradient) 13k of every 45k cycles were prologue + epilogue and the
    // output of a whole round (33 MB) hit the memory at once.
    // Virtual tile id -> tile: XCD-aware (block b runs on XCD b % 8 and v = b mod 8; each XCD takes a contiguous range of
    // tiles, N tiles fastest so that neighbours in time on one L2 share their A rows).
    auto tile_of = [&](int v, int& mt, int& nt, int& z) __attribute__((always_inline)) {
        const int xcd = v & 7, j = v >> 3;
        const int q = total_tiles >> 3, r = total_tiles & 7;
        const int t = tile0 + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;     // (tile0: this launch's first tile)
        nt = t % ntiles;
        const int rest = t / ntiles;
        mt = rest % mtiles;
        z = rest / mtiles;
    };
    // buffer descriptors cover ONE matrix of the batch (the range check is the row / column mask of loads and stores) and
    // are rebuilt when a stream enters a tile of another matrix.  (A batch offset in the instructions' scalar offset does
    // not work: the range check adds it to the lane offset -- measured: every access to matrix z > 0 was dropped.)
    __amdgpu_buffer_rsrc_t rx = make_rsrc(x, xbytes), rw = make_rsrc(ws, wsbytes), ry = make_rsrc(y, ybytes);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef U2PL_WS_STAMPS
    unsigned long long ph[4];
    ph[0] = __builtin_readcyclecounter();
#endif

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const long M = (long)g.N * g.Hout * g.Wout;
    const int K = g.R * g.S * g.Cin;
    const int nk = K / BK;
    // split-fp16: the A operand's power-of-two scale (wave-uniform; the weight planes carry theirs already)
    int ea = 0;
    float sa = 1.f;
    if constexpr (NP == 2) {
        ea = split2_exp_bits(amax_read((const unsigned*)x_amax));
        sa = split2_scale(ea);
    }
    // chunks per tile, made EVEN (an odd K / 32 gets one all-zero chunk: out-of-range offsets load zeros, the products add
    // +0): stage and register-set parity then restart at every tile and the tile loop below has ONE epilogue instance
    const int nk2 = (nk + 1) & ~1;
    const int n_my = (total_tiles - bid + G - 1) / G;      // tiles of this block (>= 1: G <= total_tiles)
    // Filter-row range of a tile.  A 128-pixel tile covers one or two image rows: where the dilation is a large part of the
    // map (ASPP: d = 24 / 36 on 97 rows) the filter row r = 0 lies above the image for the top d output rows and r = R - 1
    // below it for the bottom d -- 16 % / 25 % of those layers' chunks multiply zeros.  A tile runs only the filter rows
    // [rb, re) that reach the image for at least one of its pixels (the hull, from the tile's first and last pixel: scalar
    // arithmetic, the same in the three streams); skipped products would have added +-0.
    const int cpt = g.S * (g.Cin / BK);                     // chunks per filter row
    auto tile_taps = [&](int mt, int& rb_, int& re_) __attribute__((always_inline)) {
        rb_ = 0;
        re_ = g.R;
        if constexpr (!PW) {
            if (g.R > 1) {
                const long mf = (long)mt * BM, ml_ = mf + BM - 1, ml = ml_ < M ? ml_ : M - 1;
                const unsigned hw = (unsigned)(g.Hout * g.Wout);
                const unsigned nf = (unsigned)mf / hw, nl = (unsigned)ml / hw;
                const int hf = (int)(((unsigned)mf - nf * hw) / (unsigned)g.Wout), hl = (int)(((unsigned)ml - nl * hw) / (unsigned)g.Wout);
                const int lim = g.Hin << g.log2div;
                auto reach = [&](int r, int ha, int hb) { return hb * g.mul + g.off_h + r * g.step >= 0 && ha * g.mul + g.off_h + r * g.step < lim; };
                int lo = g.R, hi = -1;
                for (int r = 0; r < g.R; ++r) {
                    const bool ok = nf == nl ? reach(r, hf, hl)
                                             : (reach(r, hf, g.Hout - 1) || reach(r, 0, hl) || (nl > nf + 1 && reach(r, 0, g.Hout - 1)));
                    if (ok) { lo = r < lo ? r : lo; hi = r; }
                }
                if (hi >= 0) { rb_ = lo; re_ = hi + 1; } else { rb_ = 0; re_ = 1; }
                rb_ = __builtin_amdgcn_readfirstlane(rb_);
                re_ = __builtin_amdgcn_readfirstlane(re_);
            }
        }
    };

    // ---- A load stream (two chunks ahead of the products): its tile, its chunk / tap state, the rows' addresses
    const int kq = tid & 7, r0 = tid >> 3;
    const int ldxb = (int)ldx * 4;
    int a_v = bid, a_kc = 0, a_c0 = 0, a_r = 0, a_s = 0;
    int a_nk = nk, a_nk2 = nk2;                             // chunks of the A stream's tile (filter rows [rb, re) only)
    int bh[RA], bw[RA], nb[RA];
    bool mv[RA];
    int aoff[RA];         // PW: byte offset of the row's first chunk (OOB_OFF for rows past M)
    auto a_tile_setup = [&]() __attribute__((always_inline)) {
        int mt, nt, z;
        tile_of(a_v, mt, nt, z);
        rx = make_rsrc(x + (long)__builtin_amdgcn_readfirstlane(z) * zx, xbytes);
        if constexpr (!PW) {
            int rb_, re_;
            tile_taps(mt, rb_, re_);
            a_r = rb_;
            a_nk = (re_ - rb_) * cpt;
            a_nk2 = (a_nk + 1) & ~1;
        }
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const long m = (long)mt * BM + r0 + RPP * i;
            mv[i] = m < M;
            const unsigned mm = mv[i] ? (unsigned)m : 0u;
            if constexpr (PW) {
                aoff[i] = mv[i] ? (int)mm * ldxb + kq * 16 : OOB_OFF;
            } else {
                const unsigned tq = mm / (unsigned)g.Wout;
                const int wo = (int)(mm - tq * (unsigned)g.Wout);
                const unsigned n_ = tq / (unsigned)g.Hout;
                const int ho = (int)(tq - n_ * (unsigned)g.Hout);
                bh[i] = ho * g.mul + g.off_h;
                bw[i] = wo * g.mul + g.off_w;
                nb[i] = (int)n_ * g.Hin * g.Win;
            }
        }
    };
    a_tile_setup();
    // A: TWO register sets (chunks of even / odd index): the activation rows come from HBM and are requested TWO chunks
    // before they are split (a chunk lasts ~1.5 us, all 256 blocks request at the same moments: one chunk of distance
    // left 900-2000 cycles of every chunk waiting for them); B (L2-resident weight planes): one set, one chunk ahead.
    float4 ra[2][RA];
    u32x4 rb[RBU];
    auto load_a_row = [&](auto set_c, auto i_c) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_c)::value, i = decltype(i_c)::value;
        const bool live = a_kc < a_nk;
        if constexpr (PW) {
            ra[SET][i] = buf_load4s(rx, live ? aoff[i] : OOB_OFF, a_kc * (BK * 4));
        } else {
            int ih, iw;
            const bool okh = gather_coord(bh[i], a_r, g.step, g.log2div, g.Hin, ih);
            const bool okw = gather_coord(bw[i], a_s, g.step, g.log2div, g.Win, iw);
            const int off = (nb[i] + ih * g.Win + iw) * ldxb + kq * 16;
            ra[SET][i] = buf_load4s(rx, (live & mv[i] & okh & okw) ? off : OOB_OFF, a_c0 * 4);
        }
    };
    // next chunk of the stream; after a tile's last chunk the next tile of this block; after the last tile: stay (the loads
    // are unconditional -- a load under `if` would be waited for with vmcnt(0) --, what they fetch then is never used)
    auto advance_a = [&]() __attribute__((always_inline)) {
        if (a_kc + 1 < a_nk2) {
            ++a_kc;
            a_c0 += BK;
            if (a_c0 == g.Cin) {
                a_c0 = 0;
                if (++a_s == g.S) { a_s = 0; ++a_r; }
            }
        } else if (a_v + G < total_tiles) {
            a_v += G;
            a_kc = a_c0 = a_r = a_s = 0;
            a_tile_setup();         // (sets a_r to the tile's first filter row)
        }
    };
    auto load_a = [&](auto set_c) __attribute__((always_inline)) {
        static_for<0, RA>([&](auto i) __attribute__((always_inline)) { load_a_row(set_c, i); });
        advance_a();
    };
    // ---- B load stream (one chunk ahead): unit u = tid + j * NT of the [3][BN][4] 16-byte units of a chunk; piece u / (BN * 4)
    int boff[RBU];        // byte offset inside the chunk's global image, relative to (chunk, piece 0, row 0 of the tile)
    int blds[RBU];        // byte offset inside the stage's B region
#pragma unroll
    for (int j = 0; j < RBU; ++j) {
        const int u = tid + j * NT, p = u / (BN * 4), wi = u - p * (BN * 4);
        boff[j] = p * Np * WS_ROW_B + wi * 16;
        blds[j] = A_ST + p * BN * WS_ROW_B + wi * 16;
    }
    const int chunk_b = NP * Np * WS_ROW_B;                 // bytes per chunk of the split planes
    int b_v = bid, b_kc = 0, b_toff = 0;
    int b_nk = nk, b_nk2 = nk2, b_base = 0;                 // chunks of the B stream's tile, its first chunk in the planes
    auto b_tile_setup = [&]() __attribute__((always_inline)) {
        int mt, nt, z;
        tile_of(b_v, mt, nt, z);
        rw = make_rsrc(ws + (long)__builtin_amdgcn_readfirstlane(z) * zws, wsbytes);
        b_toff = __builtin_amdgcn_readfirstlane(nt * BN * WS_ROW_B);
        if constexpr (!PW) {
            int rb_, re_;
            tile_taps(mt, rb_, re_);
            b_base = rb_ * cpt;
            b_nk = (re_ - rb_) * cpt;
            b_nk2 = (b_nk + 1) & ~1;
        }
    };
    b_tile_setup();
    auto load_b_unit = [&](auto j_c) __attribute__((always_inline)) {
        constexpr int j = decltype(j_c)::value;
        rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, b_kc < b_nk ? boff[j] : OOB_OFF, b_toff + (b_base + b_kc) * chunk_b, 0);
    };
    // (Measured and dropped: the weight planes by LDS-DMA -- buffer_load_dwordx4 ... lds, a wave copying 1 KB units that are
    // contiguous in the global image and in the stage, no registers, no ds_write.  Bit-identical; with the loads issued in
    // the middle of the chunk -- behind the split, where the compiler's own waits do not cover them -- short-K shapes gained
    // 3-4 % and long-K shapes lost 10 %; issued at the chunk's start through inline assembly with hand-written waits every
    // shape of tools/bench_igemm_ws.py was slower: 5880 us against 5615 us for the set.  The builtin form cannot be placed
    // early at all: the compiler orders every LDS access that may alias a pending LDS-DMA behind it with vmcnt(0).)
    auto advance_b = [&]() __attribute__((always_inline)) {
        if (b_kc + 1 < b_nk2) ++b_kc;
        else if (b_v + G < total_tiles) {
            b_v += G;
            b_kc = 0;
            b_tile_setup();
        }
    };
    auto load_b = [&]() __attribute__((always_inline)) {
        static_for<0, RBU>([&](auto j) __attribute__((always_inline)) { load_b_unit(j); });
        advance_b();
    };
    // A piece rows: [piece][BM][64 B], k segment (kq >> 1) of row r at slot seg ^ ((r >> 2) & 3), half (kq & 1)
    int alds[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int r = r0 + RPP * i;
        alds[i] = r * WS_ROW_B + ((((kq >> 1) ^ ((r >> 2) & 3)) << 4) | ((kq & 1) << 3));
    }
    auto store_b = [&](int stage, int j) __attribute__((always_inline)) { *(u32x4*)(smem + stage * ST + blds[j]) = rb[j]; };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    const int li = lane & 31, lh = lane >> 5;
    const int sw0 = (lh ^ ((li >> 2) & 3)) << 4;            // slot of k segment lh (gk = 0); gk = 1: ^ 32
    const int afr = (wm * 32 * TM + li) * WS_ROW_B, bfr = A_ST + (wn * 32 * TN + li) * WS_ROW_B;
    using frag_t = std::conditional_t<NP == 3, bf16x8, f16x8>;
    auto lda = [&](int stage, int p, int a, int gk) __attribute__((always_inline)) {
        return __builtin_bit_cast(frag_t, *(const uint4*)(smem + stage * ST + afr + p * (BM * WS_ROW_B) + a * (32 * WS_ROW_B) + (sw0 ^ (gk << 5))));
    };
    auto ldb = [&](int stage, int p, int b, int gk) __attribute__((always_inline)) {
        return __builtin_bit_cast(frag_t, *(const uint4*)(smem + stage * ST + bfr + p * (BN * WS_ROW_B) + b * (32 * WS_ROW_B) + (sw0 ^ (gk << 5))));
    };
    struct Frag { frag_t a[NP][TM], b[NP][TN]; };
    // ---- prologue: chunk 0 into stage 0; A(1), B(1), A(2) in flight
    load_a(C0{});
    load_b();
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        uint2 p0, p1, p2;
        unsigned char* d = smem + alds[i];
        if constexpr (NP == 3) {
            split3_bf16(ra[0][i], p0, p1, p2);
            *(uint2*)(d + 2 * BM * WS_ROW_B) = p2;
        } else {
            split2_f16(ra[0][i], sa, p0, p1);
        }
        *(uint2*)d = p0;
        *(uint2*)(d + BM * WS_ROW_B) = p1;
    }
#pragma unroll
    for (int j = 0; j < RBU; ++j) store_b(0, j);
    // (issue order A(1), B(1), A(2) = the order the loop leaves its loads in at every trip: the compiler merges the wait
    // counters of the loop entry and of the back edge, a different order here costs a vmcnt(0) in the loop; and the
    // scheduler must not interleave the three groups)
    __builtin_amdgcn_sched_barrier(0);
    load_a(C1{});
    __builtin_amdgcn_sched_barrier(0);
    load_b();
    __builtin_amdgcn_sched_barrier(0);
    load_a(C0{});
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
#ifdef U2PL_WS_STAMPS
    ph[1] = __builtin_readcyclecounter();
#endif

    constexpr int PER = TM * TN, NMF = 2 * NPROD * PER;     // matrix instructions per 16-deep block product / per chunk
    constexpr int NRD = NP * (TM + TN);                     // operand reads per 16-deep block
#ifndef U2PL_WS_TAILQ2
#define U2PL_WS_TAILQ2 2
#endif
#ifndef U2PL_WS_F0PRE2
#define U2PL_WS_F0PRE2 (TM + TN)
#endif
    // products of a chunk issued BEHIND its barrier, in front of the next chunk's (they cover the latency of the next chunk's first
    // operand reads): two of the six piece products' worth at NP = 3; NP = 2: U2PL_WS_TAILQ2 x PER (<= 3 PER = all of k block 1)
    constexpr int TAIL = (NP == 2 ? U2PL_WS_TAILQ2 : 2) * PER;
    Frag f[2];
    // matrix instruction I of a chunk: k block I / (6 PER), product (I / PER) % 6, accumulator I % PER
    auto do_mfma = [&](auto i_c) __attribute__((always_inline)) {
        constexpr int I = decltype(i_c)::value;
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};       // NP == 3: a2b0 a1b1 a0b2 a1b0 a0b1 a0b0
        constexpr int QA[3] = {1, 0, 0}, QB[3] = {0, 1, 0};                         // NP == 2: a1b0 a0b1 a0b0
        constexpr int gk = I / (NPROD * PER), q = (I / PER) % NPROD, ab = I % PER, a = ab / TN, b = ab % TN;
        if constexpr (NP == 3) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[gk].a[PA[q]][a], f[gk].b[PB[q]][b], acc[a][b], 0, 0, 0);
        else acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[gk].a[QA[q]][a], f[gk].b[QB[q]][b], acc[a][b], 0, 0, 0);
    };
    // operand read J of a k block, in the order the products need them: a2.., b0.., a1.., b1.., a0.., b2..
    auto do_read = [&](auto stage_c, auto gk_c, auto j_c) __attribute__((always_inline)) {
        constexpr int stage = decltype(stage_c)::value, gk = decltype(gk_c)::value, J = decltype(j_c)::value;
        if constexpr (J < NRD) {
            constexpr int q = J / (TM + TN), w = J % (TM + TN);
            if constexpr (w < TM) f[gk].a[NP - 1 - q][w] = lda(stage, NP - 1 - q, w, gk);
            else f[gk].b[q][w - TM] = ldb(stage, q, w - TM, gk);
        }
    };
    auto zero_tail_operands = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int a = 0; a < TM; ++a) f[1].a[p][a] = __builtin_bit_cast(frag_t, make_uint4(0, 0, 0, 0));
#pragma unroll
            for (int b = 0; b < TN; ++b) f[1].b[p][b] = __builtin_bit_cast(frag_t, make_uint4(0, 0, 0, 0));
        }
    };

    // Main loop, issue order pinned (a scheduling barrier after every matrix instruction and the work placed behind it);
    // two chunks per trip so that the LDS stage and the A register set of a chunk are compile-time constants.
    // The barrier of a chunk sits INSIDE the matrix-instruction stream: the last TAIL products of a chunk (operands in
    // registers) are issued after the barrier, in front of the next chunk's, and cover the latency of its first operand
    // reads.  Before a tile's first chunk the tail runs on all-zero operands (adds +0 to +0).  Per accumulator the order of
    // the products is: k blocks ascending, six piece products each.
    // (Tried and dropped: a second wave group half a chunk out of phase -- no gain, 23 more registers; a compiler-scheduled
    // loop behind a sched_group_barrier pipeline -- same time as the pinned order, no control.)
    zero_tail_operands();
    if constexpr (ABL & 16) {   // (timing experiment: operands never read from LDS)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int a = 0; a < TM; ++a) f[0].a[p][a] = __builtin_bit_cast(frag_t, make_uint4(tid, 1, 2, 3));
#pragma unroll
            for (int b = 0; b < TN; ++b) f[0].b[p][b] = __builtin_bit_cast(frag_t, make_uint4(tid, 1, 2, 3));
        }
    }
    uint2 pc[3];          // pieces of the float4 being split
    float sp_l = 0.f, sp_h = 0.f;     // the pair being split: residuals after the pieces taken so far
    int kc = 0;                       // chunk of the current tile (the products' stream; debug stamps only)
    auto chunk = [&](auto par_c) __attribute__((always_inline)) {
        constexpr int cur = decltype(par_c)::value, nxt = cur ^ 1;
        // Per slot (= per matrix instruction) at most one operand read / store / global load and <= 5 VALU operations: a
        // wave alone then issues its matrix instructions back to back (32 cycles apart) with everything else in the gaps.
        // With the work in clumps (a whole value pair split in one slot: 11 VALU + 3 stores) the wave that wins the
        // arbitration for the matrix pipe (the older one) ran its chunk in 2160 cycles instead of 1536 while its SIMD
        // partner starved, then the partner ran its clumps with nobody's matrix instructions to cover them: 3990 cycles per
        // chunk for 3072 of matrix work (s_memtime stamps of both waves).
        //   slots 0 .. TAIL-1           tail products of the previous chunk | operand reads of k block 0
        //   TAIL .. TAIL+NRD-1          products of k block 0               | operand reads of k block 1, one per slot
        //   TAIL .. TAIL+6RA-1                                              | split of A(kc+1): 3 steps per value pair (5, 5, 1
        //                                                                     VALU), the three piece stores behind a float4's last
        //   then RBU slots                                                  | weight-piece stores, one per slot
        //   then RBU + RA slots                                             | loads B(kc+2), then A(kc+3), one per slot
        //   NP == 2 (half the matrix instructions for 2/3 of the memory operations): the split starts at slot 0 (two steps per
        //   value pair: 7 + 1 VALU), the operand reads share its slots; where the plan would overrun the chunk the loads move up
        //   beside the weight-piece stores (a store precedes the load that refills its register in the same slot)
        constexpr int F0_PRE = NP == 2 ? (U2PL_WS_F0PRE2) : TM + TN, F0_PER = (NRD - F0_PRE + TAIL - 1) / TAIL;
        constexpr int SPS = NP, SP_N = SPS * 2 * RA;          // split steps per value pair; slots of the split
        constexpr int F1_START = TAIL, SP_START = NP == 3 ? TAIL : 0, BS_START = SP_START + SP_N;
        constexpr int LD_START = (BS_START + RBU + RBU + RA <= NMF) ? BS_START + RBU : NMF - RBU - RA;
        static_assert(F1_START + NRD <= TAIL + NPROD * PER, "k block 1 operands would be read after their first use");
        static_assert(LD_START + RBU + RA <= NMF && LD_START >= BS_START && LD_START >= SP_START + SP_N, "plan does not fit the chunk");
        using CUR = std::integral_constant<int, cur>;
#ifdef U2PL_WS_STAMPS
        unsigned long long ts[14];
        ts[0] = __builtin_readcyclecounter();
#endif
        if constexpr (!(ABL & 16)) static_for<0, F0_PRE>([&](auto j) __attribute__((always_inline)) { do_read(CUR{}, C0{}, j); });
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, NMF>([&](auto sl_c) __attribute__((always_inline)) {
            constexpr int sl = decltype(sl_c)::value;
#ifdef U2PL_WS_STAMPS
            if constexpr (sl % 4 == 0 && sl != 0) ts[sl / 4] = __builtin_readcyclecounter();
#endif
            do_mfma(std::integral_constant<int, (sl < TAIL ? NMF - TAIL + sl : sl - TAIL)>{});
            if constexpr (!(ABL & 16)) {
                if constexpr (sl < TAIL)
                    static_for<0, F0_PER>([&](auto u) __attribute__((always_inline)) { do_read(CUR{}, C0{}, std::integral_constant<int, F0_PRE + sl * F0_PER + decltype(u)::value>{}); });
                if constexpr (sl >= F1_START) do_read(CUR{}, C1{}, std::integral_constant<int, sl - F1_START>{});
            }
            if constexpr (sl >= SP_START && sl < SP_START + SP_N) {
                constexpr int k = (sl - SP_START) / SPS, step = (sl - SP_START) % SPS;     // value pair k of A(kc + 1), step
                if constexpr (step == 0) {
                    const float4 v = ra[nxt][k >> 1];
                    sp_l = (k & 1) ? v.z : v.x;
                    sp_h = (k & 1) ? v.w : v.y;
                }
                unsigned w;
                if constexpr (ABL & 1) w = (__float_as_uint(sp_h) & 0xffff0000u) | (__float_as_uint(sp_l) >> 16);
                else if constexpr (NP == 2) {       // (conv_geom.h: v_fma_mixlo / mixhi, 2 + 3 instructions per value pair)
                    if constexpr (step == 0) w = split2_first(sp_l, sp_h, sa);
                    else w = split2_second(sp_l, sp_h, sa, (k & 1) ? pc[0].y : pc[0].x);
                } else {
                    if constexpr (step == 0) w = pack2_bf16_first(sp_l, sp_h);      // (first piece: overflow guard, conv_geom.h)
                    else w = pack2_bf16(sp_l, sp_h);
                    if constexpr (step < 2) {
                        sp_l = sp_l - bf16_lo_f(w);
                        sp_h = sp_h - bf16_hi_f(w);
                    }
                }
                if constexpr (k & 1) pc[step].y = w; else pc[step].x = w;
                if constexpr ((k & 1) && step == SPS - 1) {
                    if constexpr (ABL & 4) {
                        asm volatile("" ::"v"(pc[0].x), "v"(pc[0].y), "v"(pc[1].x), "v"(pc[1].y));
                        if constexpr (NP == 3) asm volatile("" ::"v"(pc[2].x), "v"(pc[2].y));
                    } else {
                        unsigned char* d = smem + nxt * ST + alds[k >> 1];
                        *(uint2*)d = pc[0];
                        *(uint2*)(d + BM * WS_ROW_B) = pc[1];
                        if constexpr (NP == 3) *(uint2*)(d + 2 * BM * WS_ROW_B) = pc[2];
                    }
                }
            }
            if constexpr (sl >= BS_START && sl < BS_START + RBU) {
                constexpr int k = sl - BS_START;
                if constexpr (ABL & 2) { asm volatile("" ::"v"(rb[k].x), "v"(rb[k].y), "v"(rb[k].z), "v"(rb[k].w)); }
                else store_b(nxt, k);
            }
            if constexpr (!(ABL & 8)) {
                if constexpr (sl >= LD_START && sl < LD_START + RBU) {              // B(kc + 2), consumed in the next chunk
                    load_b_unit(std::integral_constant<int, sl - LD_START>{});
                    if constexpr (sl == LD_START + RBU - 1) advance_b();
                }
                if constexpr (sl >= LD_START + RBU && sl < LD_START + RBU + RA) {   // A(kc + 3), consumed two chunks on
                    load_a_row(std::integral_constant<int, nxt>{}, std::integral_constant<int, sl - LD_START - RBU>{});
                    if constexpr (sl == LD_START + RBU + RA - 1) advance_a();
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
#ifdef U2PL_WS_STAMPS
        ts[12] = __builtin_readcyclecounter();
#endif
        if constexpr (!(ABL & 32)) __syncthreads();
#ifdef U2PL_WS_STAMPS
        ts[13] = __builtin_readcyclecounter();
        {
            const int bsel = blockIdx.x == 0 ? 0 : blockIdx.x == 100 ? 1 : -1;
            if (d_ws_stamps && bsel >= 0 && kc >= 8 && kc < 12 && (wave == 0 || wave == (WM * WN) / 2) && lane == 0)
                for (int i = 0; i < 14; ++i) d_ws_stamps[((bsel * 2 + (wave != 0)) * 4 + (kc - 8)) * 14 + i] = ts[i];
        }
#endif
    };

    // ---- end of a tile: the chunk's tail products, then the results leave from the accumulator registers.
    //      (Measured alternatives: swapped matrix operands put four consecutive COLUMNS of one row into a lane -- 16-byte stores,
    //      but one 16-byte piece per row and lane: 14k cycles for the tile's stores against 6.3k for these dword stores with a
    //      128-byte row segment per half wave; the LDS-transposed 16-byte stores of conv.hip's epilogue cost ~9k with their drain.)
    //      Store offsets: element (row m, column c) at (m * ldy + c) * 4 in the matrix's descriptor; a row past M lies past the
    //      descriptor's extent (ldy >= Cout) and a column past Cout is given OOB_OFF: the hardware drops both, no compares.
    int c_v = bid;
    int c_nk2 = nk2;                  // chunks (even) of the products' tile
    auto c_tile_setup = [&]() __attribute__((always_inline)) {
        if constexpr (!PW) {
            if (c_v < total_tiles) {
                int mt, nt, z, rb_, re_;
                tile_of(c_v, mt, nt, z);
                tile_taps(mt, rb_, re_);
                c_nk2 = ((re_ - rb_) * cpt + 1) & ~1;
            }
        }
    };
    c_tile_setup();
    float* red = (float*)(smem + 2 * ST);   // [4][2][BN] statistics scratch BEHIND the two stages (stage `nxt` already holds the next tile)
    unsigned y_am = 0u;                     // fused eval-BN epilogue: running max |y| over this block's tiles (published once, at the end)
#ifdef U2PL_WS_STAMPS
    int te_n = 0;
#endif
    auto tile_end = [&]() __attribute__((always_inline)) {
#ifdef U2PL_WS_STAMPS
        unsigned long long te[5];
        te[0] = __builtin_readcyclecounter();
#endif
        int mt, nt, z;
        tile_of(c_v, mt, nt, z);
        const int n0 = nt * BN;
        const long m0 = (long)mt * BM;
        // lane-derived values are recomputed per tile from an opaque copy of the thread index: hoisted out of the tile loop they
        // would live across the main loop, whose register budget is full -- they were spilled, and a spill reload behind the
        // stores waits for the stores
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
        (void)lane;
        const int zs = __builtin_amdgcn_readfirstlane(z);
        ry = make_rsrc(y + (long)zs * zy, ybytes);
        const bool bn_all = epi.mean != nullptr;
        // Every memory READ of the epilogue comes before its first STORE: loads and stores retire through one in-order
        // counter, so a load behind the tile's 64 stores -- a per-column parameter, a residual value, even the reload of a
        // spilled address -- waits for 128 KB per CU to drain (measured: +10 us per tile for the statistics variant of the
        // 256 -> 1024 convolution, whose row indices had been spilled; +170 us per launch for per-element residual loads).
        // Per-column parameters through range-checked descriptors (a column past Cout reads 0).
        const unsigned cbytes = (unsigned)g.Cout * 4u;
        auto colparam = [&](const float* p, int col) __attribute__((always_inline)) {
            return p ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(make_rsrc(p, cbytes), col * 4, 0, 0)) : 0.f;
        };
        float bias_v[TN], mu[TN], is[TN], ga[TN], be[TN], pv[TN];
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col = n0 + wn * 32 * TN + b * 32 + li;
            bias_v[b] = colparam(bias, col);
            mu[b] = is[b] = ga[b] = be[b] = pv[b] = 0.f;
            if (bn_all) { mu[b] = colparam(epi.mean, col); is[b] = colparam(epi.invstd, col); ga[b] = colparam(epi.gamma, col); be[b] = colparam(epi.beta, col); }
            if (stats) pv[b] = colparam(pivot, col);
        }
        // split-fp16: the weight matrix's scale exponent (a scalar load, in front of every store like the parameters)
        int dexp = 0;
        if constexpr (NP == 2) dexp = -(ea + split2_exp_bits(__builtin_amdgcn_readfirstlane((int)ldg(b_amax + zs))));
        // (the parameter loads are in flight under the chunk's tail products)
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, TAIL>([&](auto sl) __attribute__((always_inline)) { do_mfma(std::integral_constant<int, NMF - TAIL + decltype(sl)::value>{}); });
        if constexpr (NP == 2) {        // back to the operands' own scale: exact (a power of two), one v_ldexp_f32 per value
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[a][b][e] = __builtin_ldexpf(acc[a][b][e], dexp);
        }
#ifdef U2PL_WS_STAMPS
        te[1] = te[2] = __builtin_readcyclecounter();
#endif
        __builtin_amdgcn_sched_barrier(0);
        // Results.  Element (row m, column c) lies at (m * ld + c) * 4 in its matrix's descriptor; a row past M lies past the
        // descriptor's extent (ld >= Cout) and a column past Cout is given OOB_OFF: the hardware drops / zero-fills both, no
        // compares.  (m * ld fits 32 bits: host check.)
        //   (Measured and dropped: transposing the accumulators within each quad of lanes -- two rounds of quad-permute DPP
        //   moves + selects -- so that a lane holds four consecutive columns of one row and an instruction stores 8 rows x 128
        //   contiguous bytes as 16-byte pieces: 121 us against 110 us for these dword stores on 256 -> 1024 at 4 x 97^2.)
        const int ldyb = (int)ldy * 4;
        const int mrow = (int)m0 + wm * 32 * TM + 4 * lh;          // row of (a = 0, e = 0)
        if (bn_all) {       // eval-mode BatchNorm (+ residual, ReLU) in place: the operations of k_bn_apply, in its order
            const int ldrb = (int)epi.ldr * 4;
            const __amdgpu_buffer_rsrc_t rr = make_rsrc(epi.res, resbytes);
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int col = n0 + wn * 32 * TN + b * 32 + li;
                float rv[TM][16];
                if (epi.res) {
#pragma unroll
                    for (int a = 0; a < TM; ++a) {
                        const int rbase = col < g.Cout ? ((mrow + a * 32) * (int)epi.ldr + col) * 4 : OOB_OFF;
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            rv[a][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr, rbase + ((e & 3) + 8 * (e >> 2)) * ldrb, 0, 0));
                    }
                }
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float v = acc[a][b][e] + bias_v[b];
                        v = (v - mu[b]) * is[b] * ga[b] + be[b];
                        if (epi.res) v += rv[a][e];
                        if (epi.relu) v = fmaxf(v, 0.f);
                        acc[a][b][e] = v;
                        // (elements outside the matrix -- rows past M, columns past Cout -- are dropped by the store; their values
                        //  come from zero-filled operands and the epilogue's parameters: masked out of the maximum)
                        if (epi.y_amax && col < g.Cout && (long)(mrow + a * 32 + (e & 3) + 8 * (e >> 2)) < M) y_am = amax_bits(y_am, v);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);      // (no store may move in front of a load)
        }
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col = n0 + wn * 32 * TN + b * 32 + li;
            const float badd = bn_all ? 0.f : bias_v[b];
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int vbase = col < g.Cout ? ((mrow + a * 32) * (int)ldy + col) * 4 : OOB_OFF;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float v = bn_all ? acc[a][b][e] : acc[a][b][e] + badd;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, vbase + ((e & 3) + 8 * (e >> 2)) * ldyb, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // fused BatchNorm statistics: per 128-row tile pivot-shifted column sums [tile][2][Cout], computed BEHIND the stores
        // (register and LDS work only -- no load, no spill reload: checked in the ISA -- so it runs while the stores drain).  The additions are made in the order of k_conv_igemm<1, 2, 4, 3> (per 32-row wave block: 16
        // register values, the two lane halves, then the four 32-row blocks of the tile in ascending order) so the partial
        // sums are the same bits.
        if (stats && !bn_all) {
            static_assert(BM == 128, "statistics blocks are 128 rows");
            const long left = M - m0;                                                  // rows of this tile inside the matrix
            const int lim = (int)(left < BM ? left : BM) - (wm * 32 * TM + 4 * lh);    // row a * 32 + ro of this lane counts iff < lim
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int cl = wn * 32 * TN + b * 32 + li;
                const float sh = bias_v[b] - pv[b];
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float v = (a * 32 + (e & 3) + 8 * (e >> 2)) < lim ? acc[a][b][e] + sh : 0.f;
                        s1 += v;
                        s2 += v * v;
                    }
                    s1 += __shfl_xor(s1, 32, 64);
                    s2 += __shfl_xor(s2, 32, 64);
                    if (lh == 0) {
                        red[((wm * TM + a) * 2 + 0) * BN + cl] = s1;
                        red[((wm * TM + a) * 2 + 1) * BN + cl] = s2;
                    }
                }
            }
            __syncthreads();
            float* out = stats + (long)mt * 2 * g.Cout;
            for (int c = tid; c < BN; c += NT) {
                const int co = n0 + c;
                if (co < g.Cout) {
                    float a1 = red[0 * BN + c], a2 = red[1 * BN + c];
#pragma unroll
                    for (int r = 1; r < 4; ++r) { a1 += red[(2 * r) * BN + c]; a2 += red[(2 * r + 1) * BN + c]; }
                    out[co] = a1;
                    out[g.Cout + co] = a2;
                }
            }
            // (the next write of `red` is a whole tile -- at least one chunk barrier -- away)
        }
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
        zero_tail_operands();
        c_v += G;
        c_tile_setup();
#ifdef U2PL_WS_STAMPS
        te[3] = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        te[4] = __builtin_readcyclecounter();
        if (d_ws_stamps && blockIdx.x == 0 && (wave == 0 || wave == 5) && lane == 0 && te_n < 4)
            for (int i = 0; i < 5; ++i) d_ws_stamps[224 + 4096 + ((wave != 0) * 4 + te_n) * 5 + i] = te[i];
        ++te_n;
#endif
    };
    // (the inner loop's back edge comes from the odd chunk only and its entry states -- prologue, end of a tile -- leave the
    // loads in the same order: the compiler merges the wait-counter states at the loop header)
    for (int it = 0; it < n_my; ++it) {
        for (kc = 0; kc < c_nk2; kc += 2) {
            chunk(C0{});
            chunk(C1{});
        }
        tile_end();
    }
    if (epi.y_amax) amax_wave_publish(y_am, epi.y_amax);      // (wave-uniform branch; every wave of the block gets here)
#ifdef U2PL_WS_STAMPS
    ph[2] = ph[3] = __builtin_readcyclecounter();
    if (d_ws_stamps && blockIdx.x < 1024 && tid == 0)
        for (int i = 0; i < 4; ++i) d_ws_stamps[224 + blockIdx.x * 4 + i] = ph[i];
#endif
}

template <int TM, int TN, int WM, int WN, bool PW, int ABL = 0, int NP = 3>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) / 4) void k_igemm_ws(
    const float* __restrict__ x, long ldx, const float* __restrict__ x_amax, const unsigned short* __restrict__ ws,
    const unsigned* __restrict__ b_amax, const float* __restrict__ bias,
    float* __restrict__ y, long ldy, ConvGeom g, unsigned xbytes, unsigned wsbytes, unsigned ybytes, unsigned resbytes, int Np,
    float* __restrict__ stats, const float* __restrict__ pivot, long zx, long zws, long zy, BnEpi epi, int mtiles, int ntiles,
    int total_tiles, int tile0) {
    igemm_ws_body<TM, TN, WM, WN, PW, ABL, NP>(x, ldx, x_amax, ws, b_amax, bias, y, ldy, g, xbytes, wsbytes, ybytes, resbytes, Np, stats,
                                               pivot, zx, zws, zy, epi, mtiles, ntiles, total_tiles, tile0, (int)blockIdx.x, (int)gridDim.x);
}
// MIXED tile plan in ONE launch: blocks [0, nwide) are the persistent 128 x 256 group over the wide tiles [0, wide_tiles)
// (whole rounds: wide_tiles is a multiple of nwide), the blocks behind them take ONE 128 x 128 tile each of the remainder
// (narrow tiles [2 wide_tiles, 2 wide_tiles + narrow_tiles): a narrow tile (z, mt, 2 nt + h) is half h of the wide tile
// (z, mt, nt)).  A block needs a whole CU's LDS, so the hardware hands the narrow blocks to the CUs as the wide blocks
// retire -- the remainder round starts without a second launch's drain, gap and cold start (~15 us measured).
template <bool PW, int NP = 3>
__global__ __launch_bounds__(512, 2) void k_igemm_ws_mix(
    const float* __restrict__ x, long ldx, const float* __restrict__ x_amax, const unsigned short* __restrict__ ws,
    const unsigned* __restrict__ b_amax, const float* __restrict__ bias,
    float* __restrict__ y, long ldy, ConvGeom g, unsigned xbytes, unsigned wsbytes, unsigned ybytes, unsigned resbytes, int Np,
    float* __restrict__ stats, const float* __restrict__ pivot, long zx, long zws, long zy, BnEpi epi, int mtiles, int ntiles_w,
    int nwide, int wide_tiles, int narrow_tiles) {
    if ((int)blockIdx.x < nwide)
        igemm_ws_body<2, 2, 2, 4, PW, 0, NP>(x, ldx, x_amax, ws, b_amax, bias, y, ldy, g, xbytes, wsbytes, ybytes, resbytes, Np, stats, pivot,
                                             zx, zws, zy, epi, mtiles, ntiles_w, wide_tiles, 0, (int)blockIdx.x, nwide);
    else
        igemm_ws_body<2, 1, 2, 4, PW, 0, NP>(x, ldx, x_amax, ws, b_amax, bias, y, ldy, g, xbytes, wsbytes, ybytes, resbytes, Np, stats, pivot,
                                             zx, zws, zy, epi, mtiles, 2 * ntiles_w, narrow_tiles, 2 * wide_tiles, (int)blockIdx.x - nwide,
                                             (int)gridDim.x - nwide);
}

#define WS_NUM_CUS 256
// U2PL_WS_PERSIST = 1 (default): at most one block per CU, each working through its tiles; 0: one block per tile (A/B switch,
// same results; u2pl_igemm_ws_set_persist returns the previous value)
static double ws_fix2() {        // U2PL_WS_FIX2: fixed-cost multiplier of the tile-plan model for the split-fp16 kernels
    static double v = -1.0;
    if (v < 0) { const char* e = getenv("U2PL_WS_FIX2"); v = (e && *e) ? atof(e) : 2.0; }
    return v;
}
static int g_ws_persist = -1;
static int ws_persist() {
    if (g_ws_persist < 0) { const char* e = getenv("U2PL_WS_PERSIST"); g_ws_persist = (e && *e) ? (atoi(e) != 0) : 1; }
    return g_ws_persist;
}
U2PL_API int u2pl_igemm_ws_set_persist(int on) { const int old = ws_persist(); g_ws_persist = on != 0; return old; }
template <int TM, int TN, int WM, int WN, bool PW, int ABL = 0, int NP = 3>
static int launch_igemm_ws(const float* x, long ldx, const float* x_amax, const void* ws, const float* bias, float* y, long ldy,
                           const ConvGeom& g, hipStream_t stream, float* stats, const float* pivot, int batch, long zx,
                           long zy, const BnEpi* epi, long tile_first = 0, long tile_count = -1) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    const long M = (long)g.N * g.Hout * g.Wout;
    if (M <= 0) return 0;
    const int K = g.R * g.S * g.Cin, Np = ws_pad_rows(g.Cout);
    const BnEpi ep = epi ? *epi : BnEpi{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr};
    const size_t lds = (size_t)2 * NP * (BM + BN) * WS_ROW_B + (size_t)8 * BN * sizeof(float);      // two stages + statistics scratch
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_igemm_ws<TM, TN, WM, WN, PW, ABL, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    if (NP == 2 && !x_amax) return U2PL_EINVAL;
    const long xb = (((long)g.N * g.Hin * g.Win - 1) * ldx + g.Cin) * 4;
    const long yb = ((M - 1) * ldy + g.Cout) * 4;
    const long wsb1 = (long)(K / 32) * NP * Np * WS_ROW_B;          // one matrix
    const unsigned* b_amax = (const unsigned*)((const char*)ws + (size_t)batch * wsb1);       // split-fp16: behind the planes
    // every MATRIX of the batch within 2 GiB (32-bit byte offsets inside a matrix); the batch itself may be larger: the kernel
    // rebuilds the buffer descriptors per matrix from a 64-bit base (x + z * zx, ws + z * zws, y + z * zy)
    if (xb >= (1L << 31) || wsb1 >= (1L << 31) || yb >= (1L << 31)) return U2PL_EINVAL;
    const long resb = ep.res ? ((M - 1) * ep.ldr + g.Cout) * 4 : 0;
    if (resb >= (1L << 31)) return U2PL_EINVAL;
    const int mtiles = cdiv(M, BM), ntiles = cdiv(g.Cout, BN);
    const long all_tiles = (long)mtiles * ntiles * batch;
    if (all_tiles >= (1L << 30)) return U2PL_EINVAL;
    const long total = tile_count < 0 ? all_tiles : tile_count;      // this launch: tiles [tile_first, tile_first + total)
    if (total <= 0) return 0;
    const unsigned grid = (unsigned)((total < WS_NUM_CUS || !ws_persist()) ? total : WS_NUM_CUS);
#ifdef U2PL_WS_STAMPS
    (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(d_ws_stamps), &g_ws_stamps, sizeof(void*), 0, hipMemcpyHostToDevice, stream);
#endif
    U2PL_LAUNCH((k_igemm_ws<TM, TN, WM, WN, PW, ABL, NP>), dim3(grid), dim3(64 * WM * WN), lds, stream, x, ldx, x_amax,
                (const unsigned short*)ws, b_amax, bias, y, ldy, g, (unsigned)xb, (unsigned)wsb1, (unsigned)yb, (unsigned)resb, Np, stats, pivot, zx,
                wsb1 / 2, zy, ep, mtiles, ntiles, (int)total, (int)tile_first);
    U2PL_LAUNCH_CHECK();
    return 0;
}

template <bool PW, int NP = 3>
static int launch_igemm_ws_mix(const float* x, long ldx, const float* x_amax, const void* ws, const float* bias, float* y, long ldy,
                               const ConvGeom& g, hipStream_t stream, float* stats, const float* pivot, int batch, long zx,
                               long zy, const BnEpi* epi, long wide_tiles, long narrow_tiles) {
    const long M = (long)g.N * g.Hout * g.Wout;
    const int K = g.R * g.S * g.Cin, Np = ws_pad_rows(g.Cout);
    const BnEpi ep = epi ? *epi : BnEpi{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr};
    const size_t lds = (size_t)2 * NP * (128 + 256) * WS_ROW_B + (size_t)8 * 256 * sizeof(float);       // the wide body's
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_igemm_ws_mix<PW, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    if (NP == 2 && !x_amax) return U2PL_EINVAL;
    const long xb = (((long)g.N * g.Hin * g.Win - 1) * ldx + g.Cin) * 4;
    const long yb = ((M - 1) * ldy + g.Cout) * 4;
    const long wsb1 = (long)(K / 32) * NP * Np * WS_ROW_B;
    const unsigned* b_amax = (const unsigned*)((const char*)ws + (size_t)batch * wsb1);
    if (xb >= (1L << 31) || wsb1 >= (1L << 31) || yb >= (1L << 31)) return U2PL_EINVAL;
    const long resb = ep.res ? ((M - 1) * ep.ldr + g.Cout) * 4 : 0;
    if (resb >= (1L << 31)) return U2PL_EINVAL;
    const int mtiles = cdiv(M, 128), ntw = cdiv(g.Cout, 256);
    if ((long)mtiles * 2 * ntw * batch >= (1L << 30)) return U2PL_EINVAL;
    U2PL_LAUNCH((k_igemm_ws_mix<PW, NP>), dim3((unsigned)(WS_NUM_CUS + narrow_tiles)), dim3(512), lds, stream, x, ldx, x_amax,
                (const unsigned short*)ws, b_amax, bias, y, ldy, g, (unsigned)xb, (unsigned)wsb1, (unsigned)yb, (unsigned)resb, Np, stats, pivot, zx,
                wsb1 / 2, zy, ep, mtiles, ntw, WS_NUM_CUS, (int)wide_tiles, (int)narrow_tiles);
    U2PL_LAUNCH_CHECK();
    return 0;
}

#ifdef U2PL_WS_ABLATE
// timing experiments only (a variant build: python -m u2pl_amd.build_ext --variant abl -DU2PL_WS_ABLATE): bit mask of main-loop
// ingredients to DROP -- 1 split arithmetic, 2 weight-piece stores, 4 activation-piece stores, 8 global loads, 16 operand
// reads, 32 the barrier.  Results are garbage; see tools/bench_ws_ablate.py.
static int g_ws_abl = 0;
U2PL_API int u2pl_igemm_ws_set_ablate(int v) { const int old = g_ws_abl; g_ws_abl = v; return old; }
#endif

// tile choice: 128 x 256 on 8 waves where Cout fills it, 128 x 128 (8 waves of 64 x 32) below; Cout <= 64 is not
// served here (the callers keep those layers -- 1 % of the network's multiplies -- on k_conv_igemm)
template <int NP>
static int run_igemm_ws_np(const float* x, long ldx, const float* x_amax, const void* ws, const float* bias, float* y, long ldy,
                           const ConvGeom& g, hipStream_t stream, float* stats, const float* pivot, int batch, long zx, long zy,
                           const BnEpi* epi) {
    if (g.Cin % BK || !ws) return U2PL_EINVAL;
    if (epi && (stats || batch != 1 || (g.Cout & 3))) return U2PL_EINVAL;
    // pointwise: 1x1, stride 1, no padding, forward or data-gradient geometry alike
    const bool pw = g.R == 1 && g.S == 1 && g.mul == 1 && g.off_h == 0 && g.off_w == 0 && g.log2div == 0 &&
                    g.Hin == g.Hout && g.Win == g.Wout;
    auto go = [&](bool narrow, long first, long count) {
        if (!narrow) return pw ? launch_igemm_ws<2, 2, 2, 4, true, 0, NP>(x, ldx, x_amax, ws, bias, y, ldy, g, stream, stats, pivot, batch, zx, zy, epi, first, count)
                               : launch_igemm_ws<2, 2, 2, 4, false, 0, NP>(x, ldx, x_amax, ws, bias, y, ldy, g, stream, stats, pivot, batch, zx, zy, epi, first, count);
        return pw ? launch_igemm_ws<2, 1, 2, 4, true, 0, NP>(x, ldx, x_amax, ws, bias, y, ldy, g, stream, stats, pivot, batch, zx, zy, epi, first, count)
                  : launch_igemm_ws<2, 1, 2, 4, false, 0, NP>(x, ldx, x_amax, ws, bias, y, ldy, g, stream, stats, pivot, batch, zx, zy, epi, first, count);
    };
#ifdef U2PL_WS_ABLATE
    if (g_ws_abl && g.Cout > 128 && pw) {
#define WS_ABL(A_) case A_: return launch_igemm_ws<2, 2, 2, 4, true, A_, NP>(x, ldx, x_amax, ws, bias, y, ldy, g, stream, stats, pivot, batch, zx, zy, epi)
        switch (g_ws_abl) { WS_ABL(1); WS_ABL(2); WS_ABL(4); WS_ABL(8); WS_ABL(16); WS_ABL(32); WS_ABL(7); WS_ABL(15); WS_ABL(31); WS_ABL(63); default: return U2PL_EINVAL; }
#undef WS_ABL
    }
#endif
    // 128 x 256 or 128 x 128 tiles?  One persistent block per CU: a launch costs (tiles of the busiest block) x (time per tile).
    // Narrow tiles halve the work per tile at ~15 % more time per product (the activation split is amortised over half the
    // matrix instructions) and fill the last round better: 295 row tiles x 256 columns take 2 rounds of wide tiles but only
    // 3 rounds of half-size ones (measured 146 -> 125 us for the 1024 -> 256 1x1 convolution at 4 x 97^2).  Third plan,
    // MIXED: the full rounds as wide tiles and the remainder R <= 128 wide tiles as 2 R narrow ones in a SECOND launch (a
    // persistent block has one tile shape): 792 tiles of a Winograd batch = 3 rounds + 24 tiles cost 3.6 rounds instead of
    // 4; a narrow tile (z, mt, 2 nt + h) is half h of the wide tile (z, mt, nt), so the remainder is a contiguous range of
    // narrow tile numbers.  Both groups run in ONE launch (k_igemm_ws_mix: the narrow blocks are handed to the CUs as the wide
    // blocks retire); as two launches the plan cost ~15 us more (drain, launch gap, cold prologue: 8 units) and only paid on
    // long-K layers.  Times in units of one 32-deep chunk of a wide tile; 4 / 2 = fixed cost per tile (prologue, epilogue); fitted to
    // tools/bench_igemm_ws.py: the mixed plan wins on the long-K launches (2048 -> 256 3x3 at 4 x 97^2: 2194 -> 1953 us,
    // 2048 -> 512: 410 -> 394 us) and is not chosen for K = 256 .. 1024, where it measured 0-7 % slower.
    // U2PL_WS_NARROW = 0 (wide only) | 1 (narrow, where Cout <= 256) | 2 (wide / narrow by the model, never mixed) | unset.
    if (g.Cout <= 128) return go(true, 0, -1);
    static int force = -2;
    if (force == -2) { const char* e = getenv("U2PL_WS_NARROW"); force = (e && *e) ? atoi(e) : -1; }
    const long M_ = (long)g.N * g.Hout * g.Wout;
    const long mt = cdiv(M_, 128), nkc = (long)(g.R * g.S * g.Cin) / BK;
    const long ntw = cdiv(g.Cout, 256), ntn = cdiv(g.Cout, 128);
    const long tw = mt * ntw * batch, tn = mt * ntn * batch;
    // (split-fp16: a chunk takes half the time, the fixed cost per tile -- prologue, stores -- does not)
    const double fix = NP == 3 ? 1.0 : ws_fix2();
    const double cw1 = nkc + 4.0 * fix, cn1 = 0.5 * 1.15 * nkc + 2.0 * fix;
    const double cw = (double)cdiv(tw, WS_NUM_CUS) * cw1, cn = (double)cdiv(tn, WS_NUM_CUS) * cn1;
    if (force == 0) return go(false, 0, -1);
    if (force == 1) return go(g.Cout <= 256, 0, -1);
    const long full = tw / WS_NUM_CUS * WS_NUM_CUS, rem = tw - full;
    const bool can_mix = force != 2 && ws_persist() && ntn == 2 * ntw && full > 0 && rem > 0 && 2 * rem <= WS_NUM_CUS;
    static int two_launch = -1;      // U2PL_WS_MIX2=1: the mixed plan as two launches (A/B of the one-launch form)
    if (two_launch < 0) { const char* e = getenv("U2PL_WS_MIX2"); two_launch = (e && *e) ? (atoi(e) != 0) : 0; }
    const double cm = can_mix ? (double)(full / WS_NUM_CUS) * cw1 + cn1 + (two_launch ? 8.0 : 2.0) * fix : 1e30;
    if (cm < 0.97 * cw && cm < cn) {
        if (!two_launch)
            return pw ? launch_igemm_ws_mix<true, NP>(x, ldx, x_amax, ws, bias, y, ldy, g, stream, stats, pivot, batch, zx, zy, epi, full, 2 * rem)
                      : launch_igemm_ws_mix<false, NP>(x, ldx, x_amax, ws, bias, y, ldy, g, stream, stats, pivot, batch, zx, zy, epi, full, 2 * rem);
        const int rc = go(false, 0, full);
        return rc ? rc : go(true, 2 * full, 2 * rem);
    }
    return go(cn < 0.97 * cw, 0, -1);
    // (Measured and dropped: the 128 x 256 tile on FOUR waves of 128 x 64 -- one wave per SIMD with 512 registers, 18 operand
    // reads per 48 matrix instructions instead of 12 per 24: bit-identical, 5-13 % slower on every wide-tile shape of
    // tools/bench_igemm_ws.py; the second wave of a SIMD does cover stalls of the first.)
}

// x_amax == NULL: the bf16 three-piece planes of u2pl_weight_split3_f32 (six products); x_amax != NULL: split-fp16 -- the planes
// of u2pl_weight_split2h_f32 and the device scalar max |x| of the activation operand (three products)
static int run_igemm_ws(const float* x, long ldx, const void* ws, const float* bias, float* y, long ldy, const ConvGeom& g,
                        hipStream_t stream, float* stats = nullptr, const float* pivot = nullptr, int batch = 1,
                        long zx = 0, long zy = 0, const BnEpi* epi = nullptr, const float* x_amax = nullptr) {
    return x_amax ? run_igemm_ws_np<2>(x, ldx, x_amax, ws, bias, y, ldy, g, stream, stats, pivot, batch, zx, zy, epi)
                  : run_igemm_ws_np<3>(x, ldx, nullptr, ws, bias, y, ldy, g, stream, stats, pivot, batch, zx, zy, epi);
}

// ---- entry points: the conv.hip calls with the weight operand given as split planes (u2pl_weight_split3_f32 of the
//      [Cout][R*S*Cin] matrix for the forward, of the transposed [Cin][R*S*Cout] matrix for the data gradient)
U2PL_API int u2pl_igemm_ws_stat_blocks(int N, int Hout, int Wout) { return cdiv((long)N * Hout * Wout, 128); }

U2PL_API int u2pl_conv2d_fwd_ws_f32(const float* x, long ldx, const void* wsplit, const float* bias, float* y, long ldy,
                                    int N, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int R, int S,
                                    int stride, int pad, int dil, hipStream_t stream) {
    ConvGeom g = {N, Hin, Win, Cin, Hout, Wout, Cout, R, S, stride, -pad, -pad, dil, 0};
    return run_igemm_ws(x, ldx, wsplit, bias, y, ldy, g, stream);
}
U2PL_API int u2pl_conv2d_fwd_bnstats_ws_f32(const float* x, long ldx, const void* wsplit, const float* bias, float* y,
                                            long ldy, int N, int Hin, int Win, int Cin, int Hout, int Wout, int Cout,
                                            int R, int S, int stride, int pad, int dil, const float* pivot,
                                            float* stats_partial, hipStream_t stream) {
    ConvGeom g = {N, Hin, Win, Cin, Hout, Wout, Cout, R, S, stride, -pad, -pad, dil, 0};
    return run_igemm_ws(x, ldx, wsplit, bias, y, ldy, g, stream, stats_partial, pivot);
}
U2PL_API int u2pl_conv2d_fwd_bnact_ws_f32(const float* x, long ldx, const void* wsplit, const float* bias, float* y,
                                          long ldy, int N, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int R,
                                          int S, int stride, int pad, int dil, const float* mean, const float* invstd,
                                          const float* gamma, const float* beta, const float* res, long ldr, int relu,
                                          hipStream_t stream) {
    if (!mean || !invstd || !gamma || !beta || (Cout & 3) || (res && (ldr & 3))) return U2PL_EINVAL;
    ConvGeom g = {N, Hin, Win, Cin, Hout, Wout, Cout, R, S, stride, -pad, -pad, dil, 0};
    const BnEpi epi = {mean, invstd, gamma, beta, res, ldr, relu, nullptr};
    return run_igemm_ws(x, ldx, wsplit, bias, y, ldy, g, stream, nullptr, nullptr, 1, 0, 0, &epi);
}
U2PL_API int u2pl_conv2d_dgrad_ws_f32(const float* dy, long lddy, const void* wTsplit, float* dx, long lddx, int N, int Hin,
                                      int Win, int Cin, int Hout, int Wout, int Cout, int R, int S, int stride, int pad,
                                      int dil, hipStream_t stream) {
    int l2 = 0;
    while ((1 << l2) < stride) ++l2;
    if ((1 << l2) != stride) return U2PL_EINVAL;
    ConvGeom g = {N, Hout, Wout, Cout, Hin, Win, Cin, R, S, 1, pad, pad, -dil, l2};
    return run_igemm_ws(dy, lddy, wTsplit, nullptr, dx, lddx, g, stream);
}
// batch of row-major GEMMs Y_z[M][Nn] = X_z[M][K] * W_z[Nn][K]^T with W_z given as split planes (batch = the matrices
// of ONE u2pl_weight_split3_f32 call): the Winograd component products
U2PL_API int u2pl_gemm_batched_ws_f32(const float* x, long ldx, long zx, const void* wsplit, float* y, long ldy, long zy,
                                      long M, int K, int Nn, int batch, hipStream_t stream) {
    if (M <= 0 || batch <= 0) return 0;
    if (M >= (1L << 31)) return U2PL_EINVAL;
    ConvGeom g = {1, (int)M, 1, K, (int)M, 1, Nn, 1, 1, 1, 0, 0, 1, 0};
    return run_igemm_ws(x, ldx, wsplit, nullptr, y, ldy, g, stream, nullptr, nullptr, batch, zx, zy);
}

// ---- split-fp16 entry points (round 6): the same calls with the planes of u2pl_weight_split2h_f32 and x_amax = the device scalar
//      max |x| of the activation operand (u2pl_absmax_f32, or a producer's fused maximum; any upper bound within 2^8 of the true
//      maximum keeps fp32-class accuracy, a bound BELOW the maximum overflows fp16)
U2PL_API int u2pl_conv2d_fwd_wsh_f32(const float* x, long ldx, const float* x_amax, const void* wsplit, const float* bias, float* y, long ldy,
                                     int N, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int R, int S,
                                     int stride, int pad, int dil, hipStream_t stream) {
    if (!x_amax) return U2PL_EINVAL;
    ConvGeom g = {N, Hin, Win, Cin, Hout, Wout, Cout, R, S, stride, -pad, -pad, dil, 0};
    return run_igemm_ws(x, ldx, wsplit, bias, y, ldy, g, stream, nullptr, nullptr, 1, 0, 0, nullptr, x_amax);
}
U2PL_API int u2pl_conv2d_fwd_bnstats_wsh_f32(const float* x, long ldx, const float* x_amax, const void* wsplit, const float* bias, float* y,
                                             long ldy, int N, int Hin, int Win, int Cin, int Hout, int Wout, int Cout,
                                             int R, int S, int stride, int pad, int dil, const float* pivot,
                                             float* stats_partial, hipStream_t stream) {
    if (!x_amax) return U2PL_EINVAL;
    ConvGeom g = {N, Hin, Win, Cin, Hout, Wout, Cout, R, S, stride, -pad, -pad, dil, 0};
    return run_igemm_ws(x, ldx, wsplit, bias, y, ldy, g, stream, stats_partial, pivot, 1, 0, 0, nullptr, x_amax);
}
U2PL_API int u2pl_conv2d_fwd_bnact_wsh_f32(const float* x, long ldx, const float* x_amax, const void* wsplit, const float* bias, float* y,
                                           long ldy, int N, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int R,
                                           int S, int stride, int pad, int dil, const float* mean, const float* invstd,
                                           const float* gamma, const float* beta, const float* res, long ldr, int relu,
                                           float* y_amax, hipStream_t stream) {
    if (!x_amax || !mean || !invstd || !gamma || !beta || (Cout & 3) || (res && (ldr & 3))) return U2PL_EINVAL;
    ConvGeom g = {N, Hin, Win, Cin, Hout, Wout, Cout, R, S, stride, -pad, -pad, dil, 0};
    const BnEpi epi = {mean, invstd, gamma, beta, res, ldr, relu, (unsigned*)y_amax};
    return run_igemm_ws(x, ldx, wsplit, bias, y, ldy, g, stream, nullptr, nullptr, 1, 0, 0, &epi, x_amax);
}
U2PL_API int u2pl_conv2d_dgrad_wsh_f32(const float* dy, long lddy, const float* dy_amax, const void* wTsplit, float* dx, long lddx, int N,
                                       int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int R, int S, int stride, int pad,
                                       int dil, hipStream_t stream) {
    if (!dy_amax) return U2PL_EINVAL;
    int l2 = 0;
    while ((1 << l2) < stride) ++l2;
    if ((1 << l2) != stride) return U2PL_EINVAL;
    ConvGeom g = {N, Hout, Wout, Cout, Hin, Win, Cin, R, S, 1, pad, pad, -dil, l2};
    return run_igemm_ws(dy, lddy, wTsplit, nullptr, dx, lddx, g, stream, nullptr, nullptr, 1, 0, 0, nullptr, dy_amax);
}
U2PL_API int u2pl_gemm_batched_wsh_f32(const float* x, long ldx, long zx, const float* x_amax, const void* wsplit, float* y, long ldy,
                                       long zy, long M, int K, int Nn, int batch, hipStream_t stream) {
    if (!x_amax) return U2PL_EINVAL;
    if (M <= 0 || batch <= 0) return 0;
    if (M >= (1L << 31)) return U2PL_EINVAL;
    ConvGeom g = {1, (int)M, 1, K, (int)M, 1, Nn, 1, 1, 1, 0, 0, 1, 0};
    return run_igemm_ws(x, ldx, wsplit, nullptr, y, ldy, g, stream, nullptr, nullptr, batch, zx, zy, nullptr, x_amax);
}
