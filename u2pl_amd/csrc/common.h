// Shared device helpers for libu2pl_hip.so (gfx950 / CDNA4 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define U2PL_API extern "C" __attribute__((visibility("default")))

#define U2PL_LAUNCH_CHECK()                      \
    do {                                         \
        hipError_t e__ = hipGetLastError();      \
        if (e__ != hipSuccess) return (int)e__;  \
    } while (0)

#define U2PL_EINVAL 1001  // bad argument (reported before any launch)

// every kernel launch of the library goes through this macro: u2pl_kernel_launches() lets the bench report kernel
// launches per step next to the C-ABI calls per step (an entry point may issue several launches: tile planner bodies
// and tails, split-K reduces, Winograd component batches)
// (entry points are called concurrently from the main thread and autograd's backward thread -- ctypes releases the GIL --: the
// counter is a relaxed atomic)
#include <atomic>
extern std::atomic<unsigned long long> u2pl_kernel_launch_count;
#define U2PL_LAUNCH(...)                        \
    do {                                        \
        u2pl_kernel_launch_count.fetch_add(1, std::memory_order_relaxed);             \
        hipLaunchKernelGGL(__VA_ARGS__);        \
    } while (0)

// HBM-bound elementwise / transform kernels (BatchNorm apply + backward, column reductions, Winograd transforms) are capped
// at U2PL_HBM_MAXW waves per SIMD when that macro is defined at build time: the cap leaves wave slots (and all of the LDS)
// on every CU for an MFMA-bound kernel of another stream to run beside them (experiment: see DESIGN section 3).
#ifdef U2PL_HBM_MAXW
#define U2PL_HBM_KERNEL __attribute__((amdgpu_waves_per_eu(1, U2PL_HBM_MAXW)))
#else
#define U2PL_HBM_KERNEL
#endif

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline int grid_for(long n, int block, int max_blocks = 256 * 16) {
    long g = (n + block - 1) / block;
    if (g > max_blocks) g = max_blocks;
    if (g < 1) g = 1;
    return (int)g;
}

// ---- wave64 reductions ------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// All-lanes sum without the LDS crossbar: four DPP row rotations give every lane the total of its 16-lane row (VALU rate;
// __shfl_xor compiles to ds_bpermute_b32, an LDS-pipeline instruction), the four row totals are read as wave-uniform
// scalars.  The InfoNCE kernel does ~100 such reductions per anchor and was bound by them, not by HBM (a pure gather of
// the same rows runs at 6 TB/s: tools/micro/gather_bench.hip).
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_allsum_dpp(float v) {
    v = dpp_add<0x128>(v);   // row_ror:8
    v = dpp_add<0x124>(v);   // row_ror:4
    v = dpp_add<0x122>(v);   // row_ror:2
    v = dpp_add<0x121>(v);   // row_ror:1
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}
// wave total as a wave-uniform scalar: row rotations, then the gfx9 wave-level DPP broadcasts (row_bcast:15 adds the
// previous row's total into rows 1 and 3, row_bcast:31 adds lane 31's into rows 2 and 3): lane 63 holds the total.
// 6 DPP adds + 1 readlane.
__device__ __forceinline__ float wave_sum_sgpr(float v) {
    v = dpp_add<0x128>(v);
    v = dpp_add<0x124>(v);
    v = dpp_add<0x122>(v);
    v = dpp_add<0x121>(v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));   // row_bcast:15
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false));   // row_bcast:31
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float lane_put(float old, float val, int lane) {   // val is wave-uniform: one v_cndmask
    return (int)(threadIdx.x & 63) == lane ? val : old;
}
__device__ __forceinline__ float lane_get(float v, int lane) {                // v_readlane_b32 -> wave-uniform
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ unsigned wave_sum_u(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- fused operand maxima (split-fp16 GEMMs, conv_geom.h) ---------------------------------------------------------------
// A producer of a GEMM operand leaves max |x| of what it wrote in an "amax object" the caller zeroed: U2PL_AMAX_WORDS uint32 =
// 64 shards, one per 128-byte line; the tensor's maximum is the maximum over the shards.  Per thread a running integer maximum of
// the bit patterns of |x| (order-preserving for non-negative floats; a NaN's pattern is above every finite one, so a NaN anywhere
// makes the maximum NaN), per wave ONE fire-and-forget atomicMax on the shard (global wave index) % 64.  Why shards: same-LINE
// atomics serialise at ~88 / us (tools/micro/atomic_shard.hip: 16384 waves on one word 189 us, on 64 lines 6.6 us; a relaxed
// pre-read of the slot to skip redundant atomics costs more than it saves: the reads queue on the same line).
// Every lane of the wave must call amax_wave_publish; a consumer reads the object with amax_read (one load per lane).
#define U2PL_AMAX_SHARDS 64
#define U2PL_AMAX_STRIDE 32                                   // words between shards
#define U2PL_AMAX_WORDS (U2PL_AMAX_SHARDS * U2PL_AMAX_STRIDE)
__device__ __forceinline__ unsigned amax_bits(unsigned m, float v) {
    const unsigned b = __float_as_uint(v) & 0x7fffffffu;
    return b > m ? b : m;
}
__device__ __forceinline__ unsigned amax_bits4(unsigned m, float4 v) {
    return amax_bits(amax_bits(amax_bits(amax_bits(m, v.x), v.y), v.z), v.w);
}
__device__ __forceinline__ unsigned wave_max_u(unsigned m) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned t = (unsigned)__shfl_xor((int)m, o, 64);
        m = t > m ? t : m;
    }
    return m;
}
__device__ __forceinline__ void amax_wave_publish(unsigned m, unsigned* __restrict__ obj) {
    m = wave_max_u(m);
    if ((threadIdx.x & 63) == 0 && m != 0u) {
        const unsigned wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        atomicMax(obj + (wave & (U2PL_AMAX_SHARDS - 1)) * U2PL_AMAX_STRIDE, m);
    }
}
// the object's maximum as a wave-uniform bit pattern (all 64 lanes of the calling wave participate)
__device__ __forceinline__ unsigned amax_read(const unsigned* __restrict__ obj) {
    const unsigned v = *(const __attribute__((address_space(1))) unsigned*)(obj + (threadIdx.x & 63) * U2PL_AMAX_STRIDE);
    return (unsigned)__builtin_amdgcn_readfirstlane((int)wave_max_u(v));
}

// Same-address device atomics serialise at ~88/us (MI355X_MICROARCH "dequeue"): counters are reduced
// per BLOCK (LDS) and flushed with ONE global atomic per block; kernels that use this cap their grid at
// a few hundred blocks.
__device__ __forceinline__ void block_count_flush(unsigned cnt, unsigned* __restrict__ dst) {
    __shared__ unsigned s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    cnt = wave_sum_u(cnt);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&s_cnt, cnt);
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) atomicAdd(dst, s_cnt);
}

// order-preserving float -> uint key (NaN (positive quiet) sorts above +inf)
__device__ __forceinline__ unsigned f32_key(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_f32(unsigned k) {
    unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// align_corners=True source coordinate pieces (torch CPU arithmetic, Q8)
struct AcCoord {
    int i0, i1;
    float l0, l1;
};
__device__ __forceinline__ AcCoord ac_coord(int dst, float scale, int in_size) {
    AcCoord c;
    float src = __fmul_rn((float)dst, scale);
    c.i0 = (int)src;  // src >= 0: trunc == floor
    if (c.i0 > in_size - 1) c.i0 = in_size - 1;
    c.i1 = c.i0 + (c.i0 < in_size - 1 ? 1 : 0);
    c.l1 = __fsub_rn(src, (float)c.i0);
    c.l0 = __fsub_rn(1.0f, c.l1);
    return c;
}
static inline float ac_scale_host(long in, long out) {
    return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.0f;
}
// legacy nearest: src = min(floor(dst * float32(in/out)), in-1)
__device__ __forceinline__ int nearest_src(int dst, float scale, int in_size) {
    int s = (int)floorf(__fmul_rn((float)dst, scale));
    return s < in_size - 1 ? s : in_size - 1;
}

// ---- raw buffer loads (hardware out-of-range -> 0) --------------------------
// Masked gathers without branches or selects on data: every global read is a raw buffer load
// (buffer_load_dwordx4 ... offen) through a descriptor whose num_records is the byte size of the
// tensor; an out-of-image tap / out-of-range row gets the offset OOB_OFF, for which the hardware
// returns zeros.  All loads are unconditional, so they stay in flight under the MFMAs.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define OOB_OFF ((int)0x80000000u)   // >= num_records for every tensor (host checks bytes < 2^31)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, (short)0, (int)bytes, 0x00020000);
}
// per-lane byte offset + wave-uniform (SGPR) byte offset: a stream of rows needs no per-row address VGPRs at all
__device__ __forceinline__ float4 buf_load4s(__amdgpu_buffer_rsrc_t r, int lane_off, int uniform_off) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, uniform_off, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, int byte_off) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// ---- loads through pointers that were themselves loaded from memory -----------------------------------------------
// A pointer read out of a descriptor struct has no known address space: the compiler emits FLAT loads, which may return
// out of order with respect to other memory operations, so every wait on them is s_waitcnt vmcnt(0) lgkmcnt(0) -- a
// software pipeline built on such loads does not overlap anything.  These helpers state "global memory" explicitly
// (global_load_*, counted in order by vmcnt alone).
#define U2PL_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ T ldg(const T* p) { return *(const U2PL_GLOBAL T*)p; }

// ---- fused eval-mode BatchNorm epilogue --------------------------------------
// Eval-mode BatchNorm (+ residual, ReLU) applied in the GEMM epilogue instead of a separate pass over the conv output
// (u2pl_conv2d_fwd_bnact_f32): y = [relu]((v - mean) * invstd * gamma + beta [+ res]), the arithmetic of k_bn_apply
// (csrc/nn.hip) operation for operation, so the fused and the two-kernel forms give identical bits.  mean == NULL: off.
struct BnEpi {
    const float *mean, *invstd, *gamma, *beta, *res;
    long ldr;
    int relu;
    unsigned* y_amax;       // split-fp16: amax object that receives max |y| of the fused output (NULL: not wanted)
};

