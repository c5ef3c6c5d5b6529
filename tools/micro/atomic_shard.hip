// How fast do N waves publish one atomicMax each, by shard count and spacing?  (round 6: fused operand maxima.)
//   hipcc --offload-arch=gfx950 -O3 atomic_shard.hip -o atomic_shard && ./atomic_shard
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* slots, int shards, int stride_words, int work, const float* src, float* dst, int preread) {
    // some streaming work first so that waves retire the way an HBM-bound kernel's do
    float acc = 0.f;
    const long base = ((long)blockIdx.x * blockDim.x + threadIdx.x);
    for (int i = 0; i < work; ++i) acc += src[(base + (long)i * gridDim.x * blockDim.x) & ((1 << 24) - 1)];
    if (acc == 123.456f) dst[0] = acc;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) {
        unsigned* s = slots + (wave % shards) * stride_words;
        const unsigned m = wave * 2654435761u >> 4;      // pseudo-random magnitudes: ~log(n) new maxima
        if (!preread || m > __hip_atomic_load(s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(s, m);
    }
}
int main() {
    unsigned* slots; float *src, *dst;
    hipMalloc(&slots, 1 << 20); hipMalloc(&src, 4 << 24); hipMalloc(&dst, 64);
    hipMemset(src, 0, 4 << 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 4096;
    for (int work : {0, 8}) for (int preread : {0, 1}) for (int shards : {1, 8, 16, 64}) for (int stride : {1, 16, 32}) {
        if (shards == 1 && stride != 1) continue;
        float best = 1e9;
        for (int r = 0; r < 5; ++r) {
            hipMemset(slots, 0, 1 << 20);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, blocks, 256, 0, 0, slots, shards, stride, work, src, dst, preread);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("work %d preread %d shards %2d stride %2d words: %.1f us\n", work, preread, shards, stride, best * 1e3);
    }
    return 0;
}
