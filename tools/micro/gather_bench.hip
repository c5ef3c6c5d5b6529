// micro-benchmark: how fast can 1 KiB rows be gathered from a 604 MB bank on MI355X, random vs sorted order?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
template <int INF>
__global__ void k_gather(const float4* __restrict__ bank, const int* __restrict__ idx, int rows_per_wave, float* out) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int* my = idx + (long)wave * rows_per_wave;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int j = 0; j < rows_per_wave; j += INF) {
        float4 v[INF];
#pragma unroll
        for (int u = 0; u < INF; ++u) v[u] = bank[(long)my[j + u] * 64 + lane];
#pragma unroll
        for (int u = 0; u < INF; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    if (acc.x == 123.456f) out[wave] = acc.x + acc.y + acc.z + acc.w;
}
__global__ void k_read(const float4* __restrict__ p, long n, float* out) {
    float4 acc = make_float4(0, 0, 0, 0);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float4 v = p[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x == 123.456f) out[0] = acc.x + acc.y + acc.z + acc.w;
}
int main() {
    const long nrows = 590000;   // 604 MB of 1 KiB rows
    const int waves = 4864, rpw = 48;   // 19 jobs x 256 anchors, ~51 rows each
    float4* bank; int* idx; float* out; char* flush;
    (void)hipMalloc(&bank, nrows * 1024); (void)hipMalloc(&idx, (long)waves * rpw * 4); (void)hipMalloc(&out, waves * 4);
    (void)hipMalloc(&flush, 1L << 30);
    (void)hipMemset(bank, 0, nrows * 1024);
    (void)hipMemset(flush, 1, 1L << 30);
    std::vector<int> h((long)waves * rpw);
    srand(1);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int mode = 0; mode < 3; ++mode) {
        for (int inf = 4; inf <= 16; inf *= 2) {
            double tot = 0;
            const int reps = 5;
            for (int rep = 0; rep < reps; ++rep) {
                // fresh draws every launch; mode 0: random rows inside the wave's job bank (30k rows); 1: the same kind of draws
                // SORTED over the job and dealt to consecutive waves (row-major sweep); 2: fully sequential rows
                for (int job = 0; job < 19; ++job) {
                    std::vector<int> d(256 * rpw);
                    for (auto& x : d) x = job * 30000 + rand() % 30000;
                    if (mode == 1) std::sort(d.begin(), d.end());
                    if (mode == 2) for (size_t i = 0; i < d.size(); ++i) d[i] = job * 30000 + (int)((i + 977 * rep) % 30000);
                    for (size_t i = 0; i < d.size(); ++i) h[(long)job * 256 * rpw + i] = d[i];
                }
                (void)hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
                // evict the 256 MB Infinity Cache with CLEAN lines (a memset would leave 256 MB of dirty lines whose
                // write-back competes with the measured reads: that variant read 3.0 TB/s in every mode)
                hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, (const float4*)flush, (1L << 30) / 16, out);
                (void)hipDeviceSynchronize();
                (void)hipEventRecord(a);
                if (inf == 4) hipLaunchKernelGGL(k_gather<4>, dim3(waves / 4), dim3(256), 0, 0, bank, idx, rpw, out);
                if (inf == 8) hipLaunchKernelGGL(k_gather<8>, dim3(waves / 4), dim3(256), 0, 0, bank, idx, rpw, out);
                if (inf == 16) hipLaunchKernelGGL(k_gather<16>, dim3(waves / 4), dim3(256), 0, 0, bank, idx, rpw, out);
                (void)hipEventRecord(b); (void)hipEventSynchronize(b);
                float ms; (void)hipEventElapsedTime(&ms, a, b);
                if (rep) tot += ms;
            }
            const double us = tot / (reps - 1) * 1e3;
            printf("mode %d (0 random, 1 sorted, 2 sequential) in-flight %2d: %.1f us per launch, %.2f TB/s (cold cache)\n", mode, inf, us,
                   (double)waves * rpw * 1024 / (us * 1e-6) / 1e12);
        }
    }
    return 0;
}
