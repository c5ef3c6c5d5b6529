// Issue rate of scalar vs packed fp32 VALU operations on gfx950 (does v_pk_fma_f32 double the FMA rate of v_fma_f32?).
//   hipcc --offload-arch=gfx950 -O3 -o pk_bench pk_bench.hip && ./pk_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters) {
    f2 x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = f2{(float)threadIdx.x + i, (float)i};
    const f2 A = {a, a}, B = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) { x[i].x = __builtin_fmaf(x[i].x, a, b); x[i].y = __builtin_fmaf(x[i].y, a, b); }   // 2 v_fma_f32
            else if (MODE == 1) x[i] = __builtin_elementwise_fma(x[i], A, B);                                    // 1 v_pk_fma_f32
            else if (MODE == 2) { x[i].x = x[i].x * a; x[i].y = x[i].y * a; }                                    // 2 v_mul_f32
            else x[i] = x[i] * A;                                                                                // 1 v_pk_mul_f32
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
static void run(const char* name, float* out) {
    const int iters = 4096, blocks = 256 * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 1.0000001f, 1e-9f, iters);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, 1.0000001f, 1e-9f, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double elem_ops = (double)blocks * 256 * iters * 16;     // element operations (an FMA counts once)
    printf("%-14s %8.3f ms  %7.2f T element-ops/s\n", name, ms, elem_ops / ms / 1e9);
}
int main() {
    float* out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    run<0>("v_fma_f32", out);
    run<1>("v_pk_fma_f32", out);
    run<2>("v_mul_f32", out);
    run<3>("v_pk_mul_f32", out);
    hipDeviceSynchronize();
    return 0;
}
