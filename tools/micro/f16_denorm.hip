// Does v_mfma_f32_32x32x16_f16 honour fp16 subnormal operands on gfx950, and does v_cvt_pk_f16_f32 round to nearest even and
// produce subnormals?  (round 6: the split-fp16 GEMM relies on both.)   hipcc --offload-arch=gfx950 -O3 f16_denorm.hip && ./a.out
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void m(float av, float bv, float* c) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)av; b[i] = (_Float16)bv; }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) c[0] = acc[0];
}
__global__ void cv(const float* x, unsigned* o, int n) {
    int i = threadIdx.x;
    if (i < n) { h2 v; v[0] = (_Float16)x[i]; v[1] = (_Float16)0.f; o[i] = *(unsigned*)&v & 0xffff; }
}
int main() {
    float* c; hipMalloc(&c, 4);
    float tests[][2] = {{1.f, 1.f}, {ldexpf(1.f, -20), 1.f}, {ldexpf(1.f, -24), 1.f}, {ldexpf(1.f, -20), ldexpf(1.f, -4)}, {ldexpf(1.f,-14), 1.f}, {ldexpf(3.f,-24), 1024.f}};
    for (auto& t : tests) {
        hipLaunchKernelGGL(m, 1, 64, 0, 0, t[0], t[1], c);
        float h; hipMemcpy(&h, c, 4, hipMemcpyDeviceToHost);
        printf("mfma a=%g b=%g -> %g (expect %g)\n", t[0], t[1], h, 16.0 * t[0] * t[1]);
    }
    float xs[] = {1.0f + ldexpf(1.f, -11), 1.0f + 3 * ldexpf(1.f, -11), ldexpf(1.f, -20), ldexpf(1.f, -25), ldexpf(1.5f, -25), 65520.f, 65519.f, 1e-9f};
    float* dx; unsigned* dout; hipMalloc(&dx, sizeof xs); hipMalloc(&dout, 64);
    hipMemcpy(dx, xs, sizeof xs, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(cv, 1, 64, 0, 0, dx, dout, 8);
    unsigned ho[8]; hipMemcpy(ho, dout, 32, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i) printf("cvt %.10g -> 0x%04x\n", xs[i], ho[i]);
    return 0;
}
