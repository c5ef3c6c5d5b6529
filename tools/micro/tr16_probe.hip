// What does ds_read_b64_tr_b16 return?  LDS holds element id = row * 64 + col of a [64][64] u16 image; lane l hands in the
// address of (row = (l % 16) / 4 [+ 4 * (l / 32) * 2 ...], col = 16 * ((l / 16) % 2) + 4 * (l % 4)) -- the guess of
// cdna_hip_programming.md T10 -- and every lane prints the four ids it received.   hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out, int mode) {
    __shared__ unsigned short lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x, j = l & 15, blk = (l >> 4) & 1, hi = l >> 5;
    int row, col;
    if (mode == 0) { row = j / 4 + 8 * hi; col = blk * 16 + 4 * (j % 4); }        // [4 rows][16 cols] per 16-lane group
    else { row = j % 4 + 8 * hi; col = blk * 16 + 4 * (j / 4); }                   // alternative lane order
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + row * 64 + col));
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = (unsigned short)r[e];
}
int main() {
    unsigned short* d;
    hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (lane: (row,col) x4)\n", mode);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d:", l);
            for (int e = 0; e < 4; ++e) printf(" (%d,%d)", h[l * 4 + e] / 64, h[l * 4 + e] % 64);
            printf("\n");
        }
    }
    return 0;
}
