// ds_read_b64_tr_b16 (gfx950 transposed LDS read): prints what every lane receives from a [row][64] f16 image holding row*64+col.
// build: hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/build/tr_read scripts/ubench/tr_read.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __fp16 half4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef short short4v __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
    __shared__ __attribute__((aligned(16))) _Float16 t[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) t[i] = (_Float16)(float)(i % 2048);
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const _Float16* addr = t + (4 * g + q) * 64 + 4 * p;
    half4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) half4*)addr);
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = (float)v[i];
}
int main() { float* d; hipMalloc(&d, 1024); k<<<1, 64>>>(d); float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
 for (int l = 0; l < 64; ++l) printf("lane %2d: %g %g %g %g\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]); }
