// How does v_mfma_f32_16x16x32_f16 round its accumulation?  D = C + sum_k A[m][k] B[k][n] with C = +-2^24 (ulp 2) and dot products
// 0.5 .. 3.5 built from one or several terms: an IEEE chain rounds to nearest even, a truncating adder does not.
// build: hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/build/mfma_round scripts/ubench/mfma_round.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, float c0, float term, int nterms) {
    const int l = threadIdx.x;
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = 0; b[j] = 0; }
    // k index = 8 (l / 16) + j: the first nterms k of row / column get (term, 1)
    for (int j = 0; j < 8; ++j) { const int kk = 8 * (l >> 4) + j; if (kk < nterms) { a[j] = (_Float16)term; b[j] = (_Float16)1.0f; } }
    f32x4 c = {c0, c0, c0, c0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (l == 0) out[0] = c[0];
}
int main() {
    float* d; (void)hipMalloc(&d, 4);
    const float cs[2] = {16777216.0f, -16777216.0f};
    for (float c0 : cs) for (float sgn : {1.0f, -1.0f}) for (int nt : {1, 2, 3, 5, 7}) {
        const float term = 0.5f * sgn;
        k<<<1, 64>>>(d, c0, term, nt); float h; (void)hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        const double exact = (double)c0 + nt * (double)term;
        printf("C = %.0f + %d x %.1f: exact %.1f  mfma %.0f  (RNE of the exact sum %.0f)\n", c0, nt, term, exact, h, (double)(float)exact);
    }
}
