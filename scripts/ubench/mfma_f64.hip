// v_mfma_f64_16x16x4_f64 operand / result layout check: A[m][k] = 1 + m + 100 k, B[k][n] = 1 + n + 1000 k with the assumed layouts
// (A: lane = m + 16 k; B: lane = n + 16 k; D register j of lane l <-> row 4 (l / 16) + j, column l % 16); prints the mismatches.
// Result on MI355X: A and B as assumed; D register j of lane l <-> row 4 j + l / 16, column l % 16 (what conv2_dgrad_h_kernel uses).
// build: hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/build/mfma_f64 scripts/ubench/mfma_f64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4v __attribute__((ext_vector_type(4)));
__global__ void k(double* out) {
    const int l = threadIdx.x;
    const double a = 1 + (l & 15) + 100 * (l >> 4), b = 1 + (l & 15) + 1000 * (l >> 4);
    double4v c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = c[j];
}
int main() {
    double* d; (void)hipMalloc(&d, 256 * 8); k<<<1, 64>>>(d); double h[256]; (void)hipMemcpy(h, d, 256 * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
        const int m = 4 * (l >> 4) + j, n = l & 15; double want = 0;
        for (int kk = 0; kk < 4; ++kk) want += (1.0 + m + 100 * kk) * (1.0 + n + 1000 * kk);
        if (h[l * 4 + j] != want) { if (bad < 8) printf("lane %d reg %d: got %.0f want %.0f\n", l, j, h[l * 4 + j], want); ++bad; }
    }
    printf("mismatches: %d\n", bad);
}
