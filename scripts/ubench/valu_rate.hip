// Microbenchmark: sustained VALU issue rate per SIMD for plain and packed f32 ops at 1..4 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

using f2 = __attribute__((ext_vector_type(2))) float;

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a[8];
    f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = f2{a[i], a[i] + 1.f}; }
    const float b = 1.0001f, c = 0.0001f;
    const f2 pb = {1.0001f, 0.9999f}, pc = {0.0001f, 0.0002f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
                if (OP == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
                if (OP == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
                if (OP == 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 6) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 7) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(c));
            }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 123.456f) out[0] = s;
}

template <int OP>
void run(const char* name, float* d) {
    const int iters = 4000;
    for (int w = 1; w <= 4; ++w) {
        dim3 grid(256 * w), block(256);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        k<OP><<<grid, block>>>(d, 10);
        hipDeviceSynchronize();
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            k<OP><<<grid, block>>>(d, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double instr_per_simd = double(iters) * 32 * w;     // wave-instructions per SIMD
        const double cyc = best * 1e-3 * 2.4e9;
        printf("%-14s waves/SIMD %d: %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, w, best, cyc / instr_per_simd);
    }
}

int main() {
    float* d; hipMalloc(&d, 4);
    run<0>("v_fma_f32", d);
    run<1>("v_add_f32", d);
    run<5>("v_mul_f32", d);
    run<2>("v_pk_fma_f32", d);
    run<3>("v_pk_add_f32", d);
    run<4>("v_pk_mul_f32", d);
    run<6>("v_xor_b32", d);
    run<7>("v_mov_b32", d);
    return 0;
}
