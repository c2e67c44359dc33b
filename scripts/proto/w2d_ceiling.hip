// TIMING-ONLY prototype (results are garbage; nothing here ships): the consumer side of conv2 as a TWO-dimensional Winograd F(2x2, 3x3)
// in split precision, to price the tiling DESIGN.md section 4 names as K2's next step before anyone builds its producers.
//   per tile row (2 output rows x 32 columns = 16 tiles) and N-tile of 16 channels: 16 (xi, nu) planes x 3 v_mfma_f32_16x16x32_f16
//   = 48 MFMAs (the shipped 1-D form: 72), 128 B-operand + 64 accumulator VGPRs -> 256-register waves, two per SIMD at most.
// A fragments come from a static LDS image laid out like the shipped kernel's (160-byte records, ds_read_b128), with the real output
// transform + bias + ReLU + pool per tile row; no producers, no synchronisation: an UPPER bound for the consumers' rate.
//   build: hipcc -O3 --offload-arch=gfx950 scripts/proto/w2d_ceiling.hip -o scripts/proto/build/w2d_ceiling ; run: ./w2d_ceiling [waves=4|8]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRec = 160, kPlane = 16 * kRec, kBuf = 16 * kPlane;        // 40,960 B per tile-row buffer
constexpr int kTileRows = 40, kClipsPerWg = 16;

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void w2d_kernel(const half8* __restrict__ wts, const float* __restrict__ bias_dsc,
                                                              float* __restrict__ out, int buffers) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = wave & 3;
    for (int i = tid; i < buffers * kBuf / 4; i += WAVES * 64) reinterpret_cast<uint32_t*>(lds)[i] = 0x3c003800u + uint32_t(i & 255);
    __syncthreads();
    half8 bh[16], bl[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) { bh[p] = wts[((nt * 16 + p) * 2 + 0) * 64 + lane]; bl[p] = wts[((nt * 16 + p) * 2 + 1) * 64 + lane]; }
    const float bias = bias_dsc[16 * nt + (lane & 15)], dsc = bias_dsc[64 + 16 * nt + (lane & 15)];
    const int frag_off = (lane & 15) * kRec + (lane >> 4) * 16;
    const int rows_per_wave = WAVES == 4 ? kTileRows : kTileRows / 2;      // 8 waves: two groups share a clip's tile rows
    for (int k = 0; k < kClipsPerWg; ++k) {
        float pool = 0.f;
        for (int t = 0; t < rows_per_wave; ++t) {
            const char* buf = lds + ((k * rows_per_wave + t + (wave >> 2)) % 3) * kBuf + frag_off;
            f32x4 acc[16];
            auto frag = [&](int p, int lo) { return *reinterpret_cast<const half8*>(buf + p * kPlane + 64 * lo); };
            constexpr int PF = 3, RING = PF + 1;                 // the shipped kernel's discipline: a ring of fragments read PF steps ahead
            half8 fh[RING], fl[RING];
#pragma unroll
            for (int i = 0; i < PF; ++i) { fh[i] = frag(i, 0); fl[i] = frag(i, 1); }
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                if (p + PF < 16) { fh[(p + PF) % RING] = frag(p + PF, 0); fl[(p + PF) % RING] = frag(p + PF, 1); }
                __builtin_amdgcn_sched_barrier(0);
                const half8 ah = fh[p % RING], al = fl[p % RING];
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[p], z, 0, 0, 0);
                acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[p], acc[p], 0, 0, 0);
                acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[p], acc[p], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // Y = A^T M A, A^T = [[1,1,1,0],[0,1,-1,-1]]; M[xi][nu] = acc[4 xi + nu]
            float pv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float r0[4], r1[4];                                          // A^T M: two rows of four
#pragma unroll
                for (int nu = 0; nu < 4; ++nu) {
                    r0[nu] = acc[nu][j] + acc[4 + nu][j] + acc[8 + nu][j];
                    r1[nu] = acc[4 + nu][j] - acc[8 + nu][j] - acc[12 + nu][j];
                }
                const float y00 = r0[0] + r0[1] + r0[2], y01 = r0[1] - r0[2] - r0[3];
                const float y10 = r1[0] + r1[1] + r1[2], y11 = r1[1] - r1[2] - r1[3];
                const float v0 = fmaxf(fmaf(y00, dsc, bias), 0.f), v1 = fmaxf(fmaf(y01, dsc, bias), 0.f);
                const float v2 = fmaxf(fmaf(y10, dsc, bias), 0.f), v3 = fmaxf(fmaf(y11, dsc, bias), 0.f);
                pv[j] = (v0 + v1) + (v2 + v3);
            }
            pool += (pv[0] + pv[1]) + (pv[2] + pv[3]);
        }
        float p2 = pool + __shfl_xor(pool, 16);
        p2 += __shfl_xor(p2, 32);
        if (lane < 16 && (WAVES == 4 || wave < 4 || true)) atomicAdd(&out[(int64_t(blockIdx.x) * kClipsPerWg + k) * 64 + 16 * nt + lane], p2);
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int WAVES>
static int run(int buffers) {
    const int grid = 256, n = grid * kClipsPerWg;
    half8* wts; float* bd; float* out;
    CK(hipMalloc(&wts, 4 * 16 * 2 * 64 * sizeof(half8)));
    CK(hipMalloc(&bd, 128 * sizeof(float)));
    CK(hipMalloc(&out, size_t(n) * 64 * sizeof(float)));
    std::vector<uint16_t> hw(4 * 16 * 2 * 64 * 8);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = uint16_t(0x2c00u + (i * 2654435761u >> 22 & 0x3ffu));      // small positive halves
    CK(hipMemcpy(wts, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    std::vector<float> hb(128, 0.01f);
    CK(hipMemcpy(bd, hb.data(), 128 * 4, hipMemcpyHostToDevice));
    CK(hipMemset(out, 0, size_t(n) * 64 * 4));
    const int lds = buffers * kBuf;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(w2d_kernel<WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(w2d_kernel<WAVES>, dim3(grid), dim3(WAVES * 64), lds, 0, wts, bd, out, buffers);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(w2d_kernel<WAVES>, dim3(grid), dim3(WAVES * 64), lds, 0, wts, bd, out, buffers);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    float h = 0.f;
    CK(hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost));
    const double mfma = double(n) * kTileRows * 4 * 48;                       // MFMAs per launch (N-tiles x 48 per tile row)
    printf("waves %d, %d LDS buffers: %.4f ms per %d clips (out[0] = %g); %.0f MFMA / launch = %.3f of the pipe at 16 cycles, 2.4 GHz, 1024 SIMDs\n",
           WAVES, buffers, ms / reps, n, h, mfma, mfma * 16 / (ms / reps * 1e-3 * 2.4e9 * 1024));
    hipFree(wts); hipFree(bd); hipFree(out);
    return 0;
}

int main(int argc, char** argv) {
    const int waves = argc > 1 ? atoi(argv[1]) : 4;
    if (waves == 8) return run<8>(3);
    return run<4>(3);
}
