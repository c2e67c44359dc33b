// K2: Conv2d(3x3, pad 1)+ReLU stack + global average pool on the matrix cores.
//
// Replaces `F.relu(conv1) -> F.relu(conv2) [-> F.relu(conv3)] -> AdaptiveAvgPool2d((1,1))`
// (/root/reference/wakeword_training/train_wakeword.py:39-41,
//  /root/reference/wakeword_training_script.py:170-173).
//
// conv2 (32->64, 94.4 of the model's 96.5 MFLOP per clip) and conv3 (64->128) are implicit GEMMs:
//     M = columns of one image row, N = output channels, K = (input channel, dy, dx).
// Two arithmetics, selected process-wide (ww_set_conv_math; both parity-tested on every GPU test):
//
//   f16x3 (default)  cnn2h16_kernel<POOL> [+ cnn3h_kernel]: every fp32 operand as two f16 halves, three
//                    v_mfma_f32_16x16x32_f16 per product block, fp32 accumulate (~2^-21 relative).  12-wave workgroup:
//                    8 consumer waves (2 per SIMD) + 4 producer waves that run conv1 -- itself an MFMA -- one band ahead
//                    into a double-buffered LDS tile; the roles meet through progress counters in LDS, not barriers.
//   f32              cnn2_kernel<POOL> [+ cnn3_kernel]: v_mfma_f32_32x32x2_f32, exact fp32 (bit-for-bit an fmaf chain);
//                    conv1 on the VALU per band.
//
// Common to all: the weights of a wave's N-tile stay in VGPRs for every clip a persistent workgroup processes (no weight
// re-reads); an activation fragment read from LDS is reused for every (output row, dy) pair that touches its input
// row; conv1's 327 KB/clip output never exists in HBM; bias + ReLU + pool are fused into the MFMA epilogue with
// fixed-order reductions (deterministic, NaN-propagating like torch).
//
// Algorithmic flops per clip (T = 32): conv1 1,474,560 + conv2 94,371,840 [+ conv3 377,487,360].
// Diagnostic build: -DWW_STAMPS adds s_memtime phase stamps (never in the shipped library).
#include <cstdlib>
#include <type_traits>
#include <mutex>

#include "ww_conv1.h"
#include "ww_internal.h"

namespace ww {

constexpr int kH = WW_N_MELS;      // image rows
constexpr int kW = 32;             // image columns = lanes of an M-tile
constexpr int kRS = 34;            // LDS row stride: columns -1..32
constexpr int kMelRS = 36;         // mel tile row stride (rows -1..80, columns -1..34)
constexpr int kBand = 8;           // conv2 output rows per band
constexpr int kARows = kBand + 2;  // conv1 rows needed per band

// LDS of the conv1+conv2 kernel (floats)
constexpr int kActFloats = 32 * kARows * kRS;          // 10,880
constexpr int kMelFloats = (kH + 2) * kMelRS;          // 2,952
constexpr int kC2LdsFloats = kActFloats + kMelFloats + 4 * 32;

// ReLU that propagates NaN like torch's (F.relu(nan) = nan); fmaxf(nan, 0) would return 0
__device__ __forceinline__ float relu(float v) { return v < 0.f ? 0.f : v; }

__device__ __forceinline__ void zero_lds(float* p, int n, int tid, int nthreads) {
    for (int i = tid; i < n; i += nthreads) p[i] = 0.f;
}

// Pooling order (all conv kernels): the values of one tile step are summed as a balanced tree and the step's partial is then added to
// the running total -- the same number of adds as one long chain, but the chain is 10-40 partials long instead of 640-1280 values.
// A long chain of near-equal addends (a constant image: every position the same activation) rounds the same way at every step and
// drifted the pooled mean by 1.6e-5 relative in the exact-fp32 kernels (tests/test_gpu_guards.py, w:scaled_1e-6); partial sums of equal
// values are exact doublings.
__device__ __forceinline__ float tree4(float a, float b, float c, float d) { return (a + b) + (c + d); }
__device__ __forceinline__ float tree8(const float* v) { return tree4(v[0], v[1], v[2], v[3]) + tree4(v[4], v[5], v[6], v[7]); }
__device__ __forceinline__ float tree16(const float* v) { return tree8(v) + tree8(v + 8); }

// conv1 for one band: act[ci][q][x+1] = relu(b1[ci] + sum w1[ci][dy][dx] * mel[y+dy-1][x+dx-1]),
// y = y0 - 1 + q; zero where the position lies outside the image (that is conv2's zero padding).
// wave w computes channels 8w..8w+7 (wave-uniform -> scalar weight loads); lane = (column, row parity).
__device__ __forceinline__ void conv1_band(const float* __restrict__ melt, float* __restrict__ act,
                                           const float* __restrict__ w1, const float* __restrict__ b1, int y0,
                                           int width, int wave, int lane) {
    const int x = lane & 31, h = lane >> 5;
#pragma unroll 1
    for (int i = 0; i < kARows / 2; ++i) {
        const int q = 2 * i + h;
        const int y = y0 - 1 + q;
        const bool inside = (y >= 0) && (y < kH) && (x < width);
        const float* m = melt + (inside ? y : 0) * kMelRS + x;     // rows y-1..y+1 -> tile rows y..y+2
        const float m00 = m[0], m01 = m[1], m02 = m[2];
        const float m10 = m[kMelRS], m11 = m[kMelRS + 1], m12 = m[kMelRS + 2];
        const float m20 = m[2 * kMelRS], m21 = m[2 * kMelRS + 1], m22 = m[2 * kMelRS + 2];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int ci = 8 * wave + u;
            const float* w = w1 + ci * 9;
            float v = b1[ci];
            v = fmaf(w[0], m00, v); v = fmaf(w[1], m01, v); v = fmaf(w[2], m02, v);
            v = fmaf(w[3], m10, v); v = fmaf(w[4], m11, v); v = fmaf(w[5], m12, v);
            v = fmaf(w[6], m20, v); v = fmaf(w[7], m21, v); v = fmaf(w[8], m22, v);
            v = inside ? relu(v) : 0.f;
            act[(ci * kARows + q) * kRS + x + 1] = v;
        }
    }
}

// 144 k-steps (16 channel pairs x 3 dx x 3 dy) over 4 output rows: 576 MFMAs, 288 LDS reads.
// ap = &act[(h*ROWS + first input row)*kRS + x]; PLANE2 = floats between channel pair c and c+1.
template <int ROWS>
__device__ __forceinline__ void mfma_rows4(const float* __restrict__ ap, const float (&wb)[144], f32x16 (&acc)[4]) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            float a[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) a[q] = ap[(2 * c * ROWS + q) * kRS + dx];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r + dy], wb[(c * 3 + dy) * 3 + dx], acc[r], 0, 0, 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// conv1 + conv2 (+ pool | + store for the 3-conv model).  256 threads: wave = (row group rg, N-tile nt).
// POOL : out = pooled [n][64]
// !POOL: out = relu(conv2) as [n][80][64][32] (row, channel, column) for the conv3 kernel
// ------------------------------------------------------------------------------------------------
template <bool POOL>
__global__ __launch_bounds__(256, 2) void cnn2_kernel(const float* __restrict__ mel, int n, int width,
                                                      const float* __restrict__ w1, const float* __restrict__ b1,
                                                      const float* __restrict__ wB, const float* __restrict__ b2,
                                                      float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* act = lds;                       // [32][10][34]
    float* melt = act + kActFloats;         // [82][36]
    float* red = melt + kMelFloats;         // [4][32]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = wave & 1, rg = wave >> 1;
    const int x = lane & 31, h = lane >> 5;

    float wb[144];
#pragma unroll
    for (int i = 0; i < 144; ++i) wb[i] = wB[(nt * 144 + i) * 64 + lane];
    const float bias = b2[32 * nt + x];

    zero_lds(lds, kC2LdsFloats, tid, 256);   // halos stay zero for the life of the workgroup
    const float* ap = act + (h * kARows + rg * 4) * kRS + x;
    const float inv_area = 1.0f / float(kH * width);

    for (int clip = blockIdx.x; clip < n; clip += gridDim.x) {
        __syncthreads();   // previous clip fully consumed (and the zero fill on the first trip)
        const float* __restrict__ src = mel + int64_t(clip) * kH * width;
        for (int i = tid; i < kH * width; i += 256) {
            const int y = i / width, xx = i - y * width;
            melt[(y + 1) * kMelRS + xx + 1] = src[i];
        }
        float pool = 0.f;
        for (int band = 0; band < kH / kBand; ++band) {
            const int y0 = band * kBand;
            __syncthreads();   // mel tile ready / previous band's A reads retired
            conv1_band(melt, act, w1, b1, y0, width, wave, lane);
            __syncthreads();
            f32x16 acc[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[r][j] = 0.f;
            mfma_rows4<kARows>(ap, wb, acc);
            // D layout: lane&31 = channel, register j <-> column (j&3) + 8*(j>>2) + 4*(lane>>5)
            if constexpr (POOL) {
                float rows[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float vs[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int col = (j & 3) + 8 * (j >> 2) + 4 * h;
                        const float v = relu(acc[r][j] + bias);
                        vs[j] = (col < width) ? v : 0.f;
                    }
                    rows[r] = tree16(vs);
                }
                pool += tree4(rows[0], rows[1], rows[2], rows[3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int y = y0 + rg * 4 + r;
                    float* dst = out + ((int64_t(clip) * kH + y) * 64 + 32 * nt + x) * kW;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int col0 = 8 * g + 4 * h;
                        float4 v;
                        v.x = (col0 + 0 < width) ? relu(acc[r][4 * g + 0] + bias) : 0.f;
                        v.y = (col0 + 1 < width) ? relu(acc[r][4 * g + 1] + bias) : 0.f;
                        v.z = (col0 + 2 < width) ? relu(acc[r][4 * g + 2] + bias) : 0.f;
                        v.w = (col0 + 3 < width) ? relu(acc[r][4 * g + 3] + bias) : 0.f;
                        *reinterpret_cast<float4*>(dst + col0) = v;
                    }
                }
            }
        }
        if constexpr (POOL) {
            pool += __shfl_xor(pool, 32);
            __syncthreads();
            if (lane < 32) red[wave * 32 + lane] = pool;
            __syncthreads();
            if (tid < 64) {
                const int t_nt = tid >> 5, t_x = tid & 31;
                const float s = red[t_nt * 32 + t_x] + red[(2 + t_nt) * 32 + t_x];   // row groups 0 and 1
                out[int64_t(clip) * 64 + tid] = s * inv_area;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// conv1 + conv2 + pool with SPLIT-PRECISION conv2: every fp32 operand is carried as two f16 halves
// (x ~= hi + lo, 22 significant bits) and each product block runs as three f16 MFMAs accumulating in fp32:
//     a*w ~= a_hi*w_hi + a_hi*w_lo + a_lo*w_hi          (the dropped a_lo*w_lo is < 2^-22 relative)
// Split precision: the f16 matrix instructions do 16x the flops per cycle of v_mfma_f32_32x32x2_f32, so three of them
// per product block need 3/16 of the exact-f32 kernel's matrix-pipe cycles.  f16 products (11 x 11 bits) are exact in
// the fp32 accumulator.  Weights are pre-scaled by 2^S on the host (both halves normal f16); the accumulator is
// descaled by 2^-S.
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// Split-precision conv1+conv2 kernel (v_mfma_f32_16x16x32_f16): 12 waves per workgroup (one per CU, 3 waves per SIMD,
// <= 168 VGPRs):
//   waves 0-7   CONSUMERS = (row group of 4 rows) x (N-tile of 16 channels): 72 B-operand VGPRs each, so TWO matrix-pipe
//               waves share every SIMD and hide each other's LDS latency (one 250-VGPR wave per SIMD cannot)
//   waves 8-11  PRODUCERS: conv1 of the next band (itself an MFMA) into the other half of the double-buffered tile
// Position record = 160 bytes ([32 ci hi][32 ci lo][32 B pad]): conflict-free for the 16x16x32 A-fragment reads
// (lane = (position i, channel quarter kq) reads 16 B at position*160 + kq*16).
// ------------------------------------------------------------------------------------------------
constexpr int kPos16 = 160;
constexpr int kH16Row = kRS * kPos16;
constexpr int kH16Act = kARows * kH16Row;                   // 54,400 B per buffer
constexpr int kC2h16Lds = 2 * kH16Act + 4 * ((kH + 2) * 36) * 2 + 8 * 16 * 4;

// D: lane&31 = column x; the weight rows are permuted on the host so that register j holds channel 16*h + j:
// descale, 2*relu, hi/lo split, and the lane's 16 contiguous channels go out as two 16-byte stores per half.
__device__ __forceinline__ void conv1_row_store(const Conv1Row& r, const Conv1Scale& cs, char* __restrict__ rec) {
#pragma unroll
    for (int g8 = 0; g8 < 2; ++g8) {
        u32x4 vh, vl;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int j = 8 * g8 + 2 * d;
            uint32_t hh, ll;       // 2*relu: the factor is folded into conv2's descale
            split2(relu2(r.acc[j] * cs.sc), relu2(r.acc[j + 1] * cs.sc), hh, ll);
            vh[d] = hh;
            vl[d] = ll;
        }
        *reinterpret_cast<u32x4*>(rec + g8 * 16) = vh;
        *reinterpret_cast<u32x4*>(rec + 64 + g8 * 16) = vl;
    }
}
__device__ __forceinline__ void conv1_row_zero(char* __restrict__ rec) {
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int g8 = 0; g8 < 2; ++g8) {
        *reinterpret_cast<u32x4*>(rec + g8 * 16) = z;
        *reinterpret_cast<u32x4*>(rec + 64 + g8 * 16) = z;
    }
}

// A producer runs tile rows qa and (if SECOND) qa + 4 of the band side by side.  mh: the clip's hi plane.
// Rows outside the image are conv2's zero padding: zero records (wave-uniform branch; two rows per clip).  Columns beyond
// `width` are never stored: their records keep the zeros of the kernel's initial fill.
template <int POS, bool SECOND>
__device__ __forceinline__ void conv1_rows_mfma(const _Float16* __restrict__ mh, const GatherLanes& gl, char* __restrict__ act,
                                                half8 a1h, half8 a1l, const Conv1Scale& cs, int y0, int width, int qa, int lane) {
    const int x = lane & 31, h = lane >> 5;
    const bool col_ok = x < width;
    const int ya = y0 - 1 + qa, yb = ya + 4;
    const bool va = ya >= 0 && ya < kH, vb = yb >= 0 && yb < kH;
    char* rec = act + (x + 1) * POS + h * 32;                          // this lane's 16 channels 16h..16h+15
    char* reca = rec + qa * kRS * POS;
    char* recb = rec + (qa + 4) * kRS * POS;
    if (va && (!SECOND || vb)) {
        Conv1Row r0, r1;
        conv1_row_gather(r0, mh, gl, ya);
        if (SECOND) conv1_row_gather(r1, mh, gl, yb);
        conv1_row_mfma(r0, a1h, a1l, cs);
        if (SECOND) conv1_row_mfma(r1, a1h, a1l, cs);
        if (col_ok) {
            conv1_row_store(r0, cs, reca);
            if (SECOND) conv1_row_store(r1, cs, recb);
        }
        return;
    }
    // a row above or below the image is involved (first tile row of band 0, last of band 9): one row at a time
    if (va) {
        Conv1Row r0;
        conv1_row_gather(r0, mh, gl, ya);
        conv1_row_mfma(r0, a1h, a1l, cs);
        if (col_ok) conv1_row_store(r0, cs, reca);
    } else {
        conv1_row_zero(reca);
    }
    if (SECOND) {
        if (vb) {
            Conv1Row r1;
            conv1_row_gather(r1, mh, gl, yb);
            conv1_row_mfma(r1, a1h, a1l, cs);
            if (col_ok) conv1_row_store(r1, cs, recb);
        } else {
            conv1_row_zero(recb);
        }
    }
}

// Workgroup-local progress counters in LDS (monotonic).  signal = release add by one wave; wait = acquire poll.  The
// poll is bounded (~0.2 s): a protocol bug then cannot hang the GPU.  An expired wait is FATAL for the workgroup's
// results: it raises the workgroup's `bad` word in LDS, and before the workgroup exits every output of every clip it
// processed is overwritten with NaN (a wave that ran ahead on stale data cannot be consumed silently).  It is also
// counted in g_sync_timeouts, which the host reads with ww_sync_timeouts() (the GPU tests assert that it stays 0).
__device__ unsigned int g_sync_timeouts;

// Diagnostic build -DWW_CLOCK (never in the shipped library): consumer wave 1 of every workgroup stamps s_memtime (shader cycles) and
// s_memrealtime (100 MHz) around its loop; ww_debug_clock() returns the sums -> in-kernel clock = cycles / ticks * 100 MHz
// (MI355X_MICROARCH.md, DVFS give-back item 6).  The stamps go to a buffer of their own; no output is computed from them.
#ifdef WW_CLOCK
__device__ unsigned long long g_clk[4];
#define CLK_BEGIN() unsigned long long clk_c0 = 0, clk_r0 = 0; \
    if (wave == 1) { clk_c0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
#define CLK_END() do { if (wave == 1 && lane == 0) { \
    atomicAdd(&g_clk[0], __builtin_amdgcn_s_memtime() - clk_c0); atomicAdd(&g_clk[1], __builtin_amdgcn_s_memrealtime() - clk_r0); \
    atomicAdd(&g_clk[2], 1ull); } } while (0)
extern "C" __attribute__((visibility("default"))) int ww_debug_clock(unsigned long long* out) {
    unsigned long long z[4] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_clk), sizeof(z)) != hipSuccess) return -1;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_clk), z, sizeof(z));
    return 0;
}
#else
#define CLK_BEGIN() do {} while (0)
#define CLK_END() do {} while (0)
#endif

// The fences are executed by EVERY lane (the counter itself only by lane 0 / read by all): release and acquire order the
// LDS accesses of the work-item that executes them, and a tile is written and read by all 64 lanes of a wave.
__device__ __forceinline__ void flag_signal(uint32_t* f) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#ifndef WW_FLAG_SPINS
#define WW_FLAG_SPINS (1 << 22)
#endif
// Bounded wait on an LDS counter: 2^22 polls x s_sleep(1) ~ 0.2-0.4 s.  The bound is a poll count, not a clock, on purpose: all waves of a
// workgroup are co-resident on one CU (and are context-switched together), so a partner wave cannot be "legitimately slow" by more than
// the work of one tile step (microseconds); and a clock-bounded slow path behind the poll loop (s_memrealtime, tried in round 3) is
// inlined at every wait site and cost cnn2w_kernel<1> two more VGPR spills at its 168-register cap.
__device__ __forceinline__ void flag_wait(uint32_t* f, uint32_t target, uint32_t* bad) {
#pragma unroll 1
    for (int spin = 0; spin < WW_FLAG_SPINS; ++spin) {
        if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= target) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            return;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&g_sync_timeouts, 1u);
        __hip_atomic_store(bad, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

#ifdef WW_STAMPS
__device__ unsigned long long g_cnn_stamps[16];
#define CSTAMP(i) do { unsigned long long t__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__) :: "memory"); \
    cst[i] += t__ - clast; clast = t__; } while (0)
#else
#define CSTAMP(i) do {} while (0)
#endif

// POOL: out = pooled [n][64].   !POOL (3-conv model): out = relu(conv2) * 2^-a2 already split for conv3, as f16
// [n][80 rows][32 columns][64 ci hi | 64 ci lo] (256 bytes per position, zero beyond `width`), and apow2[clip] = 2^a2
// (one float per clip behind the records; a2 from the bound on |conv2 out|, see the conv1 notes above).
// rng: [l1 bound of conv1, max |b1|, l1 bound of conv2, max |b2|] (ww_tables.cpp).
template <bool POOL>
__global__ __launch_bounds__(768, 3) void cnn2h16_kernel(const float* __restrict__ mel, int n, int width,
                                                         const u32x4* __restrict__ w1H, const float* __restrict__ hs1,
                                                         const float* __restrict__ b1,
                                                         const u32x4* __restrict__ wH, const float* __restrict__ hs,
                                                         const float* __restrict__ b2, const float* __restrict__ rng,
                                                         float* __restrict__ out, float* __restrict__ apow2) {
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    char* act0 = ldsb;
    _Float16* melh0 = reinterpret_cast<_Float16*>(ldsb + 2 * kH16Act);       // 2 clips x (hi plane, lo plane) of [82][36] f16
    float* red = reinterpret_cast<float*>(melh0 + 4 * kMelHPlane);           // [8 consumer waves][16]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < 8;
    const int nt = wave & 3, rg = (wave >> 2) & 1;
    const int pi = lane & 15, kq = lane >> 4;
    const int ptid = tid - 512;

    half8 bh[9], bl[9];
    float bias = 0.f, descale = 0.f;
    if (consumer) {
#pragma unroll
        for (int ks = 0; ks < 9; ++ks) {
            bh[ks] = __builtin_bit_cast(half8, wH[((nt * 9 + ks) * 2 + 0) * 64 + lane]);
            bl[ks] = __builtin_bit_cast(half8, wH[((nt * 9 + ks) * 2 + 1) * 64 + lane]);
        }
        bias = b2[16 * nt + pi];
        descale = 0.5f * hs[16 * nt + pi];       // this lane's output channel; conv1 activations are stored as 2*relu(.)
    }
    half8 a1h = {}, a1l = {};
    const GatherLanes glanes = gather_lanes(lane & 31, lane >> 5);
    if (!consumer) {
        u32x4 w1h_r = w1H[lane], w1l_r = w1H[64 + lane];
        asm volatile("" : "+v"(w1h_r), "+v"(w1l_r));    // opaque: otherwise the loads are rematerialised inside the band loop
        a1h = __builtin_bit_cast(half8, w1h_r);
        a1l = __builtin_bit_cast(half8, w1l_r);
        // The producers are VALU streams sharing each SIMD's issue port with two MFMA streams (an MFMA holds the port
        // for 8 of its 16 cycles); issue is arbitrated by priority, then age.  Without this the producers starve and
        // the consumers wait a third of the time at the band barrier.
        __builtin_amdgcn_s_setprio(3);
    }
    // progress counters: tiles produced x 4 producer waves, bands consumed x 8 consumer waves, mel planes loaded x 4,
    // per-wave input maxima published x 4
    __shared__ uint32_t prod_done, cons_done[2], mel_done, xmax_done, wg_bad;
    __shared__ float xmaxw[2][4];        // [clip parity][producer wave]: max |x| over the wave's share of the clip
    __shared__ float clip_par[2][2];     // [clip parity]: 2^a (conv1 activation exponent), bound on |conv1 out|
    if (tid == 0) { prod_done = 0u; cons_done[0] = 0u; cons_done[1] = 0u; mel_done = 0u; xmax_done = 0u; wg_bad = 0u; }
    for (int i = tid; i < kC2h16Lds / 4; i += 768) reinterpret_cast<uint32_t*>(ldsb)[i] = 0u;   // halos / dead columns stay zero
    __syncthreads();

    const int my_clips = (n - int(blockIdx.x) + int(gridDim.x) - 1) / int(gridDim.x);
    const int steps = my_clips * (kH / kBand);
    const float half_inv_area = 0.5f / float(kH * width);

    const float rng_l1 = rng[0], rng_b1 = rng[1];
    // Model input of clip k -> two f16 planes (hi, lo) of x * 2^-e with a zero halo; e from the clip's max |x| (a NaN is
    // ignored by the max and propagates through the arithmetic; an infinity scales to an infinity).  The four producer
    // waves publish their maxima through LDS and meet on a counter; slot reuse (clip k + 2) is safe because the producers
    // advance tile by tile.  Returns the clip's exponents: e (input), a (conv1 activations).
    auto load_mel = [&](int k, int& e_out, int& a_out) {
        const float* __restrict__ src = mel + (int64_t(blockIdx.x) + int64_t(k) * gridDim.x) * kH * width;
        _Float16* ph = melh0 + (k & 1) * 2 * kMelHPlane;
        // thread -> (column xx, rows y0 + 8 t): no division for any width; for width 32 consecutive threads read consecutive floats
        const int xx = ptid & 31, y0 = ptid >> 5;
        const bool col_live = xx < width;
        const float* __restrict__ sp = src + y0 * width + xx;
        float v[10];
        float mx = 0.f;
#pragma unroll
        for (int t = 0; t < 10; ++t) {
            v[t] = col_live ? sp[8 * t * width] : 0.f;
            mx = fmaxf(mx, __builtin_fabsf(v[t]));
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        if (lane == 0) xmaxw[k & 1][wave - 8] = mx;
        flag_signal(&xmax_done);
        flag_wait(&xmax_done, 4u * unsigned(k + 1), &wg_bad);
        mx = fmaxf(fmaxf(xmaxw[k & 1][0], xmaxw[k & 1][1]), fmaxf(xmaxw[k & 1][2], xmaxw[k & 1][3]));
        const int e = clampi(exp_of(mx) - 14, -100, 113);
        const float bound1 = fmaf(mx, rng_l1, rng_b1);
        const int a = clampi(exp_of(bound1) - 13, -100, 100);
        if (tid == 512) { clip_par[k & 1][0] = pow2i(a); clip_par[k & 1][1] = bound1; }
        const float down = pow2i(-e);
        if (col_live) {
            _Float16* dh = ph + (y0 + 1) * kMelHRS + xx + 1;
#pragma unroll
            for (int t = 0; t < 10; ++t) {
                const float vv = v[t] * down;
                const _Float16 hi = static_cast<_Float16>(vv);
                dh[8 * t * kMelHRS] = hi;
                dh[8 * t * kMelHRS + kMelHPlane] = static_cast<_Float16>(vv - static_cast<float>(hi));
            }
        }
        e_out = e;
        a_out = a;
    };
    Conv1Scale cs;
    const int s1_exp = -exp_of(hs1[0]);               // hs1[0] = 2^-S1
    auto set_conv1_scale = [&](int e, int a) {       // channel 16h + j of this producer lane
        const int c0 = 16 * (lane >> 5);
#pragma unroll
        for (int j = 0; j < 16; ++j) cs.binit[j] = ldexpf(b1[c0 + j], s1_exp - e);
        cs.sc = ldexpf(1.0f, e - a - s1_exp);
    };
    // conv1 of step g (clip g / 10, band g % 10) into tile g & 1.  Band 0 computes all ten rows (row 0 is the zero
    // padding above the image); for the other bands tile rows 0 and 1 (image rows 8b - 1, 8b) are the previous band's
    // rows 8 and 9, still in the other tile: they are COPIED (544 x 16 bytes over the four producer waves) and only
    // eight rows are computed, two per producer.
    auto produce = [&](int g) {
        const int k = g / (kH / kBand), band = g - k * (kH / kBand);
        const _Float16* plane = melh0 + (k & 1) * 2 * kMelHPlane;      // hi plane of the clip (lo plane behind it)
        char* tile = act0 + (g & 1) * kH16Act;
        const int pw = wave - 8;
        if (band == 0) {
            conv1_rows_mfma<kPos16, true>(plane, glanes, tile, a1h, a1l, cs, 0, width, 2 + pw, lane);
            if (pw < 2) conv1_rows_mfma<kPos16, false>(plane, glanes, tile, a1h, a1l, cs, 0, width, pw, lane);
            return;
        }
        // halo rows: read first, written last -- the LDS round trip hides under the conv1 work
        const char* prev = act0 + ((g & 1) ^ 1) * kH16Act + 8 * kH16Row;
        u32x4 halo[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int c = ptid + 256 * i;                              // chunk: position c >> 3 (2 rows x 34), 16-byte part c & 7
            if (c < 2 * kRS * 8) halo[i] = *reinterpret_cast<const u32x4*>(prev + (c >> 3) * kPos16 + (c & 7) * 16);
        }
        conv1_rows_mfma<kPos16, true>(plane, glanes, tile, a1h, a1l, cs, band * kBand, width, 2 + pw, lane);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int c = ptid + 256 * i;
            if (c < 2 * kRS * 8) *reinterpret_cast<u32x4*>(tile + (c >> 3) * kPos16 + (c & 7) * 16) = halo[i];
        }
    };
    auto write_pooled = [&](int k) {     // wave 0: pooled[clip][co], co = lane: sum the two row groups
        const int64_t clip = int64_t(blockIdx.x) + int64_t(k) * gridDim.x;
        const int t_nt = lane >> 4, t_n = lane & 15;
        out[clip * 64 + lane] = (red[t_nt * 16 + t_n] + red[(4 + t_nt) * 16 + t_n]) * half_inv_area;
    };

    float pool = 0.f;
#ifdef WW_STAMPS
    unsigned long long cst[8] = {0, 0, 0, 0, 0, 0, 0, 0}, clast = 0;
    CSTAMP(7);
#endif
    // One loop per role (the consumers' resident B operands and accumulators are then not live in the producers' code)
    // and NO workgroup barrier in the steady state; the roles meet through the three counters:
    //   a consumer starts band g when tile g is complete                          (prod_done >= 4 (g + 1))
    //   a producer writes tile g once every consumer has finished band g - 2      (cons_done[g & 1] >= 8 (g / 2), same buffer)
    //   the four producers advance tile by tile                                   (prod_done >= 4 g): the halo copy
    //     reads tile g - 1, which a producer running ahead would be overwriting with tile g + 1; the same wait frees
    //     the mel planes of the previous clip for the next load
    // so a consumer wave that is done early flows into the next band instead of idling at a barrier while its SIMD
    // sibling finishes (-4 %).  The consumers' counter is split by tile buffer (band parity): consumers are NOT in lock
    // step -- a fast wave may finish band g - 1 before a slow one finishes g - 2 -- so one running total could reach
    // 8 (g - 1) with a reader of the buffer still busy; a wave cannot be two bands (one parity) ahead, because the tile it
    // would need is gated by exactly this wait.  The producers' counter is exact because they advance tile by tile.
    if (!consumer) {
        int e_nx = 0, a_nx = 0;                       // exponents of the clip whose planes were loaded last
        if (steps > 0) { load_mel(0, e_nx, a_nx); flag_signal(&mel_done); }
        for (int g = 0; g < steps; ++g) {
            const int k = g / (kH / kBand), band = g - k * (kH / kBand);
            CSTAMP(0);
            flag_wait(&prod_done, 4u * unsigned(g), &wg_bad);
            if (band == 0) {
                flag_wait(&mel_done, 4u * unsigned(k + 1), &wg_bad);
                set_conv1_scale(e_nx, a_nx);
            }
            if (g >= 2) flag_wait(&cons_done[g & 1], 8u * unsigned(g / 2), &wg_bad);
            CSTAMP(4);
            produce(g);
            flag_signal(&prod_done);
            if (band == 0 && k + 1 < my_clips) {       // planes (k + 1) & 1 were clip k - 1's, whose last tile is complete
                load_mel(k + 1, e_nx, a_nx);
                flag_signal(&mel_done);
            }
            CSTAMP(3);
        }
    } else {
    float dsc = descale, a2inv = 1.f;    // per clip: descale * 2^a;  !POOL: 2^-a2 for the stored relu(conv2)
    for (int g = 0; g < steps; ++g) {
        const int k = g / (kH / kBand), band = g - k * (kH / kBand);
        CSTAMP(0);
        flag_wait(&prod_done, 4u * unsigned(g + 1), &wg_bad);                   // tile g complete
        if (band == 0) {
            // the clip's exponents were published before its first tile; slot k & 1 is rewritten for clip k + 2 only after
            // every consumer has finished band 8 of clip k, i.e. long after this read
            dsc = descale * clip_par[k & 1][0];
            if constexpr (!POOL) {
                const float bound2 = fmaf(clip_par[k & 1][1], rng[2], rng[3]);
                const int a2 = clampi(exp_of(bound2) - 14, -100, 100);
                a2inv = pow2i(-a2);
                if (tid == 0) apow2[int64_t(blockIdx.x) + int64_t(k) * gridDim.x] = pow2i(a2);
            }
        }
        if (POOL && band == 0 && g > 0 && wave == 0) {                // red[] of clip k - 1 complete (g is even)
            flag_wait(&cons_done[0], 8u * unsigned(g / 2), &wg_bad);
            flag_wait(&cons_done[1], 8u * unsigned(g / 2), &wg_bad);
        }
        CSTAMP(4);
        {
            if (POOL && band == 0 && g > 0 && wave == 0) write_pooled(k - 1);
            const char* ap = act0 + (g & 1) * kH16Act + ((rg * 4) * kRS + pi) * kPos16 + kq * 16;
            f32x4 acc[4][2];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[r][c][j] = 0.f;
            // 36 fragment steps it = (q*3 + dx)*2 + ch (input row outermost), pipelined one step ahead.  Output row r
            // receives its last contribution with input row q = r + 2, so its bias/ReLU/pool (or store) epilogue is
            // issued right there and runs under the MFMAs of the following input rows; only row 3's is left at the end.
            auto frag = [&](int it, int half) -> half8 {
                const int q = it / 6, dx = (it / 2) % 3, ch = it & 1;
                return __builtin_bit_cast(half8, *reinterpret_cast<const u32x4*>(ap + (q * kRS + 16 * ch + dx) * kPos16 + half * 64));
            };
            if (band == 0) pool = 0.f;
            auto epilogue_row = [&](int r) {
                if constexpr (!POOL) {
                    _Float16* o16 = reinterpret_cast<_Float16*>(out);
                    const int64_t clip = int64_t(blockIdx.x) + int64_t(k) * gridDim.x;
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int xx = 16 * c + 4 * kq + j, y = band * kBand + rg * 4 + r;
                            float v = 0.5f * a2inv * relu2(fmaf(acc[r][c][j], dsc, bias));
                            v = xx < width ? v : 0.f;
                            const _Float16 hi = static_cast<_Float16>(v);
                            const int64_t rec = ((clip * kH + y) * kW + xx) * 128 + 16 * nt + pi;
                            o16[rec] = hi;
                            o16[rec + 64] = static_cast<_Float16>(v - static_cast<float>(hi));
                        }
                } else if (width == kW) {
                    float vs[8];
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int j = 0; j < 4; ++j) vs[4 * c + j] = relu2(fmaf(acc[r][c][j], dsc, bias));
                    pool += tree8(vs);
                } else {
                    float vs[8];
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float v = relu2(fmaf(acc[r][c][j], dsc, bias));
                            vs[4 * c + j] = (16 * c + 4 * kq + j < width) ? v : 0.f;
                        }
                    pool += tree8(vs);
                }
            };
            // fragments run TWO steps ahead of their MFMAs in a 3-deep register ring; the scheduling barriers keep the
            // compiler from sinking the ds_reads back down to their first use (it does, to save 8 VGPRs, and then every
            // step waits out a full LDS round trip)
#ifndef WW_K2_PF
#define WW_K2_PF 2
#endif
            constexpr int PF = WW_K2_PF, RING = PF + 1;
            half8 fh[RING], fl[RING];
#pragma unroll
            for (int i = 0; i < PF; ++i) { fh[i] = frag(i, 0); fl[i] = frag(i, 1); }
#pragma unroll
            for (int it = 0; it < 36; ++it) {
                if (it + PF < 36) { fh[(it + PF) % RING] = frag(it + PF, 0); fl[(it + PF) % RING] = frag(it + PF, 1); }
                __builtin_amdgcn_sched_barrier(0);
                const half8 ah = fh[it % RING], al = fl[it % RING];
                const int q = it / 6, dx = (it / 2) % 3, ch = it & 1;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int r = q - dy;
                    if (r < 0 || r > 3) continue;
                    const int ks = dx * 3 + dy;
                    acc[r][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[ks], acc[r][ch], 0, 0, 0);
                    acc[r][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[ks], acc[r][ch], 0, 0, 0);
                    acc[r][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[ks], acc[r][ch], 0, 0, 0);
                }
                if (it % 6 == 5 && q >= 2) epilogue_row(q - 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            CSTAMP(1);
            if (POOL && band == kH / kBand - 1) {
                float p2 = pool + __shfl_xor(pool, 16);
                p2 += __shfl_xor(p2, 32);
                if (lane < 16) red[wave * 16 + lane] = p2;
            }
            CSTAMP(2);
        }
        flag_signal(&cons_done[g & 1]);
    }
    }
    if (POOL && consumer && wave == 0 && steps > 0) {
        flag_wait(&cons_done[0], 8u * unsigned(steps / 2), &wg_bad);       // steps = 10 x clips: even
        flag_wait(&cons_done[1], 8u * unsigned(steps / 2), &wg_bad);
        write_pooled(my_clips - 1);
    }
    // an expired wait anywhere in this workgroup: poison everything it produced (pooled features, or the conv3 scale)
    __syncthreads();
    if (__hip_atomic_load(&wg_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) {
        const float nan = __uint_as_float(0x7fc00000u);
        for (int k = 0; k < my_clips; ++k) {
            const int64_t clip = int64_t(blockIdx.x) + int64_t(k) * gridDim.x;
            if constexpr (POOL) { if (tid < 64) out[clip * 64 + tid] = nan; }
            else { if (tid == 0) apow2[clip] = nan; }
        }
    }
#ifdef WW_STAMPS
    if (lane == 0 && blockIdx.x == 7 && (wave == 1 || wave == 9))
        for (int i = 0; i < 8; ++i) atomicAdd(&g_cnn_stamps[i + (wave == 9 ? 8 : 0)], cst[i]);
#endif
}
#ifdef WW_STAMPS
extern "C" __attribute__((visibility("default"))) int ww_debug_cnn_stamps(unsigned long long* out) {
    unsigned long long z[16] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cnn_stamps), sizeof(z)) != hipSuccess) return -1;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_cnn_stamps), z, sizeof(z));
    return 0;
}
#endif

// v_writelane_b32: the wave-uniform value goes into ONE lane of the register (no clang builtin in this toolchain)
template <int LANE>
__device__ __forceinline__ uint32_t write_lane(uint32_t uniform_value, uint32_t reg) {
    // s_nop 1: on gfx950 a VALU write of an SGPR (a v_cmp's ballot) needs two wait states before a VALU instruction reads it; the compiler
    // pads its own instructions but does not see the operands of inline asm.  (Round 3: a width-32 copy of the mask epilogue fed the
    // ballots straight from v_cmp into this instruction and stored wrong bits; with an s_and in between, as now, it happened to be safe.)
#ifdef WW_ABL_WRITELANE_NONOP       // timing-only A/B: the instruction without its wait states
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(reg) : "s"(uniform_value), "n"(LANE));
#else
    asm volatile("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(reg) : "s"(uniform_value), "n"(LANE));      // lane select: an inline constant
#endif
    return reg;
}
// The eight ballots of a column half (rows 0 / 1 x register j) -> lanes 0..15 of one register: 16 v_writelane behind ONE pair of wait
// states (round 4; round 3 issued each v_writelane behind its own `s_nop 1`: 30 more scalar instructions per tile row and wave of the
// training forwards).  The ballots are written by v_cmp long before this block; the s_nop covers the last of them.
__device__ __forceinline__ uint32_t ballots_to_lanes(const unsigned long long (&live0)[4], const unsigned long long (&live1)[4]) {
    uint32_t word = 0u;
    asm volatile("s_nop 1\n\t"
                 "v_writelane_b32 %0, %1, 0\n\tv_writelane_b32 %0, %2, 1\n\tv_writelane_b32 %0, %3, 2\n\tv_writelane_b32 %0, %4, 3\n\t"
                 "v_writelane_b32 %0, %5, 4\n\tv_writelane_b32 %0, %6, 5\n\tv_writelane_b32 %0, %7, 6\n\tv_writelane_b32 %0, %8, 7\n\t"
                 "v_writelane_b32 %0, %9, 8\n\tv_writelane_b32 %0, %10, 9\n\tv_writelane_b32 %0, %11, 10\n\tv_writelane_b32 %0, %12, 11\n\t"
                 "v_writelane_b32 %0, %13, 12\n\tv_writelane_b32 %0, %14, 13\n\tv_writelane_b32 %0, %15, 14\n\tv_writelane_b32 %0, %16, 15"
                 : "+v"(word)
                 : "s"(uint32_t(live0[0])), "s"(uint32_t(live0[0] >> 32)), "s"(uint32_t(live0[1])), "s"(uint32_t(live0[1] >> 32)),
                   "s"(uint32_t(live0[2])), "s"(uint32_t(live0[2] >> 32)), "s"(uint32_t(live0[3])), "s"(uint32_t(live0[3] >> 32)),
                   "s"(uint32_t(live1[0])), "s"(uint32_t(live1[0] >> 32)), "s"(uint32_t(live1[1])), "s"(uint32_t(live1[1] >> 32)),
                   "s"(uint32_t(live1[2])), "s"(uint32_t(live1[2] >> 32)), "s"(uint32_t(live1[3])), "s"(uint32_t(live1[3] >> 32)));
    return word;
}

// ------------------------------------------------------------------------------------------------
// cnn2w_kernel: conv1 + conv2 + ReLU + pool with conv2 as a ONE-DIMENSIONAL WINOGRAD F(2,3) along the image rows
// (split precision, v_mfma_f32_16x16x32_f16 x3): two output rows (2t, 2t+1) of a "tile row" t come from the four
// conv1 rows d0..d3 = 2t-1 .. 2t+2 as
//     V0 = d0 - d2,  V1 = d1 + d2,  V2 = d2 - d1,  V3 = d1 - d3                        (producers, fp32, per lane)
//     M_xi = sum_{dx, ci} V_xi[x + dx - 1][ci] * U_xi[dx][ci][co]                       (consumers, matrix cores)
//         U0 = w[dy=0],  U1 = (w0 + w1 + w2)/2,  U2 = (w0 - w1 + w2)/2,  U3 = w[dy=2]    (host, double)
//     out[2t] = M0 + M1 + M2,   out[2t+1] = M1 - M2 - M3
// 12 instead of 18 MFMA triples per (two output rows, column half, N-tile): 1.5x fewer matrix-pipe cycles than the
// direct form, which is bound by them.  The transform runs ALONG ROWS so that it is pure per-lane arithmetic in the
// producers (a lane holds one column) and the consumers keep only one tile row's eight accumulators live (B operands
// 96 + accumulators 32 VGPRs); the column direction stays a direct 3-tap convolution through shifted fragment addresses.
// The errors are those of fp32 adds on the activations (no cancellation beyond one subtraction) -- tested like the
// direct form against the float64 oracle.
// 12 waves per workgroup, one per CU:
//   waves 8-11  PRODUCERS: wave pw makes the ten CONSECUTIVE tile rows t = 10 pw .. 10 pw + 9 of every clip, so that two of a
//               tile row's four conv1 rows are the previous tile row's (kept in registers) and every conv1 row is computed
//               once: conv1 on the matrix cores, 2*relu, transform, hi/lo split, one record per (xi, column);
//   waves 0-7   CONSUMERS = (N-tile of 16 channels) x (group g): group g takes the tile rows of producers 2g and 2g + 1,
//               alternately (the pool is a sum over all positions: the order of the tile rows is free, and fixed);
//               24 fragment steps x 3 MFMAs per tile row.
// Each group has a ring of 3 tile-row buffers in LDS.  Synchronisation: per buffer a FULL counter (producer -> consumers)
// and a FREE counter (4 consumer waves -> producer), monotonic, bounded polls, poisoning as in cnn2h16_kernel.
// ------------------------------------------------------------------------------------------------
constexpr int kWTileRows = kH / 2;                  // 40
constexpr int kWRec = 160;                          // [32 ci hi][32 ci lo][32 B pad]: conflict-free 16x16x32 fragment reads
constexpr int kWPlane = kRS * kWRec;                // one xi plane of a tile row: columns -1..32
constexpr int kWBuf = 4 * kWPlane;                  // 21,760 B per tile row
constexpr int kWRing = 3;                           // tile-row buffers per consumer group
constexpr int kWNB = 2 * kWRing;
constexpr int kWPerProd = kWTileRows / 4;           // 10 tile rows per producer and clip
constexpr int kWPerGroup = kWTileRows / 2;          // 20 tile rows per consumer group and clip
constexpr int kCWLds = kWNB * kWBuf + 4 * kMelHPlane * 2 + 2 * 8 * 16 * 4;
static_assert(kWTileRows % 4 == 0, "four producers with equal shares");

// OUT 1 (POOL): out = pooled [n][64].   OUT 0 (3-conv model): out = relu(conv2) as float32 [n][80 rows][32 columns][64 channels] (zero
// beyond `width`) for cnn3w_kernel, and apow2[clip] = 2^a2, the exponent that kernel gives its transformed conv3 inputs.
// OUT 2 (training forward of the 2-conv model): pooled as under OUT 1, plus the ReLU mask of conv2 as the accumulator ballots themselves,
// bits[n][40 tile rows][4 N-tiles][2 column halves][2 rows][4 j] x 64 bits (ww_train_h.hip: mask2_byte), and bits1[n][80][32] = 32 bits per position, bit c = [relu(conv1)[c] > 0]
// (2 x uint16, one per producer half-wave): all the backward pass needs of the activations (ww_train_h.hip).
// OUT 3 (training forward of the 3-conv model): relu(conv2) and apow2 as under OUT 0, plus bits1.
template <int OUT>
__global__ __launch_bounds__(768, 3) void cnn2w_kernel(const float* __restrict__ mel, int n, int width,
                                                       const u32x4* __restrict__ w1H, const float* __restrict__ hs1,
                                                       const float* __restrict__ b1,
                                                       const u32x4* __restrict__ wH, const float* __restrict__ hs,
                                                       const float* __restrict__ b2, const float* __restrict__ rng,
                                                       float* __restrict__ out, float* __restrict__ apow2, uint16_t* __restrict__ bits,
                                                       uint16_t* __restrict__ bits1) {
#ifdef WW_ABL_NOBITS            // timing-only ablations of the training forward: 1 = no conv2 mask image, 2 = no conv1 sign image either
    constexpr bool POOL = OUT == 1 || OUT == 2, BITS = false, BITS1 = WW_ABL_NOBITS < 2 && (OUT == 2 || OUT == 3);
#else
    constexpr bool POOL = OUT == 1 || OUT == 2, BITS = OUT == 2, BITS1 = OUT == 2 || OUT == 3;
#endif
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    char* act0 = ldsb;
    _Float16* melh0 = reinterpret_cast<_Float16*>(ldsb + kWNB * kWBuf);      // 2 clips x (hi plane, lo plane) of [82][36] f16
    float* red = reinterpret_cast<float*>(melh0 + 4 * kMelHPlane);           // [clip parity][8 consumer waves][16]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < 8;
    const int nt = wave & 3, grp = (wave >> 2) & 1;
    const int pi = lane & 15, kq = lane >> 4;
    const int ptid = tid - 512;

    __shared__ uint32_t full_cnt[kWNB], free_cnt[kWNB], mel_done, xmax_done, clip_done[2], wg_bad;
    __shared__ float xmaxw[2][4];
    __shared__ float clip_par[2][2];
    if (tid < kWNB) { full_cnt[tid] = 0u; free_cnt[tid] = 0u; }
    if (tid == 0) { mel_done = 0u; xmax_done = 0u; clip_done[0] = 0u; clip_done[1] = 0u; wg_bad = 0u; }
    for (int i = tid; i < kCWLds / 4; i += 768) reinterpret_cast<uint32_t*>(ldsb)[i] = 0u;   // column halos / dead columns stay zero
    __syncthreads();

    const int my_clips = (n - int(blockIdx.x) + int(gridDim.x) - 1) / int(gridDim.x);
    const int steps = my_clips * kWTileRows;

    if (!consumer) {
        // ================================================= producers =================================================
        const int pw = wave - 8;
        u32x4 w1h_r = w1H[lane], w1l_r = w1H[64 + lane];
        asm volatile("" : "+v"(w1h_r), "+v"(w1l_r));
        const half8 a1h = __builtin_bit_cast(half8, w1h_r), a1l = __builtin_bit_cast(half8, w1l_r);
        const GatherLanes glanes = gather_lanes(lane & 31, lane >> 5);
#ifndef WW_WINO_PPRIO
#define WW_WINO_PPRIO 3
#endif
        __builtin_amdgcn_s_setprio(WW_WINO_PPRIO);
        const float rng_l1 = rng[0], rng_b1 = rng[1];
        const int s1_exp = -exp_of(hs1[0]);
        // model input of clip k -> two f16 planes of x * 2^-e (see cnn2h16_kernel).  All four producers meet inside (the xmax
        // counter), which also tells each of them that nobody still reads the planes being overwritten.
        auto load_mel = [&](int k, int& e_out, int& a_out) {
            const float* __restrict__ src = mel + (int64_t(blockIdx.x) + int64_t(k) * gridDim.x) * kH * width;
            _Float16* ph = melh0 + (k & 1) * 2 * kMelHPlane;
            const int xx = ptid & 31, y0 = ptid >> 5;
            const bool col_live = xx < width;
            const float* __restrict__ sp = src + y0 * width + xx;
            float v[10];
            float mx = 0.f;
#pragma unroll
            for (int t = 0; t < 10; ++t) {
                v[t] = col_live ? sp[8 * t * width] : 0.f;
                mx = fmaxf(mx, __builtin_fabsf(v[t]));
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
            if (lane == 0) xmaxw[k & 1][pw] = mx;
            flag_signal(&xmax_done);
            flag_wait(&xmax_done, 4u * unsigned(k + 1), &wg_bad);
            mx = fmaxf(fmaxf(xmaxw[k & 1][0], xmaxw[k & 1][1]), fmaxf(xmaxw[k & 1][2], xmaxw[k & 1][3]));
            const int e = clampi(exp_of(mx) - 14, -100, 113);
            const float bound1 = fmaf(mx, rng_l1, rng_b1);
            const int a = clampi(exp_of(bound1) - 12, -100, 100);       // |V| <= 2 * (2 relu): one more bit of headroom than the direct form
            if (tid == 512) { clip_par[k & 1][0] = pow2i(a); clip_par[k & 1][1] = bound1; }
            const float down = pow2i(-e);
            if (col_live) {
                _Float16* dh = ph + (y0 + 1) * kMelHRS + xx + 1;
#pragma unroll
                for (int t = 0; t < 10; ++t) {
                    const float vv = v[t] * down;
                    const _Float16 hi = static_cast<_Float16>(vv);
                    dh[8 * t * kMelHRS] = hi;
                    dh[8 * t * kMelHRS + kMelHPlane] = static_cast<_Float16>(vv - static_cast<float>(hi));
                }
            }
            e_out = e;
            a_out = a;
        };
        Conv1Scale cs;
        auto set_conv1_scale = [&](int e, int a) {
            const int c0 = 16 * (lane >> 5);
#pragma unroll
            for (int j = 0; j < 16; ++j) cs.binit[j] = ldexpf(b1[c0 + j], s1_exp - e);
            cs.sc = ldexpf(1.0f, e - a - s1_exp);
        };
        const int x = lane & 31, h = lane >> 5;
        const bool col_ok = x < width;
        auto store16 = [&](const f32x16& v, char* rec) {
#pragma unroll
            for (int g8 = 0; g8 < 2; ++g8) {
                u32x4 vh, vl;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    uint32_t hh, ll;
                    split2(v[8 * g8 + 2 * d], v[8 * g8 + 2 * d + 1], hh, ll);
                    vh[d] = hh;
                    vl[d] = ll;
                }
                *reinterpret_cast<u32x4*>(rec + g8 * 16) = vh;
                *reinterpret_cast<u32x4*>(rec + 64 + g8 * 16) = vl;
            }
        };
        int e_nx = 0, a_nx = 0;
        if (steps > 0) { load_mel(0, e_nx, a_nx); flag_signal(&mel_done); }
        const int pgrp = pw >> 1, podd = pw & 1;
        f32x16 d0, d1, d2, d3;                                // 2 relu(conv1) * 2^-a of image rows 2t-1 .. 2t+2
        // one conv1 row (zero outside the image: a clamped row times 0 -- conv2's zero padding; a NaN there can only meet outputs
        // that already see it through the image row itself)
        auto conv1_rows2 = [&](const _Float16* plane, int ya, f32x16& ra, f32x16& rb) {    // rows ya, ya + 1: stage by stage
            Conv1Row r0, r1;
            const bool va = unsigned(ya) < unsigned(kH), vb = unsigned(ya + 1) < unsigned(kH);
            conv1_row_gather(r0, plane, glanes, va ? ya : (ya < 0 ? 0 : kH - 1));
            conv1_row_gather(r1, plane, glanes, vb ? ya + 1 : (ya + 1 < 0 ? 0 : kH - 1));
            conv1_row_mfma(r0, a1h, a1l, cs);
            conv1_row_mfma(r1, a1h, a1l, cs);
            const float sa = va ? cs.sc : 0.f, sb = vb ? cs.sc : 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                ra[j] = relu2(r0.acc[j] * sa);
                rb[j] = relu2(r1.acc[j] * sb);
            }
        };
        for (int k = 0; k < my_clips; ++k) {
            flag_wait(&mel_done, 4u * unsigned(k + 1), &wg_bad);              // the clip's planes are complete
            set_conv1_scale(e_nx, a_nx);
            if (k + 1 < my_clips) { load_mel(k + 1, e_nx, a_nx); flag_signal(&mel_done); }
            const _Float16* plane = melh0 + (k & 1) * 2 * kMelHPlane;
#pragma unroll 1
            for (int i = 0; i < kWPerProd; ++i) {
                const int t = kWPerProd * pw + i;
                const int q = k * kWPerGroup + 2 * i + podd;                  // position in the group's sequence of tile rows
                const int b = pgrp * kWRing + q % kWRing;
#ifndef WW_ABL_NOPROD          // timing-only ablation: consumers alone (results are garbage)
                if (i == 0) conv1_rows2(plane, 2 * t - 1, d0, d1);
                else { d0 = d2; d1 = d3; }                                    // rows 2t-1, 2t are the previous tile row's 2t'+1, 2t'+2
                conv1_rows2(plane, 2 * t + 1, d2, d3);
                if constexpr (BITS1) {
                    // sign bits of conv1, every image row once: rows 2t+1, 2t+2 here; row 0 by the first producer's first tile row
                    auto sign16 = [&](const f32x16& v) {
                        uint32_t m = 0u;
#pragma unroll
                        for (int j = 0; j < 16; ++j) m |= (col_ok && v[j] > 0.f) ? (1u << j) : 0u;
                        return uint16_t(m);
                    };
                    uint16_t* o1 = bits1 + (((int64_t(blockIdx.x) + int64_t(k) * gridDim.x) * kH) * kW + x) * 2 + h;
                    if (t == 0 && i == 0) o1[0] = sign16(d1);
                    o1[(2 * t + 1) * kW * 2] = sign16(d2);
                    if (2 * t + 2 < kH) o1[(2 * t + 2) * kW * 2] = sign16(d3);
                }
#endif
#ifndef WW_ABL_X_NORING
                flag_wait(&free_cnt[b], 4u * unsigned(q / kWRing), &wg_bad);  // the group's four waves are done with the buffer's previous tile row
#endif
#ifndef WW_ABL_NOPROD
                if (col_ok) {
                    char* rec = act0 + b * kWBuf + (x + 1) * kWRec + h * 32;
                    store16(d0 - d2, rec);
                    store16(d1 + d2, rec + kWPlane);
                    store16(d2 - d1, rec + 2 * kWPlane);
                    store16(d1 - d3, rec + 3 * kWPlane);
                }
#endif
                flag_signal(&full_cnt[b]);
            }
        }
    } else {
        // ================================================= consumers =================================================
        half8 bh[12], bl[12];
#pragma unroll
        for (int ks = 0; ks < 12; ++ks) {
            bh[ks] = __builtin_bit_cast(half8, wH[((nt * 12 + ks) * 2 + 0) * 64 + lane]);
            bl[ks] = __builtin_bit_cast(half8, wH[((nt * 12 + ks) * 2 + 1) * 64 + lane]);
        }
        const float bias = b2[16 * nt + pi];
        const float descale = 0.5f * hs[16 * nt + pi];
        float dsc = descale, pool = 0.f;
#ifdef WW_WINO_CPRIO
        __builtin_amdgcn_s_setprio(WW_WINO_CPRIO);
#endif
        const int gsteps = my_clips * kWPerGroup;
        CLK_BEGIN();
        for (int q = 0; q < gsteps; ++q) {
            const int k = q / kWPerGroup, sq = q - k * kWPerGroup;
            const int b = grp * kWRing + q % kWRing;
#ifndef WW_ABL_X_NORING
            flag_wait(&full_cnt[b], unsigned(q / kWRing) + 1u, &wg_bad);
#endif
            if (sq == 0) {
                dsc = descale * clip_par[k & 1][0];
                pool = 0.f;
                if constexpr (!POOL) {
                    if (wave == 0 && lane == 0) {     // |V3| <= 2 max relu(conv2): one more bit of headroom than a plain split needs
                        const float bound2 = fmaf(clip_par[k & 1][1], rng[2], rng[3]);
                        apow2[int64_t(blockIdx.x) + int64_t(k) * gridDim.x] = pow2i(clampi(exp_of(bound2) - 13, -100, 100));
                    }
                }
            }
            const char* ap = act0 + b * kWBuf + pi * kWRec + kq * 16;
            f32x4 acc[4][2];
#pragma unroll
            for (int xi = 0; xi < 4; ++xi)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[xi][c][j] = 0.f;
            // 24 fragment steps it = dx*8 + c*4 + xi: consecutive steps hit different accumulators
            auto frag = [&](int it, int half) -> half8 {
                const int dx = it >> 3, c = (it >> 2) & 1, xi = it & 3;
#ifdef WW_ABL_HALFREAD      // timing-only ablation: the lo fragment is not read (the hi one stands in): half the consumers' LDS reads
                half = 0;
#endif
                return __builtin_bit_cast(half8, *reinterpret_cast<const u32x4*>(ap + xi * kWPlane + (16 * c + dx) * kWRec + half * 64));
            };
#ifndef WW_WINO_PF
#define WW_WINO_PF 2
#endif
            constexpr int PF = WW_WINO_PF, RING = PF + 1;
            half8 fh[RING], fl[RING];
#pragma unroll
            for (int i = 0; i < PF; ++i) { fh[i] = frag(i, 0); fl[i] = frag(i, 1); }
#ifdef WW_ABL_NOCONS           // timing-only ablation: producers alone
            constexpr int kIts = 0;
#else
            constexpr int kIts = 24;
#endif
#pragma unroll
            for (int it = 0; it < kIts; ++it) {
                if (it + PF < 24) { fh[(it + PF) % RING] = frag(it + PF, 0); fl[(it + PF) % RING] = frag(it + PF, 1); }
                __builtin_amdgcn_sched_barrier(0);
                const half8 ah = fh[it % RING], al = fl[it % RING];
                const int dx = it >> 3, c = (it >> 2) & 1, xi = it & 3;
                const int ks = xi * 3 + dx;
                acc[xi][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[ks], acc[xi][c], 0, 0, 0);
                acc[xi][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[ks], acc[xi][c], 0, 0, 0);
                acc[xi][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[ks], acc[xi][c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#ifndef WW_ABL_X_NORING
            flag_signal(&free_cnt[b]);                       // the buffer is free as soon as its fragments are in the accumulators
#endif
            // output transform + bias + 2*relu + pool (D: lane & 15 = channel, register j <-> column 16 c + 4 kq + j)
            const int trow = kWPerProd * (2 * grp + (sq & 1)) + (sq >> 1);      // the tile row this step holds (producer 2 grp + (sq & 1), its i-th)
            float pv[8];                                        // the tile row's eight (v0 + v1) pairs: summed as a tree, then into `pool`
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                unsigned long long live0[4], live1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#ifdef WW_ABL_X_NOFIN
                    const float v0 = (j == 0 && c == 0) ? ((acc[1][c][j] + acc[2][c][j]) + (acc[0][c][j] + acc[3][c][j])) + ((acc[1][1][2] + acc[2][1][2]) + (acc[0][1][2] + acc[3][1][2])) : 0.f, v1 = 0.f;
#else
                    const float m12 = acc[1][c][j] + acc[2][c][j], m1m2 = acc[1][c][j] - acc[2][c][j];
                    const float y0 = acc[0][c][j] + m12, y1 = m1m2 - acc[3][c][j];
                    const float v0 = relu2(fmaf(y0, dsc, bias)), v1 = relu2(fmaf(y1, dsc, bias));
#endif
                    const bool col_live = width == kW || 16 * c + 4 * kq + j < width;
                    if constexpr (POOL) {
                        pv[4 * c + j] = col_live ? v0 + v1 : 0.f;
                    } else {
                        const int64_t clip = int64_t(blockIdx.x) + int64_t(k) * gridDim.x;
                        float* o = out + ((clip * kH + 2 * trow) * kW + 16 * c + 4 * kq + j) * 64 + 16 * nt + pi;
                        o[0] = col_live ? 0.5f * v0 : 0.f;
                        o[kW * 64] = col_live ? 0.5f * v1 : 0.f;
                    }
                    if constexpr (BITS) {       // bit (16 kq + pi) of the ballot <-> channel 16 nt + pi at column 16 c + 4 kq + j
                        // the column mask is ANDed in as a scalar: ballot(col_live && v > 0) is lowered to v_cndmask + v_cmp_ne behind
                        // the compare (two more vector instructions per ballot, 32 per tile row)
                        // (nor as ballot(col_live): that is rebuilt from the lane mask the same way -- scalar arithmetic on the uniform width)
                        unsigned long long cm = 0ull;
#pragma unroll
                        for (int q = 0; q < 4; ++q) cm |= (width == kW || 16 * c + 4 * q + j < width) ? 0xFFFFull << (16 * q) : 0ull;
                        live0[j] = __builtin_amdgcn_ballot_w64(v0 > 0.f) & cm;
                        live1[j] = __builtin_amdgcn_ballot_w64(v1 > 0.f) & cm;
                    }
                }
                if constexpr (BITS) {           // the eight ballots go out as they are (see cnn3w_kernel<true>): [r][j] x 64 bits per (tile row, N-tile, c)
#ifdef WW_ABL_WRITELANE_EACH      // A/B: round 3's form, every v_writelane behind its own wait states
                    uint32_t word = 0u;
                    word = write_lane<0>(uint32_t(live0[0]), word);  word = write_lane<1>(uint32_t(live0[0] >> 32), word);
                    word = write_lane<2>(uint32_t(live0[1]), word);  word = write_lane<3>(uint32_t(live0[1] >> 32), word);
                    word = write_lane<4>(uint32_t(live0[2]), word);  word = write_lane<5>(uint32_t(live0[2] >> 32), word);
                    word = write_lane<6>(uint32_t(live0[3]), word);  word = write_lane<7>(uint32_t(live0[3] >> 32), word);
                    word = write_lane<8>(uint32_t(live1[0]), word);  word = write_lane<9>(uint32_t(live1[0] >> 32), word);
                    word = write_lane<10>(uint32_t(live1[1]), word); word = write_lane<11>(uint32_t(live1[1] >> 32), word);
                    word = write_lane<12>(uint32_t(live1[2]), word); word = write_lane<13>(uint32_t(live1[2] >> 32), word);
                    word = write_lane<14>(uint32_t(live1[3]), word); word = write_lane<15>(uint32_t(live1[3] >> 32), word);
#else
                    const uint32_t word = ballots_to_lanes(live0, live1);
#endif
                    {   // the store's address = a scalar base + 4 x the lane index, the index recomputed here (mbcnt): kept across the tile
                        // loop the compiler held `bits + lane` in a register pair it SPILLED at this kernel's 168 registers, and every
                        // store waited (s_waitcnt vmcnt(0)) on a scratch reload of it
                        uint32_t l16 = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
                        asm volatile("" : "+v"(l16));
                        if (l16 < 16u) {
                            const int64_t clip = int64_t(blockIdx.x) + int64_t(k) * gridDim.x;
                            uint32_t* wp = reinterpret_cast<uint32_t*>(bits) + ((((clip * kWTileRows + trow) * 4 + nt) * 2 + c) * 16);
                            wp[l16] = word;
                        }
                    }
                }
            }
            if constexpr (POOL) pool += tree8(pv);
            if (POOL && sq == kWPerGroup - 1) {               // this wave's last tile row of the clip
                float p2 = pool + __shfl_xor(pool, 16);
                p2 += __shfl_xor(p2, 32);
                float* rk = red + (k & 1) * 8 * 16;
                if (lane < 16) rk[wave * 16 + lane] = p2;
                // the last of the eight consumer waves to arrive writes the clip's pooled features
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                uint32_t old = 0u;
                if (lane == 0) old = __hip_atomic_fetch_add(&clip_done[k & 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                old = __builtin_amdgcn_readfirstlane(old);
                if (old + 1u == 8u * unsigned(k / 2 + 1)) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    const int64_t clip = int64_t(blockIdx.x) + int64_t(k) * gridDim.x;
                    // once per clip: the lane index and the scale are RECOMPUTED here (mbcnt; a scalar divide) instead of being kept in
                    // VGPRs across the MFMA loop, where at the 168-register cap they were spilled to scratch
                    const int ln = int(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
                    const float scale = 0.5f / float(kH * __builtin_amdgcn_readfirstlane(width));
                    out[clip * 64 + ln] = (rk[ln] + rk[64 + ln]) * scale;
                }
            }
        }
        CLK_END();
    }
    // an expired wait anywhere in this workgroup: poison everything it produced (pooled features, or the conv3 scale)
    __syncthreads();
    if (__hip_atomic_load(&wg_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) {
        const float nan = __uint_as_float(0x7fc00000u);
        for (int k = 0; k < my_clips; ++k) {
            const int64_t clip = int64_t(blockIdx.x) + int64_t(k) * gridDim.x;
            if constexpr (POOL) { if (tid < 64) out[clip * 64 + tid] = nan; }
            else { if (tid == 0) apow2[clip] = nan; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// cnn2x_kernel: the same 1-D Winograd F(2,3) conv1 + conv2 + ReLU + pool as cnn2w_kernel with the consumers on
// v_mfma_f32_32x32x16_f16 (round 4).  Why: cnn2w_kernel is bound by the SIMDs' vector-issue port, not by the matrix pipe
// (profiles/r04_k2_census.txt): a 16x16x32 MFMA holds the port for 8 of its 16 cycles, a 32x32x16 one for 8 of its 32 -- half the
// issue cost per flop -- and an N-tile of 32 channels halves the fragment reads from LDS (each activation fragment now feeds 32
// output channels).  The price is the register file: a wave with all four xi of a 32-channel N-tile would need 192 B-operand + 64
// accumulator VGPRs.  So the four xi are SPLIT OVER TWO WAVES:
//   wave A (xh = 0) accumulates M1, M0      wave B (xh = 1) accumulates M2, M3          (96 B-operand + 32 accumulator VGPRs each)
//   out[2t]   = M0 + (M1 + M2)  is finished by A, which needs M2 from B
//   out[2t+1] = (M1 - M2) - M3  is finished by B, which needs M1 from A
// Each wave runs the xi it gives away FIRST and stores that accumulator (4 KB) into the pair's exchange slot in LDS while the MFMAs of
// its second xi run; at the end of the tile row it reads the partner's 4 KB and does bias + 2 relu + pool for ITS output row.  The
// slots are one tile row deep (FULL / FREE counters per direction): partners stay within half a tile row of each other.
//   waves 0-7   CONSUMERS = (N-tile nt of 32 channels) x (xh) x (group g); wave = nt + 2 xh + 4 g; group g takes the tile rows of
//               producers 2g and 2g + 1 alternately, as in cnn2w_kernel; 12 fragment steps x 3 MFMAs per tile row and wave
//   waves 8-11  PRODUCERS, as in cnn2w_kernel
// Tile row in LDS: [4 planes pi = V1, V0, V2, V3][hi, lo][34 positions][32 ci] f16 = 64-byte records; the four 16-byte chunks of a
// record are XOR-swizzled with (position >> 2) & 3, which makes the 32x32x16 A-fragment reads (lane = (position, k half): 16 bytes at
// position * 64 + chunk * 16) conflict-free under the ds_read_b128 lane-group rule (checked by enumeration, scripts/proto/swz.py).
// ------------------------------------------------------------------------------------------------
constexpr int kXRec = 64;
constexpr int kXPlane = kRS * kXRec;                // 2,176: one (plane, half)
constexpr int kXTile = 8 * kXPlane;                 // 17,408 B per tile row
constexpr int kXNB = 2 * kWRing;
constexpr int kXExch = 4 * 2 * 4096;                // [pair = 2 g + nt][direction][16 registers x 64 lanes]
constexpr int kCXLds = kXNB * kXTile + 4 * kMelHPlane * 2 + 2 * 8 * 32 * 4 + kXExch;
static_assert(kCXLds + 256 <= 160 * 1024, "LDS budget of cnn2x_kernel");

// OUT 1: out = pooled [n][64].
template <int OUT>
__global__ __launch_bounds__(768, 3) void cnn2x_kernel(const float* __restrict__ mel, int n, int width,
                                                       const u32x4* __restrict__ w1H, const float* __restrict__ hs1,
                                                       const float* __restrict__ b1,
                                                       const u32x4* __restrict__ wX, const float* __restrict__ hs,
                                                       const float* __restrict__ b2, const float* __restrict__ rng,
                                                       float* __restrict__ out) {
    static_assert(OUT == 1, "only the pooled form is built");
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    char* act0 = ldsb;
    _Float16* melh0 = reinterpret_cast<_Float16*>(ldsb + kXNB * kXTile);      // 2 clips x (hi plane, lo plane) of [82][36] f16
    float* red = reinterpret_cast<float*>(melh0 + 4 * kMelHPlane);           // [clip parity][8 consumer waves][32]
    char* exch = reinterpret_cast<char*>(red + 2 * 8 * 32);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < 8;
    const int ptid = tid - 512;

    __shared__ uint32_t full_cnt[kXNB], free_cnt[kXNB], xfull[4][2], xfree[4][2], mel_done, xmax_done, clip_done[2], wg_bad;
    __shared__ float xmaxw[2][4];
    __shared__ float clip_par[2][2];
    if (tid < kXNB) { full_cnt[tid] = 0u; free_cnt[tid] = 0u; }
    if (tid < 8) { xfull[tid >> 1][tid & 1] = 0u; xfree[tid >> 1][tid & 1] = 0u; }
    if (tid == 0) { mel_done = 0u; xmax_done = 0u; clip_done[0] = 0u; clip_done[1] = 0u; wg_bad = 0u; }
    for (int i = tid; i < kCXLds / 4; i += 768) reinterpret_cast<uint32_t*>(ldsb)[i] = 0u;   // column halos / dead columns stay zero
    __syncthreads();

    const int my_clips = (n - int(blockIdx.x) + int(gridDim.x) - 1) / int(gridDim.x);
#ifdef WW_STAMPS
    unsigned long long cst[8] = {0, 0, 0, 0, 0, 0, 0, 0}, clast = 0;
    CSTAMP(7);
#endif

    if (!consumer) {
        // ================================================= producers =================================================
        const int pw = wave - 8;
        u32x4 w1h_r = w1H[lane], w1l_r = w1H[64 + lane];
        asm volatile("" : "+v"(w1h_r), "+v"(w1l_r));
        const half8 a1h = __builtin_bit_cast(half8, w1h_r), a1l = __builtin_bit_cast(half8, w1l_r);
        const GatherLanes glanes = gather_lanes(lane & 31, lane >> 5);
        __builtin_amdgcn_s_setprio(WW_WINO_PPRIO);
        const float rng_l1 = rng[0], rng_b1 = rng[1];
        const int s1_exp = -exp_of(hs1[0]);
        auto load_mel = [&](int k, int& e_out, int& a_out) {      // as in cnn2w_kernel
            const float* __restrict__ src = mel + (int64_t(blockIdx.x) + int64_t(k) * gridDim.x) * kH * width;
            _Float16* ph = melh0 + (k & 1) * 2 * kMelHPlane;
            const int xx = ptid & 31, y0 = ptid >> 5;
            const bool col_live = xx < width;
            const float* __restrict__ sp = src + y0 * width + xx;
            float v[10];
            float mx = 0.f;
#pragma unroll
            for (int t = 0; t < 10; ++t) {
                v[t] = col_live ? sp[8 * t * width] : 0.f;
                mx = fmaxf(mx, __builtin_fabsf(v[t]));
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
            if (lane == 0) xmaxw[k & 1][pw] = mx;
            flag_signal(&xmax_done);
            flag_wait(&xmax_done, 4u * unsigned(k + 1), &wg_bad);
            mx = fmaxf(fmaxf(xmaxw[k & 1][0], xmaxw[k & 1][1]), fmaxf(xmaxw[k & 1][2], xmaxw[k & 1][3]));
            const int e = clampi(exp_of(mx) - 14, -100, 113);
            const float bound1 = fmaf(mx, rng_l1, rng_b1);
            const int a = clampi(exp_of(bound1) - 12, -100, 100);
            if (tid == 512) { clip_par[k & 1][0] = pow2i(a); clip_par[k & 1][1] = bound1; }
            const float down = pow2i(-e);
            if (col_live) {
                _Float16* dh = ph + (y0 + 1) * kMelHRS + xx + 1;
#pragma unroll
                for (int t = 0; t < 10; ++t) {
                    const float vv = v[t] * down;
                    const _Float16 hi = static_cast<_Float16>(vv);
                    dh[8 * t * kMelHRS] = hi;
                    dh[8 * t * kMelHRS + kMelHPlane] = static_cast<_Float16>(vv - static_cast<float>(hi));
                }
            }
            e_out = e;
            a_out = a;
        };
        Conv1Scale cs;
        auto set_conv1_scale = [&](int e, int a) {
            const int c0 = 16 * (lane >> 5);
#pragma unroll
            for (int j = 0; j < 16; ++j) cs.binit[j] = ldexpf(b1[c0 + j], s1_exp - e);
            cs.sc = ldexpf(1.0f, e - a - s1_exp);
        };
        const int x = lane & 31, h = lane >> 5;
        const bool col_ok = x < width;
        // this lane's 16 channels 16 h .. 16 h + 15 are chunks 2 h and 2 h + 1 of position x + 1 (swizzled)
        const int wr0 = (x + 1) * kXRec + (((2 * h) ^ (((x + 1) >> 2) & 3)) << 4);
        auto store16 = [&](const f32x16& v, char* plane) {     // plane: the hi half of one of the four planes of a tile row
#pragma unroll
            for (int g8 = 0; g8 < 2; ++g8) {
                u32x4 vh, vl;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    uint32_t hh, ll;
                    split2(v[8 * g8 + 2 * d], v[8 * g8 + 2 * d + 1], hh, ll);
                    vh[d] = hh;
                    vl[d] = ll;
                }
                char* rec = plane + (wr0 ^ (g8 << 4));
                *reinterpret_cast<u32x4*>(rec) = vh;
                *reinterpret_cast<u32x4*>(rec + kXPlane) = vl;
            }
        };
        int e_nx = 0, a_nx = 0;
        if (my_clips > 0) { load_mel(0, e_nx, a_nx); flag_signal(&mel_done); }
        const int pgrp = pw >> 1, podd = pw & 1;
        f32x16 d0, d1, d2, d3;
        auto conv1_rows2 = [&](const _Float16* plane, int ya, f32x16& ra, f32x16& rb) {
            Conv1Row r0, r1;
            const bool va = unsigned(ya) < unsigned(kH), vb = unsigned(ya + 1) < unsigned(kH);
            conv1_row_gather(r0, plane, glanes, va ? ya : (ya < 0 ? 0 : kH - 1));
            conv1_row_gather(r1, plane, glanes, vb ? ya + 1 : (ya + 1 < 0 ? 0 : kH - 1));
            conv1_row_mfma(r0, a1h, a1l, cs);
            conv1_row_mfma(r1, a1h, a1l, cs);
            const float sa = va ? cs.sc : 0.f, sb = vb ? cs.sc : 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                ra[j] = relu2(r0.acc[j] * sa);
                rb[j] = relu2(r1.acc[j] * sb);
            }
        };
        for (int k = 0; k < my_clips; ++k) {
            flag_wait(&mel_done, 4u * unsigned(k + 1), &wg_bad);
            set_conv1_scale(e_nx, a_nx);
            if (k + 1 < my_clips) { load_mel(k + 1, e_nx, a_nx); flag_signal(&mel_done); }
            const _Float16* plane = melh0 + (k & 1) * 2 * kMelHPlane;
#pragma unroll 1
            for (int i = 0; i < kWPerProd; ++i) {
                const int t = kWPerProd * pw + i;
                const int q = k * kWPerGroup + 2 * i + podd;
                const int b = pgrp * kWRing + q % kWRing;
                CSTAMP(0);
#ifndef WW_ABL_NOPROD
                if (i == 0) conv1_rows2(plane, 2 * t - 1, d0, d1);
                else { d0 = d2; d1 = d3; }
                conv1_rows2(plane, 2 * t + 1, d2, d3);
#endif
#ifdef WW_STAMPS
                asm volatile("" :: "v"(d2[0]), "v"(d3[15]));
#endif
                CSTAMP(1);
#ifndef WW_ABL_X_NORING
                flag_wait(&free_cnt[b], 4u * unsigned(q / kWRing), &wg_bad);
#endif
                CSTAMP(2);
#ifndef WW_ABL_NOPROD
                if (col_ok) {
                    char* tile = act0 + b * kXTile;
                    store16(d1 + d2, tile);                      // plane 0 = V1
                    store16(d0 - d2, tile + 2 * kXPlane);        // plane 1 = V0
                    store16(d2 - d1, tile + 4 * kXPlane);        // plane 2 = V2
                    store16(d1 - d3, tile + 6 * kXPlane);        // plane 3 = V3
                }
#endif
                flag_signal(&full_cnt[b]);
                CSTAMP(3);
            }
        }
    } else {
        // ================================================= consumers =================================================
        const int nt = wave & 1, xh = (wave >> 1) & 1, grp = wave >> 2;
        const int pair = 2 * grp + nt;
        const int r = lane & 31, hh = lane >> 5;
        // B operands: the wave's two xi (first = the one it gives away: 1 | 2, second: 0 | 3), steps s = dx * 2 + c
        const int xi_f = xh ? 2 : 1, xi_s = xh ? 3 : 0;
        half8 bfh[6], bfl[6], bsh[6], bsl[6];
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            bfh[s] = __builtin_bit_cast(half8, wX[(((nt * 4 + xi_f) * 6 + s) * 2 + 0) * 64 + lane]);
            bfl[s] = __builtin_bit_cast(half8, wX[(((nt * 4 + xi_f) * 6 + s) * 2 + 1) * 64 + lane]);
            bsh[s] = __builtin_bit_cast(half8, wX[(((nt * 4 + xi_s) * 6 + s) * 2 + 0) * 64 + lane]);
            bsl[s] = __builtin_bit_cast(half8, wX[(((nt * 4 + xi_s) * 6 + s) * 2 + 1) * 64 + lane]);
        }
        const float bias = b2[32 * nt + r];
        const float descale = 0.5f * hs[32 * nt + r];
        float dsc = descale, pool = 0.f;
        const float sg = xh ? -1.f : 1.f;
        // fragment offsets inside a (plane, half): step (dx, c): position r + dx, chunk 2 c + hh, swizzled
        int fo[6];
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            const int dx = s >> 1, c = s & 1, p = r + dx;
            fo[s] = p * kXRec + (((2 * c + hh) ^ ((p >> 2) & 3)) << 4) + 4 * xh * kXPlane;     // the wave's planes 2 xh, 2 xh + 1
        }
        char* xw = exch + (pair * 2 + xh) * 4096 + lane * 16;             // what this wave gives
        const char* xr = exch + (pair * 2 + (xh ^ 1)) * 4096 + lane * 16;  // what it takes
        const int gsteps = my_clips * kWPerGroup;
        CLK_BEGIN();
        for (int q = 0; q < gsteps; ++q) {
            const int k = q / kWPerGroup, sq = q - k * kWPerGroup;
            const int b = grp * kWRing + q % kWRing;
            CSTAMP(0);
#ifndef WW_ABL_X_NORING     // timing-only ablation: consumers that never wait for a tile row
            flag_wait(&full_cnt[b], unsigned(q / kWRing) + 1u, &wg_bad);
#endif
            CSTAMP(1);
            if (sq == 0) {
                dsc = descale * clip_par[k & 1][0];
                pool = 0.f;
            }
            const char* tile = act0 + b * kXTile;
            f32x16 accf, accs;
#pragma unroll
            for (int j = 0; j < 16; ++j) { accf[j] = 0.f; accs[j] = 0.f; }
            // 12 fragment steps: it < 6 the first xi (plane 2 xh), then the second (plane 2 xh + 1)
            auto frag = [&](int it, int half) -> half8 {
                const int sel = it / 6, s = it % 6;
                return __builtin_bit_cast(half8, *reinterpret_cast<const u32x4*>(tile + fo[s] + (2 * sel + half) * kXPlane));
            };
#ifndef WW_X32_PF
#define WW_X32_PF 1
#endif
            constexpr int PF = WW_X32_PF, RING = PF + 1;
            half8 fh[RING], fl[RING];
#pragma unroll
            for (int i = 0; i < PF; ++i) { fh[i] = frag(i, 0); fl[i] = frag(i, 1); }
#ifdef WW_ABL_NOCONS           // timing-only ablation: producers alone (no MFMAs, no exchange)
#define WW_ABL_X_NOXCH
            constexpr int kIts = 0;
#else
            constexpr int kIts = 12;
#endif
#pragma unroll
            for (int it = 0; it < kIts; ++it) {
                if (it + PF < 12) { fh[(it + PF) % RING] = frag(it + PF, 0); fl[(it + PF) % RING] = frag(it + PF, 1); }
                __builtin_amdgcn_sched_barrier(0);
                const half8 ah = fh[it % RING], al = fl[it % RING];
                if (it < 6) {
                    accf = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bfh[it], accf, 0, 0, 0);
                    accf = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bfl[it], accf, 0, 0, 0);
                    accf = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bfh[it], accf, 0, 0, 0);
                } else {
                    accs = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bsh[it - 6], accs, 0, 0, 0);
                    accs = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bsl[it - 6], accs, 0, 0, 0);
                    accs = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bsh[it - 6], accs, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#ifndef WW_ABL_X_NOXCH      // timing-only ablation: no exchange at all (results are garbage)
                if (it == 7) {
                    // the first accumulator is complete: give it to the partner (who read the slot's previous content at the end of ITS
                    // previous tile row) while the second xi's MFMAs run
                    CSTAMP(2);
#ifndef WW_ABL_X_NOWAIT     // timing-only ablation: the exchange without its waits
                    flag_wait(&xfree[pair][xh], unsigned(q), &wg_bad);
#endif
                    CSTAMP(3);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x4 v = {accf[4 * i], accf[4 * i + 1], accf[4 * i + 2], accf[4 * i + 3]};
                        *reinterpret_cast<f32x4*>(xw + i * 1024) = v;
                    }
                    flag_signal(&xfull[pair][xh]);
                    CSTAMP(4);
                    __builtin_amdgcn_sched_barrier(0);
                }
#endif
            }
            CSTAMP(2);
#ifndef WW_ABL_X_NORING
            flag_signal(&free_cnt[b]);                       // the buffer is free as soon as its fragments are in the accumulators
#endif
            CSTAMP(5);
            // the partner's accumulator; then this wave's output row: A (xh 0): row 2t = M0 + (M1 + M2); B: row 2t + 1 = (M1 - M2) - M3
#if !defined(WW_ABL_X_NOWAIT) && !defined(WW_ABL_X_NOXCH)
            flag_wait(&xfull[pair][xh ^ 1], unsigned(q) + 1u, &wg_bad);
#endif
            CSTAMP(6);
            f32x16 oth;
#ifndef WW_ABL_X_NOXCH
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(xr + i * 1024);
                oth[4 * i] = v[0]; oth[4 * i + 1] = v[1]; oth[4 * i + 2] = v[2]; oth[4 * i + 3] = v[3];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            flag_signal(&xfree[pair][xh ^ 1]);
#else
#pragma unroll
            for (int j = 0; j < 16; ++j) oth[j] = bias;
#endif
            // one instruction stream for both: y = sg accs + (sg accf + oth), sg = +1 (A) | -1 (B); an fma by +-1 rounds once, like the add
            float pv[16];
#ifdef WW_ABL_X_NOFIN       // timing-only ablation: no output transform / bias / relu (one add per accumulator keeps the MFMAs alive)
#pragma unroll
            for (int j = 0; j < 16; ++j) pv[j] = 0.f;
            pv[0] = accs[0] + accf[5] + oth[3];
#else
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float y = fmaf(sg, accs[j], fmaf(sg, accf[j], oth[j]));
                pv[j] = relu2(fmaf(y, dsc, bias));
            }
#endif
            if (width != kW) {                       // wave-uniform; D: register j <-> column (j & 3) + 8 (j >> 2) + 4 hh
#pragma unroll
                for (int j = 0; j < 16; ++j) pv[j] = (j & 3) + 8 * (j >> 2) + 4 * hh < width ? pv[j] : 0.f;
            }
            pool += tree16(pv);
            if (sq == kWPerGroup - 1) {               // this wave's last tile row of the clip
                const float p2 = pool + __shfl_xor(pool, 32);
                float* rk = red + (k & 1) * 8 * 32;
                if (lane < 32) rk[wave * 32 + lane] = p2;
                // the last of the eight consumer waves to arrive writes the clip's pooled features
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                uint32_t old = 0u;
                if (lane == 0) old = __hip_atomic_fetch_add(&clip_done[k & 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                old = __builtin_amdgcn_readfirstlane(old);
                if (old + 1u == 8u * unsigned(k / 2 + 1)) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    const int64_t clip = int64_t(blockIdx.x) + int64_t(k) * gridDim.x;
                    const int ln = int(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
                    const float scale = 0.5f / float(kH * __builtin_amdgcn_readfirstlane(width));
                    // channel ln = 32 nt + r: waves nt + 2 xh + 4 g; fixed order (rows 2t, 2t+1 of group 0, then of group 1)
                    const float* w0 = rk + (ln >> 5) * 32 + (ln & 31);
                    out[clip * 64 + ln] = ((w0[0] + w0[2 * 32]) + (w0[4 * 32] + w0[6 * 32])) * scale;
                }
            }
        }
        CLK_END();
    }
#ifdef WW_STAMPS
    CSTAMP(7);
    if (lane == 0 && blockIdx.x == 7 && (wave == 1 || wave == 9))
        for (int i = 0; i < 8; ++i) atomicAdd(&g_cnn_stamps[i + (wave == 9 ? 8 : 0)], cst[i]);
#endif
    // an expired wait anywhere in this workgroup: poison everything it produced
    __syncthreads();
    if (__hip_atomic_load(&wg_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) {
        const float nan = __uint_as_float(0x7fc00000u);
        for (int k = 0; k < my_clips; ++k) {
            const int64_t clip = int64_t(blockIdx.x) + int64_t(k) * gridDim.x;
            if (tid < 64) out[clip * 64 + tid] = nan;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// cnn3w_kernel: conv3 (64 -> 128) + ReLU + pool of the 3-conv WakewordModel as the same 1-D Winograd F(2,3) along the rows.
// Input: relu(conv2) as float32 [row][column][64 channels] (cnn2w_kernel<false>).  8 waves = 8 N-tiles of 16 channels; a
// wave's 192 B-operand VGPRs (24 k-steps = (xi, dx, channel block) x hi/lo) stay resident.  Per tile row the 512 threads
// fetch the four input rows (one float4 = 4 channels of one column per thread and row), form V0..V3 in fp32, scale by the
// clip's 2^-a2, split and write the [4 xi][34][64 hi | 64 lo] tile (288-byte records: conflict-free fragment reads) -- the
// fetch is issued at the top of a step, transformed and written to the OTHER tile buffer behind the step's MFMAs.
// 96 instead of 144 MFMA triples per (tile row, N-tile) of the direct form.
// ------------------------------------------------------------------------------------------------
constexpr int kW3Rec = 288;
constexpr int kW3Plane = kRS * kW3Rec;              // 9,792
constexpr int kW3Buf = 4 * kW3Plane;                // 39,168
constexpr int kC3wLds = 2 * kW3Buf;

// BITS (training forward): also the ReLU mask of conv3, [relu(conv3) > 0], as the accumulator ballots themselves:
// bits3[n][40 tile rows][8 N-tiles][2 column halves][2 rows][4 j] x 64 bits (ww_train_h.hip: mask3_byte).
template <bool BITS>
__global__ __launch_bounds__(512, 2) void cnn3w_kernel(const float* __restrict__ mid, const float* __restrict__ apow2, int n, int width,
                                                       const u32x4* __restrict__ wH, const float* __restrict__ hs,
                                                       const float* __restrict__ b3, float* __restrict__ out, uint16_t* __restrict__ bits3) {
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int nt = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pi = lane & 15, kq = lane >> 4;

    half8 bh[24], bl[24];
#pragma unroll
    for (int ks = 0; ks < 24; ++ks) {
        bh[ks] = __builtin_bit_cast(half8, wH[((nt * 24 + ks) * 2 + 0) * 64 + lane]);
        bl[ks] = __builtin_bit_cast(half8, wH[((nt * 24 + ks) * 2 + 1) * 64 + lane]);
    }
    const float bias = b3[16 * nt + pi];
    const float descale = hs[16 * nt + pi];                 // this lane's output channel (the inputs are relu(conv2) itself, not 2*relu)
    for (int i = tid; i < kC3wLds / 4; i += 512) reinterpret_cast<uint32_t*>(ldsb)[i] = 0u;   // column halos stay zero
    __syncthreads();

    const int my_clips = (n - int(blockIdx.x) + int(gridDim.x) - 1) / int(gridDim.x);
    const int steps = my_clips * kWTileRows;
    const float inv_area = 1.0f / float(kH * width);
    const int lx = tid >> 4, lcg = tid & 15;                 // loader role: column, group of 4 channels

    float4 pre[4];
    float pre_ap = 1.f;
    auto fetch = [&](int g) {            // rows 2t-1 .. 2t+2 of step g (clip g / 40, tile row g % 40) -> registers
        const int k = g / kWTileRows, t = g - k * kWTileRows;
        const int64_t clip = int64_t(blockIdx.x) + int64_t(k) * gridDim.x;
        // (tried: a uniform row pointer + the thread's 32-bit offset, which removes the two scratch reloads of the spilled pointer pair at
        // the top of every step -- the register allocator then spills a WEIGHT fragment instead and reloads it inside the MFMA stream:
        // 3.22 vs 3.12 ms for the 3-conv stack, scripts/isa_scratch.py shows both)
        const float* src = mid + (clip * kH * kW + lx) * 64 + 4 * lcg;
        pre_ap = apow2[clip];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int y = 2 * t - 1 + i;
            const bool ok = y >= 0 && y < kH;
            const float4 v = *reinterpret_cast<const float4*>(src + int64_t(ok ? y : 0) * (kW * 64));
            pre[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stash = [&](int g) {            // registers -> transform -> LDS tile g & 1
        const float s = __uint_as_float(0x7f000000u - __float_as_uint(pre_ap));     // 2^-a2 (exact for a power of two; NaN stays NaN)
        char* dst = ldsb + (g & 1) * kW3Buf + (lx + 1) * kW3Rec + lcg * 8;
        auto put = [&](float4 v, char* rec) {
            uint32_t h0, l0, h1, l1;
            split2(v.x * s, v.y * s, h0, l0);
            split2(v.z * s, v.w * s, h1, l1);
            *reinterpret_cast<uint2*>(rec) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(rec + 128) = make_uint2(l0, l1);
        };
        const float4 d0 = pre[0], d1 = pre[1], d2 = pre[2], d3 = pre[3];
        put(make_float4(d0.x - d2.x, d0.y - d2.y, d0.z - d2.z, d0.w - d2.w), dst);
        put(make_float4(d1.x + d2.x, d1.y + d2.y, d1.z + d2.z, d1.w + d2.w), dst + kW3Plane);
        put(make_float4(d2.x - d1.x, d2.y - d1.y, d2.z - d1.z, d2.w - d1.w), dst + 2 * kW3Plane);
        put(make_float4(d1.x - d3.x, d1.y - d3.y, d1.z - d3.z, d1.w - d3.w), dst + 3 * kW3Plane);
    };

    if (steps > 0) { fetch(0); stash(0); }
    __syncthreads();
    float pool = 0.f, pool4 = 0.f, dsc = descale;     // pool4: partial over four tile rows (pooling order: see tree4 above)
    for (int g = 0; g < steps; ++g) {
        const int k = g / kWTileRows, t = g - k * kWTileRows;
        if (t == 0) {
            dsc = descale * apow2[int64_t(blockIdx.x) + int64_t(k) * gridDim.x];    // the tile holds V * 2^-a2
            pool = 0.f;
            pool4 = 0.f;
        }
        if (g + 1 < steps) fetch(g + 1);
        const char* ap = ldsb + (g & 1) * kW3Buf + pi * kW3Rec + kq * 16;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            f32x4 acc[4];
#pragma unroll
            for (int xi = 0; xi < 4; ++xi)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[xi][j] = 0.f;
            // 24 fragment steps it = (dx*2 + cb)*4 + xi: consecutive steps hit different accumulators
            auto frag = [&](int it, int half) -> half8 {
                const int dx = it >> 3, cb = (it >> 2) & 1, xi = it & 3;
                return __builtin_bit_cast(half8, *reinterpret_cast<const u32x4*>(ap + xi * kW3Plane + (16 * c + dx) * kW3Rec + half * 128 + cb * 64));
            };
            half8 fh[2], fl[2];
            fh[0] = frag(0, 0); fl[0] = frag(0, 1);
#pragma unroll
            for (int it = 0; it < 24; ++it) {
                if (it + 1 < 24) { fh[(it + 1) & 1] = frag(it + 1, 0); fl[(it + 1) & 1] = frag(it + 1, 1); }
                __builtin_amdgcn_sched_barrier(0);
                const half8 ah = fh[it & 1], al = fl[it & 1];
                const int dx = it >> 3, cb = (it >> 2) & 1, xi = it & 3;
                const int ks = (xi * 3 + dx) * 2 + cb;
                acc[xi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[ks], acc[xi], 0, 0, 0);
                acc[xi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[ks], acc[xi], 0, 0, 0);
                acc[xi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[ks], acc[xi], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            unsigned long long live0[4], live1[4];
            float pv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float m12 = acc[1][j] + acc[2][j], m1m2 = acc[1][j] - acc[2][j];
                const float y0 = acc[0][j] + m12, y1 = m1m2 - acc[3][j];
                const float v0 = relu2(fmaf(y0, dsc, bias)), v1 = relu2(fmaf(y1, dsc, bias));
                const bool col_live = width == kW || 16 * c + 4 * kq + j < width;
                pv[j] = col_live ? v0 + v1 : 0.f;
                if constexpr (BITS) {           // bit (16 kq + pi) of the ballot <-> channel 16 nt + pi at column 16 c + 4 kq + j
                    unsigned long long cm = 0ull;   // the column mask as scalar arithmetic (see cnn2w_kernel<2>)
#pragma unroll
                    for (int q = 0; q < 4; ++q) cm |= (width == kW || 16 * c + 4 * q + j < width) ? 0xFFFFull << (16 * q) : 0ull;
                    live0[j] = __builtin_amdgcn_ballot_w64(v0 > 0.f) & cm;
                    live1[j] = __builtin_amdgcn_ballot_w64(v1 > 0.f) & cm;
                }
            }
            if constexpr (BITS) pool += tree4(pv[0], pv[1], pv[2], pv[3]);     // training forward: at its 256-VGPR cap a second accumulator costs 6 spills
            else pool4 += tree4(pv[0], pv[1], pv[2], pv[3]);
            if constexpr (BITS) {
                // the eight ballots of (row r, register j) go out as they are: 64 contiguous bytes per (tile row, N-tile, column half) =
                // [r][j] x 64 bits, bit 16 kq + pi <-> column 16 c + 4 kq + j, channel 16 nt + pi.  v_writelane moves each half into
                // lane 2 (4 r + j) + half of one register: 16 scalar-to-lane moves and ONE 4-byte store by 16 lanes.
#if 1       // one v_writelane per wait-state pair here: the single block of cnn2w_kernel<2> (ballots_to_lanes) measured 2.5 % SLOWER in this
            // kernel (3-conv training step 6.00 vs 5.85 ms): sixteen scalar operands live at once at its 256-register cap
                uint32_t word = 0u;
                word = write_lane<0>(uint32_t(live0[0]), word);  word = write_lane<1>(uint32_t(live0[0] >> 32), word);
                word = write_lane<2>(uint32_t(live0[1]), word);  word = write_lane<3>(uint32_t(live0[1] >> 32), word);
                word = write_lane<4>(uint32_t(live0[2]), word);  word = write_lane<5>(uint32_t(live0[2] >> 32), word);
                word = write_lane<6>(uint32_t(live0[3]), word);  word = write_lane<7>(uint32_t(live0[3] >> 32), word);
                word = write_lane<8>(uint32_t(live1[0]), word);  word = write_lane<9>(uint32_t(live1[0] >> 32), word);
                word = write_lane<10>(uint32_t(live1[1]), word); word = write_lane<11>(uint32_t(live1[1] >> 32), word);
                word = write_lane<12>(uint32_t(live1[2]), word); word = write_lane<13>(uint32_t(live1[2] >> 32), word);
                word = write_lane<14>(uint32_t(live1[3]), word); word = write_lane<15>(uint32_t(live1[3] >> 32), word);
#else
                const uint32_t word = ballots_to_lanes(live0, live1);
#endif
                {   // scalar base + 4 x a lane index recomputed here (see cnn2w_kernel<2>: no register pair held, or spilled, across the loop)
                    uint32_t l16 = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
                    asm volatile("" : "+v"(l16));
                    if (l16 < 16u) {
                        const int64_t clip = int64_t(blockIdx.x) + int64_t(k) * gridDim.x;
                        uint32_t* wp = reinterpret_cast<uint32_t*>(bits3) + ((((clip * kWTileRows + t) * 8 + nt) * 2 + c) * 16);
                        wp[l16] = word;
                    }
                }
            }
            // the next tile row goes into the OTHER buffer (last read in step g-1) between the two column halves: its loads have
            // landed by now, and this wave's transform + stores run under its SIMD sibling's MFMAs instead of in front of the barrier
            if (c == 0 && g + 1 < steps) stash(g + 1);
        }
        if (!BITS && (t & 3) == 3) { pool += pool4; pool4 = 0.f; }      // kWTileRows = 40: the last partial is folded in before the reduction
        if (t == kWTileRows - 1) {
            float p2 = pool + __shfl_xor(pool, 16);
            p2 += __shfl_xor(p2, 32);
            const int64_t clip = int64_t(blockIdx.x) + int64_t(k) * gridDim.x;
            if (lane < 16) out[clip * 128 + 16 * nt + lane] = p2 * 0.5f * inv_area;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// conv3 (64->128) + ReLU + pool for the 3-conv WakewordModel.  512 threads: wave = (K-half kh, N-tile nt);
// the two K-halves of an N-tile are summed through LDS before bias/ReLU.  Band = 4 output rows.
// in = relu(conv2) as [n][80][64][32]; out = pooled [n][128].
// ------------------------------------------------------------------------------------------------
constexpr int kC3Rows = 6;
constexpr int kC3ActFloats = 64 * kC3Rows * kRS;     // 13,056
constexpr int kC3XchFloats = 4 * 4 * 16 * 64;        // [nt][r][j][lane] 16,384
constexpr int kC3LdsFloats = kC3ActFloats + kC3XchFloats;

// STORE (training forward): relu(conv3) is also kept, as [n][80][128][32] (row, channel, column; zero beyond `width`).
template <bool STORE>
__global__ __launch_bounds__(512, 2) void cnn3_kernel(const float* __restrict__ in, int n, int width,
                                                      const float* __restrict__ wB, const float* __restrict__ b3,
                                                      float* __restrict__ out, float* __restrict__ mid3) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* act = lds;                        // [64][6][34]
    float* xch = act + kC3ActFloats;         // [4][4][16][64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = wave & 3, kh = wave >> 2;
    const int x = lane & 31, h = lane >> 5;

    float wb[144];
#pragma unroll
    for (int i = 0; i < 144; ++i) wb[i] = wB[(nt * 288 + kh * 144 + i) * 64 + lane];
    const float bias = b3[32 * nt + x];

    zero_lds(lds, kC3ActFloats, tid, 512);
    const float* ap = act + ((32 * kh + h) * kC3Rows) * kRS + x;
    const float inv_area = 1.0f / float(kH * width);

    for (int clip = blockIdx.x; clip < n; clip += gridDim.x) {
        float pool = 0.f;
        for (int band = 0; band < kH / 4; ++band) {
            const int y0 = band * 4;
            __syncthreads();   // previous band's reads of act / xch retired
            // tile rows y0-1 .. y0+4; one image row = 64 channels x 32 columns contiguous in `in`
#pragma unroll 1
            for (int q = 0; q < kC3Rows; ++q) {
                const int y = y0 - 1 + q;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (y >= 0 && y < kH)
                    v = *reinterpret_cast<const float4*>(in + (int64_t(clip) * kH + y) * (64 * kW) + tid * 4);
                const int ci = tid >> 3, c0 = (tid & 7) * 4;
                float* d = act + (ci * kC3Rows + q) * kRS + c0 + 1;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
            __syncthreads();
            f32x16 acc[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[r][j] = 0.f;
            mfma_rows4<kC3Rows>(ap, wb, acc);
            if (kh == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 16; ++j) xch[((nt * 4 + r) * 16 + j) * 64 + lane] = acc[r][j];
            }
            __syncthreads();
            if (kh == 0) {
                float rows[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float vs[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int col = (j & 3) + 8 * (j >> 2) + 4 * h;
                        const float v = relu(acc[r][j] + xch[((nt * 4 + r) * 16 + j) * 64 + lane] + bias);
                        vs[j] = (col < width) ? v : 0.f;
                    }
                    rows[r] = tree16(vs);
                    if constexpr (STORE) {
                        float* dst = mid3 + ((int64_t(clip) * kH + y0 + r) * 128 + 32 * nt + x) * kW + 4 * h;
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            *reinterpret_cast<float4*>(dst + 8 * g) = make_float4(vs[4 * g], vs[4 * g + 1], vs[4 * g + 2], vs[4 * g + 3]);
                    }
                }
                pool += tree4(rows[0], rows[1], rows[2], rows[3]);
            }
        }
        if (kh == 0) {
            pool += __shfl_xor(pool, 32);
            if (lane < 32) out[int64_t(clip) * 128 + 32 * nt + lane] = pool * inv_area;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// conv3 (64 -> 128) + ReLU + pool, split precision (3-conv WakewordModel).  Input = relu(conv2) as written by
// cnn2h16_kernel<false>: f16 records [row][column][64 ci hi | 64 ci lo].  8 waves = 8 N-tiles of 16 channels; a wave's
// 144 B-operand VGPRs (18 k-steps x hi/lo) stay resident; band = 4 output rows; the 6-row tile (records padded to 288
// bytes: conflict-free 16x16x32 fragment reads) is double-buffered and the next band's tile is fetched into registers
// at the top of a step and written to LDS behind the MFMAs (issue early / write late).
// ------------------------------------------------------------------------------------------------
constexpr int kRec3 = 288;
constexpr int kT3Rows = 6;
constexpr int kT3Bytes = kT3Rows * kRS * kRec3;              // 58,752
constexpr int kC3hLds = 2 * kT3Bytes;

__global__ __launch_bounds__(512, 2) void cnn3h_kernel(const _Float16* __restrict__ in, const float* __restrict__ apow2, int n,
                                                       int width, const u32x4* __restrict__ wH, const float* __restrict__ hs,
                                                       const float* __restrict__ b3, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int nt = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pi = lane & 15, kq = lane >> 4;

    half8 bh[18], bl[18];
#pragma unroll
    for (int ks = 0; ks < 18; ++ks) {
        bh[ks] = __builtin_bit_cast(half8, wH[((nt * 18 + ks) * 2 + 0) * 64 + lane]);
        bl[ks] = __builtin_bit_cast(half8, wH[((nt * 18 + ks) * 2 + 1) * 64 + lane]);
    }
    const float bias = b3[16 * nt + pi];
    const float descale = hs[16 * nt + pi];          // this lane's output channel
    for (int i = tid; i < kC3hLds / 4; i += 512) reinterpret_cast<uint32_t*>(ldsb)[i] = 0u;
    __syncthreads();

    const int my_clips = (n - int(blockIdx.x) + int(gridDim.x) - 1) / int(gridDim.x);
    const int bands = kH / 4, steps = my_clips * bands;
    const float half_inv_area = 0.5f / float(kH * width);
    const int t_pos = tid >> 4, t_piece = tid & 15;                   // tile fetch role: (column, 16-byte piece of its record)

    u32x4 pre[kT3Rows];
    auto fetch = [&](int g) {            // tile of step g (clip g / 20, band g % 20) -> registers
        const int k = g / bands, band = g - k * bands;
        const int64_t clip = int64_t(blockIdx.x) + int64_t(k) * gridDim.x;
        const char* src = reinterpret_cast<const char*>(in) + (clip * kH * kW) * 256 + tid * 16;
#pragma unroll
        for (int q = 0; q < kT3Rows; ++q) {
            const int y = band * 4 - 1 + q;
            const bool ok = y >= 0 && y < kH;
            const u32x4 v = *reinterpret_cast<const u32x4*>(src + int64_t(ok ? y : 0) * (kW * 256));
            pre[q] = ok ? v : u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto stash = [&](int g) {            // registers -> LDS tile g & 1
        char* dst = ldsb + (g & 1) * kT3Bytes + (t_pos + 1) * kRec3 + t_piece * 16;
#pragma unroll
        for (int q = 0; q < kT3Rows; ++q) *reinterpret_cast<u32x4*>(dst + q * kRS * kRec3) = pre[q];
    };

    if (steps > 0) { fetch(0); stash(0); }
    __syncthreads();
    float pool = 0.f, dsc = descale;
    for (int g = 0; g < steps; ++g) {
        const int k = g / bands, band = g - k * bands;
        if (band == 0) dsc = descale * apow2[int64_t(blockIdx.x) + int64_t(k) * gridDim.x];   // the records hold relu(conv2) * 2^-a2
        if (g + 1 < steps) fetch(g + 1);
        const char* ap = ldsb + (g & 1) * kT3Bytes + pi * kRec3 + kq * 16;
        f32x4 acc[4][2];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[r][c][j] = 0.f;
        // 72 fragment steps it = ((cb*3 + dx)*6 + q)*2 + ch
        auto frag = [&](int it, int half) -> half8 {
            const int cb = it / 36, dx = (it / 12) % 3, q = (it / 2) % 6, ch = it & 1;
            return __builtin_bit_cast(half8, *reinterpret_cast<const u32x4*>(ap + (q * kRS + 16 * ch + dx) * kRec3 + half * 128 + cb * 64));
        };
        // fragment ring two steps ahead of the MFMAs, pinned with scheduling barriers (see cnn2h16_kernel)
        constexpr int PF = 2, RING = PF + 1;
        half8 fh[RING], fl[RING];
#pragma unroll
        for (int i = 0; i < PF; ++i) { fh[i] = frag(i, 0); fl[i] = frag(i, 1); }
#pragma unroll
        for (int it = 0; it < 72; ++it) {
            if (it + PF < 72) { fh[(it + PF) % RING] = frag(it + PF, 0); fl[(it + PF) % RING] = frag(it + PF, 1); }
            __builtin_amdgcn_sched_barrier(0);
            const half8 ah = fh[it % RING], al = fl[it % RING];
            const int cb = it / 36, dx = (it / 12) % 3, q = (it / 2) % 6, ch = it & 1;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int r = q - dy;
                if (r < 0 || r > 3) continue;
                const int ks = (cb * 3 + dx) * 3 + dy;
                acc[r][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[ks], acc[r][ch], 0, 0, 0);
                acc[r][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[ks], acc[r][ch], 0, 0, 0);
                acc[r][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[ks], acc[r][ch], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (band == 0) pool = 0.f;
        {
            float rows[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float vs[8];
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = relu2(fmaf(acc[r][c][j], dsc, bias));
                        vs[4 * c + j] = (16 * c + 4 * kq + j < width) ? v : 0.f;
                    }
                rows[r] = tree8(vs);
            }
            pool += tree4(rows[0], rows[1], rows[2], rows[3]);
        }
        if (band == bands - 1) {
            float p2 = pool + __shfl_xor(pool, 16);
            p2 += __shfl_xor(p2, 32);
            const int64_t clip = int64_t(blockIdx.x) + int64_t(k) * gridDim.x;
            if (lane < 16) out[clip * 128 + 16 * nt + lane] = p2 * half_inv_area;
        }
        if (g + 1 < steps) stash(g + 1);     // the other buffer: last read in step g-1, retired by the barrier below
        __syncthreads();
    }
}

int sync_timeouts(unsigned int* count) {
    WW_HIP(hipMemcpyFromSymbol(count, HIP_SYMBOL(g_sync_timeouts), sizeof(unsigned int)));
    return WW_OK;
}

// relu(conv2) of the 3-conv model: 655,360 bytes per clip in either arithmetic, + one float per clip (f16x3: 2^a2)
static int64_t mid_bytes(int64_t n) { return n * int64_t(kH) * 64 * kW * int64_t(sizeof(float)); }
int64_t cnn_scratch_bytes(int64_t n, int n_conv) {
    return n_conv == 3 ? mid_bytes(n) + ((n * int64_t(sizeof(float)) + 255) & ~int64_t(255)) : 0;
}

// > 64 KiB of dynamic LDS needs an opt-in per kernel, once per device
static int opt_in_lds() {
    static std::mutex mu;
    static bool done[64] = {};
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    WW_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(WW_EINVAL, "device ordinal out of range");
    if (done[dev]) return WW_OK;
    WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cnn2h16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kC2h16Lds));
    WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cnn2h16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kC2h16Lds));
    WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cnn2w_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kCWLds));
    WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cnn2w_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kCWLds));
    WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cnn2w_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, kCWLds));
    WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cnn3w_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kC3wLds));
    WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cnn3w_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kC3wLds));
    WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cnn2w_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, kCWLds));
    WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cnn2x_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, kCXLds));
    WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cnn3h_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kC3hLds));
    WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cnn3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               int(sizeof(float) * kC3LdsFloats)));
    WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cnn3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               int(sizeof(float) * kC3LdsFloats)));
    done[dev] = true;
    return WW_OK;
}

// training forward of the 2-conv model in split precision (ww_train.hip): the inference kernel with the ReLU mask as a second output.
// `packed`: a packed image whose conv1 / conv2-Winograd / range entries were written on the device from the live parameters.
int launch_cnn2w_pool_bits(const float* mel, int64_t n, int width, const float* packed, float* pooled, uint32_t* bits, uint32_t* bits1,
                           hipStream_t stream) {
    if (n == 0) return WW_OK;
    const PackedLayout L = packed_layout(2);
    if (int rc = opt_in_lds()) return rc;
    const int cus = device_cu_count();
    hipLaunchKernelGGL(cnn2w_kernel<2>, dim3(int(n < cus ? n : cus)), dim3(768), kCWLds, stream, mel, int(n), width,
                       reinterpret_cast<const u32x4*>(packed + L.conv1_h), packed + L.conv1_hs, packed + L.conv1_b,
                       reinterpret_cast<const u32x4*>(packed + L.conv2_hw), packed + L.conv2_hws, packed + L.conv2_b, packed + L.range, pooled,
                       static_cast<float*>(nullptr), reinterpret_cast<uint16_t*>(bits), reinterpret_cast<uint16_t*>(bits1));
    WW_HIP(hipGetLastError());
    return WW_OK;
}

// training forward of the 3-conv model in split precision: the inference kernels; relu(conv2) stays in `mid2` as float32
// [n][80 rows][32 columns][64 channels], the ReLU mask of conv3 and the sign bits of conv1 as bit images.
int launch_cnn3w_pool_bits(const float* mel, int64_t n, int width, const float* packed, float* mid2, float* apow2, float* pooled,
                           uint32_t* bits3, uint32_t* bits1, hipStream_t stream) {
    if (n == 0) return WW_OK;
    const PackedLayout L = packed_layout(3);
    if (int rc = opt_in_lds()) return rc;
    const int cus = device_cu_count();
    const int grid1 = int(n < cus ? n : cus);
    hipLaunchKernelGGL(cnn2w_kernel<3>, dim3(grid1), dim3(768), kCWLds, stream, mel, int(n), width,
                       reinterpret_cast<const u32x4*>(packed + L.conv1_h), packed + L.conv1_hs, packed + L.conv1_b,
                       reinterpret_cast<const u32x4*>(packed + L.conv2_hw), packed + L.conv2_hws, packed + L.conv2_b, packed + L.range, mid2,
                       apow2, static_cast<uint16_t*>(nullptr), reinterpret_cast<uint16_t*>(bits1));
    WW_HIP(hipGetLastError());
    hipLaunchKernelGGL(cnn3w_kernel<true>, dim3(grid1), dim3(512), kC3wLds, stream, static_cast<const float*>(mid2),
                       static_cast<const float*>(apow2), int(n), width, reinterpret_cast<const u32x4*>(packed + L.conv3_hw),
                       packed + L.conv3_hws, packed + L.conv3_b, pooled, reinterpret_cast<uint16_t*>(bits3));
    WW_HIP(hipGetLastError());
    return WW_OK;
}

// training forward (ww_train.hip): conv1 + conv2 + ReLU in exact fp32, relu(conv2) kept as [n][80][64][32]
int launch_cnn2_f32_mid(const float* mel, int64_t n, int width, const float* w1, const float* b1, const float* wB, const float* b2, float* mid,
                        hipStream_t stream) {
    if (n == 0) return WW_OK;
    const int64_t cus = device_cu_count();
    const int grid2 = int(n < 2 * cus ? n : 2 * cus);
    hipLaunchKernelGGL(cnn2_kernel<false>, dim3(grid2), dim3(256), sizeof(float) * kC2LdsFloats, stream, mel, int(n), width, w1, b1, wB, b2, mid);
    WW_HIP(hipGetLastError());
    return WW_OK;
}

// training forward of the 3-conv model: conv3 + ReLU + pool in exact fp32 from relu(conv2) [n][80][64][32], keeping relu(conv3)
int launch_cnn3_f32_store(const float* mid2, int64_t n, int width, const float* wB, const float* b3, float* pooled, float* mid3,
                          hipStream_t stream) {
    if (n == 0) return WW_OK;
    if (int rc = opt_in_lds()) return rc;
    const int64_t cus = device_cu_count();
    hipLaunchKernelGGL(cnn3_kernel<true>, dim3(int(n < cus ? n : cus)), dim3(512), sizeof(float) * kC3LdsFloats, stream, mid2, int(n), width, wB, b3,
                       pooled, mid3);
    WW_HIP(hipGetLastError());
    return WW_OK;
}

// Experiment knob (WW_CNN3_SUBBATCH, read once): clips per conv2 -> conv3 sub-batch of the 3-conv model's Winograd path.
static int64_t cnn3_subbatch() {
    static const int64_t v = [] {
        const char* e = getenv("WW_CNN3_SUBBATCH");
        const long long x = e ? atoll(e) : 0;
        return int64_t(x > 0 ? x : 0);
    }();
    return v;
}

// Which kernel runs conv2 of the 2-conv model under WW_CONV_MATH_F16X3: cnn2w_kernel (v_mfma_f32_16x16x32_f16, the shipped form) or, with
// WW_K2_FORM=x32 in the environment (read once; experiments and tests/test_gpu_guards.py), cnn2x_kernel (v_mfma_f32_32x32x16_f16: parity-
// correct, 20 % slower on MI355X -- DESIGN.md section 4, K2).  -DWW_K2_X32=1 makes x32 the build's default.
#ifndef WW_K2_X32
#define WW_K2_X32 0
#endif
static bool k2_form_x32() {
    static const bool v = [] {
        const char* e = getenv("WW_K2_FORM");
        if (!e || !*e) return bool(WW_K2_X32);
        return e[0] == 'x';
    }();
    return v;
}

int launch_cnn_pool(const float* mel, int64_t n, int width, const float* packed, int n_conv, void* scratch,
                    float* pooled, hipStream_t stream) {
    if (n == 0) return WW_OK;
    if (n_conv == 3 && !scratch) return fail(WW_EINVAL, "n_conv == 3 needs ww_cnn_scratch_bytes() of scratch");
    const PackedLayout L = packed_layout(n_conv);
    if (int rc = opt_in_lds()) return rc;
    const int cus = device_cu_count();
    const int grid1 = int(n < cus ? n : cus);                  // persistent: one workgroup per CU
    const int cmath = conv_math_mode();                        // read once per launch (thread override, else the process default)
    if (cmath != 0) {                                          // f16x3
        const u32x4* w1h = reinterpret_cast<const u32x4*>(packed + L.conv1_h);
        const u32x4* w2h = reinterpret_cast<const u32x4*>(packed + L.conv2_h16);
        if (cmath == 1) {                                      // conv2 (and conv3) as 1-D Winograd
            const u32x4* w2w = reinterpret_cast<const u32x4*>(packed + L.conv2_hw);
            if (n_conv == 2 && k2_form_x32()) {
                hipLaunchKernelGGL(cnn2x_kernel<1>, dim3(grid1), dim3(768), kCXLds, stream, mel, int(n), width, w1h,
                                   packed + L.conv1_hs, packed + L.conv1_b, reinterpret_cast<const u32x4*>(packed + L.conv2_hx),
                                   packed + L.conv2_hws, packed + L.conv2_b, packed + L.range, pooled);
                WW_HIP(hipGetLastError());
                return WW_OK;
            }
            if (n_conv == 2) {
                hipLaunchKernelGGL(cnn2w_kernel<true>, dim3(grid1), dim3(768), kCWLds, stream, mel, int(n), width, w1h,
                                   packed + L.conv1_hs, packed + L.conv1_b, w2w, packed + L.conv2_hws, packed + L.conv2_b,
                                   packed + L.range, pooled, static_cast<float*>(nullptr), static_cast<uint16_t*>(nullptr),
                                   static_cast<uint16_t*>(nullptr));
                WW_HIP(hipGetLastError());
                return WW_OK;
            }
            float* apw = reinterpret_cast<float*>(static_cast<char*>(scratch) + mid_bytes(n));
            // conv2 -> conv3 over sub-batches that reuse the FRONT of the scratch buffer: the 655 KB per clip of relu(conv2) are then
            // written and read back inside the 256 MB Infinity Cache instead of going through HBM (cnn3_subbatch() clips; 0 = one pass)
            const int64_t sub = cnn3_subbatch() > 0 ? cnn3_subbatch() : n;
            for (int64_t s0 = 0; s0 < n; s0 += sub) {
                const int64_t cnt = n - s0 < sub ? n - s0 : sub;
                const int g = int(cnt < cus ? cnt : cus);
                hipLaunchKernelGGL(cnn2w_kernel<false>, dim3(g), dim3(768), kCWLds, stream, mel + s0 * 80 * width, int(cnt), width, w1h,
                                   packed + L.conv1_hs, packed + L.conv1_b, w2w, packed + L.conv2_hws, packed + L.conv2_b,
                                   packed + L.range, static_cast<float*>(scratch), apw, static_cast<uint16_t*>(nullptr), static_cast<uint16_t*>(nullptr));
                WW_HIP(hipGetLastError());
                hipLaunchKernelGGL(cnn3w_kernel<false>, dim3(g), dim3(512), kC3wLds, stream, static_cast<const float*>(scratch),
                                   static_cast<const float*>(apw), int(cnt), width, reinterpret_cast<const u32x4*>(packed + L.conv3_hw),
                                   packed + L.conv3_hws, packed + L.conv3_b, pooled + s0 * 128, static_cast<uint16_t*>(nullptr));
                WW_HIP(hipGetLastError());
            }
            return WW_OK;
        }
        if (n_conv == 2) {
            hipLaunchKernelGGL(cnn2h16_kernel<true>, dim3(grid1), dim3(768), kC2h16Lds, stream, mel, int(n), width, w1h,
                               packed + L.conv1_hs, packed + L.conv1_b, w2h, packed + L.conv2_hs, packed + L.conv2_b,
                               packed + L.range, pooled, static_cast<float*>(nullptr));
            WW_HIP(hipGetLastError());
            return WW_OK;
        }
        float* apow2 = reinterpret_cast<float*>(static_cast<char*>(scratch) + mid_bytes(n));
        hipLaunchKernelGGL(cnn2h16_kernel<false>, dim3(grid1), dim3(768), kC2h16Lds, stream, mel, int(n), width, w1h,
                           packed + L.conv1_hs, packed + L.conv1_b, w2h, packed + L.conv2_hs, packed + L.conv2_b,
                           packed + L.range, static_cast<float*>(scratch), apow2);
        WW_HIP(hipGetLastError());
        hipLaunchKernelGGL(cnn3h_kernel, dim3(grid1), dim3(512), kC3hLds, stream, static_cast<const _Float16*>(scratch),
                           static_cast<const float*>(apow2), int(n), width, reinterpret_cast<const u32x4*>(packed + L.conv3_h),
                           packed + L.conv3_hs, packed + L.conv3_b, pooled);
        WW_HIP(hipGetLastError());
        return WW_OK;
    }
    // exact f32
    const int grid2 = int(n < 2 * int64_t(cus) ? n : 2 * int64_t(cus));   // two 4-wave workgroups per CU
    const size_t lds2 = sizeof(float) * kC2LdsFloats;
    if (n_conv == 2) {
        hipLaunchKernelGGL(cnn2_kernel<true>, dim3(grid2), dim3(256), lds2, stream, mel, int(n), width,
                           packed + L.conv1_w, packed + L.conv1_b, packed + L.conv2_w, packed + L.conv2_b, pooled);
        WW_HIP(hipGetLastError());
        return WW_OK;
    }
    float* mid = static_cast<float*>(scratch);
    hipLaunchKernelGGL(cnn2_kernel<false>, dim3(grid2), dim3(256), lds2, stream, mel, int(n), width,
                       packed + L.conv1_w, packed + L.conv1_b, packed + L.conv2_w, packed + L.conv2_b, mid);
    WW_HIP(hipGetLastError());
    hipLaunchKernelGGL(cnn3_kernel<false>, dim3(grid1), dim3(512), sizeof(float) * kC3LdsFloats, stream, mid, int(n), width,
                       packed + L.conv3_w, packed + L.conv3_b, pooled, static_cast<float*>(nullptr));
    WW_HIP(hipGetLastError());
    return WW_OK;
}

}  // namespace ww
