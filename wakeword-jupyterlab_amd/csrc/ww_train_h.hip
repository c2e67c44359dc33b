// Split-precision (f16 matrix-core) training kernels of the conv stack, both models: device-side weight
// packing for the forward, and the weight / data gradients of conv2 (2-conv SimpleWakewordModel; dense form for the 3-conv model) and
// conv3 (3-conv WakewordModel) -- the kernels that took 85 % of the exact-fp32 training step (ww_train.hip).
//
// What is computed is the same as in ww_train.hip (/root/reference/wakeword_training/train_wakeword.py:109-115,
// wakeword_training_script.py:250-258: loss.backward()).  For the LAST conv of a model (its ReLU feeds only the global average pool)
//     dW[co][ci][dy][dx] = sum_b gp[b,co] * S[b][co][ci][dy][dx],     S = sum_{y,x} mask[b,co,y,x] * a[b,ci,y+dy-1,x+dx-1]
//     da[b,ci,y,x]       = sum_{co,dy,dx} mask[b,co,y-dy+1,x-dx+1] * (gp[b,co] * W[co][ci][dy][dx])
// with mask = [relu(conv) > 0] and gp = d loss / d pooled / (80 T): the gradient is rank one, so ONE operand of either product is a
// 0/1 matrix -- exact in f16.  The other operand (the input activations, or gp * W rebuilt per clip) is carried as two f16 halves
// (x * 2^-e = hi + lo), every product block is TWO v_mfma_f32_16x16x32_f16 (mask * hi + mask * lo), all products are exact in the fp32
// accumulator and gp enters in fp32.  Below the last conv the gradient is dense: both operands as two halves, three MFMAs per block.
// The results differ from the exact-fp32 kernels by the 2^-24 of the splits and by the accumulation order (the f16 MFMA rounds the exact
// 32-term sum once: scripts/ubench/mfma_round.hip).  16x the matrix rate of v_mfma_f32_32x32x2_f32 at 2-3 instructions per product.
//
// The masks travel as BITS written by the forward kernels: maskbits[b][row][col] = COUT bits, bits1[b][row][col] = conv1's 32 sign bits.
#include <mutex>

#include "ww_conv1.h"
#include "ww_internal.h"

namespace ww {

using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

constexpr int kTH = WW_N_MELS, kTW = 32;


// Byte `cb` (channels 8 cb .. 8 cb + 7) of position (row, col) in conv3's mask image as cnn3w_kernel<true> writes it (its accumulator
// ballots: [40 tile rows][8 N-tiles][2 column halves][2 rows][4 j] x 64 bits, bit 16 kq + pi <-> column 16 c + 4 kq + j, channel 16 nt + pi)
__device__ __forceinline__ int64_t mask3_byte(int64_t clip, int row, int col, int cb) {
    const int t = row >> 1, r = row & 1, nt = cb >> 1, c = col >> 4, kq = (col >> 2) & 3, j = col & 3;
    return ((((((clip * (kTH / 2) + t) * 8 + nt) * 2 + c) * 2 + r) * 4 + j) * 8) + 2 * kq + (cb & 1);
}
// the same for conv2's mask image (cnn2w_kernel<2>: four N-tiles)
__device__ __forceinline__ int64_t mask2_byte(int64_t clip, int row, int col, int cb) {
    const int t = row >> 1, r = row & 1, nt = cb >> 1, c = col >> 4, kq = (col >> 2) & 3, j = col & 3;
    return ((((((clip * (kTH / 2) + t) * 4 + nt) * 2 + c) * 2 + r) * 4 + j) * 8) + 2 * kq + (cb & 1);
}
// ds_read_b64_tr_b16: within a group of 16 lanes, lane 4q + p supplies the address of row q, 16-bit columns 4p .. 4p+3 of a 4 x 16 block;
// lane i receives column i, row q in element q (scripts/ubench/tr_read.hip).  EXEC must be all ones.
__device__ __forceinline__ fp16x4 lds_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(p));
}
__device__ __forceinline__ half8 cat8(fp16x4 a, fp16x4 b) {
    const u32x2 ua = __builtin_bit_cast(u32x2, a), ub = __builtin_bit_cast(u32x2, b);
    const u32x4 v = {ua[0], ua[1], ub[0], ub[1]};
    return __builtin_bit_cast(half8, v);
}

// ------------------------------------------------------------------------------------------------
// The split-precision images of conv1 and conv2 (1-D Winograd) written ON THE DEVICE from the live torch-layout parameters: the
// weights change with every optimiser step, so the host packers of ww_tables.cpp (pack_conv1_f16x3, pack_conv_wino_f16x3, the
// range bounds) are restated here value for value (same double arithmetic, same exponents, same operand order) -- a test
// compares the two images bit for bit.  img: a packed image (packed_layout(2)); only the entries named below are written.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int scale_exp_dev(float wmax) {
    if (!(wmax > 0.f) || !__builtin_isfinite(wmax)) return 0;
    int e = 0;
    (void)frexpf(wmax, &e);                     // wmax = f * 2^e, f in [0.5, 1)
    const int S = 13 - e;                       // wmax * 2^S in [2^12, 2^13)
    return S > 120 ? 120 : (S < -100 ? -100 : S);
}
__device__ __forceinline__ void split_f16_dev(double v, uint16_t& hb, uint16_t& lb) {
    const float vf = float(v);
    const _Float16 hi = static_cast<_Float16>(vf);
    const _Float16 lo = static_cast<_Float16>(vf - static_cast<float>(hi));
    hb = __builtin_bit_cast(uint16_t, hi);
    lb = __builtin_bit_cast(uint16_t, lo);
}
struct PackOffsets { int64_t conv1_h, conv1_hs, conv1_b, conv2_hw, conv2_hws, conv2_b, range; };

// one workgroup of 256 threads: conv1 operand + scale, both biases, range[0], [1], [3]; range[2] = 0 for the conv2 kernel's atomic max
__global__ __launch_bounds__(256) void pack_conv1_h_dev_kernel(const float* __restrict__ w1, const float* __restrict__ b1,
                                                               const float* __restrict__ b2, float* __restrict__ img, PackOffsets o) {
    __shared__ float red[256];
    __shared__ double redd[32];
    const int tid = threadIdx.x;
    float m = 0.f;
    for (int i = tid; i < 32 * 9; i += 256) m = fmaxf(m, __builtin_fabsf(w1[i]));
    red[tid] = m;
    if (tid < 32) {
        double s = 0.0;
        for (int t = 0; t < 9; ++t) s += fabs(double(w1[tid * 9 + t]));
        redd[tid] = s;
    }
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) red[tid] = fmaxf(red[tid], red[tid + off]);
        __syncthreads();
    }
    const int S = scale_exp_dev(red[0]);
    __syncthreads();
    if (tid < 4) img[o.conv1_hs + tid] = tid == 0 ? ldexpf(1.0f, -S) : 0.f;
    uint16_t* o16 = reinterpret_cast<uint16_t*>(img + o.conv1_h);
    for (int i = tid; i < 64 * 8; i += 256) {
        const int lane = i >> 3, j = i & 7;
        const int mrow = lane & 31, k = 8 * (lane >> 5) + j;
        const int mch = (mrow & 3) + 4 * (mrow >> 3) + 16 * ((mrow >> 2) & 1);
        uint16_t hb, lb;
        split_f16_dev(k < 9 ? ldexp(double(w1[mch * 9 + k]), S) : 0.0, hb, lb);
        o16[lane * 8 + j] = hb;
        o16[64 * 8 + lane * 8 + j] = lb;
    }
    if (tid < 32) img[o.conv1_b + tid] = b1[tid];
    if (tid < 64) img[o.conv2_b + tid] = b2[tid];
    // range: |conv1 out| <= max|in| * range[0] + range[1]; the same for conv2 with [2], [3]
    red[tid] = tid < 32 ? __builtin_fabsf(b1[tid]) : 0.f;
    __syncthreads();
    if (tid == 0) {
        double best = 0.0;
        float bm = 0.f;
        for (int c = 0; c < 32; ++c) { best = fmax(best, redd[c]); bm = fmaxf(bm, red[c]); }
        img[o.range + 0] = float(best * (1.0 + 1e-6));
        img[o.range + 1] = bm;
        float b2m = 0.f;
        for (int c = 0; c < 64; ++c) b2m = fmaxf(b2m, __builtin_fabsf(b2[c]));
        img[o.range + 2] = 0.f;
        img[o.range + 3] = b2m;
#pragma unroll
        for (int i = 4; i < 8; ++i) img[o.range + i] = 0.f;
    }
}

// one workgroup per output channel co, thread = (ci, dx): U0 = w[dy=0], U1 = (w0+w1+w2)/2, U2 = (w0-w1+w2)/2, U3 = w[dy=2] in double;
// the channel's exponent from max |U|; operand order [nt][ks = (xi*3+dx)*(CIN/32) + cb][hi,lo][lane][8] (pack_conv_wino_f16x3).
// range_at >= 0 (conv2): the row's l1 norm feeds img[range_at] through an atomic max.
template <int CIN>
__global__ __launch_bounds__(CIN * 4) void pack_conv_wino_dev_kernel(const float* __restrict__ w, float* __restrict__ img, int64_t at_hw,
                                                                      int64_t at_hws, int64_t range_at) {
    constexpr int T = CIN * 4, NCB = CIN / 32;
    __shared__ double red[T];
    const int co = blockIdx.x, tid = threadIdx.x;
    const int ci = tid / 3, dx = tid - 3 * ci;
    double u[4] = {0., 0., 0., 0.};
    if (tid < CIN * 3) {
        const double w0 = w[((co * CIN + ci) * 3 + 0) * 3 + dx], w1v = w[((co * CIN + ci) * 3 + 1) * 3 + dx], w2v = w[((co * CIN + ci) * 3 + 2) * 3 + dx];
        u[0] = w0; u[1] = 0.5 * (w0 + w1v + w2v); u[2] = 0.5 * (w0 - w1v + w2v); u[3] = w2v;
    }
    red[tid] = fmax(fmax(fabs(u[0]), fabs(u[1])), fmax(fabs(u[2]), fabs(u[3])));
    __syncthreads();
    for (int off = T / 2; off > 0; off >>= 1) {
        if (tid < off) red[tid] = fmax(red[tid], red[tid + off]);
        __syncthreads();
    }
    const int S = scale_exp_dev(float(red[0]) * 1.0000001f);
    if (tid == 0) {                              // the row's l1 norm summed in the host's order by one thread
        img[at_hws + co] = ldexpf(1.0f, -S);
        if (range_at >= 0) {
            double s = 0.0;
            for (int i = 0; i < CIN * 9; ++i) s += fabs(double(w[co * CIN * 9 + i]));
            atomicMax(reinterpret_cast<unsigned int*>(img + range_at), __float_as_uint(float(s * (1.0 + 1e-6))));
        }
    }
    if (tid < CIN * 3) {
        uint16_t* o16 = reinterpret_cast<uint16_t*>(img + at_hw);
        const int nt = co >> 4, cb = ci >> 5, lane = (co & 15) + 16 * ((ci & 31) >> 3), j = ci & 7;
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            const int ks = (xi * 3 + dx) * NCB + cb;
            uint16_t hb, lb;
            split_f16_dev(ldexp(u[xi], S), hb, lb);
            const int64_t base = ((int64_t(nt) * 12 * NCB + ks) * 2) * 64 * 8;
            o16[base + lane * 8 + j] = hb;
            o16[base + 64 * 8 + lane * 8 + j] = lb;
        }
    }
}
__global__ void copy_floats_kernel(const float* __restrict__ src, float* __restrict__ dst, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

// n_conv 2: conv1 + conv2 (Winograd) + range;  n_conv 3: also conv3 (Winograd) and its bias
int launch_pack_conv_h_dev(const ww_train_params* p, float* img, hipStream_t st) {
    const PackedLayout L = packed_layout(p->n_conv);
    const PackOffsets o{L.conv1_h, L.conv1_hs, L.conv1_b, L.conv2_hw, L.conv2_hws, L.conv2_b, L.range};
    hipLaunchKernelGGL(pack_conv1_h_dev_kernel, dim3(1), dim3(256), 0, st, p->conv_weight[0], p->conv_bias[0], p->conv_bias[1], img, o);
    hipLaunchKernelGGL(pack_conv_wino_dev_kernel<32>, dim3(64), dim3(128), 0, st, p->conv_weight[1], img, L.conv2_hw, L.conv2_hws, L.range + 2);
    if (p->n_conv == 3) {
        hipLaunchKernelGGL(pack_conv_wino_dev_kernel<64>, dim3(128), dim3(256), 0, st, p->conv_weight[2], img, L.conv3_hw, L.conv3_hws, int64_t(-1));
        hipLaunchKernelGGL(copy_floats_kernel, dim3(1), dim3(128), 0, st, p->conv_bias[2], img + L.conv3_b, 128);
    }
    WW_HIP(hipGetLastError());
    return WW_OK;
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of conv2 (32 -> 64) with relu(conv1) recomputed from the log-mel tile.
//   D[m = co][n = ci] += A[m][k] B[k][n] on v_mfma_f32_16x16x32_f16, k = the 32 columns of one image row:
//   A = mask row y (exact), B = a1 row y + dy - 1 shifted by dx - 1 (hi, lo): S[dy][dx] += A_y (B_hi + B_lo).
// Both operands sum over POSITIONS while the LDS tiles are channels-last records (one record per position, so that the dx shift is a
// record offset): the fragments come in through ds_read_b64_tr_b16, the transposing read.  The k <-> column map is a free choice (the
// same for A and B): lane group g, element j takes column 8g + 2j (first read) / 8g + 2j + 1 (second read), which makes every read's
// eight rows tile the 64 banks exactly with 144-byte records.
// 12 waves, one workgroup per CU, persistent over clips:
//   waves 0-7   CONSUMERS = (16 co) x (16 ci): nine 16x16 accumulators S (one per tap) for the current clip, nine more for
//               dW = sum_b gp[b,co] 2^a S_b (fp32, added at the end of every clip: gp never meets f16), one for the bias count;
//               per a1 row: 3 x 2 B fragments (12 transposed reads), one new mask row (2 reads; the rows y-1, y, y+1 roll through
//               registers), 18 + 1 MFMAs;
//   waves 8-11  PRODUCERS: per step of four rows, conv1 + ReLU on the vector ALU (thread = column x 4 channels), scaled by the clip's
//               2^-a, split, 8-byte stores; the mask rows from the bit image through a 256-entry LUT (byte -> eight f16 0/1).
// LDS: a ring of 10 mask rows (rows 4s-1 .. 4s+4 are read while 4s+5 .. 4s+8 are written), two halves of four a1 rows with zero
// halo columns, two log-mel tiles, the LUT.  One workgroup barrier per step (20 per clip).
// Output: one partial [64][32][9] + [64] per workgroup (the layout of conv_wgrad_kernel) -> reduce_partials_kernel.
// ------------------------------------------------------------------------------------------------
constexpr int kHRows = 4, kHSteps = kTH / kHRows, kHGRing = 10;
// DENSE (the 3-conv model's conv2, whose gradient is not rank one): the A operand is dz2 itself, pre-split by conv3_dgrad_h_kernel
// (f16 records [row][column][64 hi | 64 lo] of dz2 2^-edz, dzs[clip] = 2^edz): 272-byte records, three MFMAs per block
// (A_hi B_hi + A_hi B_lo + A_lo B_hi), dW += 2^(edz + a - 1) S.
template <bool DENSE>
struct WgHT {
    static constexpr int CIN = 32, COUT = 64;
    static constexpr int kARec = CIN * 4 + 16, kGRec = DENSE ? COUT * 4 + 16 : COUT * 2 + 16;     // 144; 144 | 272 bytes
    static constexpr int kARow = 34 * kARec, kGRow = 32 * kGRec;
    static constexpr int kOffA = kHGRing * kGRow;
    static constexpr int kOffMel = kOffA + 2 * kHRows * kARow;
    static constexpr int kOffLut = kOffMel + 2 * 2 * kMelHPlane * 2;      // two clips x (hi, lo) f16 planes
    static constexpr int kLds = kOffLut + 256 * 16;
    static constexpr int kPartial = COUT * CIN * 9 + COUT;
};
using WgH = WgHT<false>;
static_assert(WgHT<true>::kLds <= 160 * 1024 && WgH::kOffA % 16 == 0 && WgH::kOffMel % 16 == 0 && WgH::kOffLut % 16 == 0 &&
              WgHT<true>::kOffA % 16 == 0 && WgHT<true>::kOffMel % 16 == 0, "LDS map");

// maskbits: !DENSE the ReLU bit image [n][80][32][8 bytes];  DENSE the dz2 records (16 x 16 bytes per position).  gp: !DENSE [n][64];  DENSE dzs [n].
template <bool DENSE>
__global__ __launch_bounds__(768, 3) void conv2_wgrad_h_kernel(const float* __restrict__ mel, const uint8_t* __restrict__ maskbits,
                                                               const float* __restrict__ gp, int n, int width,
                                                               const u32x4* __restrict__ w1H, const float* __restrict__ hs1,
                                                               const float* __restrict__ b1, const float* __restrict__ rng,
                                                               float* __restrict__ partial) {
    using L = WgHT<DENSE>;
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    char* gring = ldsb;
    char* aring = ldsb + L::kOffA;
    _Float16* melh0 = reinterpret_cast<_Float16*>(ldsb + L::kOffMel);       // 2 clips x (hi plane, lo plane) of [82][36] f16
    u32x4* lut = reinterpret_cast<u32x4*>(ldsb + L::kOffLut);
    __shared__ uint32_t melmax[2];
    __shared__ int clip_ea[2][2];                              // per clip parity: input exponent e, activation exponent a
    __shared__ float clip_up[2], clip_dz[2];                   // 2^a / 2 (x 2^edz): the tile holds 2 relu(conv1) 2^-a;  2^edz

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < 8;
    const int my_clips = (n - int(blockIdx.x) + int(gridDim.x) - 1) / int(gridDim.x);
    const int total = my_clips * kHSteps;

    for (int i = tid; i < L::kLds / 4; i += 768) reinterpret_cast<uint32_t*>(ldsb)[i] = 0u;    // halo columns / dead columns stay zero
    if (tid == 0) { melmax[0] = 0u; melmax[1] = 0u; }
    __syncthreads();
    if (tid < 256) {
        u32x4 m;
#pragma unroll
        for (int d = 0; d < 4; ++d) m[d] = ((tid >> (2 * d)) & 1 ? 0x3c00u : 0u) | ((tid >> (2 * d + 1)) & 1 ? 0x3c000000u : 0u);
        lut[tid] = m;
    }
    // Model input of clip k -> two f16 planes (hi, lo) of x * 2^-e with a zero halo (the forward kernels' load_mel), in two stages
    // separated by a workgroup barrier: the clip's max |x| (mel_max), then e, a and the planes (mel_planes).  Producer threads only:
    // thread -> (column xx, rows y0 + 8 t).
    const int ptid = tid - 512;
    auto mel_max = [&](int k) {
        const float* __restrict__ sp = mel + (int64_t(blockIdx.x) + int64_t(k) * gridDim.x) * kTH * width + (ptid >> 5) * width + (ptid & 31);
        float mx = 0.f;
        if ((ptid & 31) < width) {
#pragma unroll
            for (int t = 0; t < 10; ++t) mx = fmaxf(mx, __builtin_fabsf(sp[8 * t * width]));
        }
        atomicMax(&melmax[k & 1], __float_as_uint(mx));
    };
    auto mel_planes = [&](int k) {
        const float mx = __uint_as_float(melmax[k & 1]);
        const int e = clampi(exp_of(mx) - 14, -100, 113);
        const int a = clampi(exp_of(fmaf(mx, rng[0], rng[1])) - 13, -100, 100);
        if (ptid == 0) {
            clip_ea[k & 1][0] = e; clip_ea[k & 1][1] = a;
            const float dz = DENSE ? gp[int64_t(blockIdx.x) + int64_t(k) * gridDim.x] : 1.0f;      // 2^edz
            clip_up[k & 1] = 0.5f * pow2i(a) * dz;
            clip_dz[k & 1] = dz;
        }
        const int xx = ptid & 31, y0 = ptid >> 5;
        if (xx < width) {
            const float* __restrict__ sp = mel + (int64_t(blockIdx.x) + int64_t(k) * gridDim.x) * kTH * width + y0 * width + xx;
            _Float16* dh = melh0 + (k & 1) * 2 * kMelHPlane + (y0 + 1) * kMelHRS + xx + 1;
            const float down = pow2i(-e);
#pragma unroll
            for (int t = 0; t < 10; ++t) {
                const float vv = sp[8 * t * width] * down;
                const _Float16 hi = static_cast<_Float16>(vv);
                dh[8 * t * kMelHRS] = hi;
                dh[8 * t * kMelHRS + kMelHPlane] = static_cast<_Float16>(vv - static_cast<float>(hi));
            }
        }
    };
    if (!consumer) mel_max(0);
    __syncthreads();
    if (!consumer) mel_planes(0);
    __syncthreads();

    if (!consumer) {
    // ================================================= producers =================================================
    // conv1 on the matrix cores (ww_conv1.h): producer wave pw makes a1 row 4 s + pw of every step
    const int pw = wave - 8;
    u32x4 w1h_r = w1H[lane], w1l_r = w1H[64 + lane];
    asm volatile("" : "+v"(w1h_r), "+v"(w1l_r));
    const half8 a1h = __builtin_bit_cast(half8, w1h_r), a1l = __builtin_bit_cast(half8, w1l_r);
    const GatherLanes glanes = gather_lanes(lane & 31, lane >> 5);
    const int s1_exp = -exp_of(hs1[0]);               // hs1[0] = 2^-S1
    Conv1Scale cs;
    const int x = lane & 31, h = lane >> 5;
    const bool col_ok = x < width;
    const int mcol = ptid >> 3, mcg = ptid & 7;                  // mask role: column, byte (8 channels) of the position's 64 bits
    auto produce = [&](int gs) {
        const int k = gs / kHSteps, s = gs - k * kHSteps;
        const int64_t clip = int64_t(blockIdx.x) + int64_t(k) * gridDim.x;
        // mask rows of this step: 0 .. 4 for the clip's first step, else 4s+1 .. 4s+4 (rows beyond 79 do not exist)
        const int g0 = s == 0 ? 0 : 4 * s + 1, g1 = 4 * s + 4 < kTH ? 4 * s + 4 : kTH - 1;
        uint8_t mb[5];
        u32x4 dzh[DENSE ? 5 : 1], dzl[DENSE ? 5 : 1];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int g = g0 + i;
            if constexpr (DENSE) {
                const u32x4* rec = reinterpret_cast<const u32x4*>(maskbits) + ((clip * kTH + (g <= g1 ? g : g1)) * kTW + mcol) * 16 + mcg;
                dzh[i] = rec[0];
                dzl[i] = rec[8];
            } else {
                mb[i] = g <= g1 ? maskbits[mask2_byte(clip, g, mcol, mcg)] : uint8_t(0);
            }
        }
        if (s == 0) {                                            // the clip's scales: bias in the accumulator's scale, descale
            const int e = clip_ea[k & 1][0], a = clip_ea[k & 1][1];
#pragma unroll
            for (int j = 0; j < 16; ++j) cs.binit[j] = ldexpf(b1[16 * h + j], s1_exp - e);
            cs.sc = ldexpf(1.0f, e - a - s1_exp);
        }
        // a1 row 4s + pw -> half gs & 1 of the a ring: 2 relu(conv1) 2^-a, split, the lane's 16 channels as 4 x 16 bytes
        {
            Conv1Row r;
            conv1_row_gather(r, melh0 + (k & 1) * 2 * kMelHPlane, glanes, kHRows * s + pw);
            conv1_row_mfma(r, a1h, a1l, cs);
            if (col_ok) {
                char* rec = aring + ((gs & 1) * kHRows + pw) * L::kARow + (x + 1) * L::kARec + h * 32;
#pragma unroll
                for (int g8 = 0; g8 < 2; ++g8) {
                    u32x4 vh, vl;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        uint32_t hh, ll;
                        split2(relu2(r.acc[8 * g8 + 2 * d] * cs.sc), relu2(r.acc[8 * g8 + 2 * d + 1] * cs.sc), hh, ll);
                        vh[d] = hh;
                        vl[d] = ll;
                    }
                    *reinterpret_cast<u32x4*>(rec + g8 * 16) = vh;
                    *reinterpret_cast<u32x4*>(rec + L::CIN * 2 + g8 * 16) = vl;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int g = g0 + i;
            if (g <= g1) {
                const int slot = (k * kTH + g) % kHGRing;
                char* rec = gring + slot * L::kGRow + mcol * L::kGRec + mcg * 16;
                if constexpr (DENSE) {
                    *reinterpret_cast<u32x4*>(rec) = dzh[i];
                    *reinterpret_cast<u32x4*>(rec + L::COUT * 2) = dzl[i];
                } else {
                    *reinterpret_cast<u32x4*>(rec) = lut[mb[i]];
                }
            }
        }
        // the next clip's planes and exponents, well ahead of its first step (barriers separate the three stages)
        if (k + 1 < my_clips) {
            if (s == 4 && ptid == 0) melmax[(k + 1) & 1] = 0u;
            if (s == 8) mel_max(k + 1);
            if (s == 12) mel_planes(k + 1);
        }
    };
    if (total > 0) produce(0);
    __syncthreads();
    for (int gs = 0; gs < total; ++gs) {
#ifndef WW_HABL_NOPROD            // timing-only ablation: consumers alone (results are garbage)
        if (gs + 1 < total) produce(gs + 1);
#endif
        __syncthreads();
    }
    } else {
    // ================================================= consumers =================================================
    const int ct = wave & 3, it = (wave >> 2) & 1;
    const int grp = lane >> 4, q = (lane >> 2) & 3, p = lane & 3, ln = lane & 15;
    // transposed-read bases: row = column 8 grp + 2 q of the image row (+1 for the second read), 16-bit columns 4p .. 4p+3 of the tile
    const char* abase = gring + (8 * grp + 2 * q) * L::kGRec + (16 * ct + 4 * p) * 2;
    const char* bbase = aring + (8 * grp + 2 * q) * L::kARec + (16 * it + 4 * p) * 2;
    f32x4 S[9], dW[9], cnt, dbv;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) { S[t][j] = 0.f; dW[t][j] = 0.f; }
#pragma unroll
    for (int j = 0; j < 4; ++j) { cnt[j] = 0.f; dbv[j] = 0.f; }
    struct APair { half8 h, l; };                                // A fragment: the mask (l unused), or dz2's hi and lo halves
    APair a_prev, a_cur, a_next;
    half8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a_prev.h[j] = 0; a_cur.h[j] = 0; a_next.h[j] = 0; a_prev.l[j] = 0; a_cur.l[j] = 0; a_next.l[j] = 0;
        ones[j] = static_cast<_Float16>(1.0f);
    }
    float4 g4 = make_float4(1.f, 1.f, 1.f, 1.f);
    auto load_a = [&](int k, int g) -> APair {
        const char* r = abase + ((k * kTH + g) % kHGRing) * L::kGRow;
        APair f;
        f.h = cat8(lds_tr16(r), lds_tr16(r + L::kGRec));
        if constexpr (DENSE) f.l = cat8(lds_tr16(r + L::COUT * 2), lds_tr16(r + L::kGRec + L::COUT * 2));
        else f.l = f.h;
        return f;
    };
    auto consume = [&](int gs) {
        const int k = gs / kHSteps, s = gs - k * kHSteps;
        const int64_t clip = int64_t(blockIdx.x) + int64_t(k) * gridDim.x;
        if (s == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { a_prev.h[j] = 0; a_prev.l[j] = 0; }
            a_cur = load_a(k, 0);
            if constexpr (!DENSE) g4 = *reinterpret_cast<const float4*>(gp + clip * L::COUT + 16 * ct + 4 * grp);
        }
#pragma unroll
        for (int i = 0; i < kHRows; ++i) {
            const int r = kHRows * s + i;
            if (r + 1 < kTH) a_next = load_a(k, r + 1);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) { a_next.h[j] = 0; a_next.l[j] = 0; }
            }
            const char* br = bbase + ((gs & 1) * kHRows + i) * L::kARow;
            half8 bh[3], bl[3];
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const char* b0 = br + dx * L::kARec;
                bh[dx] = cat8(lds_tr16(b0), lds_tr16(b0 + L::kARec));
                bl[dx] = cat8(lds_tr16(b0 + L::CIN * 2), lds_tr16(b0 + L::kARec + L::CIN * 2));
            }
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const APair ay = dy == 0 ? a_next : (dy == 1 ? a_cur : a_prev);       // gradient row r - dy + 1
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    S[dy * 3 + dx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ay.h, bh[dx], S[dy * 3 + dx], 0, 0, 0);
                    S[dy * 3 + dx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ay.h, bl[dx], S[dy * 3 + dx], 0, 0, 0);
                    if constexpr (DENSE) S[dy * 3 + dx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ay.l, bh[dx], S[dy * 3 + dx], 0, 0, 0);
                }
            }
            cnt = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_cur.h, ones, cnt, 0, 0, 0);  // sum over positions of the gradient row, per channel
            if constexpr (DENSE) cnt = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_cur.l, ones, cnt, 0, 0, 0);
            a_prev = a_cur;
            a_cur = a_next;
        }
        if (s == kHSteps - 1) {
            // dW += gp[b, co] * (2^a / 2) S (the tile holds 2 relu(conv1) 2^-a): D register j <-> co = 16 ct + 4 grp + j, lane & 15 <-> ci
            const float up = clip_up[k & 1];
            const float gj[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) { dW[t][j] = fmaf(gj[j], S[t][j] * up, dW[t][j]); S[t][j] = 0.f; }
#pragma unroll
            for (int j = 0; j < 4; ++j) { dbv[j] = fmaf(gj[j] * clip_dz[k & 1], cnt[j], dbv[j]); cnt[j] = 0.f; }
        }
    };

    __syncthreads();
    for (int gs = 0; gs < total; ++gs) {
#ifndef WW_HABL_NOCONS            // timing-only ablation: producers alone
        consume(gs);
#endif
        __syncthreads();
    }
    {
        float* outp = partial + int64_t(blockIdx.x) * L::kPartial;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) outp[((16 * ct + 4 * grp + j) * L::CIN + 16 * it + ln) * 9 + t] = dW[t][j];
        if (it == 0 && ln == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) outp[L::COUT * L::CIN * 9 + 16 * ct + 4 * grp + j] = dbv[j];
        }
    }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of conv3 (64 -> 128) of the 3-conv WakewordModel: the conv2 kernel's arithmetic (mask x hi/lo, transposed reads) with
//   * the B operand = relu(conv2) from HBM (float32 channels last, written by the forward) scaled by the clip's 2^-a2 and split by the
//     producers (thread = column x 8 channels: two 16-byte loads, two 16-byte stores per row);
//   * the 128 co in two PASSES of 64 per clip (a pass = the conv2 kernel's clip: 20 steps; mask records hold the pass's 64 channels), so
//     that a consumer wave = (16 co) x (2 x 16 ci) keeps 18 accumulators (72 VGPRs) and the kernel its 12 waves at 3 per SIMD;
//   * dW accumulated in the workgroup's own partial in GLOBAL memory, in accumulator (lane) order: after every pass
//     partial[pass][wave][pair*9 + tap][j][lane] += gp[b,co] 2^a2 S  (72 coalesced read-modify-writes per wave and pass; no atomics: the
//     region is private to the wave), un-permuted by reduce_wgrad3_h_kernel.
// ------------------------------------------------------------------------------------------------
struct Wg3H {
    static constexpr int CIN = 64, COUT = 128;
    static constexpr int kARec = CIN * 4 + 16, kGRec = 64 * 2 + 16;        // 272, 144 bytes (mask records: the pass's 64 co)
    static constexpr int kARow = 34 * kARec, kGRow = 32 * kGRec;
    static constexpr int kOffA = kHGRing * kGRow;
    static constexpr int kOffLut = kOffA + 2 * kHRows * kARow;
    static constexpr int kLds = kOffLut + 256 * 16;
    static constexpr int kPartial = COUT * CIN * 9 + COUT;                   // = conv_wgrad_kernel<64, 128>'s partial
};
static_assert(Wg3H::kLds <= 160 * 1024 && Wg3H::kOffA % 16 == 0 && Wg3H::kOffLut % 16 == 0, "LDS map");

__global__ __launch_bounds__(768, 3) void conv3_wgrad_h_kernel(const float* __restrict__ act2 /*[n][80][32][64]*/, const float* __restrict__ apow2,
                                                               const uint8_t* __restrict__ maskbits /*conv3 mask image: mask3_byte*/,
                                                               const float* __restrict__ gp /*[n][128]*/, int n, float* __restrict__ partial) {
    using L = Wg3H;
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    char* gring = ldsb;
    char* aring = ldsb + L::kOffA;
    u32x4* lut = reinterpret_cast<u32x4*>(ldsb + L::kOffLut);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < 8;
    const int my_clips = (n - int(blockIdx.x) + int(gridDim.x) - 1) / int(gridDim.x);
    const int total = my_clips * 2 * kHSteps;                     // global step gs = (clip k, pass hf, step s) = ((k*2 + hf)*20 + s)

    for (int i = tid; i < L::kLds / 4; i += 768) reinterpret_cast<uint32_t*>(ldsb)[i] = 0u;    // halo columns stay zero
    __syncthreads();
    if (tid < 256) {
        u32x4 m;
#pragma unroll
        for (int d = 0; d < 4; ++d) m[d] = ((tid >> (2 * d)) & 1 ? 0x3c00u : 0u) | ((tid >> (2 * d + 1)) & 1 ? 0x3c000000u : 0u);
        lut[tid] = m;
    }
    __syncthreads();
    float* mypart = partial + int64_t(blockIdx.x) * L::kPartial;

    if (!consumer) {
    // ================================================= producers =================================================
    const int ptid = tid - 512;
    const int mcol = ptid >> 3, mcg = ptid & 7;                  // column, byte of the pass's 64 mask bits / group of 8 input channels
    auto produce = [&](int gs) {
        const int vc = gs / kHSteps, s = gs - vc * kHSteps;        // vc = k*2 + hf: a pass is this kernel's "clip"
        const int k = vc >> 1, hf = vc & 1;
        const int64_t clip = int64_t(blockIdx.x) + int64_t(k) * gridDim.x;
        const int g0 = s == 0 ? 0 : 4 * s + 1, g1 = 4 * s + 4 < kTH ? 4 * s + 4 : kTH - 1;
        uint8_t mb[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int g = g0 + i;
            mb[i] = g <= g1 ? maskbits[mask3_byte(clip, g, mcol, 8 * hf + mcg)] : uint8_t(0);
        }
        // relu(conv2) rows 4s .. 4s+3, this thread's column and 8 channels
        const float down = __uint_as_float(0x7f000000u - __float_as_uint(apow2[clip]));       // 2^-a2
        const float* src = act2 + ((clip * kTH + kHRows * s) * kTW + mcol) * L::CIN + 8 * mcg;
        float4 va[kHRows], vb[kHRows];
#pragma unroll
        for (int i = 0; i < kHRows; ++i) {
            va[i] = *reinterpret_cast<const float4*>(src + int64_t(i) * kTW * L::CIN);
            vb[i] = *reinterpret_cast<const float4*>(src + int64_t(i) * kTW * L::CIN + 4);
        }
        char* arow = aring + ((gs & 1) * kHRows) * L::kARow + (mcol + 1) * L::kARec + mcg * 16;
#pragma unroll
        for (int i = 0; i < kHRows; ++i) {
            u32x4 vh, vl;
            uint32_t hh, ll;
            split2(va[i].x * down, va[i].y * down, hh, ll); vh[0] = hh; vl[0] = ll;
            split2(va[i].z * down, va[i].w * down, hh, ll); vh[1] = hh; vl[1] = ll;
            split2(vb[i].x * down, vb[i].y * down, hh, ll); vh[2] = hh; vl[2] = ll;
            split2(vb[i].z * down, vb[i].w * down, hh, ll); vh[3] = hh; vl[3] = ll;
            *reinterpret_cast<u32x4*>(arow + i * L::kARow) = vh;
            *reinterpret_cast<u32x4*>(arow + i * L::kARow + L::CIN * 2) = vl;
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int g = g0 + i;
            if (g <= g1) {
                const int slot = (vc * kTH + g) % kHGRing;
                *reinterpret_cast<u32x4*>(gring + slot * L::kGRow + mcol * L::kGRec + mcg * 16) = lut[mb[i]];
            }
        }
    };
    if (total > 0) produce(0);
    __syncthreads();
    for (int gs = 0; gs < total; ++gs) {
        if (gs + 1 < total) produce(gs + 1);
        __syncthreads();
    }
    } else {
    // ================================================= consumers =================================================
    const int ct = wave & 3, ip = (wave >> 2) & 1;                 // 16 co of the pass x ci tiles 2 ip, 2 ip + 1
    const int grp = lane >> 4, q = (lane >> 2) & 3, p = lane & 3, ln = lane & 15;
    const char* abase = gring + (8 * grp + 2 * q) * L::kGRec + (16 * ct + 4 * p) * 2;
    const char* bbase = aring + (8 * grp + 2 * q) * L::kARec + (32 * ip + 4 * p) * 2;
    f32x4 S[2][9], cnt;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) S[u][t][j] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) cnt[j] = 0.f;
    half8 a_prev, a_cur, a_next, ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a_prev[j] = 0; a_cur[j] = 0; a_next[j] = 0; ones[j] = static_cast<_Float16>(1.0f); }
    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float up = 0.f;
    auto load_a = [&](int vc, int g) -> half8 {
        const char* r = abase + ((vc * kTH + g) % kHGRing) * L::kGRow;
        return cat8(lds_tr16(r), lds_tr16(r + L::kGRec));
    };
    __syncthreads();
    for (int gs = 0; gs < total; ++gs) {
        const int vc = gs / kHSteps, s = gs - vc * kHSteps;
        const int k = vc >> 1, hf = vc & 1;
        const int64_t clip = int64_t(blockIdx.x) + int64_t(k) * gridDim.x;
        if (s == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a_prev[j] = 0;
            a_cur = load_a(vc, 0);
            g4 = *reinterpret_cast<const float4*>(gp + clip * L::COUT + 64 * hf + 16 * ct + 4 * grp);
            up = apow2[clip];
        }
#pragma unroll
        for (int i = 0; i < kHRows; ++i) {
            const int r = kHRows * s + i;
            if (r + 1 < kTH) a_next = load_a(vc, r + 1);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) a_next[j] = 0;
            }
            const char* br = bbase + ((gs & 1) * kHRows + i) * L::kARow;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                half8 bh[3], bl[3];
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const char* b0 = br + dx * L::kARec + u * 32;
                    bh[dx] = cat8(lds_tr16(b0), lds_tr16(b0 + L::kARec));
                    bl[dx] = cat8(lds_tr16(b0 + L::CIN * 2), lds_tr16(b0 + L::kARec + L::CIN * 2));
                }
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const half8 ay = dy == 0 ? a_next : (dy == 1 ? a_cur : a_prev);       // mask row r - dy + 1
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        S[u][dy * 3 + dx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ay, bh[dx], S[u][dy * 3 + dx], 0, 0, 0);
                        S[u][dy * 3 + dx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ay, bl[dx], S[u][dy * 3 + dx], 0, 0, 0);
                    }
                }
            }
            cnt = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_cur, ones, cnt, 0, 0, 0);
            a_prev = a_cur;
            a_cur = a_next;
        }
        if (s == kHSteps - 1) {
            // partial += gp[b, co] * 2^a2 * S; D register j <-> co = 64 hf + 16 ct + 4 grp + j, lane & 15 <-> ci = 16 (2 ip + u) + ln
            const float gj[4] = {g4.x, g4.y, g4.z, g4.w};
            float* pp = mypart + ((hf * 8 + wave) * 18) * 256 + lane;
            const bool first = k == 0;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float* at = pp + ((u * 9 + t) * 4 + j) * 64;
                        const float v = gj[j] * (S[u][t][j] * up);
                        *at = first ? v : *at + v;
                        S[u][t][j] = 0.f;
                    }
            if (ip == 0 && ln == 0) {
                float* db = mypart + L::COUT * L::CIN * 9 + 64 * hf + 16 * ct + 4 * grp;
#pragma unroll
                for (int j = 0; j < 4; ++j) db[j] = first ? gj[j] * cnt[j] : db[j] + gj[j] * cnt[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) cnt[j] = 0.f;
        }
        __syncthreads();
    }
    }
}

// sum of the workgroups' lane-order partials in a fixed order (four slices of the groups per output, added in order), un-permuted to
// torch's [128][64][3][3] (+ bias).  Thread -> SOURCE index (coalesced reads); the destination follows from it.
__global__ __launch_bounds__(256) void reduce_wgrad3_h_kernel(const float* __restrict__ partial, int groups, float* __restrict__ dw,
                                                              float* __restrict__ db) {
    constexpr int kW = 128 * 64 * 9, kP = kW + 128;
    __shared__ float part[4][64];
    const int c = threadIdx.x & 63, sl = threadIdx.x >> 6;
    for (int i0 = blockIdx.x * 64; i0 < kP; i0 += gridDim.x * 64) {
        const int src = i0 + c;
        float s = 0.f;
        if (src < kP)
            for (int g = sl; g < groups; g += 4) s += partial[int64_t(g) * kP + src];
        part[sl][c] = s;
        __syncthreads();
        if (sl == 0 && src < kP) {
            int dst = src;
            if (src < kW) {      // src = ((((hf*8 + wave)*18) + u*9 + t)*4 + j)*64 + lane
                const int lane = src & 63, j = (src >> 6) & 3, ut = (src >> 8) % 18, hw = src / (18 * 256);
                const int u = ut / 9, t = ut - 9 * u, wave = hw & 7, hf = hw >> 3;
                const int co = 64 * hf + 16 * (wave & 3) + 4 * (lane >> 4) + j, ci = 16 * (2 * (wave >> 2) + u) + (lane & 15);
                dst = (co * 64 + ci) * 9 + t;
            }
            const float v = ((part[0][c] + part[1][c]) + part[2][c]) + part[3][c];
            if (src < kW) dw[dst] = v;
            else db[src - kW] = v;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Data gradient of conv2 and, from it, the weight gradient of conv1 -- the first layer: da1 is consumed where it is produced.
//   da1[ci][y][x] = sum_{co,dy,dx} mask2[co][y-dy+1][x-dx+1] * G[co][ci][dy][dx],     G = gp[b,co] * W2   (rebuilt per clip)
//   D[m = column][n = ci] += A[m][k] B[k][n] on v_mfma_f32_16x16x32_f16, k = 32 of the 64 co of one tap:
//   A = mask2 at the shifted position (exact 0/1, one ds_read_b128 of the channels-last record), B = G 2^-eg as hi + lo: two MFMAs.
//   dz1 = da1 * [relu(conv1) > 0]  (the forward's sign bits);  dW1[ci][t] = sum_pos dz1[ci][pos] * mel[pos + t],  db1 = sum dz1
//   as one more split-precision block per row (M = ci, N = the 9 taps + a column of ones, K = the row's 32 positions; see the
//   epilogue): a conv1 weight gradient is a sum of ~10^7 products with heavy cancellation -- float inside one clip (2560 positions),
//   double across clips.
// 8 waves (256 VGPRs each) = (16 ci) x (row of a four-row step); the wave's 18 k-steps x (hi, lo) = 144 B-operand VGPRs are rebuilt
// at the start of every clip from the pre-ordered fp32 weights (L1/L2) times the clip's gp, scaled by 2^-eg (eg from max|gp[b]| max|W2|).
// All 512 threads fill the next step's mask rows (bit image -> LUT -> 160-byte records) and conv1 sign rows; one barrier per step.
// Output: one partial [32][9] + [32] (float; summed over the workgroup in double) per workgroup -> reduce_partials_kernel.
// ------------------------------------------------------------------------------------------------
// DENSE (the 3-conv model's conv2): A = dz2 pre-split by conv3_dgrad_h_kernel (f16 records [64 hi | 64 lo], 288 bytes in LDS), B = W2 2^-ew
// as hi + lo built ONCE per launch, three MFMAs per block (A_hi B_hi + A_hi B_lo + A_lo B_hi), da1 = 2^(ew + edz) x the accumulator.
template <bool DENSE>
struct DgHT {
    static constexpr int CIN = 32, COUT = 64;
    static constexpr int kGRec = DENSE ? 288 : 160, kGRow = 34 * kGRec;     // columns -1..32, conflict-free 16-byte row reads
    static constexpr int kMelRS = 36, kMelFloats = (kTH + 2) * kMelRS;      // fp32 log-mel tile with a zero halo
    static constexpr int kOffMel = (kHGRing + 1) * kGRow;                   // ring + one all-zero row (the rows above / below the image)
    static constexpr int kOffS1 = kOffMel + 2 * kMelFloats * 4;             // conv1 sign words: 2 steps x 4 rows x 32 columns
    static constexpr int kOffLut = kOffS1 + 2 * kHRows * kTW * 4;
    static constexpr int kLds = kOffLut + 256 * 16;
    static constexpr int kPartial = CIN * 9 + CIN;
};
using DgH = DgHT<false>;
static_assert(DgHT<true>::kLds <= 160 * 1024 && DgH::kOffMel % 16 == 0 && DgH::kOffS1 % 16 == 0 && DgH::kOffLut % 16 == 0 &&
              DgHT<true>::kOffMel % 16 == 0 && DgHT<true>::kOffS1 % 16 == 0, "LDS map");

// W2 [64][32][3][3] -> the lane order of the B operand: wp[((nt*18 + ks)*64 + lane)*8 + j] = W2[32 kb + 8 g + j][16 nt + n][dy][dx],
// ks = (dy*3 + dx)*2 + kb, lane = n + 16 g; wmax = max |W2| (float bits, zeroed by the caller)
__global__ void pack_dgrad_h_dev_kernel(const float* __restrict__ w2, float* __restrict__ wp, unsigned int* __restrict__ wmax) {
    float m = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * 18 * 64 * 8; i += gridDim.x * blockDim.x) {
        const int j = i & 7, lane = (i >> 3) & 63, ks = (i >> 9) % 18, nt = i / (18 * 512);
        const int kb = ks & 1, tap = ks >> 1, co = 32 * kb + 8 * (lane >> 4) + j, ci = 16 * nt + (lane & 15);
        const float v = w2[(co * 32 + ci) * 9 + tap];
        wp[i] = v;
        m = fmaxf(m, __builtin_fabsf(v));
    }
    atomicMax(wmax, __float_as_uint(m));
}
// gp[b][co] = dpooled[b][co] * s, gpmax[b] = max_co |gp[b][co]|: one wave per clip (64 channels)
__global__ void gp_max_kernel(const float* __restrict__ dpooled, float s, int n, float* __restrict__ gp, float* __restrict__ gpmax) {
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= n) return;
    const float v = dpooled[int64_t(b) * 64 + lane] * s;
    gp[int64_t(b) * 64 + lane] = v;
    float m = __builtin_fabsf(v);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (lane == 0) gpmax[b] = m;
}

// Diagnostic build -DWW_DG_STAMPS (never shipped): every wave of workgroup 7 adds up the shader cycles (s_memtime) it spends in the step's phases --
// 0 rebuild + issuing the next rows' loads, 1 the 72-MFMA loop, 2 the dW1 / db1 block, 3 expanding the next rows into LDS, 4 the barrier.
#ifdef WW_DG_STAMPS
__device__ unsigned long long g_dg_stamps[40];
#define DSTAMP(i) do { unsigned long long t__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__) :: "memory"); \
    dst[i] += t__ - dlast; dlast = t__; } while (0)
extern "C" __attribute__((visibility("default"))) int ww_debug_dg_stamps(unsigned long long* out) {
    unsigned long long z[40] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dg_stamps), sizeof(z)) != hipSuccess) return -1;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dg_stamps), z, sizeof(z));
    return 0;
}
#else
#define DSTAMP(i) do {} while (0)
#endif

// maskbits: !DENSE the ReLU bit image of conv2;  DENSE the dz2 records.  gpmax: !DENSE max |gp| per clip;  DENSE dzs [n] (2^edz).  gp unused if DENSE.
template <bool DENSE>
__global__ __launch_bounds__(512, 2) void conv2_dgrad_h_kernel(const float* __restrict__ mel, const uint8_t* __restrict__ maskbits,
                                                               const uint32_t* __restrict__ bits1, const float* __restrict__ gp,
                                                               const float* __restrict__ gpmax, const float* __restrict__ wp,
                                                               const float* __restrict__ w2max, int n, int width,
                                                               float* __restrict__ partial) {
    using L = DgHT<DENSE>;
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    char* gring = ldsb;
    float* meltile = reinterpret_cast<float*>(ldsb + L::kOffMel);
    uint32_t* s1rows = reinterpret_cast<uint32_t*>(ldsb + L::kOffS1);
    u32x4* lut = reinterpret_cast<u32x4*>(ldsb + L::kOffLut);

    __shared__ uint32_t melmax[2];                                  // float bits of max |mel| of clip k at [k & 1]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = wave & 1, rg = wave >> 1;                       // 16 ci x row of the step
    if (tid == 0) { melmax[0] = 0u; melmax[1] = 0u; }
    const int ln = lane & 15, grp = lane >> 4;
    const int my_clips = (n - int(blockIdx.x) + int(gridDim.x) - 1) / int(gridDim.x);
    const int total = my_clips * kHSteps;

    for (int i = tid; i < L::kLds / 4; i += 512) reinterpret_cast<uint32_t*>(ldsb)[i] = 0u;      // halo columns stay zero
    __syncthreads();
    if (tid < 256) {
        u32x4 m;
#pragma unroll
        for (int d = 0; d < 4; ++d) m[d] = ((tid >> (2 * d)) & 1 ? 0x3c00u : 0u) | ((tid >> (2 * d + 1)) & 1 ? 0x3c000000u : 0u);
        lut[tid] = m;
    }
    auto load_mel = [&](int k) {                                    // + the clip's max |mel| (the exponent of its f16 image in the epilogue)
        const float* src = mel + (int64_t(blockIdx.x) + int64_t(k) * gridDim.x) * kTH * width;
        float* melt = meltile + (k & 1) * L::kMelFloats;
        float mx = 0.f;
        for (int i = tid; i < kTH * width; i += 512) {
            const int y = i / width, xx = i - y * width;
            const float v = src[i];
            melt[(y + 1) * L::kMelRS + xx + 1] = v;
            mx = fmaxf(mx, __builtin_fabsf(v));
        }
        atomicMax(&melmax[k & 1], __float_as_uint(mx));
    };
    auto mel_exp = [&](int k) { return clampi(exp_of(__uint_as_float(melmax[k & 1])) - 14, -100, 100); };      // |mel| 2^-em < 2^15
    auto scale_mel = [&](int k) {                                   // a barrier after load_mel(k): the tile becomes mel 2^-em (f16 range)
        float* melt = meltile + (k & 1) * L::kMelFloats;
        const float down = pow2i(-mel_exp(k));
        for (int i = tid; i < kTH * width; i += 512) {
            const int y = i / width, xx = i - y * width;
            melt[(y + 1) * L::kMelRS + xx + 1] *= down;
        }
    };
    // the rows a step needs beyond what is already in LDS: mask rows 0..4 (first step of a clip) or 4s+1..4s+4, conv1 sign rows 4s..4s+3
    const int mcg = tid & 7, mpos = tid >> 3;                       // mask role: byte (8 channels), position 0..63 (+64 per pass)
    const int dchunk = tid & 15, dpos = tid >> 4;                   // DENSE: 16-byte chunk of the 256-byte record, position 0..31 (+32 per pass)
    uint8_t mb[3];
    u32x4 dzr[DENSE ? 5 : 1];
    uint32_t sw_next = 0u;
    // Every global address below is a UNIFORM base (scalar registers: clip, step) + a 32-bit per-thread offset.  Written as one 64-bit
    // index per thread, the compiler kept `pointer + thread part` in register pairs across the step loop, spilled them at this kernel's
    // 256 registers, and reloaded them from scratch in front of the loads -- with s_waitcnt vmcnt(0), i.e. every step first waited
    // for the loads it had just issued to run ahead of its matrix work (scripts/isa_scratch.py lists such reloads).
    const uint32_t mthread = uint32_t(mask2_byte(0, 0, mpos & 31, mcg));          // the thread's part of mask2_byte: N-tile, column, byte
    const uint32_t dthread = uint32_t(dpos * 16 + dchunk);
    auto fill_load = [&](int gs) {                                 // issue the global loads early ...
        const int k = gs / kHSteps, s = gs - k * kHSteps;
        const int64_t clip = int64_t(blockIdx.x) + int64_t(k) * gridDim.x;
        const int g0 = s == 0 ? 0 : 4 * s + 1, g1 = 4 * s + 4 < kTH ? 4 * s + 4 : kTH - 1;
        if constexpr (DENSE) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int g = g0 + i <= g1 ? g0 + i : g1;
                const u32x4* rowp = reinterpret_cast<const u32x4*>(maskbits) + (clip * kTH + g) * kTW * 16;
                dzr[i] = rowp[dthread];
            }
        } else {
            const uint8_t* clipp = maskbits + clip * (kTH / 2) * 512;              // mask2_byte(clip, 0, 0, 0)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int pos = mpos + 64 * i, g = g0 + (pos >> 5);
                const uint32_t off = uint32_t((g >> 1) * 512 + (g & 1) * 32) + mthread;      // mask2_byte(clip, g, pos & 31, mcg) - the clip's base
                mb[i] = g <= g1 ? clipp[off] : uint8_t(0);
            }
        }
        const uint32_t* s1p = bits1 + (clip * kTH + kHRows * s) * kTW;
        if (tid < kHRows * kTW) sw_next = s1p[uint32_t(tid)];
    };
    auto fill_store = [&](int gs) {                                // ... and expand them into LDS behind the step's matrix work
        const int k = gs / kHSteps, s = gs - k * kHSteps;
        const int g0 = s == 0 ? 0 : 4 * s + 1, g1 = 4 * s + 4 < kTH ? 4 * s + 4 : kTH - 1;
        if constexpr (DENSE) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int g = g0 + i;
                if (g <= g1) {
                    const int slot = (k * kTH + g) % kHGRing;
                    *reinterpret_cast<u32x4*>(gring + slot * L::kGRow + (dpos + 1) * L::kGRec + dchunk * 16) = dzr[i];
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int pos = mpos + 64 * i, g = g0 + (pos >> 5);
                if (g <= g1) {
                    const int slot = (k * kTH + g) % kHGRing;
                    *reinterpret_cast<u32x4*>(gring + slot * L::kGRow + ((pos & 31) + 1) * L::kGRec + mcg * 16) = lut[mb[i]];
                }
            }
        }
        if (tid < kHRows * kTW) s1rows[(gs & 1) * kHRows * kTW + tid] = sw_next;
        if (s == 8 && k + 1 < my_clips) load_mel(k + 1);
        if (s == 12 && k + 1 < my_clips) scale_mel(k + 1);
    };

    // B operand of this wave: G[k-step][hi, lo], rebuilt per clip
    half8 bh[18], bl[18];
    // dW1 / db1: row (register j) ci = 16 nt + 4 grp + j, column (lane & 15) = tap, 9 = bias: one clip in float (cacc), all clips in double
    double dacc[4] = {0., 0., 0., 0.};
    f32x4 cacc = {0.f, 0.f, 0.f, 0.f};
    // dz1 in the accumulator's units is below 576 x 2^13 (!DENSE: G 2^-eg < 2^13) or 576 x 2^14 x 2^13 (DENSE: records < 2^14): scaled into f16
    constexpr float kDzDown = DENSE ? 0x1p-22f : 0x1p-10f;
    float dscale = 0.f, mel_up = 1.f;
    auto rebuild = [&](int k) {
        const int64_t clip = int64_t(blockIdx.x) + int64_t(k) * gridDim.x;
        // !DENSE: G = gp[b] W2 2^-eg, per clip.  DENSE: W2 2^-ew once (k == 0), the clip only changes the descale 2^(ew + edz)
        const int eg = clampi(exp_of((DENSE ? 1.0f : gpmax[clip]) * w2max[0]) - 12, -100, 100);
        const float down = pow2i(-eg);
        dscale = DENSE ? pow2i(eg) * gpmax[clip] : pow2i(eg);
        mel_up = pow2i(mel_exp(k));
        if (DENSE && k > 0) return;
        float gv[16];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int q4 = 0; q4 < 2; ++q4) {
                float4 v = make_float4(1.f, 1.f, 1.f, 1.f);
                if constexpr (!DENSE) v = *reinterpret_cast<const float4*>(gp + clip * L::COUT + 32 * kb + 8 * grp + 4 * q4);
                gv[8 * kb + 4 * q4] = v.x * down; gv[8 * kb + 4 * q4 + 1] = v.y * down; gv[8 * kb + 4 * q4 + 2] = v.z * down; gv[8 * kb + 4 * q4 + 3] = v.w * down;
            }
        const float4* w4 = reinterpret_cast<const float4*>(wp) + (int64_t(nt) * 18 * 64 + lane) * 2;
#pragma unroll
        for (int ks = 0; ks < 18; ++ks) {
            const float4 wa = w4[ks * 128], wb = w4[ks * 128 + 1];
            const int kb = ks & 1;
            const float g0 = wa.x * gv[8 * kb], g1 = wa.y * gv[8 * kb + 1], g2 = wa.z * gv[8 * kb + 2], g3 = wa.w * gv[8 * kb + 3],
                        g4 = wb.x * gv[8 * kb + 4], g5 = wb.y * gv[8 * kb + 5], g6 = wb.z * gv[8 * kb + 6], g7 = wb.w * gv[8 * kb + 7];
            u32x4 vh, vl;
            uint32_t hh, ll;
            split2(g0, g1, hh, ll); vh[0] = hh; vl[0] = ll;
            split2(g2, g3, hh, ll); vh[1] = hh; vl[1] = ll;
            split2(g4, g5, hh, ll); vh[2] = hh; vl[2] = ll;
            split2(g6, g7, hh, ll); vh[3] = hh; vl[3] = ll;
            bh[ks] = __builtin_bit_cast(half8, vh);
            bl[ks] = __builtin_bit_cast(half8, vl);
        }
    };

    load_mel(0);
    if (total > 0) fill_load(0);
    __syncthreads();                                               // the LUT is complete
    if (total > 0) scale_mel(0);
    if (total > 0) fill_store(0);
    __syncthreads();
    const char* abase = gring + (ln + 1) * L::kGRec + grp * 16;       // A: row = column ln of the m-tile, k = 8 channels of the lane group
#ifdef WW_DG_STAMPS
    unsigned long long dst[5] = {0, 0, 0, 0, 0}, dlast;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dlast) :: "memory");
#endif
    for (int gs = 0; gs < total; ++gs) {
        const int k = gs / kHSteps, s = gs - k * kHSteps;
        DSTAMP(4);                                                      // (behind the previous step's barrier)
        if (s == 0) rebuild(k);
        if (gs + 1 < total) fill_load(gs + 1);
        DSTAMP(0);
        const int y = kHRows * s + rg;
        f32x4 acc[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[mt][j] = 0.f;
        // 36 fragment steps it = ((dy*3 + dx)*2 + kb)*2 + mt; the fragments run PF steps ahead of their MFMAs in a pinned register ring
        // (left alone the compiler hoists dozens of the 36 loads at once and spills).  A mask row outside the image reads the ring's
        // all-zero eleventh row: no branch in the stream.
        const char* rowp[3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = y + 1 - dy;                              // gradient row of tap dy
            rowp[dy] = abase + ((yy >= 0 && yy < kTH) ? (k * kTH + yy) % kHGRing : kHGRing) * L::kGRow;
        }
        auto frag = [&](int it, int half) -> half8 {
            const int mt = it & 1, kb = (it >> 1) & 1, tp = it >> 2, dy = tp / 3, dx = tp - 3 * dy;
            return __builtin_bit_cast(half8, *reinterpret_cast<const u32x4*>(rowp[dy] + (16 * mt + 1 - dx) * L::kGRec + half * 128 + kb * 64));
        };
#ifndef WW_DG_PF
#define WW_DG_PF 3
#endif
        constexpr int PF = WW_DG_PF, RING = PF + 1;
        half8 fh[RING], fl[DENSE ? RING : 1];
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            fh[i] = frag(i, 0);
            if constexpr (DENSE) fl[i] = frag(i, 1);
        }
#pragma unroll
        for (int it = 0; it < 36; ++it) {
            if (it + PF < 36) {
                fh[(it + PF) % RING] = frag(it + PF, 0);
                if constexpr (DENSE) fl[(it + PF) % RING] = frag(it + PF, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            const int mt = it & 1, ks = it >> 1;
            const half8 a = fh[it % RING];
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bh[ks], acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bl[ks], acc[mt], 0, 0, 0);
            if constexpr (DENSE) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[it % RING], bh[ks], acc[mt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        DSTAMP(1);
        // epilogue: D register j <-> column 16 mt + 4 grp + j, lane & 15 <-> ci = 16 nt + ln.  That IS the A-operand layout of a
        // 16x16x32 block (m = ci, the lane group's eight k = its eight columns 4 grp + 0..3, 16 + 4 grp + 0..3), so the row's
        // dW1 / db1 contribution is one k = 32 block: A = dz1 2^-down as hi + lo, B[k][n = tap] = mel at the same eight columns
        // shifted by the tap (n = 9: ones -> db1) as hi + lo, three MFMAs into a float tile that lives for ONE clip (2560 positions)
        // and is then added to the double sums.  (Round 3 ran eight v_mfma_f64_16x16x4 per row here: 64 matrix-pipe cycles each on
        // this part, 30 % of the kernel's pipe time, issued as one dependent chain behind every step.)
        const float* melt = meltile + (k & 1) * L::kMelFloats;
        const uint32_t* s1 = s1rows + ((gs & 1) * kHRows + rg) * kTW;
        const int tap = ln, ty = tap / 3, tx = tap - 3 * ty;
        const float* mp = melt + (y + ty) * L::kMelRS + tx;           // + column: taps of position (y, col) are tile rows y..y+2, columns col..col+2
        u32x4 ah, al, mh, ml;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const u32x4 sw = *reinterpret_cast<const u32x4*>(s1 + 16 * mt + 4 * grp);
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                const int col = 16 * mt + 4 * grp + 2 * jp;
                const float a0 = ((sw[2 * jp] >> (16 * nt + ln)) & 1u) ? acc[mt][2 * jp] * kDzDown : 0.f;
                const float a1 = ((sw[2 * jp + 1] >> (16 * nt + ln)) & 1u) ? acc[mt][2 * jp + 1] * kDzDown : 0.f;
                const float m0 = tap < 9 ? mp[col] : (tap == 9 ? 1.0f : 0.f);
                const float m1 = tap < 9 ? mp[col + 1] : (tap == 9 ? 1.0f : 0.f);
                uint32_t hh, ll;
                split2(a0, a1, hh, ll); ah[2 * mt + jp] = hh; al[2 * mt + jp] = ll;
                split2(m0, m1, hh, ll); mh[2 * mt + jp] = hh; ml[2 * mt + jp] = ll;
            }
        }
        cacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, ah), __builtin_bit_cast(half8, mh), cacc, 0, 0, 0);
        cacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, ah), __builtin_bit_cast(half8, ml), cacc, 0, 0, 0);
        cacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, al), __builtin_bit_cast(half8, mh), cacc, 0, 0, 0);
        if (s == kHSteps - 1) {                                        // the clip's tile -> the double sums, with the clip's scale
            const double up = double(dscale) * double(1.0f / kDzDown) * double(tap < 9 ? mel_up : 1.0f);
            if (tid == 0) melmax[k & 1] = 0u;                          // read at s == 0 (rebuild), next written by load_mel(k + 2) at step 8 of clip k + 1
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dacc[j] += double(cacc[j]) * up;
                cacc[j] = 0.f;
            }
        }
        DSTAMP(2);
        if (gs + 1 < total) fill_store(gs + 1);
        DSTAMP(3);
        __syncthreads();
    }
    DSTAMP(4);
#ifdef WW_DG_STAMPS
    if (!DENSE && blockIdx.x == 7 && lane == 0)
        for (int i = 0; i < 5; ++i) atomicAdd(&g_dg_stamps[wave * 5 + i], dst[i]);
#endif
    // the four row waves of a ci tile, in fixed order -> this workgroup's partial
    double* red = reinterpret_cast<double*>(ldsb);                   // [8 waves][4][64]
#pragma unroll
    for (int j = 0; j < 4; ++j) red[(wave * 4 + j) * 64 + lane] = dacc[j];
    __syncthreads();
    if (rg == 0 && ln < 10) {
        float* outp = partial + int64_t(blockIdx.x) * L::kPartial;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double sum = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) sum += red[((2 * r + nt) * 4 + j) * 64 + lane];
            const int ci = 16 * nt + 4 * grp + j;
            if (ln < 9) outp[ci * 9 + ln] = float(sum);
            else outp[L::CIN * 9 + ci] = float(sum);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Data gradient of conv3 (3-conv model):  dz2 = [relu(conv2) > 0] * sum_{co,dy,dx} mask3[co][y-dy+1][x-dx+1] * G[co][ci][dy][dx],
// G = gp[b,co] * W3 rebuilt per clip (conv2_dgrad_h_kernel's arithmetic: mask exact, G 2^-eg as hi + lo, two MFMAs per block).
// K = 128 co x 9 taps = 36 k-steps per N-tile of 16 ci: 288 B-operand VGPRs -- split over TWO waves (kh = co half, 144 VGPRs each);
// 8 waves = (4 N-tiles) x (2 K halves), every wave runs both rows of a two-row step.  The halves meet through LDS: wave kh finishes
// row 2s + kh -- it parks the partial tile of the OTHER row in the exchange buffer, and after the step's barrier adds its partner's partial
// of its own row, applies relu(conv2)'s sign (float32 channels last, loaded a step ahead) and stores dz2 ALREADY SPLIT for the conv2
// kernels below: f16 records [row][column][64 hi | 64 lo] of dz2 2^-edz with one exponent per clip, edz = eg + 10 (|dz2| <= 1152 max|G|,
// so the records stay below 2^14 whatever the clip; typical values sit 2^6 lower, where hi and lo are still normal f16), dzs[clip] = 2^edz.
// Mask ring: 6 rows of 288-byte records (128 channels; conflict-free 16-byte row reads).
// ------------------------------------------------------------------------------------------------
constexpr int kH3Rows = 2, kH3Steps = kTH / kH3Rows, kH3Ring = 6;
struct Dg3H {
    static constexpr int CIN = 64, COUT = 128;
    static constexpr int kGRec = 288, kGRow = 34 * kGRec;
    static constexpr int kOffX = kH3Ring * kGRow;                            // exchange: [2 steps][4 nt][2 kh (writer)][2 mt][4 j][64 lanes] floats
    static constexpr int kOffLut = kOffX + 2 * 4 * 2 * 2 * 4 * 64 * 4;
    static constexpr int kLds = kOffLut + 256 * 16;
};
static_assert(Dg3H::kLds <= 160 * 1024 && Dg3H::kOffX % 16 == 0 && Dg3H::kOffLut % 16 == 0, "LDS map");

// W3 [128][64][3][3] -> wp[(((nt*2 + kh)*18 + ks)*64 + lane)*8 + j] = W3[64 kh + 32 kb + 8 g + j][16 nt + n][dy][dx], ks = (dy*3+dx)*2 + kb
__global__ void pack_dgrad3_h_dev_kernel(const float* __restrict__ w3, float* __restrict__ wp, unsigned int* __restrict__ wmax) {
    float m = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 4 * 2 * 18 * 64 * 8; i += gridDim.x * blockDim.x) {
        const int j = i & 7, lane = (i >> 3) & 63, ks = (i >> 9) % 18, kh = (i / (18 * 512)) & 1, nt = i / (2 * 18 * 512);
        const int kb = ks & 1, tap = ks >> 1, co = 64 * kh + 32 * kb + 8 * (lane >> 4) + j, ci = 16 * nt + (lane & 15);
        const float v = w3[(co * 64 + ci) * 9 + tap];
        wp[i] = v;
        m = fmaxf(m, __builtin_fabsf(v));
    }
    atomicMax(wmax, __float_as_uint(m));
}
// gp[b][co] = dpooled[b][co] * s, gpmax[b] = max_co |gp[b][co]|: one wave per clip, 128 channels
__global__ void gp_max128_kernel(const float* __restrict__ dpooled, float s, int n, float* __restrict__ gp, float* __restrict__ gpmax) {
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= n) return;
    const float v0 = dpooled[int64_t(b) * 128 + lane] * s, v1 = dpooled[int64_t(b) * 128 + 64 + lane] * s;
    gp[int64_t(b) * 128 + lane] = v0;
    gp[int64_t(b) * 128 + 64 + lane] = v1;
    float m = fmaxf(__builtin_fabsf(v0), __builtin_fabsf(v1));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (lane == 0) gpmax[b] = m;
}

__global__ __launch_bounds__(512, 2) void conv3_dgrad_h_kernel(const float* __restrict__ act2 /*[n][80][32][64]*/,
                                                               const uint8_t* __restrict__ maskbits /*conv3 mask image: mask3_byte*/,
                                                               const float* __restrict__ gp, const float* __restrict__ gpmax,
                                                               const float* __restrict__ wp, const float* __restrict__ w3max, int n,
                                                               _Float16* __restrict__ dz2h /*[n][80][32][64 hi | 64 lo]*/,
                                                               float* __restrict__ dzs /*[n]: 2^edz*/) {
    using L = Dg3H;
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    char* gring = ldsb;
    float* xch = reinterpret_cast<float*>(ldsb + L::kOffX);
    u32x4* lut = reinterpret_cast<u32x4*>(ldsb + L::kOffLut);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = wave & 3, kh = wave >> 2;                       // 16 ci x co half
    const int ln = lane & 15, grp = lane >> 4;
    const int my_clips = (n - int(blockIdx.x) + int(gridDim.x) - 1) / int(gridDim.x);
    const int total = my_clips * kH3Steps;

    for (int i = tid; i < L::kLds / 4; i += 512) reinterpret_cast<uint32_t*>(ldsb)[i] = 0u;      // halo columns stay zero
    __syncthreads();
    if (tid < 256) {
        u32x4 m;
#pragma unroll
        for (int d = 0; d < 4; ++d) m[d] = ((tid >> (2 * d)) & 1 ? 0x3c00u : 0u) | ((tid >> (2 * d + 1)) & 1 ? 0x3c000000u : 0u);
        lut[tid] = m;
    }
    // mask rows a step needs beyond what is in LDS: rows 0..2 (first step of a clip) or 2s+1, 2s+2; thread = (byte of 16, position + 32 per pass)
    const int mcg = tid & 15, mpos = tid >> 4;
    uint8_t mb[3];
    auto fill_load = [&](int gs) {
        const int k = gs / kH3Steps, s = gs - k * kH3Steps;
        const int64_t clip = int64_t(blockIdx.x) + int64_t(k) * gridDim.x;
        const int g0 = s == 0 ? 0 : 2 * s + 1, g1 = 2 * s + 2 < kTH ? 2 * s + 2 : kTH - 1;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int g = g0 + i;
            mb[i] = g <= g1 ? maskbits[mask3_byte(clip, g, mpos, mcg)] : uint8_t(0);
        }
    };
    auto fill_store = [&](int gs) {
        const int k = gs / kH3Steps, s = gs - k * kH3Steps;
        const int g0 = s == 0 ? 0 : 2 * s + 1, g1 = 2 * s + 2 < kTH ? 2 * s + 2 : kTH - 1;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int g = g0 + i;
            if (g <= g1) {
                const int slot = (k * kTH + g) % kH3Ring;
                *reinterpret_cast<u32x4*>(gring + slot * L::kGRow + (mpos + 1) * L::kGRec + mcg * 16) = lut[mb[i]];
            }
        }
    };

    half8 bh[18], bl[18];
    auto rebuild = [&](int k) {
        const int64_t clip = int64_t(blockIdx.x) + int64_t(k) * gridDim.x;
        const int eg = clampi(exp_of(gpmax[clip] * w3max[0]) - 12, -100, 100);
        const float down = pow2i(-eg);
        if (tid == 0) dzs[clip] = pow2i(eg + 10);
        float gv[16];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int q4 = 0; q4 < 2; ++q4) {
                const float4 v = *reinterpret_cast<const float4*>(gp + clip * L::COUT + 64 * kh + 32 * kb + 8 * grp + 4 * q4);
                gv[8 * kb + 4 * q4] = v.x * down; gv[8 * kb + 4 * q4 + 1] = v.y * down; gv[8 * kb + 4 * q4 + 2] = v.z * down; gv[8 * kb + 4 * q4 + 3] = v.w * down;
            }
        const float4* w4 = reinterpret_cast<const float4*>(wp) + (int64_t(nt * 2 + kh) * 18 * 64 + lane) * 2;
#pragma unroll
        for (int ks = 0; ks < 18; ++ks) {
            const float4 wa = w4[ks * 128], wb = w4[ks * 128 + 1];
            const int kb = ks & 1;
            u32x4 vh, vl;
            uint32_t hh, ll;
            split2(wa.x * gv[8 * kb], wa.y * gv[8 * kb + 1], hh, ll); vh[0] = hh; vl[0] = ll;
            split2(wa.z * gv[8 * kb + 2], wa.w * gv[8 * kb + 3], hh, ll); vh[1] = hh; vl[1] = ll;
            split2(wb.x * gv[8 * kb + 4], wb.y * gv[8 * kb + 5], hh, ll); vh[2] = hh; vl[2] = ll;
            split2(wb.z * gv[8 * kb + 6], wb.w * gv[8 * kb + 7], hh, ll); vh[3] = hh; vl[3] = ll;
            bh[ks] = __builtin_bit_cast(half8, vh);
            bl[ks] = __builtin_bit_cast(half8, vl);
        }
    };

    if (total > 0) fill_load(0);
    __syncthreads();                                               // the LUT is complete
    if (total > 0) fill_store(0);
    __syncthreads();
    const char* abase = gring + (ln + 1) * L::kGRec + kh * 128 + grp * 16;   // A: row = column ln of the m-tile, this wave's 64 co, 8 per lane group
    f32x4 keep[2];                                                  // this wave's own partial of the row it finishes (row 2s + kh), per m-tile
    float a2v[2][4];                                                // relu(conv2) at that row: columns 16 mt + 4 grp + j, channel 16 nt + ln
    int64_t keep_clip = 0;
    int keep_y = -1;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) { keep[mt][j] = 0.f; a2v[mt][j] = 0.f; }
    // finish the row parked by the previous step: partner's partial from the exchange buffer `buf`
    auto finish = [&](int buf) {
        if (keep_y < 0) return;
        const float* xp = xch + ((((buf * 4 + nt) * 2 + (kh ^ 1)) * 2) * 4) * 64 + lane;
        _Float16* out = dz2h + ((keep_clip * kTH + keep_y) * kTW + 4 * grp) * 128 + 16 * nt + ln;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = a2v[mt][j] > 0.f ? (keep[mt][j] + xp[(mt * 4 + j) * 64]) * 0x1p-10f : 0.f;   // da 2^-eg 2^-10 = dz2 2^-edz
                const _Float16 hi = static_cast<_Float16>(v);
                out[(16 * mt + j) * 128] = hi;
                out[(16 * mt + j) * 128 + 64] = static_cast<_Float16>(v - static_cast<float>(hi));
            }
    };
    for (int gs = 0; gs < total; ++gs) {
        const int k = gs / kH3Steps, s = gs - k * kH3Steps;
        const int64_t clip = int64_t(blockIdx.x) + int64_t(k) * gridDim.x;
        finish((gs + 1) & 1);                                       // the previous step's buffer
        if (s == 0) rebuild(k);
        if (gs + 1 < total) fill_load(gs + 1);
        // relu(conv2) of the row this wave will finish after the barrier (row 2s + kh)
        {
            const float* ap = act2 + ((clip * kTH + kH3Rows * s + kh) * kTW + 4 * grp) * L::CIN + 16 * nt + ln;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) a2v[mt][j] = ap[(16 * mt + j) * L::CIN];
        }
        f32x4 acc[2][2];                                            // [row of the step][m-tile]
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[r][mt][j] = 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int y = kH3Rows * s + r;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int yy = y + 1 - dy;                          // mask row of tap dy
                if (yy >= 0 && yy < kTH) {                          // wave-uniform
                    const char* rowp = abase + ((k * kTH + yy) % kH3Ring) * L::kGRow;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb) {
                            const int ks = (dy * 3 + dx) * 2 + kb;
#pragma unroll
                            for (int mt = 0; mt < 2; ++mt) {
                                const half8 a = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4*>(rowp + (16 * mt + 1 - dx) * L::kGRec + kb * 64));
                                acc[r][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bh[ks], acc[r][mt], 0, 0, 0);
                                acc[r][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bl[ks], acc[r][mt], 0, 0, 0);
                            }
                        }
                }
            }
        }
        // park the other row's partial for the partner, keep this wave's own row
        {
            float* xw = xch + (((((gs & 1) * 4 + nt) * 2 + kh) * 2) * 4) * 64 + lane;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) xw[(mt * 4 + j) * 64] = kh == 0 ? acc[1][mt][j] : acc[0][mt][j];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) keep[mt] = kh == 0 ? acc[0][mt] : acc[1][mt];
            keep_clip = clip;
            keep_y = kH3Rows * s + kh;
        }
        if (gs + 1 < total) fill_store(gs + 1);
        __syncthreads();
    }
    finish((total + 1) & 1);
}

// ------------------------------------------------------------------------------------------------
// test / diagnostic: the last conv's mask image in canonical order, out[b][row][col][C/8 bytes], byte cb = channels 8 cb .. 8 cb + 7
// ------------------------------------------------------------------------------------------------
__global__ void decode_mask_image_kernel(const uint8_t* __restrict__ img, int64_t n, int C, uint8_t* __restrict__ out) {
    const int64_t total = n * kTH * kTW * (C / 8);
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
        const int cb = int(i % (C / 8)), col = int((i / (C / 8)) % kTW), row = int((i / (C / 8) / kTW) % kTH);
        const int64_t clip = i / (int64_t(C / 8) * kTW * kTH);
        out[i] = img[C == 64 ? mask2_byte(clip, row, col, cb) : mask3_byte(clip, row, col, cb)];
    }
}
int launch_decode_mask_image(const uint32_t* img, int64_t n, int C, uint8_t* out, hipStream_t st) {
    hipLaunchKernelGGL(decode_mask_image_kernel, dim3(1024), dim3(256), 0, st, reinterpret_cast<const uint8_t*>(img), n, C, out);
    WW_HIP(hipGetLastError());
    return WW_OK;
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static int train_h_opt_in() {
    static std::mutex mu;
    static bool done[64] = {};
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    WW_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(WW_EINVAL, "device ordinal out of range");
    if (done[dev]) return WW_OK;
    WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv2_wgrad_h_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, WgH::kLds));
    WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv2_wgrad_h_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, WgHT<true>::kLds));
    WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv2_dgrad_h_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, DgH::kLds));
    WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv2_dgrad_h_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, DgHT<true>::kLds));
    WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wgrad_h_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, Wg3H::kLds));
    WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_dgrad_h_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, Dg3H::kLds));
    done[dev] = true;
    return WW_OK;
}

int launch_conv2_wgrad_h(const float* mel, const uint32_t* maskbits, const float* gp, int64_t n, int width, const float* packed,
                         float* partial, int grid, hipStream_t st) {
    if (int rc = train_h_opt_in()) return rc;
    const PackedLayout P = packed_layout(2);
    hipLaunchKernelGGL(conv2_wgrad_h_kernel<false>, dim3(grid), dim3(768), WgH::kLds, st, mel, reinterpret_cast<const uint8_t*>(maskbits), gp, int(n),
                       width, reinterpret_cast<const u32x4*>(packed + P.conv1_h), packed + P.conv1_hs, packed + P.conv1_b, packed + P.range,
                       partial);
    WW_HIP(hipGetLastError());
    return WW_OK;
}

// conv2 weight gradient of the 3-conv model from the pre-split dense gradient dz2h (conv3_dgrad_h_kernel)
int launch_conv2_wgrad_h_dense(const float* mel, const float* dz2h, const float* dzs, int64_t n, int width, const float* packed, float* partial,
                               int grid, hipStream_t st) {
    if (int rc = train_h_opt_in()) return rc;
    const PackedLayout P = packed_layout(3);
    hipLaunchKernelGGL(conv2_wgrad_h_kernel<true>, dim3(grid), dim3(768), WgHT<true>::kLds, st, mel, reinterpret_cast<const uint8_t*>(dz2h), dzs,
                       int(n), width, reinterpret_cast<const u32x4*>(packed + P.conv1_h), packed + P.conv1_hs, packed + P.conv1_b,
                       packed + P.range, partial);
    WW_HIP(hipGetLastError());
    return WW_OK;
}

// conv3 weight gradient of the 3-conv model: partial [grid][128*64*9 + 128] (lane order) -> dw [128][64][3][3], db [128]
int launch_conv3_wgrad_h(const float* act2, const float* apow2, const uint32_t* maskbits, const float* gp, int64_t n, float* partial,
                         float* dw, float* db, int grid, hipStream_t st) {
    if (int rc = train_h_opt_in()) return rc;
    hipLaunchKernelGGL(conv3_wgrad_h_kernel, dim3(grid), dim3(768), Wg3H::kLds, st, act2, apow2, reinterpret_cast<const uint8_t*>(maskbits), gp,
                       int(n), partial);
    hipLaunchKernelGGL(reduce_wgrad3_h_kernel, dim3(1024), dim3(256), 0, st, partial, grid, dw, db);
    WW_HIP(hipGetLastError());
    return WW_OK;
}

// dgrad scratch (floats): wp [2*18*64*8 | 4*2*18*64*8], wmax [4], gpmax [n]
static int64_t wp_floats(int n_conv) { return n_conv == 3 ? 4 * 2 * 18 * 64 * 8 : 2 * 18 * 64 * 8; }
int64_t dgrad_h_scratch_floats(int64_t n, int n_conv) { return wp_floats(n_conv) + 4 + ((n + 3) & ~int64_t(3)) + (n_conv == 3 ? wp_floats(2) + 4 + ((n + 3) & ~int64_t(3)) : 0); }
// 3-conv layout of the scratch: [conv3: wp, wmax, gpmax[n]] [conv2: wp, wmax] [dzs[n]]
float* dgrad_h_scratch2(float* scratch, int64_t n) { return scratch + wp_floats(3) + 4 + ((n + 3) & ~int64_t(3)); }
float* dgrad_h_dzs(float* scratch, int64_t n) { return dgrad_h_scratch2(scratch, n) + wp_floats(2) + 4; }

int launch_gp_max(const float* dpooled, float s, int64_t n, int n_conv, float* gp, float* scratch, hipStream_t st) {
    float* gpmax = scratch + wp_floats(n_conv) + 4;
    if (n_conv == 3) hipLaunchKernelGGL(gp_max128_kernel, dim3(unsigned((n + 3) / 4)), dim3(256), 0, st, dpooled, s, int(n), gp, gpmax);
    else hipLaunchKernelGGL(gp_max_kernel, dim3(unsigned((n + 3) / 4)), dim3(256), 0, st, dpooled, s, int(n), gp, gpmax);
    WW_HIP(hipGetLastError());
    return WW_OK;
}

int launch_conv3_dgrad_h(const float* act2, const uint32_t* maskbits, const float* gp, const float* w3, float* scratch, int64_t n, float* dz2h,
                         float* dzs, int grid, hipStream_t st) {
    if (int rc = train_h_opt_in()) return rc;
    float* wp = scratch;
    float* wmax = scratch + wp_floats(3);
    float* gpmax = wmax + 4;
    WW_HIP(hipMemsetAsync(wmax, 0, 16, st));
    hipLaunchKernelGGL(pack_dgrad3_h_dev_kernel, dim3(144), dim3(256), 0, st, w3, wp, reinterpret_cast<unsigned int*>(wmax));
    hipLaunchKernelGGL(conv3_dgrad_h_kernel, dim3(grid), dim3(512), Dg3H::kLds, st, act2, reinterpret_cast<const uint8_t*>(maskbits), gp, gpmax,
                       wp, wmax, int(n), reinterpret_cast<_Float16*>(dz2h), dzs);
    WW_HIP(hipGetLastError());
    return WW_OK;
}

int launch_conv2_dgrad_h(const float* mel, const uint32_t* maskbits, const uint32_t* bits1, const float* gp, const float* w2, float* scratch,
                         int64_t n, int width, float* partial, int grid, hipStream_t st) {
    if (int rc = train_h_opt_in()) return rc;
    float* wp = scratch;
    float* wmax = scratch + wp_floats(2);
    float* gpmax = wmax + 4;
    WW_HIP(hipMemsetAsync(wmax, 0, 16, st));
    hipLaunchKernelGGL(pack_dgrad_h_dev_kernel, dim3(36), dim3(256), 0, st, w2, wp, reinterpret_cast<unsigned int*>(wmax));
    hipLaunchKernelGGL(conv2_dgrad_h_kernel<false>, dim3(grid), dim3(512), DgH::kLds, st, mel, reinterpret_cast<const uint8_t*>(maskbits), bits1, gp,
                       gpmax, wp, wmax, int(n), width, partial);
    WW_HIP(hipGetLastError());
    return WW_OK;
}

// conv2 data gradient + conv1 weight gradient of the 3-conv model from the pre-split dense gradient dz2h; scratch2: wp [2*18*64*8] + wmax [4]
int launch_conv2_dgrad_h_dense(const float* mel, const float* dz2h, const float* dzs, const uint32_t* bits1, const float* w2, float* scratch2,
                               int64_t n, int width, float* partial, int grid, hipStream_t st) {
    if (int rc = train_h_opt_in()) return rc;
    float* wp = scratch2;
    float* wmax = scratch2 + wp_floats(2);
    WW_HIP(hipMemsetAsync(wmax, 0, 16, st));
    hipLaunchKernelGGL(pack_dgrad_h_dev_kernel, dim3(36), dim3(256), 0, st, w2, wp, reinterpret_cast<unsigned int*>(wmax));
    hipLaunchKernelGGL(conv2_dgrad_h_kernel<true>, dim3(grid), dim3(512), DgHT<true>::kLds, st, mel, reinterpret_cast<const uint8_t*>(dz2h), bits1,
                       static_cast<const float*>(nullptr), dzs, wp, wmax, int(n), width, partial);
    WW_HIP(hipGetLastError());
    return WW_OK;
}

}  // namespace ww
