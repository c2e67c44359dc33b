// Training step of both models: train-mode forward and the backward pass -- orchestration, the head's
// backward, and the EXACT-FP32 conv kernels (ww_set_train_math(WW_TRAIN_MATH_F32)); the default split-precision conv kernels are in ww_train_h.hip.
//
// Replaces, for one batch, the body of the reference's training loops
//     output = model(data); loss = criterion(output, target); loss.backward()
// (/root/reference/wakeword_training/train_wakeword.py:109-115, wakeword_training_script.py:250-258): the forward in train
// mode (nn.LSTM's inter-layer dropout and nn.Dropout before fc, train_wakeword.py:34-35,46-47) and d loss / d parameter
// for every parameter, given d loss / d logits (CrossEntropyLoss and the optimiser stay with the caller).
// Exact fp32 throughout (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 and fp32 VALU): gradients feed an optimiser.
//
// What makes the backward pass cheap to state: relu(conv2) is consumed only by the global average pool, so
//     d loss / d conv2[b, co, y, x] = gp[b, co] * [conv2[b, co, y, x] > 0],      gp = d loss / d pooled / (80 T)
// and with gm = that product
//     dW2[co][ci][dy][dx] = sum_{b,y,x} gm[b,co,y,x] * a1[b,ci,y+dy-1,x+dx-1]                    (conv2_wgrad_kernel)
//     da1[b,ci,y,x]       = sum_{co,dy,dx} gm[b,co,y-dy+1,x-dx+1] * W2[co][ci][dy][dx]           (conv2_dgrad_kernel)
//     dW1[ci][dy][dx]     = sum_{b,y,x} da1 * [a1 > 0] * mel[b,y+dy-1,x+dx-1],   db1 likewise     (its epilogue)
// a1 = relu(conv1) is recomputed from the log-mel tile where it is needed (0.74 MMAC per clip) instead of being stored.
// The forward keeps relu(conv2) (the exact-f32 cnn2_kernel<false> of ww_cnn.hip), the gate activations and dropout factors.
// The LSTM step with zero state has the closed form of ww_head.hip; its backward is elementwise (lstm_gates_bwd_kernel) plus
// six small GEMMs (mfma_gemm_kernel).  W_hh gradients are exactly zero (h0 = 0) and are left to the caller to zero-fill.
// Every reduction over clips runs in a fixed order (per-workgroup partials + reduce_partials_kernel): bitwise repeatable.
#include <mutex>

#include "ww_internal.h"

namespace ww {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kTH = WW_N_MELS, kTW = 32, kTRS = 34, kTMelRS = 36;
constexpr int kTMelFloats = (kTH + 2) * kTMelRS;          // log-mel tile with a zero halo: rows -1..80, columns -1..34

__device__ __forceinline__ float relu_t(float v) { return v < 0.f ? 0.f : v; }

// ------------------------------------------------------------------------------------------------
// weights on the device, torch layout -> MFMA operand order (the weights change every optimiser step)
// ------------------------------------------------------------------------------------------------
// conv weight [Cout][Cin][3][3] -> B operand of the forward kernel: out[(nt*KS + (c*3+dy)*3+dx)*64 + lane] =
// W[32 nt + (lane&31)][2c + (lane>>5)][dy][dx], KS = Cin/2*9 (pack_conv_b_operand of ww_tables.cpp)
__global__ void pack_conv_b_dev_kernel(const float* __restrict__ w, int cout, int cin, float* __restrict__ out) {
    const int ks = cin / 2 * 9, total = cout / 32 * ks * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int lane = i & 63, r = i >> 6, nt = r / ks, s = r - nt * ks, c = s / 9, dy = (s % 9) / 3, dx = s % 3;
        const int co = 32 * nt + (lane & 31), ci = 2 * c + (lane >> 5);
        out[i] = w[((co * cin + ci) * 3 + dy) * 3 + dx];
    }
}
// conv weight [COUT][CIN][3][3] -> B operands of the data-gradient kernel (a correlation of gm with the FLIPPED, transposed weights):
// out[((nt*KQ + kq)*144 + (c*3+dy)*3+dx)*64 + lane] = W[32 kq + 2c + (lane>>5)][32 nt + (lane&31)][2-dy][2-dx],  KQ = COUT/32
__global__ void pack_dgrad_b_dev_kernel(const float* __restrict__ w, int cout, int cin, float* __restrict__ out) {
    const int kqn = cout / 32, total = (cin / 32) * kqn * 144 * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int lane = i & 63, r = i >> 6, blk = r / 144, s = r - blk * 144, nt = blk / kqn, kq = blk - nt * kqn;
        const int c = s / 9, dy = (s % 9) / 3, dx = s % 3;
        const int co = 32 * kq + 2 * c + (lane >> 5), ci = 32 * nt + (lane & 31);
        out[i] = w[((co * cin + ci) * 3 + (2 - dy)) * 3 + (2 - dx)];
    }
}

// ------------------------------------------------------------------------------------------------
// pooled[b][co] = mean over (80, width) of relu(conv2), mid = [b][row][co][col] (cnn2_kernel<false>); one workgroup per clip
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool_mid_kernel(const float* __restrict__ mid, int n, int width, float* __restrict__ pooled) {
    const int clip = blockIdx.x, co = threadIdx.x >> 2, part = threadIdx.x & 3;
    const float* src = mid + (int64_t(clip) * kTH * 64 + co) * kTW + part * 8;
    float acc = 0.f;
    for (int y = 0; y < kTH; ++y) {
        const float4 a = *reinterpret_cast<const float4*>(src + int64_t(y) * 64 * kTW);
        const float4 b = *reinterpret_cast<const float4*>(src + int64_t(y) * 64 * kTW + 4);
        acc += ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w));       // columns beyond `width` hold zeros
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    if (part == 0) pooled[int64_t(clip) * 64 + co] = acc / float(kTH * width);
}

// ------------------------------------------------------------------------------------------------
// shared pieces of the two conv-backward kernels
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_mel_tile(const float* __restrict__ src, int width, float* __restrict__ melt, int tid, int nthreads) {
    for (int i = tid; i < kTH * width; i += nthreads) {
        const int y = i / width, xx = i - y * width;
        melt[(y + 1) * kTMelRS + xx + 1] = src[i];
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of a 3x3 convolution CIN -> COUT.  512 threads = 8 waves: wave = (block of 32 co x 32 ci, row split).
//   dW[co][ci][tap] += sum over the band's positions gm[co][y][x] * a[ci][y+dy-1][x+dx-1]
// as v_mfma_f32_32x32x2_f32 with M = co, N = ci, K = two neighbouring columns; a wave's nine 32x32 accumulators (144 VGPRs) stay in
// registers over ALL clips of the persistent workgroup.
//   gm   !DENSE: gp[b][co] * [oact > 0] (the conv feeds the global average pool: rank-one gradient);  DENSE: the tensor dz itself
//   a    FROM_MEL: relu(conv1) recomputed from the log-mel tile (CIN = 32);  else loaded from the stored activations iact
// All activation / gradient tensors are [b][row][channel][32 columns] floats (cnn2_kernel<false>'s layout).
// conv2: <32, 64, 8>: 2 blocks x 4 row pairs.  conv3: <64, 128, 4>: 8 blocks, each wave all 4 rows of the band.
// Output: one partial [COUT][CIN][9] + [COUT] (bias) per workgroup -> reduce_partials_kernel.
// ------------------------------------------------------------------------------------------------
template <int CIN, int COUT, int BAND>
struct WgradCfg {
    static constexpr int kBlocks = (COUT / 32) * (CIN / 32);
    static constexpr int kSplits = 8 / kBlocks;                   // row splits of a band over waves
    static constexpr int kRowsPerWave = BAND / kSplits;
    static constexpr int kActCi = (BAND + 2) * kTRS + 1;          // odd stride: conflict-free across ci
    static constexpr int kActFloats = CIN * kActCi;
    static constexpr int kGmFloats = BAND * COUT * 33;
    static constexpr int kLdsFloats = kTMelFloats + kActFloats + kGmFloats;
    static constexpr int kPartial = COUT * CIN * 9 + COUT;
    static_assert(kBlocks == 2 || kBlocks == 8, "eight waves: 2 blocks x 4 row splits, or 8 blocks");
    static_assert(BAND % kSplits == 0 && kLdsFloats * 4 <= 160 * 1024, "band does not fit");
};

template <int CIN, int COUT, int BAND, bool DENSE, bool FROM_MEL>
__global__ __launch_bounds__(512, 2) void conv_wgrad_kernel(const float* __restrict__ mel, const float* __restrict__ iact,
                                                            const float* __restrict__ oact_or_dz, const float* __restrict__ gp /*[n][COUT]*/,
                                                            int n, int width, const float* __restrict__ w1, const float* __restrict__ b1,
                                                            float* __restrict__ partial) {
    using Cfg = WgradCfg<CIN, COUT, BAND>;
    static_assert(!FROM_MEL || CIN == 32, "conv1 feeds a 32-channel input");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* melt = lds;
    float* act = melt + kTMelFloats;
    float* gm = act + Cfg::kActFloats;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = wave % Cfg::kBlocks, rp = wave / Cfg::kBlocks;
    const int cob = blk % (COUT / 32), cib = blk / (COUT / 32);
    const int m = lane & 31, kk = lane >> 5;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    constexpr int kGmPasses = COUT / 64;                          // a row of gm = COUT x 32 floats = kGmPasses x (512 threads x float4)
    float db[kGmPasses];
#pragma unroll
    for (int p = 0; p < kGmPasses; ++p) db[p] = 0.f;
    for (int i = tid; i < Cfg::kLdsFloats; i += 512) lds[i] = 0.f;

    const int lrow8 = tid >> 3, lcol = (tid & 7) * 4;              // loader role: channel (tid >> 3) + 64 * pass, four columns
    const int cx = tid & 31, cg = tid >> 5;                        // conv1 role: column, channels 2 cg, 2 cg + 1
    float w1r[2][9], b1r[2];
    if constexpr (FROM_MEL) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int t = 0; t < 9; ++t) w1r[u][t] = w1[(2 * cg + u) * 9 + t];
            b1r[u] = b1[2 * cg + u];
        }
    }

    for (int clip = blockIdx.x; clip < n; clip += gridDim.x) {
        __syncthreads();
        if constexpr (FROM_MEL) load_mel_tile(mel + int64_t(clip) * kTH * width, width, melt, tid, 512);
        float g[kGmPasses];
#pragma unroll
        for (int p = 0; p < kGmPasses; ++p) g[p] = DENSE ? 0.f : gp[int64_t(clip) * COUT + lrow8 + 64 * p];
        for (int band = 0; band < kTH / BAND; ++band) {
            const int y0 = band * BAND;
            __syncthreads();          // mel tile ready / the previous band's operand reads retired
#pragma unroll
            for (int r = 0; r < BAND; ++r)
#pragma unroll
                for (int p = 0; p < kGmPasses; ++p) {
                    const int co = lrow8 + 64 * p;
                    const float4 v = *reinterpret_cast<const float4*>(oact_or_dz + ((int64_t(clip) * kTH + y0 + r) * COUT + co) * kTW + lcol);
                    float* d = gm + (r * COUT + co) * 33 + lcol;
                    float g0, g1, g2, g3;
                    if constexpr (DENSE) { g0 = v.x; g1 = v.y; g2 = v.z; g3 = v.w; }
                    else { g0 = v.x > 0.f ? g[p] : 0.f; g1 = v.y > 0.f ? g[p] : 0.f; g2 = v.z > 0.f ? g[p] : 0.f; g3 = v.w > 0.f ? g[p] : 0.f; }
                    d[0] = g0; d[1] = g1; d[2] = g2; d[3] = g3;
                    db[p] += (g0 + g1) + (g2 + g3);
                }
            // input activations of the band: rows y0-1 .. y0+BAND, zero outside the image and beyond `width`
            if constexpr (FROM_MEL) {
#pragma unroll 1
                for (int q = 0; q < BAND + 2; ++q) {
                    const int y = y0 - 1 + q;
                    const bool inside = y >= 0 && y < kTH && cx < width;
                    const float* mp = melt + (inside ? y : 0) * kTMelRS + cx;
                    const float m00 = mp[0], m01 = mp[1], m02 = mp[2], m10 = mp[kTMelRS], m11 = mp[kTMelRS + 1], m12 = mp[kTMelRS + 2],
                                m20 = mp[2 * kTMelRS], m21 = mp[2 * kTMelRS + 1], m22 = mp[2 * kTMelRS + 2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        float v = b1r[u];
                        v = fmaf(w1r[u][0], m00, v); v = fmaf(w1r[u][1], m01, v); v = fmaf(w1r[u][2], m02, v);
                        v = fmaf(w1r[u][3], m10, v); v = fmaf(w1r[u][4], m11, v); v = fmaf(w1r[u][5], m12, v);
                        v = fmaf(w1r[u][6], m20, v); v = fmaf(w1r[u][7], m21, v); v = fmaf(w1r[u][8], m22, v);
                        act[(2 * cg + u) * Cfg::kActCi + q * kTRS + cx + 1] = inside ? relu_t(v) : 0.f;
                    }
                }
            } else {
#pragma unroll 1
                for (int q = 0; q < BAND + 2; ++q) {
                    const int y = y0 - 1 + q;
#pragma unroll
                    for (int p = 0; p < CIN / 64; ++p) {
                        const int ci = lrow8 + 64 * p;
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (y >= 0 && y < kTH) v = *reinterpret_cast<const float4*>(iact + ((int64_t(clip) * kTH + y) * CIN + ci) * kTW + lcol);
                        float* d = act + ci * Cfg::kActCi + q * kTRS + lcol + 1;
                        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                    }
                }
            }
            __syncthreads();
#pragma unroll 1
            for (int rr = 0; rr < Cfg::kRowsPerWave; ++rr) {
                const int qo = rp * Cfg::kRowsPerWave + rr;
                const float* ga = gm + (qo * COUT + 32 * cob + m) * 33 + kk;
                const float* ba = act + (32 * cib + m) * Cfg::kActCi + qo * kTRS + kk;
#pragma unroll 2
                for (int x0 = 0; x0 < kTW; x0 += 2) {
                    const float a = ga[x0];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx)
                            acc[dy * 3 + dx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, ba[dy * kTRS + x0 + dx], acc[dy * 3 + dx], 0, 0, 0);
                }
            }
        }
    }
    // sum the row splits through LDS in fixed order, tap by tap; write this workgroup's partial
    __syncthreads();
    float* xch = lds;                                        // [8 waves][16][64] = 8,192 floats per tap
    float* outp = partial + int64_t(blockIdx.x) * Cfg::kPartial;
#pragma unroll
    for (int t = 0; t < 9; ++t) {             // unrolled: a runtime index would put the accumulators in scratch memory
#pragma unroll
        for (int j = 0; j < 16; ++j) xch[(wave * 16 + j) * 64 + lane] = acc[t][j];
        __syncthreads();
        if (rp == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float sum = 0.f;
#pragma unroll
                for (int r = 0; r < Cfg::kSplits; ++r) sum += xch[((r * Cfg::kBlocks + blk) * 16 + j) * 64 + lane];
                const int co = 32 * cob + (j & 3) + 8 * (j >> 2) + 4 * kk, ci = 32 * cib + m;      // D: lane&31 = n, register j <-> row m
                outp[(co * CIN + ci) * 9 + t] = sum;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int p = 0; p < kGmPasses; ++p) {
        float d = db[p];
        d += __shfl_xor(d, 1);
        d += __shfl_xor(d, 2);
        d += __shfl_xor(d, 4);
        if ((tid & 7) == 0) outp[COUT * CIN * 9 + lrow8 + 64 * p] = d;
    }
}

// ------------------------------------------------------------------------------------------------
// Data gradient of a 3x3 convolution CIN -> COUT (gradient wrt its CIN-channel input) = correlation of gm (COUT channels) with the
// flipped, transposed weights: the forward kernels' mfma_rows4 structure, M = 32 columns, N = 32 of the CIN channels, K = (co, tap) in
// chunks of 144 k-steps = 32 co.  Waves = (K chunk kq of COUT/32, N tile nt of CIN/32, row group rg of 4 rows), 144 resident B-operand
// VGPRs each; the K chunks are summed through LDS in a fixed chain kq = last -> ... -> 0.
//   conv2: <32, 64>: 2 chunks x 1 tile x 2 row groups = 4 waves, 8-row bands.   conv3: <64, 128>: 4 x 2 x 1 = 8 waves, 4-row bands.
//   TO_CONV1: epilogue = conv1's sign recomputed from the log-mel tile, dW1 / db1 accumulated (float64) over all clips -> partial [32*9 + 32]
//   else:     epilogue = dz_out[b][row][ci][col] = da * [iact > 0]   (the dense gradient the next-lower layer's kernels take)
// ------------------------------------------------------------------------------------------------
template <int CIN, int COUT>
struct DgradCfg {
    static constexpr int kKq = COUT / 32, kNt = CIN / 32;
    static constexpr int kRg = 8 / (kKq * kNt) >= 2 ? 2 : 1;       // conv2: 2 row groups (4 waves), conv3: 1 (8 waves)
    static constexpr int kWaves = kKq * kNt * kRg;
    static constexpr int kBand = 4 * kRg;
    static constexpr int kGmFloats = COUT * (kBand + 2) * kTRS;
    static constexpr int kXchFloats = kRg * kNt * 4 * 16 * 64;
    static constexpr int kLdsFloats = kTMelFloats + kGmFloats + kXchFloats;
    static_assert(kLdsFloats * 4 <= 160 * 1024, "band does not fit");
};
constexpr int kDgPartial = 32 * 9 + 32;

template <int CIN, int COUT, bool DENSE, bool TO_CONV1>
__global__ __launch_bounds__((DgradCfg<CIN, COUT>::kWaves * 64), 1) void conv_dgrad_kernel(
    const float* __restrict__ mel, const float* __restrict__ iact, const float* __restrict__ oact_or_dz, const float* __restrict__ gp, int n,
    int width, const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ wB, float* __restrict__ out) {
    using Cfg = DgradCfg<CIN, COUT>;
    constexpr int T = Cfg::kWaves * 64, R = Cfg::kBand + 2;
    static_assert(T / 4 == COUT, "loader: COUT channels x 4 threads x 8 columns");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* melt = lds;
    float* gmt = melt + kTMelFloats;                        // [COUT][R][34]
    float* xch = gmt + Cfg::kGmFloats;                      // [rg][nt][4][16][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = wave % Cfg::kKq, nt = (wave / Cfg::kKq) % Cfg::kNt, rg = wave / (Cfg::kKq * Cfg::kNt);
    const int ln = lane & 31, h = lane >> 5;                // D: lane & 31 = input channel within the tile; A: lane & 31 = column

    float wb[144];
#pragma unroll
    for (int i = 0; i < 144; ++i) wb[i] = wB[((nt * Cfg::kKq + kq) * 144 + i) * 64 + lane];
    float w1r[9], b1r = 0.f;
    double dw1[9] = {0., 0., 0., 0., 0., 0., 0., 0., 0.}, db1 = 0.;
    if constexpr (TO_CONV1) {
#pragma unroll
        for (int t = 0; t < 9; ++t) w1r[t] = w1[ln * 9 + t];
        b1r = b1[ln];
    }
    for (int i = tid; i < Cfg::kLdsFloats; i += T) lds[i] = 0.f;

    const int lco = tid >> 2, lcol = (tid & 3) * 8;          // loader role: COUT channels x 4 threads x 8 columns
    const float* ap = gmt + ((32 * kq + h) * R + rg * 4) * kTRS + ln;
    float* xw = xch + ((rg * Cfg::kNt + nt) * 4) * 16 * 64 + lane;

    for (int clip = blockIdx.x; clip < n; clip += gridDim.x) {
        __syncthreads();
        if constexpr (TO_CONV1) load_mel_tile(mel + int64_t(clip) * kTH * width, width, melt, tid, T);
        const float g = DENSE ? 0.f : gp[int64_t(clip) * COUT + lco];
        for (int band = 0; band < kTH / Cfg::kBand; ++band) {
            const int y0 = band * Cfg::kBand;
            __syncthreads();
#pragma unroll 1
            for (int q = 0; q < R; ++q) {                     // gm band with halo: rows y0-1 .. y0+kBand, zero outside the image
                const int y = y0 - 1 + q;
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
                if (y >= 0 && y < kTH) {
                    const float* src = oact_or_dz + ((int64_t(clip) * kTH + y) * COUT + lco) * kTW + lcol;
                    a = *reinterpret_cast<const float4*>(src);
                    b = *reinterpret_cast<const float4*>(src + 4);
                }
                float* d = gmt + (lco * R + q) * kTRS + lcol + 1;
                if constexpr (DENSE) {
                    d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
                } else {
                    d[0] = a.x > 0.f ? g : 0.f; d[1] = a.y > 0.f ? g : 0.f; d[2] = a.z > 0.f ? g : 0.f; d[3] = a.w > 0.f ? g : 0.f;
                    d[4] = b.x > 0.f ? g : 0.f; d[5] = b.y > 0.f ? g : 0.f; d[6] = b.z > 0.f ? g : 0.f; d[7] = b.w > 0.f ? g : 0.f;
                }
            }
            __syncthreads();
            f32x16 acc[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[r][j] = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    float a[6];
#pragma unroll
                    for (int q = 0; q < 6; ++q) a[q] = ap[(2 * c * R + q) * kTRS + dx];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r + dy], wb[(c * 3 + dy) * 3 + dx], acc[r], 0, 0, 0);
                }
            }
            // K chunks summed in the fixed chain kq = last -> ... -> 0 through one exchange slot per (rg, nt)
#pragma unroll
            for (int st = Cfg::kKq - 1; st >= 1; --st) {
                if (kq == st) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int j = 0; j < 16; ++j) xw[(r * 16 + j) * 64] = acc[r][j];
                }
                __syncthreads();
                if (kq == st - 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int j = 0; j < 16; ++j) acc[r][j] += xw[(r * 16 + j) * 64];
                }
                if (st > 1) __syncthreads();
            }
            if (kq == 0) {
                // D: lane&31 = input channel 32 nt + ln, register j <-> column (j&3) + 8 (j>>2) + 4 h
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int y = y0 + rg * 4 + r;
                    if constexpr (TO_CONV1) {
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            const int col = (j & 3) + 8 * (j >> 2) + 4 * h;
                            const float da = acc[r][j];
                            const float* mp = melt + y * kTMelRS + col;          // taps: tile rows y..y+2, columns col..col+2
                            const float m0 = mp[0], m1 = mp[1], m2 = mp[2], m3 = mp[kTMelRS], m4 = mp[kTMelRS + 1], m5 = mp[kTMelRS + 2],
                                        m6 = mp[2 * kTMelRS], m7 = mp[2 * kTMelRS + 1], m8 = mp[2 * kTMelRS + 2];
                            float z = b1r;
                            z = fmaf(w1r[0], m0, z); z = fmaf(w1r[1], m1, z); z = fmaf(w1r[2], m2, z);
                            z = fmaf(w1r[3], m3, z); z = fmaf(w1r[4], m4, z); z = fmaf(w1r[5], m5, z);
                            z = fmaf(w1r[6], m6, z); z = fmaf(w1r[7], m7, z); z = fmaf(w1r[8], m8, z);
                            const double dz = (z > 0.f && col < width) ? double(da) : 0.0;
                            dw1[0] = fma(dz, double(m0), dw1[0]); dw1[1] = fma(dz, double(m1), dw1[1]); dw1[2] = fma(dz, double(m2), dw1[2]);
                            dw1[3] = fma(dz, double(m3), dw1[3]); dw1[4] = fma(dz, double(m4), dw1[4]); dw1[5] = fma(dz, double(m5), dw1[5]);
                            dw1[6] = fma(dz, double(m6), dw1[6]); dw1[7] = fma(dz, double(m7), dw1[7]); dw1[8] = fma(dz, double(m8), dw1[8]);
                            db1 += dz;
                        }
                    } else {
                        const int64_t at = ((int64_t(clip) * kTH + y) * CIN + 32 * nt + ln) * kTW + 4 * h;
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            const float4 ia = *reinterpret_cast<const float4*>(iact + at + 8 * gq);
                            float4 dz;
                            dz.x = ia.x > 0.f ? acc[r][4 * gq + 0] : 0.f;
                            dz.y = ia.y > 0.f ? acc[r][4 * gq + 1] : 0.f;
                            dz.z = ia.z > 0.f ? acc[r][4 * gq + 2] : 0.f;
                            dz.w = ia.w > 0.f ? acc[r][4 * gq + 3] : 0.f;
                            *reinterpret_cast<float4*>(out + at + 8 * gq) = dz;
                        }
                    }
                }
            }
        }
    }
    if constexpr (TO_CONV1) {
        // the two column halves (h) and the row groups, in fixed order -> this workgroup's partial
        __syncthreads();
        double* red = reinterpret_cast<double*>(xch);            // [rg][h][10][32] doubles
        if (kq == 0) {
#pragma unroll
            for (int t = 0; t < 9; ++t) red[((rg * 2 + h) * 10 + t) * 32 + ln] = dw1[t];
            red[((rg * 2 + h) * 10 + 9) * 32 + ln] = db1;
        }
        __syncthreads();
        float* outp = out + int64_t(blockIdx.x) * kDgPartial;
        if (tid < 32) {
#pragma unroll
            for (int t = 0; t < 10; ++t) {
                double sum = 0.0;
#pragma unroll
                for (int i = 0; i < 2 * Cfg::kRg; ++i) sum += red[(i * 10 + t) * 32 + tid];
                if (t < 9) outp[tid * 9 + t] = float(sum);
                else outp[32 * 9 + tid] = float(sum);
            }
        }
    }
}

// out[i] = sum over workgroups g of partial[g][i] in a fixed order: 256 threads = 64 outputs x 4 slices of the groups (g = 4 q + slice),
// the four slice sums added in order
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, int groups, int len, int w_len,
                                                              float* __restrict__ dw, float* __restrict__ db) {
    __shared__ float part[4][64];
    const int c = threadIdx.x & 63, sl = threadIdx.x >> 6;
    for (int i0 = blockIdx.x * 64; i0 < len; i0 += gridDim.x * 64) {
        const int i = i0 + c;
        float s = 0.f;
        if (i < len)
            for (int g = sl; g < groups; g += 4) s += partial[int64_t(g) * len + i];
        part[sl][c] = s;
        __syncthreads();
        if (sl == 0 && i < len) {                          // the weight part and the bias part go straight to their gradient tensors
            const float v = ((part[0][c] + part[1][c]) + part[2][c]) + part[3][c];
            if (i < w_len) dw[i] = v;
            else db[i - w_len] = v;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// LSTM step backward, elementwise part.  h = go * tc, hd = h * mask (saved: gi, gg, go, tc, mask); given dhd:
//   dh = dhd * mask;  d g_o = dh * tc * go (1 - go);  dc = dh * go * (1 - tc^2);
//   d g_i = dc * gg * gi (1 - gi);  d g_g = dc * gi * (1 - gg^2);  d g_f = 0 (c0 = 0)
// dg: [n][1024] in torch's gate-row order (i, f, g, o), so that dW_ih = dg^T x is the torch-layout gradient.
// ------------------------------------------------------------------------------------------------
__global__ void lstm_gates_bwd_kernel(const float* __restrict__ dhd, const float* __restrict__ gates, const float* __restrict__ mask, int n,
                                      float* __restrict__ dg) {
    const int64_t plane = int64_t(n) * kHidden;
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < plane; i += int64_t(gridDim.x) * blockDim.x) {
        const int64_t b = i / kHidden;
        const int u = int(i - b * kHidden);
        const float gi = gates[i], gg = gates[plane + i], go = gates[2 * plane + i], tc = gates[3 * plane + i];
        const float dh = dhd[i] * mask[i];
        const float dc = dh * go * (1.0f - tc * tc);
        float* o = dg + b * 4 * kHidden + u;
        o[0] = dc * gg * gi * (1.0f - gi);
        o[kHidden] = 0.f;
        o[2 * kHidden] = dc * gi * (1.0f - gg * gg);
        o[3 * kHidden] = dh * tc * go * (1.0f - go);
    }
}

// ------------------------------------------------------------------------------------------------
// C[M][N] = sum_k A(m,k) B(k,n) with general strides (the six small GEMMs of the head's backward: M*N*K <= 1.1e9) on
// v_mfma_f32_32x32x2_f32, exact fp32.  A workgroup of 4 waves owns a 32 x 32 tile of C; wave w takes the k pairs w, w + 4, ...
// (neighbouring k of the four waves share cache lines where k is the contiguous index), operands straight from global memory
// (one dword per lane and MFMA for each: A lane = (m, k parity), B lane = (n, k parity)); the four partial tiles are summed
// through LDS in the fixed order 0, 1, 2, 3.  Rows / columns beyond M / N load zeros and are not stored.
// ------------------------------------------------------------------------------------------------
// rowsum (nullable): rowsum[m] = sum_k A(m,k) in the same fixed order -- the bias gradients are the row sums of the weight-gradient GEMMs' A.
// Split K (round 3): the weight-gradient GEMMs have K = the batch (4096) and only 8-256 output tiles, i.e. one 4-wave workgroup per CU or
// fewer, each waiting out its own global loads (0.31 ms for the six GEMMs, ~11 % of the f32 matrix peak).  gridDim.z = S slices of K, each
// slice writes its partial tile to `part` ([S][M][N], + [S][M] row sums), combine_splitk_kernel adds the slices in the fixed order 0..S-1:
// S times the waves and loads in flight per CU, still bitwise repeatable.  S = 1 writes C directly.
__global__ __launch_bounds__(256) void mfma_gemm_kernel(const float* __restrict__ A, int64_t sam, int64_t sak, const float* __restrict__ B,
                                                        int64_t sbk, int64_t sbn, float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                        float* __restrict__ rowsum, float* __restrict__ part) {
    __shared__ float xch[3][16][64];
    __shared__ float rs[4][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mn = lane & 31, kk = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const bool a_ok = m0 + mn < M, b_ok = n0 + mn < N;
    const float* ap = A + int64_t(a_ok ? m0 + mn : 0) * sam;
    const float* bp = B + int64_t(b_ok ? n0 + mn : 0) * sbn;
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    float asum = 0.f;
    const int splits = int(gridDim.z), z = int(blockIdx.z);
    const int all_pairs = (K + 1) / 2, per = (all_pairs + splits - 1) / splits;
    const int p_begin = z * per, pairs = p_begin + per < all_pairs ? p_begin + per : all_pairs;      // this slice: k pairs [p_begin, pairs)
    int p = p_begin + wave;
    for (; p + 28 < pairs; p += 32) {                        // eight k pairs per trip: sixteen loads in flight under the MFMAs
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = 2 * (p + 4 * u) + kk;
            const bool k_ok = k < K;
            a[u] = (a_ok && k_ok) ? ap[int64_t(k) * sak] : 0.f;
            b[u] = (b_ok && k_ok) ? bp[int64_t(k) * sbk] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0); asum += a[u]; }
    }
    for (; p < pairs; p += 4) {
        const int k = 2 * p + kk;
        const bool k_ok = k < K;
        const float a = (a_ok && k_ok) ? ap[int64_t(k) * sak] : 0.f;
        const float b = (b_ok && k_ok) ? bp[int64_t(k) * sbk] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        asum += a;
    }
    if (rowsum != nullptr && blockIdx.x == 0) rs[wave][lane] = asum;
    if (wave > 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) xch[wave - 1][j][lane] = acc[j];
    }
    __syncthreads();
    if (wave == 0) {
        float* cdst = splits > 1 ? part + int64_t(z) * M * N : C;
        const int64_t cld = splits > 1 ? N : ldc;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float v = ((acc[j] + xch[0][j][lane]) + xch[1][j][lane]) + xch[2][j][lane];
            const int m = m0 + (j & 3) + 8 * (j >> 2) + 4 * kk;          // D: lane & 31 = n, register j <-> row m
            if (m < M && b_ok) cdst[int64_t(m) * cld + n0 + mn] = v;
        }
        if (rowsum != nullptr && blockIdx.x == 0 && lane < 32 && a_ok) {       // (wave 0..3) x (k parity 0, 1), fixed order
            const float r = (((rs[0][lane] + rs[0][lane + 32]) + (rs[1][lane] + rs[1][lane + 32])) + (rs[2][lane] + rs[2][lane + 32])) +
                            (rs[3][lane] + rs[3][lane + 32]);
            if (splits > 1) part[int64_t(splits) * M * N + int64_t(z) * M + m0 + lane] = r;
            else rowsum[m0 + lane] = r;
        }
    }
}

// C[m][n] = part[0][m][n] + part[1][m][n] + ... (fixed order); rowsum[m] likewise from the [S][M] block behind the tiles
__global__ void combine_splitk_kernel(const float* __restrict__ part, int splits, int M, int N, float* __restrict__ C, int64_t ldc,
                                      float* __restrict__ rowsum) {
    const int64_t mn = int64_t(M) * N;
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < mn + (rowsum ? M : 0); i += int64_t(gridDim.x) * blockDim.x) {
        if (i < mn) {
            float v = part[i];
            for (int zz = 1; zz < splits; ++zz) v += part[int64_t(zz) * mn + i];
            const int64_t m = i / N;
            C[m * ldc + (i - m * N)] = v;
        } else {
            const int64_t m = i - mn;
            const float* rp = part + int64_t(splits) * mn;
            float v = rp[m];
            for (int zz = 1; zz < splits; ++zz) v += rp[int64_t(zz) * M + m];
            rowsum[m] = v;
        }
    }
}

// `part`: scratch of at least 8 * (M * N + M) floats for the GEMMs that are split (the reused gradient-partial block of the workspace)
static void sgemm(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn, float* C, int64_t ldc, int M, int N, int K,
                  hipStream_t st, float* rowsum = nullptr, float* part = nullptr) {
    const int tiles = ((N + 31) / 32) * ((M + 31) / 32);
    int splits = 1;
    if (part != nullptr && K >= 512) {
        while (splits < 8 && tiles * splits < 1024 && K / (2 * splits) >= 128) splits *= 2;     // ~4 workgroups per CU, >= 64 k pairs per slice
    }
    hipLaunchKernelGGL(mfma_gemm_kernel, dim3((N + 31) / 32, (M + 31) / 32, splits), dim3(256), 0, st, A, sam, sak, B, sbk, sbn, C, ldc, M, N, K, rowsum,
                       part);
    if (splits > 1) {
        const int64_t work = int64_t(M) * N + (rowsum ? M : 0);
        hipLaunchKernelGGL(combine_splitk_kernel, dim3(unsigned((work + 255) / 256 < 1024 ? (work + 255) / 256 : 1024)), dim3(256), 0, st, part, splits, M,
                           N, C, ldc, rowsum);
    }
}

// gp[b][co] = dpooled[b][co] / (80 * width)
__global__ void scale_kernel(const float* __restrict__ x, float s, int64_t len, float* __restrict__ out) {
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < len; i += int64_t(gridDim.x) * blockDim.x) out[i] = x[i] * s;
}

// ------------------------------------------------------------------------------------------------
// workspace and orchestration
// ------------------------------------------------------------------------------------------------
int launch_cnn2_f32_mid(const float* mel, int64_t n, int width, const float* w1, const float* b1, const float* wB, const float* b2, float* mid,
                        hipStream_t stream);   // ww_cnn.hip
int launch_cnn3_f32_store(const float* mid2, int64_t n, int width, const float* wB, const float* b3, float* pooled, float* mid3,
                          hipStream_t stream);   // ww_cnn.hip
int launch_lstm_fc_train(const float* pooled, int64_t n, int C, const float* w_ih0, const float* b_ih0, const float* b_hh0,
                         const float* w_ih1, const float* b_ih1, const float* b_hh1, const float* fcw, const float* fcb, float* packed_ws,
                         float* gates0, float* mask0, float* hd0, float* gates1, float* mask1, float* hd1, float p_lstm, float p_fc,
                         uint64_t seed, float* logits, hipStream_t stream);   // ww_head.hip

constexpr int kWg2Partial = WgradCfg<32, 64, 8>::kPartial, kWg3Partial = WgradCfg<64, 128, 4>::kPartial;
constexpr size_t kWg2Lds = sizeof(float) * WgradCfg<32, 64, 8>::kLdsFloats, kWg3Lds = sizeof(float) * WgradCfg<64, 128, 4>::kLdsFloats;
constexpr size_t kDg2Lds = sizeof(float) * DgradCfg<32, 64>::kLdsFloats, kDg3Lds = sizeof(float) * DgradCfg<64, 128>::kLdsFloats;
constexpr int kMaxGroups = 256;                      // grids never exceed the CU count (256 on MI355X)

struct TrainWs {
    float *mid2, *mid3, *dz2, *pooled, *gates0, *mask0, *hd0, *gates1, *mask1, *hd1, *lstm_packed, *conv2_b_op, *conv3_b_op, *dgrad2_b_op, *dgrad3_b_op;
    float *dhd1, *dg1, *dhd0, *dg0, *dpooled, *gp, *partial;
    uint32_t* maskbits;                                  // [n][80][32][c_last / 32]: [relu(last conv) > 0]
    float* wpk;                                          // split precision: packed image written on the device
    uint32_t* bits1;                                     // split precision: [n][80][32] sign bits of conv1
    float* dgh;                                          // scratch of the split-precision data-gradient kernel
    float* apow2;                                        // 3-conv model, split precision: 2^a2 per clip (cnn2w_kernel<3> -> cnn3w_kernel)
    int64_t total;
};
static int64_t a256(int64_t floats) { return (floats * 4 + 255) / 256 * 64; }      // floats, 256-byte granules
// Which arithmetic wrote a workspace: the layout and the meaning of its contents follow the forward's mode, so the backward (and the
// diagnostics) of a step must not run under another one.  Host-side note per workspace pointer, written by train_forward; a backward
// under a different mode fails with WW_EINVAL instead of reading bit images that are not there.  (Bounded: the newest 64 workspaces.)
static std::mutex g_ws_mu;
static struct { const void* ws; int mode; } g_ws_mode[64];
static int g_ws_next = 0;
static void note_workspace_mode(const void* ws, int mode) {
    std::lock_guard<std::mutex> lock(g_ws_mu);
    for (auto& e : g_ws_mode)
        if (e.ws == ws) { e.mode = mode; return; }
    g_ws_mode[g_ws_next] = {ws, mode};
    g_ws_next = (g_ws_next + 1) % 64;
}
static int workspace_mode(const void* ws) {            // -1: unknown (never seen, or evicted)
    std::lock_guard<std::mutex> lock(g_ws_mu);
    for (auto& e : g_ws_mode)
        if (e.ws == ws && ws != nullptr) return e.mode;
    return -1;
}

static TrainWs carve_train(void* base, int64_t n, int n_conv, int mode) {
    TrainWs w{};
    float* p = static_cast<float*>(base);
    int64_t o = 0;
    auto take = [&](int64_t floats) { float* at = p ? p + o : nullptr; o += a256(floats); return at; };
    const int c_last = n_conv == 3 ? 128 : 64;
    // the activations the exact-fp32 kernels keep; under the split arithmetic only the 3-conv model's relu(conv2) is stored (the rest
    // travels as bit images), so the layout depends on the arithmetic: query, forward and backward must agree on ww_set_train_math
    const bool split = (mode >= 0 ? mode : train_math_mode()) == WW_TRAIN_MATH_F16X3;   // -1 (diagnostics on a workspace nobody noted): the process default
    w.mid2 = (!split || n_conv == 3) ? take(n * kTH * 64 * kTW) : nullptr;
    w.mid3 = (n_conv == 3 && !split) ? take(n * kTH * 128 * kTW) : nullptr;
    w.dz2 = n_conv == 3 ? take(n * kTH * 64 * kTW) : nullptr;
    w.pooled = take(n * c_last);
    w.gates0 = take(4 * n * kHidden); w.mask0 = take(n * kHidden); w.hd0 = take(n * kHidden);
    w.gates1 = take(4 * n * kHidden); w.mask1 = take(n * kHidden); w.hd1 = take(n * kHidden);
    w.lstm_packed = take((c_last + kHidden) * kGateCols + 2 * kGateCols);
    w.conv2_b_op = take(2 * 144 * 64);
    w.conv3_b_op = n_conv == 3 ? take(4 * 288 * 64) : nullptr;
    w.dgrad2_b_op = take(2 * 144 * 64);
    w.dgrad3_b_op = n_conv == 3 ? take(8 * 144 * 64) : nullptr;
    w.dhd1 = take(n * kHidden); w.dg1 = take(n * 4 * kHidden); w.dhd0 = take(n * kHidden); w.dg0 = take(n * 4 * kHidden);
    w.dpooled = take(n * c_last); w.gp = take(n * c_last);
    w.partial = take(int64_t(kMaxGroups) * (n_conv == 3 ? kWg3Partial : kWg2Partial));      // reused by every partial-producing kernel in turn
    w.maskbits = reinterpret_cast<uint32_t*>(take(n * kTH * kTW * (c_last / 32)));
    w.wpk = take(packed_layout(n_conv).total);
    w.bits1 = reinterpret_cast<uint32_t*>(take(n * kTH * kTW));
    w.apow2 = n_conv == 3 ? take(n) : nullptr;
    w.dgh = take(dgrad_h_scratch_floats(n, n_conv));
    w.total = o * 4;
    return w;
}

int64_t train_workspace_bytes(int64_t n, int n_conv, int mode) { return carve_train(nullptr, n, n_conv, mode).total; }

static int train_opt_in() {
    static std::mutex mu;
    static bool done[64] = {};
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    WW_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(WW_EINVAL, "device ordinal out of range");
    if (done[dev]) return WW_OK;
    auto opt = [](const void* f, int floats) { return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, floats * 4); };
    WW_HIP(opt(reinterpret_cast<const void*>(conv_wgrad_kernel<32, 64, 8, false, true>), WgradCfg<32, 64, 8>::kLdsFloats));
    WW_HIP(opt(reinterpret_cast<const void*>(conv_wgrad_kernel<32, 64, 8, true, true>), WgradCfg<32, 64, 8>::kLdsFloats));
    WW_HIP(opt(reinterpret_cast<const void*>(conv_wgrad_kernel<64, 128, 4, false, false>), WgradCfg<64, 128, 4>::kLdsFloats));
    WW_HIP(opt(reinterpret_cast<const void*>(conv_dgrad_kernel<32, 64, false, true>), DgradCfg<32, 64>::kLdsFloats));
    WW_HIP(opt(reinterpret_cast<const void*>(conv_dgrad_kernel<32, 64, true, true>), DgradCfg<32, 64>::kLdsFloats));
    WW_HIP(opt(reinterpret_cast<const void*>(conv_dgrad_kernel<64, 128, false, false>), DgradCfg<64, 128>::kLdsFloats));
    done[dev] = true;
    return WW_OK;
}

// test / diagnostic: copies of the dropout factors the last forward on this workspace drew ([n][256] each)
int train_masks(const void* workspace, int64_t n, int n_conv, float* mask0, float* mask1, hipStream_t st) {
    TrainWs w = carve_train(const_cast<void*>(workspace), n, n_conv, workspace_mode(workspace));
    WW_HIP(hipMemcpyAsync(mask0, w.mask0, sizeof(float) * n * kHidden, hipMemcpyDeviceToDevice, st));
    WW_HIP(hipMemcpyAsync(mask1, w.mask1, sizeof(float) * n * kHidden, hipMemcpyDeviceToDevice, st));
    return WW_OK;
}

// test / diagnostic: the packed image the last split-precision forward of the 2-conv model wrote on the device
int train_packed_image(const void* workspace, int64_t n, int n_conv, float* img, hipStream_t st) {
    if (workspace_mode(workspace) == WW_TRAIN_MATH_F32) return fail(WW_EINVAL, "this workspace's forward ran in exact fp32: no packed image");
    TrainWs w = carve_train(const_cast<void*>(workspace), n, n_conv, workspace_mode(workspace));
    WW_HIP(hipMemcpyAsync(img, w.wpk, sizeof(float) * packed_layout(n_conv).total, hipMemcpyDeviceToDevice, st));
    return WW_OK;
}

// test / diagnostic: the ReLU bit images the last split-precision forward left in the workspace, in canonical order
int train_bit_images(const void* workspace, int64_t n, int n_conv, uint8_t* mask_last, uint32_t* sign1, hipStream_t st) {
    if (workspace_mode(workspace) == WW_TRAIN_MATH_F32) return fail(WW_EINVAL, "this workspace's forward ran in exact fp32: no bit images");
    TrainWs w = carve_train(const_cast<void*>(workspace), n, n_conv, workspace_mode(workspace));
    if (int rc = launch_decode_mask_image(w.maskbits, n, n_conv == 3 ? 128 : 64, mask_last, st)) return rc;
    WW_HIP(hipMemcpyAsync(sign1, w.bits1, sizeof(uint32_t) * n * kTH * kTW, hipMemcpyDeviceToDevice, st));
    return WW_OK;
}

int train_forward(const float* mel, int64_t n, int width, const ww_train_params* p, float p_lstm, float p_fc, uint64_t seed, int mode,
                  void* workspace, int64_t workspace_bytes, float* logits, hipStream_t st) {
    if (device_cu_count() > kMaxGroups) return fail(WW_EUNSUPPORTED, "more than 256 CUs: the workspace is sized for 256 partials");
    const int nc = p->n_conv, c_last = nc == 3 ? 128 : 64;
    TrainWs w = carve_train(workspace, n, nc, mode);
    if (workspace_bytes < w.total)
        return fail(WW_EINVAL, "training workspace of %lld bytes, but %lld clips of the %d-conv model under train math %d need %lld "
                               "(ww_train_workspace_bytes with the same mode)", (long long)workspace_bytes, (long long)n, nc, mode, (long long)w.total);
    note_workspace_mode(workspace, mode);
    if (mode == WW_TRAIN_MATH_F16X3) {
        // split precision: the inference kernels (convs as 1-D Winograd on the f16 matrix cores) with the ReLU masks as extra outputs.
        // 2 convs: relu(conv2) itself is never stored.  3 convs: relu(conv2) stays as float32 [row][column][64].
        // The backward pass must run under the same arithmetic (it reads the bit images).
        WW_HIP(hipMemsetAsync(w.wpk, 0, sizeof(float) * packed_layout(nc).total, st));
        if (int rc = launch_pack_conv_h_dev(p, w.wpk, st)) return rc;
        if (nc == 2) {
            if (int rc = launch_cnn2w_pool_bits(mel, n, width, w.wpk, w.pooled, w.maskbits, w.bits1, st)) return rc;
        } else {
            if (int rc = launch_cnn3w_pool_bits(mel, n, width, w.wpk, w.mid2, w.apow2, w.pooled, w.maskbits, w.bits1, st)) return rc;
        }
        return launch_lstm_fc_train(w.pooled, n, c_last, p->lstm_weight_ih[0], p->lstm_bias_ih[0], p->lstm_bias_hh[0], p->lstm_weight_ih[1],
                                    p->lstm_bias_ih[1], p->lstm_bias_hh[1], p->fc_weight, p->fc_bias, w.lstm_packed, w.gates0, w.mask0, w.hd0,
                                    w.gates1, w.mask1, w.hd1, p_lstm, p_fc, seed, logits, st);
    }
    hipLaunchKernelGGL(pack_conv_b_dev_kernel, dim3(72), dim3(256), 0, st, p->conv_weight[1], 64, 32, w.conv2_b_op);
    WW_HIP(hipGetLastError());
    if (int rc = launch_cnn2_f32_mid(mel, n, width, p->conv_weight[0], p->conv_bias[0], w.conv2_b_op, p->conv_bias[1], w.mid2, st)) return rc;
    if (nc == 3) {
        hipLaunchKernelGGL(pack_conv_b_dev_kernel, dim3(288), dim3(256), 0, st, p->conv_weight[2], 128, 64, w.conv3_b_op);
        WW_HIP(hipGetLastError());
        if (int rc = launch_cnn3_f32_store(w.mid2, n, width, w.conv3_b_op, p->conv_bias[2], w.pooled, w.mid3, st)) return rc;
    } else {
        hipLaunchKernelGGL(pool_mid_kernel, dim3(int(n)), dim3(256), 0, st, w.mid2, int(n), width, w.pooled);
        WW_HIP(hipGetLastError());
    }
    return launch_lstm_fc_train(w.pooled, n, c_last, p->lstm_weight_ih[0], p->lstm_bias_ih[0], p->lstm_bias_hh[0], p->lstm_weight_ih[1],
                                p->lstm_bias_ih[1], p->lstm_bias_hh[1], p->fc_weight, p->fc_bias, w.lstm_packed, w.gates0, w.mask0, w.hd0,
                                w.gates1, w.mask1, w.hd1, p_lstm, p_fc, seed, logits, st);
}

// partial[groups][len] -> the weight part and the bias part of the gradient
static int reduce_to(const TrainWs& w, int groups, int len, int w_len, float* dw, float* db, int b_len, hipStream_t st) {
    (void)b_len;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((len + 63) / 64 < 1024 ? (len + 63) / 64 : 1024), dim3(256), 0, st, w.partial, groups, len, w_len, dw, db);
    WW_HIP(hipGetLastError());
    return WW_OK;
}

int train_backward(const float* mel, int64_t n, int width, const ww_train_params* p, const float* dlogits, int mode, void* workspace,
                   int64_t workspace_bytes, const ww_train_grads* g, hipStream_t st) {
    if (int rc = train_opt_in()) return rc;
    const int nc = p->n_conv, c_last = nc == 3 ? 128 : 64;
    const int cus = device_cu_count();
    const int grid = int(n < cus ? n : cus);         // one persistent workgroup per CU (117-149 KB of LDS each)
    if (const int fwd = workspace_mode(workspace); fwd >= 0 && fwd != mode)
        return fail(WW_EINVAL, "the forward of this workspace ran under train math %d, the backward is asked under %d: pass the same mode "
                               "to both halves of a step", fwd, mode);
    TrainWs w = carve_train(workspace, n, nc, mode);
    if (workspace_bytes < w.total)
        return fail(WW_EINVAL, "training workspace of %lld bytes, but %lld clips of the %d-conv model under train math %d need %lld",
                    (long long)workspace_bytes, (long long)n, nc, mode, (long long)w.total);
    const int N = int(n), H = kHidden;
    const bool bits = mode == WW_TRAIN_MATH_F16X3;       // the forward left bit images (and channels-last relu(conv2))
    const bool split = bits && nc == 2;                               // the kernels of ww_train_h.hip
    // fc: dW = dlogits^T hd1, db = the row sums of dlogits^T, dhd1 = dlogits W_fc
    // the split-K scratch: the gradient-partial block, not yet in use at this point of the backward (largest need: 8 x (1024 x 256 + 1024) floats)
    float* gpart = w.partial;
    static_assert(int64_t(kMaxGroups) * kWg2Partial >= 8 * (1024 * 256 + 1024), "split-K scratch fits the partial block");
    sgemm(dlogits, 1, 2, w.hd1, H, 1, g->fc_weight, H, 2, H, N, st, g->fc_bias, gpart);
    sgemm(dlogits, 2, 1, p->fc_weight, H, 1, w.dhd1, H, N, H, 2, st);
    // layer 1
    hipLaunchKernelGGL(lstm_gates_bwd_kernel, dim3(1024), dim3(256), 0, st, w.dhd1, w.gates1, w.mask1, N, w.dg1);
    sgemm(w.dg1, 1, 4 * H, w.hd0, H, 1, g->lstm_weight_ih[1], H, 4 * H, H, N, st, g->lstm_bias[1], gpart);   // [1024][256] = dg1^T hd0; bias = its A's row sums
    sgemm(w.dg1, 4 * H, 1, p->lstm_weight_ih[1], H, 1, w.dhd0, H, N, H, 4 * H, st, nullptr, gpart);            // [n][256] = dg1 W_ih_l1
    // layer 0
    hipLaunchKernelGGL(lstm_gates_bwd_kernel, dim3(1024), dim3(256), 0, st, w.dhd0, w.gates0, w.mask0, N, w.dg0);
    sgemm(w.dg0, 1, 4 * H, w.pooled, c_last, 1, g->lstm_weight_ih[0], c_last, 4 * H, c_last, N, st, g->lstm_bias[0], gpart);   // [1024][C] = dg0^T pooled
    sgemm(w.dg0, 4 * H, 1, p->lstm_weight_ih[0], c_last, 1, w.dpooled, c_last, N, c_last, 4 * H, st, nullptr, gpart);      // [n][C] = dg0 W_ih_l0
    if (bits) {
        if (int rc = launch_gp_max(w.dpooled, 1.0f / float(kTH * width), n, nc, w.gp, w.dgh, st)) return rc;
    } else {
        hipLaunchKernelGGL(scale_kernel, dim3(256), dim3(256), 0, st, w.dpooled, 1.0f / float(kTH * width), n * c_last, w.gp);
        WW_HIP(hipGetLastError());
    }
    // conv stack, top down.  The last conv feeds the pool (rank-one gradient gp * [act > 0]); below it the gradient is dense.
    const float* w1 = p->conv_weight[0];
    const float* b1 = p->conv_bias[0];
    if (nc == 3) {
        if (!bits) hipLaunchKernelGGL(pack_dgrad_b_dev_kernel, dim3(288), dim3(256), 0, st, p->conv_weight[2], 128, 64, w.dgrad3_b_op);
        if (bits) {
            if (int rc = launch_conv3_wgrad_h(w.mid2, w.apow2, w.maskbits, w.gp, n, w.partial, g->conv_weight[2], g->conv_bias[2], grid, st)) return rc;
        } else {
            hipLaunchKernelGGL((conv_wgrad_kernel<64, 128, 4, false, false>), dim3(grid), dim3(512), kWg3Lds, st,
                               mel, w.mid2, w.mid3, w.gp, N, width, w1, b1, w.partial);
            WW_HIP(hipGetLastError());
            if (int rc = reduce_to(w, grid, kWg3Partial, 128 * 64 * 9, g->conv_weight[2], g->conv_bias[2], 128, st)) return rc;
        }
        if (bits) {
            if (int rc = launch_conv3_dgrad_h(w.mid2, w.maskbits, w.gp, p->conv_weight[2], w.dgh, n, w.dz2, dgrad_h_dzs(w.dgh, n), grid, st)) return rc;
        } else
            hipLaunchKernelGGL((conv_dgrad_kernel<64, 128, false, false>), dim3(grid), dim3(512), kDg3Lds, st,
                               mel, w.mid2, w.mid3, w.gp, N, width, w1, b1, w.dgrad3_b_op, w.dz2);
        WW_HIP(hipGetLastError());
    }
    if (!bits) hipLaunchKernelGGL(pack_dgrad_b_dev_kernel, dim3(72), dim3(256), 0, st, p->conv_weight[1], 64, 32, w.dgrad2_b_op);
    if (nc == 3 && bits) {
        if (int rc = launch_conv2_wgrad_h_dense(mel, w.dz2, dgrad_h_dzs(w.dgh, n), n, width, w.wpk, w.partial, grid, st)) return rc;
    } else if (nc == 3)
        hipLaunchKernelGGL((conv_wgrad_kernel<32, 64, 8, true, true>), dim3(grid), dim3(512), kWg2Lds, st,
                           mel, static_cast<const float*>(nullptr), w.dz2, w.gp, N, width, w1, b1, w.partial);
    else if (split) {
        if (int rc = launch_conv2_wgrad_h(mel, w.maskbits, w.gp, n, width, w.wpk, w.partial, grid, st)) return rc;
    } else
        hipLaunchKernelGGL((conv_wgrad_kernel<32, 64, 8, false, true>), dim3(grid), dim3(512), kWg2Lds, st,
                           mel, static_cast<const float*>(nullptr), w.mid2, w.gp, N, width, w1, b1, w.partial);
    WW_HIP(hipGetLastError());
    if (int rc = reduce_to(w, grid, kWg2Partial, 64 * 32 * 9, g->conv_weight[1], g->conv_bias[1], 64, st)) return rc;
    if (nc == 3 && bits) {
        if (int rc = launch_conv2_dgrad_h_dense(mel, w.dz2, dgrad_h_dzs(w.dgh, n), w.bits1, p->conv_weight[1], dgrad_h_scratch2(w.dgh, n), n, width,
                                                w.partial, grid, st)) return rc;
    } else if (nc == 3)
        hipLaunchKernelGGL((conv_dgrad_kernel<32, 64, true, true>), dim3(grid), dim3(256), kDg2Lds, st,
                           mel, static_cast<const float*>(nullptr), w.dz2, w.gp, N, width, w1, b1, w.dgrad2_b_op, w.partial);
    else if (split) {
        if (int rc = launch_conv2_dgrad_h(mel, w.maskbits, w.bits1, w.gp, p->conv_weight[1], w.dgh, n, width, w.partial, grid, st)) return rc;
    } else
        hipLaunchKernelGGL((conv_dgrad_kernel<32, 64, false, true>), dim3(grid), dim3(256), kDg2Lds, st,
                           mel, static_cast<const float*>(nullptr), w.mid2, w.gp, N, width, w1, b1, w.dgrad2_b_op, w.partial);
    WW_HIP(hipGetLastError());
    return reduce_to(w, grid, kDgPartial, 32 * 9, g->conv_weight[0], g->conv_bias[0], 32, st);
}

}  // namespace ww
