// K3: 2-layer LSTM, ONE time step, zero (h0, c0), + Linear(256, 2) [+ softmax p(wakeword)].
//
// Replaces `lstm_out, _ = self.lstm(x.unsqueeze(1)); x = lstm_out[:, -1, :]; x = self.fc(self.dropout(x))`
// (/root/reference/wakeword_training/train_wakeword.py:42-48, wakeword_training_script.py:175-182; eval mode).
// With seq_len 1 and zero state each layer is exactly
//     g = W_ih x + (b_ih + b_hh);  c = sigmoid(g_i) * tanh(g_g);  h = sigmoid(g_o) * tanh(c)
// so the gate GEMM is [clips, K] x [K, 768] (i, g, o columns; the forget gate and W_hh are dead).
//
// One 8-wave workgroup per 16 clips (256 workgroups at 4096 clips: every CU busy).  Wave w owns hidden units 32w..32w+31:
// six 16x16 f32-MFMA tiles (gates i, g, o x two halves) that share the A operand, so the gate non-linearity runs in
// registers on matching layouts.  Layer-0 output goes to LDS and is layer 1's A operand; weights stream from L2.
#include "ww_internal.h"

namespace ww {


constexpr int kHS = kHidden + 4;   // LDS row stride for [32 clips][256]: 16-byte aligned rows, 4 banks apart

#ifdef WW_K3_PRECISE_MATH
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f / (expf(2.0f * x) + 1.0f); }
#else
// hardware exp2 / reciprocal (v_exp_f32, v_rcp_f32: ~1 ulp each).  sigmoid(x) = 1/(1 + 2^(-x log2 e));
// tanh(x) = 1 - 2/(2^(2x log2 e) + 1): monotone, saturates cleanly to +-1, NaN in -> NaN out.
__device__ __forceinline__ float sigmoidf_(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float tanhf_(float x) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.88539008177792681f * x) + 1.0f);
}
#endif

// LDS activation rows use the same within-16 permutation as the packed weights: element k of a row sits at
// 16*(k>>4) + slot16(k), slot16 = (k&3)*4 + ((k&15)>>2), so that lane (clip, kq) reads its A values of four
// consecutive k-steps (k = 4s + kq) with ONE ds_read_b128.
__device__ __forceinline__ int perm16(int k) { return (k & ~15) | ((k & 3) << 2) | ((k & 15) >> 2); }

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int kClipsPerBlock = 16;

// xs: LDS [16][xstride] activations (row = clip, permuted); wt: packed [K/16][768][16]; hb = this wave's hidden block of
// 32 units = six 16x16 tiles (gate i,g,o x two halves).  Writes h[clip][perm16(32*hb + u)] into hout.
// Weight loads run two 16-k groups ahead of the MFMAs.
// TRAIN: the gate activations sigma(g_i), tanh(g_g), sigma(g_o) and tanh(c) of every (clip, unit) go to `save` ([4][n][256], what the
// backward pass needs) and the layer output is multiplied by the unit's dropout factor (0 or 1 / (1 - p); also saved).
struct TrainSave {
    float* gates;        // [4][n][256]
    float* mask;         // [n][256]
    int n, clip0;
    float p;             // drop probability
    uint32_t seed_lo, seed_hi, layer;
};
// counter-based keep / drop decision for (layer, clip, unit): splitmix64 finaliser of the key -> uniform in [0, 1)
__device__ __forceinline__ float dropout_factor(const TrainSave& ts, int clip, int unit) {
    if (ts.p <= 0.f) return 1.0f;
    unsigned long long x = (static_cast<unsigned long long>(ts.seed_hi) << 32 | ts.seed_lo) + 0x9E3779B97F4A7C15ull * (1ull + ts.layer)
                           + (static_cast<unsigned long long>(clip) << 20) + static_cast<unsigned long long>(unit);
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    const float u = static_cast<float>(static_cast<uint32_t>(x >> 40)) * (1.0f / 16777216.0f);      // 24 bits
    return u < ts.p ? 0.f : 1.0f / (1.0f - ts.p);
}

template <int K, bool TRAIN = false>
__device__ __forceinline__ void lstm_layer(const float* __restrict__ xs, int xstride, const float* __restrict__ wt,
                                           const float* __restrict__ bias, int hb, int lane, float* __restrict__ hout,
                                           int hstride, const TrainSave* ts = nullptr) {
    const int col = lane & 15, kq = lane >> 4;
    f32x4 acc[3][2];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[g][h][j] = 0.f;
    const float4* a_p = reinterpret_cast<const float4*>(xs + col * xstride + kq * 4);
    const float4* b_p = reinterpret_cast<const float4*>(wt + (int64_t(hb) * 96 + col) * 16 + kq * 4);
    constexpr int G = K / 16, D = 2;                 // 16-k groups, prefetch depth
    float4 bw[D][6];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int t = 0; t < 6; ++t) bw[d][t] = b_p[(int64_t(d) * kGateCols + 16 * t) * 4];
#pragma unroll
    for (int kg = 0; kg < G; ++kg) {
        const float4 a = a_p[kg * 4];
        float4 bc[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) bc[t] = bw[kg % D][t];
        if (kg + D < G) {
#pragma unroll
            for (int t = 0; t < 6; ++t) bw[kg % D][t] = b_p[(int64_t(kg + D) * kGateCols + 16 * t) * 4];
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the weight loads D groups ahead (the scheduler sinks them to their use)
        const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < 6; ++t) {              // tile t = gate*2 + half: packed columns 32*gate + 16*half ..
                const float bv = s == 0 ? bc[t].x : s == 1 ? bc[t].y : s == 2 ? bc[t].z : bc[t].w;
                acc[t >> 1][t & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv, acc[t >> 1][t & 1], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    // D: lane&15 = hidden unit within the half, register j <-> clip 4*(lane>>4) + j
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int u = 16 * h + col;
        const float b_i = bias[hb * 96 + u], b_g = bias[hb * 96 + 32 + u], b_o = bias[hb * 96 + 64 + u];
        const int ucol = perm16(32 * hb + u);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int clip = 4 * kq + j;
            float gi, gg, go, tc;
            if constexpr (TRAIN) {
                // gradients multiply these by dlogits that nearly cancel over the batch: 1 - 2/(e^2x + 1) loses ~5 bits of tanh(x) near
                // x = 0 (fine for the 1e-3 logit tolerance of inference, visible at 1e-4 in d fc.weight): the library functions here
                gi = 1.0f / (1.0f + expf(-(acc[0][h][j] + b_i)));
                gg = tanhf(acc[1][h][j] + b_g);
                go = 1.0f / (1.0f + expf(-(acc[2][h][j] + b_o)));
                tc = tanhf(gi * gg);
            } else {
                gi = sigmoidf_(acc[0][h][j] + b_i);
                gg = tanhf_(acc[1][h][j] + b_g);
                go = sigmoidf_(acc[2][h][j] + b_o);
                tc = tanhf_(gi * gg);
            }
            float hv = go * tc;
            if constexpr (TRAIN) {
                const int gclip = ts->clip0 + clip, unit = 32 * hb + u;
                if (gclip < ts->n) {
                    const int64_t at = int64_t(gclip) * kHidden + unit, plane = int64_t(ts->n) * kHidden;
                    const float m = dropout_factor(*ts, gclip, unit);
                    ts->gates[at] = gi;
                    ts->gates[plane + at] = gg;
                    ts->gates[2 * plane + at] = go;
                    ts->gates[3 * plane + at] = tc;
                    ts->mask[at] = m;
                    hv *= m;
                }
            }
            hout[clip * hstride + ucol] = hv;
        }
    }
}

__global__ __launch_bounds__(512) void lstm_fc_kernel(const float* __restrict__ pooled, int n, int C,
                                                      const float* __restrict__ w0, const float* __restrict__ b0,
                                                      const float* __restrict__ w1, const float* __restrict__ b1,
                                                      const float* __restrict__ fcw, const float* __restrict__ fcb,
                                                      float* __restrict__ logits, float* __restrict__ prob) {
    __shared__ __attribute__((aligned(16))) float xs[kClipsPerBlock * 132];
    __shared__ __attribute__((aligned(16))) float h0[kClipsPerBlock * kHS];
    __shared__ __attribute__((aligned(16))) float h1[kClipsPerBlock * kHS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int clip0 = blockIdx.x * kClipsPerBlock;
    const int xstride = C + 4;

    for (int i = tid; i < kClipsPerBlock * C; i += 512) {
        const int r = i / C, k = i - r * C;
        xs[r * xstride + perm16(k)] = (clip0 + r < n) ? pooled[int64_t(clip0 + r) * C + k] : 0.f;
    }
    __syncthreads();
    if (C == 64) lstm_layer<64>(xs, xstride, w0, b0, wave, lane, h0, kHS);
    else lstm_layer<128>(xs, xstride, w0, b0, wave, lane, h0, kHS);
    __syncthreads();
    lstm_layer<kHidden>(h0, kHS, w1, b1, wave, lane, h1, kHS);
    __syncthreads();
    {
        // fc: 32 outputs (16 clips x 2 classes), each a 256-term dot product split over 16 lanes, summed in fixed order
        const int o = tid >> 4, part = tid & 15, clip = o >> 1, cls = o & 1;
        const float* hrow = h1 + clip * kHS;
        const float* wrow = fcw + cls * kHidden;
        float acc = 0.f;
#pragma unroll 8
        for (int k = part * 16; k < part * 16 + 16; ++k) acc = fmaf(hrow[perm16(k)], wrow[k], acc);
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        acc += __shfl_xor(acc, 4);
        acc += __shfl_xor(acc, 8);
        const float logit = acc + fcb[cls];
        const float other = __shfl_xor(logit, 16);         // the other class of the same clip
        if (part == 0 && clip0 + clip < n) {
            logits[int64_t(clip0 + clip) * 2 + cls] = logit;
            if (prob && cls == 1) prob[clip0 + clip] = 1.0f / (1.0f + expf(other - logit));   // softmax(logits)[1]
        }
    }
}

// Training-mode head (the reference's model.train() forward: nn.LSTM(dropout=p) between the layers and
// nn.Dropout(p) before fc, train_wakeword.py:34-35,46-47): the exact-f32 kernel above, saving what the backward pass needs.
// hd1 [n][256]: the dropped layer-1 output (fc's input).
__global__ __launch_bounds__(512) void lstm_fc_train_kernel(const float* __restrict__ pooled, int n, int C,
                                                            const float* __restrict__ w0, const float* __restrict__ b0,
                                                            const float* __restrict__ w1, const float* __restrict__ b1,
                                                            const float* __restrict__ fcw, const float* __restrict__ fcb,
                                                            float* __restrict__ gates0, float* __restrict__ mask0, float* __restrict__ hd0,
                                                            float* __restrict__ gates1, float* __restrict__ mask1, float* __restrict__ hd1,
                                                            float p_lstm, float p_fc, uint32_t seed_lo, uint32_t seed_hi,
                                                            float* __restrict__ logits) {
    __shared__ __attribute__((aligned(16))) float xs[kClipsPerBlock * 132];
    __shared__ __attribute__((aligned(16))) float h0[kClipsPerBlock * kHS];
    __shared__ __attribute__((aligned(16))) float h1[kClipsPerBlock * kHS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int clip0 = blockIdx.x * kClipsPerBlock;
    const int xstride = C + 4;
    for (int i = tid; i < kClipsPerBlock * C; i += 512) {
        const int r = i / C, k = i - r * C;
        xs[r * xstride + perm16(k)] = (clip0 + r < n) ? pooled[int64_t(clip0 + r) * C + k] : 0.f;
    }
    __syncthreads();
    const TrainSave t0{gates0, mask0, n, clip0, p_lstm, seed_lo, seed_hi, 0u};
    const TrainSave t1{gates1, mask1, n, clip0, p_fc, seed_lo, seed_hi, 1u};
    if (C == 64) lstm_layer<64, true>(xs, xstride, w0, b0, wave, lane, h0, kHS, &t0);
    else lstm_layer<128, true>(xs, xstride, w0, b0, wave, lane, h0, kHS, &t0);
    __syncthreads();
    lstm_layer<kHidden, true>(h0, kHS, w1, b1, wave, lane, h1, kHS, &t1);
    __syncthreads();
    for (int i = tid; i < kClipsPerBlock * kHidden; i += 512) {          // the dropped activations, natural order
        const int r = i >> 8, k = i & 255;
        if (clip0 + r < n) {
            hd0[int64_t(clip0 + r) * kHidden + k] = h0[r * kHS + perm16(k)];
            hd1[int64_t(clip0 + r) * kHidden + k] = h1[r * kHS + perm16(k)];
        }
    }
    {
        const int o = tid >> 4, part = tid & 15, clip = o >> 1, cls = o & 1;
        const float* hrow = h1 + clip * kHS;
        const float* wrow = fcw + cls * kHidden;
        float acc = 0.f;
#pragma unroll 8
        for (int k = part * 16; k < part * 16 + 16; ++k) acc = fmaf(hrow[perm16(k)], wrow[k], acc);
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        acc += __shfl_xor(acc, 4);
        acc += __shfl_xor(acc, 8);
        if (part == 0 && clip0 + clip < n) logits[int64_t(clip0 + clip) * 2 + cls] = acc + fcb[cls];
    }
}

// W_ih [4H][K] + biases (torch layout, DEVICE memory: the weights change every optimiser step) -> the k-major packed image of
// pack_lstm (ww_tables.cpp): wt[(kg * 768 + col) * 16 + slot], col = (hb*3 + gate)*32 + u, slot = (k & 3) * 4 + ((k & 15) >> 2).
__global__ void pack_lstm_dev_kernel(const float* __restrict__ w_ih, const float* __restrict__ b_ih, const float* __restrict__ b_hh, int K,
                                     float* __restrict__ wt, float* __restrict__ b) {
    const int goff[3] = {0, 2 * kHidden, 3 * kHidden};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < K * kGateCols; i += gridDim.x * blockDim.x) {
        const int col = i / K, k = i - col * K;                       // consecutive threads read consecutive k of one row
        const int hb = col / 96, g = (col % 96) / 32, u = col % 32, row = goff[g] + 32 * hb + u;
        wt[(int64_t(k >> 4) * kGateCols + col) * 16 + ((k & 3) << 2 | ((k & 15) >> 2))] = w_ih[int64_t(row) * K + k];
        if (k == 0) b[col] = b_ih[row] + b_hh[row];
    }
}

int launch_lstm_fc_train(const float* pooled, int64_t n, int C, const float* w_ih0, const float* b_ih0, const float* b_hh0,
                         const float* w_ih1, const float* b_ih1, const float* b_hh1, const float* fcw, const float* fcb,
                         float* packed_ws /* (C + 256) * 768 + 2 * 768 floats */, float* gates0, float* mask0, float* hd0,
                         float* gates1, float* mask1, float* hd1, float p_lstm, float p_fc, uint64_t seed, float* logits,
                         hipStream_t stream) {
    float* w0 = packed_ws;
    float* b0 = w0 + int64_t(C) * kGateCols;
    float* w1 = b0 + kGateCols;
    float* b1 = w1 + int64_t(kHidden) * kGateCols;
    hipLaunchKernelGGL(pack_lstm_dev_kernel, dim3(192), dim3(256), 0, stream, w_ih0, b_ih0, b_hh0, C, w0, b0);
    hipLaunchKernelGGL(pack_lstm_dev_kernel, dim3(768), dim3(256), 0, stream, w_ih1, b_ih1, b_hh1, kHidden, w1, b1);
    WW_HIP(hipGetLastError());
    const int grid = int((n + kClipsPerBlock - 1) / kClipsPerBlock);
    hipLaunchKernelGGL(lstm_fc_train_kernel, dim3(grid), dim3(512), 0, stream, pooled, int(n), C, w0, b0, w1, b1, fcw, fcb, gates0, mask0, hd0,
                       gates1, mask1, hd1, p_lstm, p_fc, uint32_t(seed), uint32_t(seed >> 32), logits);
    WW_HIP(hipGetLastError());
    return WW_OK;
}

// ------------------------------------------------------------------------------------------------
// Split-precision form (conv math f16x3): the gate GEMMs on v_mfma_f32_16x16x32_f16 with activations and weights as
// f16 hi + lo pairs (three MFMAs per product block, fp32 accumulate), 3/16 of the exact-f32 form's matrix-pipe cycles.
// The workgroup is bound by its own latency chain (layer 0 -> layer 1 -> fc on one CU): the kernel takes the same time
// at 16 and at 4096 clips, and two thirds of it were f32 MFMA cycles.
// Activations live in LDS as two f16 planes [16 clips][K + 8]: lane (clip m, kq) reads its 8 consecutive k of a 32-k block
// with one ds_read_b128 per plane (row stride (K + 8) halfs: conflict-free).
// ------------------------------------------------------------------------------------------------
using half8_t = __attribute__((ext_vector_type(8))) _Float16;
using u32x4_t = __attribute__((ext_vector_type(4))) uint32_t;

// descale: [768] 2^-S per packed gate column (the weights' per-row scales); rowscale: LDS [16] 2^e per clip (the input row was
// stored as x * 2^-e), or nullptr when the rows are unscaled (layer 1: |h| < 1).
template <int K>
__device__ __forceinline__ void lstm_layer_h(const _Float16* __restrict__ xh, const _Float16* __restrict__ xl,
                                             const u32x4_t* __restrict__ wt, const float* __restrict__ descale,
                                             const float* __restrict__ rowscale, const float* __restrict__ bias,
                                             int hb, int lane, _Float16* __restrict__ out_h, _Float16* __restrict__ out_l,
                                             float* __restrict__ out_f, int out_stride) {
    constexpr int KS = K + 8, KB = K / 32, D = 2;
    const int col = lane & 15, kq = lane >> 4;
    f32x4 acc[3][2];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[g][h][j] = 0.f;
    const _Float16* ah_p = xh + col * KS + kq * 8;
    const _Float16* al_p = xl + col * KS + kq * 8;
    // wave hb's six N-tiles are packed tiles 6 hb .. 6 hb + 5; per k-block and tile: [hi,lo][64 lanes] x 16 bytes
    const u32x4_t* b_p = wt + (int64_t(hb) * 6 * 2) * 64 + lane;
    constexpr int kTileStride = 2 * 64, kKbStride = (kGateCols / 16) * 2 * 64;
    u32x4_t bw[D][6][2];
#pragma unroll
    for (int d = 0; d < D && d < KB; ++d)
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            bw[d][t][0] = b_p[int64_t(d) * kKbStride + t * kTileStride];
            bw[d][t][1] = b_p[int64_t(d) * kKbStride + t * kTileStride + 64];
        }
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const half8_t ah = *reinterpret_cast<const half8_t*>(ah_p + 32 * kb);
        const half8_t al = *reinterpret_cast<const half8_t*>(al_p + 32 * kb);
        u32x4_t bc[6][2];
#pragma unroll
        for (int t = 0; t < 6; ++t) { bc[t][0] = bw[kb % D][t][0]; bc[t][1] = bw[kb % D][t][1]; }
        if (kb + D < KB) {
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                bw[kb % D][t][0] = b_p[int64_t(kb + D) * kKbStride + t * kTileStride];
                bw[kb % D][t][1] = b_p[int64_t(kb + D) * kKbStride + t * kTileStride + 64];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const half8_t bh = __builtin_bit_cast(half8_t, bc[t][0]), bl = __builtin_bit_cast(half8_t, bc[t][1]);
            acc[t >> 1][t & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[t >> 1][t & 1], 0, 0, 0);
            acc[t >> 1][t & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[t >> 1][t & 1], 0, 0, 0);
            acc[t >> 1][t & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[t >> 1][t & 1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // D: lane&15 = hidden unit within the half, register j <-> clip 4*(lane>>4) + j
    float rs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rs[j] = rowscale ? rowscale[4 * kq + j] : 1.0f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int u = 16 * h + col;
        const float b_i = bias[hb * 96 + u], b_g = bias[hb * 96 + 32 + u], b_o = bias[hb * 96 + 64 + u];
        const float d_i = descale[hb * 96 + u], d_g = descale[hb * 96 + 32 + u], d_o = descale[hb * 96 + 64 + u];
        const int unit = 32 * hb + u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int clip = 4 * kq + j;
            const float c = sigmoidf_(fmaf(acc[0][h][j] * rs[j], d_i, b_i)) * tanhf_(fmaf(acc[1][h][j] * rs[j], d_g, b_g));
            const float hv = sigmoidf_(fmaf(acc[2][h][j] * rs[j], d_o, b_o)) * tanhf_(c);
            if (out_f) {
                out_f[clip * out_stride + unit] = hv;
            } else {
                const _Float16 hi = static_cast<_Float16>(hv);
                out_h[clip * out_stride + unit] = hi;
                out_l[clip * out_stride + unit] = static_cast<_Float16>(hv - static_cast<float>(hi));
            }
        }
    }
}

__global__ __launch_bounds__(512) void lstm_fc_h_kernel(const float* __restrict__ pooled, int n, int C,
                                                        const u32x4_t* __restrict__ w0, const float* __restrict__ b0,
                                                        const u32x4_t* __restrict__ w1, const float* __restrict__ b1,
                                                        const float* __restrict__ hs, const float* __restrict__ fcw,
                                                        const float* __restrict__ fcb, float* __restrict__ logits,
                                                        float* __restrict__ prob) {
    constexpr int kXS = 128 + 8, kHS16 = kHidden + 8;
    __shared__ __attribute__((aligned(16))) _Float16 xh[kClipsPerBlock * kXS], xl[kClipsPerBlock * kXS];
    __shared__ __attribute__((aligned(16))) _Float16 h0h[kClipsPerBlock * kHS16], h0l[kClipsPerBlock * kHS16];
    __shared__ __attribute__((aligned(16))) float h1[kClipsPerBlock * kHS];
    __shared__ float rowscale[kClipsPerBlock];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int clip0 = blockIdx.x * kClipsPerBlock;
    const int xs = C + 8;

    {
        // row r = tid >> 5 (32 threads per clip).  The row is stored as x * 2^-e with e from its max |x| (max |x'| in
        // [2^14, 2^15)): pooled features of any magnitude keep both f16 halves in range; 2^e goes back in at the gates.
        const int r = tid >> 5, k0 = tid & 31;
        const bool live = clip0 + r < n;
        float v[4];
        float mx = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = (live && k0 + 32 * i < C) ? pooled[int64_t(clip0 + r) * C + k0 + 32 * i] : 0.f;
            mx = fmaxf(mx, __builtin_fabsf(v[i]));
        }
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        int e = int((__float_as_uint(mx) >> 23) & 0xffu) - 127 - 14;
        e = e < -100 ? -100 : (e > 113 ? 113 : e);
        const float down = __uint_as_float(uint32_t(127 - e) << 23);
        if (k0 == 0) rowscale[r] = __uint_as_float(uint32_t(127 + e) << 23);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (k0 + 32 * i < C) {
                const float vv = v[i] * down;
                const _Float16 hi = static_cast<_Float16>(vv);
                xh[r * xs + k0 + 32 * i] = hi;
                xl[r * xs + k0 + 32 * i] = static_cast<_Float16>(vv - static_cast<float>(hi));
            }
    }
    __syncthreads();
    if (C == 64) lstm_layer_h<64>(xh, xl, w0, hs, rowscale, b0, wave, lane, h0h, h0l, nullptr, kHS16);
    else lstm_layer_h<128>(xh, xl, w0, hs, rowscale, b0, wave, lane, h0h, h0l, nullptr, kHS16);
    __syncthreads();
    lstm_layer_h<kHidden>(h0h, h0l, w1, hs + kGateCols, nullptr, b1, wave, lane, nullptr, nullptr, h1, kHS);
    __syncthreads();
    {
        const int o = tid >> 4, part = tid & 15, clip = o >> 1, cls = o & 1;
        const float* hrow = h1 + clip * kHS;
        const float* wrow = fcw + cls * kHidden;
        float acc = 0.f;
#pragma unroll 8
        for (int k = part * 16; k < part * 16 + 16; ++k) acc = fmaf(hrow[k], wrow[k], acc);
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        acc += __shfl_xor(acc, 4);
        acc += __shfl_xor(acc, 8);
        const float logit = acc + fcb[cls];
        const float other = __shfl_xor(logit, 16);
        if (part == 0 && clip0 + clip < n) {
            logits[int64_t(clip0 + clip) * 2 + cls] = logit;
            if (prob && cls == 1) prob[clip0 + clip] = 1.0f / (1.0f + expf(other - logit));
        }
    }
}

int launch_lstm_fc(const float* pooled, int64_t n, const float* packed, int n_conv, float* logits, float* prob,
                   hipStream_t stream) {
    if (n == 0) return WW_OK;
    const PackedLayout L = packed_layout(n_conv);
    const int grid = int((n + kClipsPerBlock - 1) / kClipsPerBlock);
    if (conv_math_mode() != 0) {
        hipLaunchKernelGGL(lstm_fc_h_kernel, dim3(grid), dim3(512), 0, stream, pooled, int(n), L.c_last,
                           reinterpret_cast<const u32x4_t*>(packed + L.l0_h), packed + L.l0_b,
                           reinterpret_cast<const u32x4_t*>(packed + L.l1_h), packed + L.l1_b, packed + L.lstm_hs,
                           packed + L.fc_w, packed + L.fc_b, logits, prob);
        WW_HIP(hipGetLastError());
        return WW_OK;
    }
    hipLaunchKernelGGL(lstm_fc_kernel, dim3(grid), dim3(512), 0, stream, pooled, int(n), L.c_last, packed + L.l0_w,
                       packed + L.l0_b, packed + L.l1_w, packed + L.l1_b, packed + L.fc_w, packed + L.fc_b, logits,
                       prob);
    WW_HIP(hipGetLastError());
    return WW_OK;
}

}  // namespace ww
