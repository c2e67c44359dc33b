// K3: 2-layer LSTM, ONE time step, zero (h0, c0), + Linear(256, 2) [+ softmax p(wakeword)].
//
// Replaces `lstm_out, _ = self.lstm(x.unsqueeze(1)); x = lstm_out[:, -1, :]; x = self.fc(self.dropout(x))`
// (/root/reference/wakeword_training/train_wakeword.py:42-48, wakeword_training_script.py:175-182; eval mode).
// With seq_len 1 and zero state each layer is exactly
//     g = W_ih x + (b_ih + b_hh);  c = sigmoid(g_i) * tanh(g_g);  h = sigmoid(g_o) * tanh(c)
// so the gate GEMM is [clips, K] x [K, 768] (i, g, o columns; the forget gate and W_hh are dead).
//
// One 8-wave workgroup per 32 clips.  Wave w owns hidden units 32w..32w+31: three 32x32 f32-MFMA tiles
// (i, g, o) that share the A operand, so the gate non-linearity runs in registers on matching layouts.
// Layer-0 output goes to LDS and is layer 1's A operand; weights stream from L2 as B operands.
#include "ww_internal.h"

namespace ww {

using f32x16 = __attribute__((ext_vector_type(16))) float;

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

// LDS activation rows use the same within-8 permutation as the packed weights: element k of a row sits at
// 8*(k>>3) + slot8(k), slot8 = (k&1)*4 + ((k&7)>>1), so that lane (clip, kh) reads its A values of four
// consecutive k-steps with ONE ds_read_b128.
__device__ __forceinline__ int perm8(int k) { return (k & ~7) | ((k & 1) << 2) | ((k & 7) >> 1); }

// xs: LDS [32][xstride] activations (row = clip, permuted); wt: packed [K/8][768][8]; hb = this wave's hidden block.
// Writes h[clip][perm8(32*hb + u)] into hout (LDS, stride kHS4).  Weight loads run two 8-k groups ahead of the MFMAs.
template <int K>
__device__ __forceinline__ void lstm_layer(const float* __restrict__ xs, int xstride, const float* __restrict__ wt,
                                           const float* __restrict__ bias, int hb, int lane, float* __restrict__ hout,
                                           int hstride) {
    const int row = lane & 31, kh = lane >> 5;
    f32x16 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[g][j] = 0.f;
    const float4* a_p = reinterpret_cast<const float4*>(xs + row * xstride + kh * 4);
    const float4* b_p = reinterpret_cast<const float4*>(wt + (int64_t(hb) * 96 + row) * 8 + kh * 4);
    constexpr int G = K / 8, D = 2;                  // groups, prefetch depth
    float4 bw[D][3];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int g = 0; g < 3; ++g) bw[d][g] = b_p[(int64_t(d) * kGateCols + 32 * g) * 2];
#pragma unroll
    for (int kg = 0; kg < G; ++kg) {
        const float4 a = a_p[kg * 2];
        float4 bc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) bc[g] = bw[kg % D][g];
        if (kg + D < G) {
#pragma unroll
            for (int g = 0; g < 3; ++g) bw[kg % D][g] = b_p[(int64_t(kg + D) * kGateCols + 32 * g) * 2];
        }
        const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const float bv = s == 0 ? bc[g].x : s == 1 ? bc[g].y : s == 2 ? bc[g].z : bc[g].w;
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv, acc[g], 0, 0, 0);
            }
    }
    // D: lane&31 = hidden unit u, register j <-> clip (j&3) + 8*(j>>2) + 4*(lane>>5)
    const float b_i = bias[hb * 96 + row], b_g = bias[hb * 96 + 32 + row], b_o = bias[hb * 96 + 64 + row];
    const int ucol = perm8(32 * hb + row);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int clip = (j & 3) + 8 * (j >> 2) + 4 * kh;
        const float c = sigmoidf_(acc[0][j] + b_i) * tanhf_(acc[1][j] + b_g);
        hout[clip * hstride + ucol] = sigmoidf_(acc[2][j] + b_o) * tanhf_(c);
    }
}

__global__ __launch_bounds__(512) void lstm_fc_kernel(const float* __restrict__ pooled, int n, int C,
                                                      const float* __restrict__ w0, const float* __restrict__ b0,
                                                      const float* __restrict__ w1, const float* __restrict__ b1,
                                                      const float* __restrict__ fcw, const float* __restrict__ fcb,
                                                      float* __restrict__ logits, float* __restrict__ prob) {
    __shared__ __attribute__((aligned(16))) float xs[32 * 132];
    __shared__ __attribute__((aligned(16))) float h0[32 * kHS];
    __shared__ __attribute__((aligned(16))) float h1[32 * kHS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int clip0 = blockIdx.x * 32;
    const int xstride = C + 4;

    for (int i = tid; i < 32 * C; i += 512) {
        const int r = i / C, k = i - r * C;
        xs[r * xstride + perm8(k)] = (clip0 + r < n) ? pooled[int64_t(clip0 + r) * C + k] : 0.f;
    }
    __syncthreads();
    if (C == 64) lstm_layer<64>(xs, xstride, w0, b0, wave, lane, h0, kHS);
    else lstm_layer<128>(xs, xstride, w0, b0, wave, lane, h0, kHS);
    __syncthreads();
    lstm_layer<kHidden>(h0, kHS, w1, b1, wave, lane, h1, kHS);
    __syncthreads();
    {
        // fc: 64 outputs (32 clips x 2 classes), each a 256-term dot product split over 8 lanes, summed in fixed order
        const int o = tid >> 3, part = tid & 7, clip = o >> 1, cls = o & 1;
        const float* hrow = h1 + clip * kHS;
        const float* wrow = fcw + cls * kHidden;
        float acc = 0.f;
#pragma unroll 8
        for (int k = part * 32; k < part * 32 + 32; ++k) acc = fmaf(hrow[perm8(k)], wrow[k], acc);
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        acc += __shfl_xor(acc, 4);
        const float logit = acc + fcb[cls];
        const float other = __shfl_xor(logit, 8);          // the other class of the same clip
        if (part == 0 && clip0 + clip < n) {
            logits[int64_t(clip0 + clip) * 2 + cls] = logit;
            if (prob && cls == 1) prob[clip0 + clip] = 1.0f / (1.0f + expf(other - logit));   // softmax(logits)[1]
        }
    }
}

int launch_lstm_fc(const float* pooled, int64_t n, const float* packed, int n_conv, float* logits, float* prob,
                   hipStream_t stream) {
    if (n == 0) return WW_OK;
    const PackedLayout L = packed_layout(n_conv);
    const int grid = int((n + 31) / 32);
    hipLaunchKernelGGL(lstm_fc_kernel, dim3(grid), dim3(512), 0, stream, pooled, int(n), L.c_last, packed + L.l0_w,
                       packed + L.l0_b, packed + L.l1_w, packed + L.l1_b, packed + L.fc_w, packed + L.fc_b, logits,
                       prob);
    WW_HIP(hipGetLastError());
    return WW_OK;
}

}  // namespace ww
