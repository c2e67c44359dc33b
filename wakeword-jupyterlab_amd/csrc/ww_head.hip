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

constexpr int kHS = kHidden + 1;   // LDS row stride for [32 clips][256]: odd -> conflict-free column reads

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
    // 1 - 2/(e^{2x}+1): monotone, saturates cleanly to +-1 for large |x|
    return 1.0f - 2.0f / (expf(2.0f * x) + 1.0f);
}

// xs: LDS [32][xstride] activations (row = clip); wt: [K][768] k-major; hb = this wave's hidden block.
// Writes h[clip][32*hb + u] into hout (LDS, stride kHS).
__device__ __forceinline__ void lstm_layer(const float* __restrict__ xs, int xstride, int K,
                                           const float* __restrict__ wt, const float* __restrict__ bias, int hb,
                                           int lane, float* __restrict__ hout) {
    const int row = lane & 31, kh = lane >> 5;
    f32x16 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[g][j] = 0.f;
    const float* a_p = xs + row * xstride + kh;
    const float* b_p = wt + int64_t(kh) * kGateCols + hb * 96 + row;
#pragma unroll 8
    for (int s = 0; s < K / 2; ++s) {
        const float a = a_p[2 * s];
        const float* b = b_p + int64_t(2 * s) * kGateCols;
        const float bi = b[0], bg = b[32], bo = b[64];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bi, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bg, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bo, acc[2], 0, 0, 0);
    }
    // D: lane&31 = hidden unit u, register j <-> clip (j&3) + 8*(j>>2) + 4*(lane>>5)
    const float b_i = bias[hb * 96 + row], b_g = bias[hb * 96 + 32 + row], b_o = bias[hb * 96 + 64 + row];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int clip = (j & 3) + 8 * (j >> 2) + 4 * kh;
        const float c = sigmoidf_(acc[0][j] + b_i) * tanhf_(acc[1][j] + b_g);
        hout[clip * kHS + 32 * hb + row] = sigmoidf_(acc[2][j] + b_o) * tanhf_(c);
    }
}

__global__ __launch_bounds__(512) void lstm_fc_kernel(const float* __restrict__ pooled, int n, int C,
                                                      const float* __restrict__ w0, const float* __restrict__ b0,
                                                      const float* __restrict__ w1, const float* __restrict__ b1,
                                                      const float* __restrict__ fcw, const float* __restrict__ fcb,
                                                      float* __restrict__ logits, float* __restrict__ prob) {
    __shared__ float xs[32 * 129];
    __shared__ float h0[32 * kHS];
    __shared__ float h1[32 * kHS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int clip0 = blockIdx.x * 32;
    const int xstride = C + 1;

    for (int i = tid; i < 32 * C; i += 512) {
        const int r = i / C, k = i - r * C;
        xs[r * xstride + k] = (clip0 + r < n) ? pooled[int64_t(clip0 + r) * C + k] : 0.f;
    }
    __syncthreads();
    lstm_layer(xs, xstride, C, w0, b0, wave, lane, h0);
    __syncthreads();
    lstm_layer(h0, kHS, kHidden, w1, b1, wave, lane, h1);
    __syncthreads();
    if (tid < 32 && clip0 + tid < n) {
        const float* hrow = h1 + tid * kHS;
        float l0 = fcb[0], l1 = fcb[1];
        for (int k = 0; k < kHidden; ++k) {
            l0 = fmaf(hrow[k], fcw[k], l0);
            l1 = fmaf(hrow[k], fcw[kHidden + k], l1);
        }
        logits[int64_t(clip0 + tid) * 2 + 0] = l0;
        logits[int64_t(clip0 + tid) * 2 + 1] = l1;
        if (prob) prob[clip0 + tid] = 1.0f / (1.0f + expf(l0 - l1));   // softmax(logits)[1]
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
