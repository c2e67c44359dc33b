// Training step of SimpleWakewordModel: train-mode forward and the backward pass.
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
// six small GEMMs (sgemm_kernel).  W_hh gradients are exactly zero (h0 = 0) and are left to the caller to zero-fill.
// Every reduction over clips runs in a fixed order (per-workgroup partials + reduce_partials_kernel): bitwise repeatable.
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
// conv2 weight [64][32][3][3] -> B operand of the data-gradient kernel (a correlation of gm with the FLIPPED, transposed
// weights): out[(kh*144 + (c*3+dy)*3+dx)*64 + lane] = W[32 kh + 2c + (lane>>5)][lane&31][2-dy][2-dx]
__global__ void pack_dgrad_b_dev_kernel(const float* __restrict__ w, float* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * 144 * 64; i += gridDim.x * blockDim.x) {
        const int lane = i & 63, r = i >> 6, kh = r / 144, s = r - kh * 144, c = s / 9, dy = (s % 9) / 3, dx = s % 3;
        const int co = 32 * kh + 2 * c + (lane >> 5), ci = lane & 31;
        out[i] = w[((co * 32 + ci) * 3 + (2 - dy)) * 3 + (2 - dx)];
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
// conv2 weight gradient.  512 threads = 8 waves: wave = (co half ch, row pair rp); 8-row bands.
//   dW2[co][ci][tap] += sum over the band's positions gm[co][y][x] * a1[ci][y+dy-1][x+dx-1]
// as v_mfma_f32_32x32x2_f32 with M = co (32 of the half), N = ci, K = two neighbouring columns; the nine taps' 32x32
// accumulators (144 VGPRs) stay in registers over ALL clips of the persistent workgroup.  LDS: log-mel tile, a1 band
// [32 ci][10 rows][34] (ci stride 341: conflict-free across ci), gm band [8 rows][64 co][33].
// Output: one partial [64][32][9] + [64] (bias) per workgroup -> reduce_partials_kernel.
// ------------------------------------------------------------------------------------------------
constexpr int kWgActCi = 10 * kTRS + 1;                     // 341
constexpr int kWgActFloats = 32 * kWgActCi;                 // 10,912
constexpr int kWgGmFloats = 8 * 64 * 33;                    // 16,896
constexpr int kWgLdsFloats = kTMelFloats + kWgActFloats + kWgGmFloats;
constexpr int kWgPartial = 64 * 32 * 9 + 64;                // floats per workgroup

__global__ __launch_bounds__(512, 2) void conv2_wgrad_kernel(const float* __restrict__ mel, const float* __restrict__ mid,
                                                             const float* __restrict__ gp /*[n][64]*/, int n, int width,
                                                             const float* __restrict__ w1, const float* __restrict__ b1,
                                                             float* __restrict__ partial /*[grid][kWgPartial]*/) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* melt = lds;
    float* act = melt + kTMelFloats;
    float* gm = act + kWgActFloats;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ch = wave & 1, rp = wave >> 1;
    const int m = lane & 31, kk = lane >> 5;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    float db2 = 0.f;                                        // loader role: fixed co = tid >> 3 for every row
    for (int i = tid; i < kWgLdsFloats; i += 512) lds[i] = 0.f;

    const int lco = tid >> 3, lcol = (tid & 7) * 4;
    const int cx = tid & 31, cg = tid >> 5;                 // conv1 role: column, channels 2 cg, 2 cg + 1
    float w1r[2][9], b1r[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int t = 0; t < 9; ++t) w1r[u][t] = w1[(2 * cg + u) * 9 + t];
        b1r[u] = b1[2 * cg + u];
    }

    for (int clip = blockIdx.x; clip < n; clip += gridDim.x) {
        __syncthreads();
        load_mel_tile(mel + int64_t(clip) * kTH * width, width, melt, tid, 512);
        const float g = gp[int64_t(clip) * 64 + lco];
        for (int band = 0; band < kTH / 8; ++band) {
            const int y0 = band * 8;
            __syncthreads();          // mel tile ready / the previous band's operand reads retired
            // gm band: gp * [relu(conv2) > 0]
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float4 v = *reinterpret_cast<const float4*>(mid + ((int64_t(clip) * kTH + y0 + r) * 64 + lco) * kTW + lcol);
                float* d = gm + (r * 64 + lco) * 33 + lcol;
                const float g0 = v.x > 0.f ? g : 0.f, g1 = v.y > 0.f ? g : 0.f, g2 = v.z > 0.f ? g : 0.f, g3 = v.w > 0.f ? g : 0.f;
                d[0] = g0; d[1] = g1; d[2] = g2; d[3] = g3;
                db2 += (g0 + g1) + (g2 + g3);
            }
            // a1 band: rows y0-1 .. y0+8, zero outside the image and beyond `width`
#pragma unroll 1
            for (int q = 0; q < 10; ++q) {
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
                    act[(2 * cg + u) * kWgActCi + q * kTRS + cx + 1] = inside ? relu_t(v) : 0.f;
                }
            }
            __syncthreads();
            // this wave: output rows 2 rp, 2 rp + 1 of the band, co half ch
#pragma unroll 1
            for (int rr = 0; rr < 2; ++rr) {
                const int qo = 2 * rp + rr;
                const float* ga = gm + (qo * 64 + 32 * ch + m) * 33 + kk;
                const float* ba = act + m * kWgActCi + qo * kTRS + kk;
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
    // sum the four row pairs through LDS in fixed order, write this workgroup's partial
    __syncthreads();
    float* xch = lds;                                        // [rp][ch][tap][16][64] = 73,728 floats > the tiles: go tap by tap
    float* outp = partial + int64_t(blockIdx.x) * kWgPartial;
#pragma unroll
    for (int t = 0; t < 9; ++t) {             // unrolled: a runtime index would put the accumulators in scratch memory
#pragma unroll
        for (int j = 0; j < 16; ++j) xch[((rp * 2 + ch) * 16 + j) * 64 + lane] = acc[t][j];
        __syncthreads();
        if (rp == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) s += xch[((r * 2 + ch) * 16 + j) * 64 + lane];
                const int co = 32 * ch + (j & 3) + 8 * (j >> 2) + 4 * kk, ci = m;      // D: lane&31 = n, register j <-> row m
                outp[(co * 32 + ci) * 9 + t] = s;
            }
        }
        __syncthreads();
    }
    db2 += __shfl_xor(db2, 1);
    db2 += __shfl_xor(db2, 2);
    db2 += __shfl_xor(db2, 4);
    if ((tid & 7) == 0) outp[64 * 32 * 9 + lco] = db2;
}

// ------------------------------------------------------------------------------------------------
// conv2 data gradient + conv1 gradients.  256 threads = 4 waves: wave = (K half kh: 32 of the 64 co, row group rg of 4 rows);
// 8-row bands; da1 = correlation of gm (64 channels) with the flipped weights, M = 32 columns, N = ci, K = (co, tap):
// the forward kernels' mfma_rows4 structure with 144 resident B-operand VGPRs per wave, the two K halves summed through LDS.
// Epilogue per da1 value: the sign of conv1 recomputed from the log-mel tile (broadcast LDS reads), dW1 / db1 accumulated in
// registers over all clips.  LDS: log-mel tile, gm band with halo [64 co][10 rows][34], exchange [2 rg][4][16][64].
// ------------------------------------------------------------------------------------------------
constexpr int kDgGmFloats = 64 * 10 * kTRS;                 // 21,760
constexpr int kDgXchFloats = 2 * 4 * 16 * 64;               // 8,192
constexpr int kDgLdsFloats = kTMelFloats + kDgGmFloats + kDgXchFloats;
constexpr int kDgPartial = 32 * 9 + 32;

__global__ __launch_bounds__(256, 1) void conv2_dgrad_kernel(const float* __restrict__ mel, const float* __restrict__ mid,
                                                             const float* __restrict__ gp, int n, int width,
                                                             const float* __restrict__ w1, const float* __restrict__ b1,
                                                             const float* __restrict__ wB /*pack_dgrad_b_dev_kernel*/,
                                                             float* __restrict__ partial /*[grid][kDgPartial]*/) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* melt = lds;
    float* gmt = melt + kTMelFloats;                        // [64][10][34]
    float* xch = gmt + kDgGmFloats;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave & 1, rg = wave >> 1;
    const int ci = lane & 31, h = lane >> 5;

    float wb[144];
#pragma unroll
    for (int i = 0; i < 144; ++i) wb[i] = wB[(kh * 144 + i) * 64 + lane];
    float w1r[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w1r[t] = w1[ci * 9 + t];
    const float b1r = b1[ci];
    // float64 accumulators: a conv1 weight gradient is a sum of ~10^5 products dz * mel per workgroup with heavy cancellation
    // (mel ~ -35 +- 15 dB against a dz that sums to almost nothing); in float32 the order of summation alone moves it by 1e-4
    double dw1[9] = {0., 0., 0., 0., 0., 0., 0., 0., 0.}, db1 = 0.;
    for (int i = tid; i < kDgLdsFloats; i += 256) lds[i] = 0.f;

    const int lco = tid >> 2, lcol = (tid & 3) * 8;          // loader role: 64 co x 4 threads x 8 columns
    const float* ap = gmt + ((32 * kh + h) * 10 + rg * 4) * kTRS + ci;     // A operand: lane&31 = column here (see below)

    for (int clip = blockIdx.x; clip < n; clip += gridDim.x) {
        __syncthreads();
        load_mel_tile(mel + int64_t(clip) * kTH * width, width, melt, tid, 256);
        const float g = gp[int64_t(clip) * 64 + lco];
        for (int band = 0; band < kTH / 8; ++band) {
            const int y0 = band * 8;
            __syncthreads();
            // gm band with halo: rows y0-1 .. y0+8, zero outside the image
#pragma unroll 1
            for (int q = 0; q < 10; ++q) {
                const int y = y0 - 1 + q;
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
                if (y >= 0 && y < kTH) {
                    const float* src = mid + ((int64_t(clip) * kTH + y) * 64 + lco) * kTW + lcol;
                    a = *reinterpret_cast<const float4*>(src);
                    b = *reinterpret_cast<const float4*>(src + 4);
                }
                float* d = gmt + (lco * 10 + q) * kTRS + lcol + 1;
                d[0] = a.x > 0.f ? g : 0.f; d[1] = a.y > 0.f ? g : 0.f; d[2] = a.z > 0.f ? g : 0.f; d[3] = a.w > 0.f ? g : 0.f;
                d[4] = b.x > 0.f ? g : 0.f; d[5] = b.y > 0.f ? g : 0.f; d[6] = b.z > 0.f ? g : 0.f; d[7] = b.w > 0.f ? g : 0.f;
            }
            __syncthreads();
            // 144 k-steps (16 co pairs of this half x 3 x 3) over this wave's 4 rows.  A[m = column][k = co parity], B = wb.
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
                    for (int q = 0; q < 6; ++q) a[q] = ap[(2 * c * 10 + q) * kTRS + dx];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r + dy], wb[(c * 3 + dy) * 3 + dx], acc[r], 0, 0, 0);
                }
            }
            if (kh == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 16; ++j) xch[((rg * 4 + r) * 16 + j) * 64 + lane] = acc[r][j];
            }
            __syncthreads();
            if (kh == 0) {
                // D: lane&31 = n = ci, register j <-> column (j&3) + 8 (j>>2) + 4 h
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int y = y0 + rg * 4 + r;
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int col = (j & 3) + 8 * (j >> 2) + 4 * h;
                        const float da = acc[r][j] + xch[((rg * 4 + r) * 16 + j) * 64 + lane];
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
                }
            }
        }
    }
    // the two column halves (lane ^ 32) and the two row groups, in fixed order -> this workgroup's partial
    __syncthreads();
    double* red = reinterpret_cast<double*>(xch);            // [rg][h][10][32] doubles = 10 KB of the 32 KB exchange area
    if (kh == 0) {
#pragma unroll
        for (int t = 0; t < 9; ++t) red[((rg * 2 + h) * 10 + t) * 32 + ci] = dw1[t];
        red[((rg * 2 + h) * 10 + 9) * 32 + ci] = db1;
    }
    __syncthreads();
    float* outp = partial + int64_t(blockIdx.x) * kDgPartial;
    if (tid < 32) {
#pragma unroll
        for (int t = 0; t < 10; ++t) {
            const double s = (red[(0 * 10 + t) * 32 + tid] + red[(1 * 10 + t) * 32 + tid]) + (red[(2 * 10 + t) * 32 + tid] + red[(3 * 10 + t) * 32 + tid]);
            if (t < 9) outp[tid * 9 + t] = float(s);
            else outp[32 * 9 + tid] = float(s);
        }
    }
}

// out[i] = sum over workgroups g (fixed order) of partial[g][i]
__global__ void reduce_partials_kernel(const float* __restrict__ partial, int groups, int len, float* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int g = 0; g < groups; ++g) s += partial[int64_t(g) * len + i];
        out[i] = s;
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

// out[j] = sum_b x[b][j] in a fixed order (bias gradients): 1024 threads = 64 columns x 16 row slices, slices summed through LDS
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ x, int rows, int cols, float* __restrict__ out) {
    __shared__ float part[16][65];
    const int c = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + c;
    float s = 0.f;
    if (j < cols)
        for (int b = sl; b < rows; b += 16) s += x[int64_t(b) * cols + j];
    part[sl][c] = s;
    __syncthreads();
    if (sl == 0 && j < cols) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += part[i][c];
        out[j] = t;
    }
}

// ------------------------------------------------------------------------------------------------
// C[M][N] = sum_k A(m,k) B(k,n) with general strides (the six small GEMMs of the head's backward: M*N*K <= 1.1e9).
// T x T tiles (T = 64: 4x4 outputs per thread; T = 32: 2x2, four times the workgroups for the K = batch weight-gradient GEMMs),
// 256 threads, K in steps of 16 through LDS; fp32 FMA, fixed order.
// ------------------------------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(256) void sgemm_kernel(const float* __restrict__ A, int64_t sam, int64_t sak, const float* __restrict__ B, int64_t sbk,
                                                    int64_t sbn, float* __restrict__ C, int64_t ldc, int M, int N, int K) {
    constexpr int R = T / 16;
    __shared__ float As[16][T + 1], Bs[16][T + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.y * T, n0 = blockIdx.x * T;
    float acc[R][R] = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int i = threadIdx.x; i < 16 * T; i += 256) {
            const int kk = i / T, mm = i - kk * T;              // consecutive threads walk m (or n)
            As[kk][mm] = (m0 + mm < M && k0 + kk < K) ? A[int64_t(m0 + mm) * sam + int64_t(k0 + kk) * sak] : 0.f;
            Bs[kk][mm] = (n0 + mm < N && k0 + kk < K) ? B[int64_t(k0 + kk) * sbk + int64_t(n0 + mm) * sbn] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[R], b[R];
#pragma unroll
            for (int i = 0; i < R; ++i) { a[i] = As[kk][ty * R + i]; b[i] = Bs[kk][tx * R + i]; }
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int j = 0; j < R; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (m0 + ty * R + i < M && n0 + tx * R + j < N) C[int64_t(m0 + ty * R + i) * ldc + n0 + tx * R + j] = acc[i][j];
}

static void sgemm(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn, float* C, int64_t ldc, int M, int N, int K,
                  hipStream_t st) {
    if (int64_t(M) * N <= 512 * 512)     // few output tiles (the weight gradients, K = batch): small tiles, more workgroups
        hipLaunchKernelGGL(sgemm_kernel<32>, dim3((N + 31) / 32, (M + 31) / 32), dim3(256), 0, st, A, sam, sak, B, sbk, sbn, C, ldc, M, N, K);
    else
        hipLaunchKernelGGL(sgemm_kernel<64>, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, st, A, sam, sak, B, sbk, sbn, C, ldc, M, N, K);
}

// gp[b][co] = dpooled[b][co] / (80 * width)
__global__ void scale_kernel(const float* __restrict__ x, float s, int64_t len, float* __restrict__ out) {
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < len; i += int64_t(gridDim.x) * blockDim.x) out[i] = x[i] * s;
}

// ------------------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------------------
int launch_cnn2_f32_mid(const float* mel, int64_t n, int width, const float* w1, const float* b1, const float* wB, const float* b2, float* mid,
                        hipStream_t stream);   // ww_cnn.hip
int launch_lstm_fc_train(const float* pooled, int64_t n, int C, const float* w_ih0, const float* b_ih0, const float* b_hh0,
                         const float* w_ih1, const float* b_ih1, const float* b_hh1, const float* fcw, const float* fcb, float* packed_ws,
                         float* gates0, float* mask0, float* hd0, float* gates1, float* mask1, float* hd1, float p_lstm, float p_fc,
                         uint64_t seed, float* logits, hipStream_t stream);   // ww_head.hip

struct TrainWs {
    float *mid, *pooled, *gates0, *mask0, *hd0, *gates1, *mask1, *hd1, *lstm_packed, *conv2_b_op, *dgrad_b_op;
    float *dhd1, *dg1, *dhd0, *dg0, *dpooled, *gp, *wg_partial, *dg_partial;
    int64_t total;
};
static int64_t a256(int64_t floats) { return (floats * 4 + 255) / 256 * 64; }      // floats, 256-byte granules
static TrainWs carve_train(void* base, int64_t n, int grid_w, int grid_d) {
    TrainWs w{};
    float* p = static_cast<float*>(base);
    int64_t o = 0;
    auto take = [&](int64_t floats) { float* at = p ? p + o : nullptr; o += a256(floats); return at; };
    w.mid = take(n * kTH * 64 * kTW);
    w.pooled = take(n * 64);
    w.gates0 = take(4 * n * kHidden); w.mask0 = take(n * kHidden); w.hd0 = take(n * kHidden);
    w.gates1 = take(4 * n * kHidden); w.mask1 = take(n * kHidden); w.hd1 = take(n * kHidden);
    w.lstm_packed = take((64 + kHidden) * kGateCols + 2 * kGateCols);
    w.conv2_b_op = take(2 * 144 * 64);
    w.dgrad_b_op = take(2 * 144 * 64);
    w.dhd1 = take(n * kHidden); w.dg1 = take(n * 4 * kHidden); w.dhd0 = take(n * kHidden); w.dg0 = take(n * 4 * kHidden);
    w.dpooled = take(n * 64); w.gp = take(n * 64);
    w.wg_partial = take(int64_t(grid_w) * kWgPartial);
    w.dg_partial = take(int64_t(grid_d) * kDgPartial);
    w.total = o * 4;
    return w;
}
static void train_grids(int64_t n, int& grid_w, int& grid_d) {
    const int cus = device_cu_count();
    grid_w = int(n < cus ? n : cus);                 // one 8-wave workgroup per CU (117 KB of LDS)
    grid_d = int(n < cus ? n : cus);                 // one 4-wave workgroup per CU (128 KB of LDS)
}

int64_t train_workspace_bytes(int64_t n) {
    return carve_train(nullptr, n, 256, 256).total;  // grids never exceed the CU count (256 on MI355X)
}

static int train_opt_in() {
    static bool done[64] = {};
    int dev = 0;
    WW_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(WW_EINVAL, "device ordinal out of range");
    if (done[dev]) return WW_OK;
    WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv2_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(sizeof(float) * kWgLdsFloats)));
    WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv2_dgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(sizeof(float) * kDgLdsFloats)));
    done[dev] = true;
    return WW_OK;
}

// test / diagnostic: copies of the dropout factors the last forward on this workspace drew ([n][256] each)
int train_masks(const void* workspace, int64_t n, float* mask0, float* mask1, hipStream_t st) {
    TrainWs w = carve_train(const_cast<void*>(workspace), n, 256, 256);
    WW_HIP(hipMemcpyAsync(mask0, w.mask0, sizeof(float) * n * kHidden, hipMemcpyDeviceToDevice, st));
    WW_HIP(hipMemcpyAsync(mask1, w.mask1, sizeof(float) * n * kHidden, hipMemcpyDeviceToDevice, st));
    return WW_OK;
}

int train_forward(const float* mel, int64_t n, int width, const ww_train_params* p, float p_lstm, float p_fc, uint64_t seed, void* workspace,
                  float* logits, hipStream_t st) {
    int gw, gd;
    train_grids(n, gw, gd);
    if (gw > 256 || gd > 256) return fail(WW_EUNSUPPORTED, "more than 256 CUs: the workspace is sized for 256 partials");
    TrainWs w = carve_train(workspace, n, 256, 256);
    hipLaunchKernelGGL(pack_conv_b_dev_kernel, dim3(72), dim3(256), 0, st, p->conv_weight[1], 64, 32, w.conv2_b_op);
    WW_HIP(hipGetLastError());
    if (int rc = launch_cnn2_f32_mid(mel, n, width, p->conv_weight[0], p->conv_bias[0], w.conv2_b_op, p->conv_bias[1], w.mid, st)) return rc;
    hipLaunchKernelGGL(pool_mid_kernel, dim3(int(n)), dim3(256), 0, st, w.mid, int(n), width, w.pooled);
    WW_HIP(hipGetLastError());
    return launch_lstm_fc_train(w.pooled, n, 64, p->lstm_weight_ih[0], p->lstm_bias_ih[0], p->lstm_bias_hh[0], p->lstm_weight_ih[1],
                                p->lstm_bias_ih[1], p->lstm_bias_hh[1], p->fc_weight, p->fc_bias, w.lstm_packed, w.gates0, w.mask0, w.hd0,
                                w.gates1, w.mask1, w.hd1, p_lstm, p_fc, seed, logits, st);
}

int train_backward(const float* mel, int64_t n, int width, const ww_train_params* p, const float* dlogits, void* workspace,
                   const ww_train_grads* g, hipStream_t st) {
    if (int rc = train_opt_in()) return rc;
    int gw, gd;
    train_grids(n, gw, gd);
    TrainWs w = carve_train(workspace, n, 256, 256);
    const int N = int(n), H = kHidden;
    // fc: dW = dlogits^T hd1, db = colsum(dlogits), dhd1 = dlogits W_fc
    sgemm(dlogits, 1, 2, w.hd1, H, 1, g->fc_weight, H, 2, H, N, st);
    hipLaunchKernelGGL(colsum_kernel, dim3(1), dim3(1024), 0, st, dlogits, N, 2, g->fc_bias);
    sgemm(dlogits, 2, 1, p->fc_weight, H, 1, w.dhd1, H, N, H, 2, st);
    // layer 1
    hipLaunchKernelGGL(lstm_gates_bwd_kernel, dim3(1024), dim3(256), 0, st, w.dhd1, w.gates1, w.mask1, N, w.dg1);
    sgemm(w.dg1, 1, 4 * H, w.hd0, H, 1, g->lstm_weight_ih[1], H, 4 * H, H, N, st);             // [1024][256] = dg1^T hd0
    hipLaunchKernelGGL(colsum_kernel, dim3(16), dim3(1024), 0, st, w.dg1, N, 4 * H, g->lstm_bias[1]);
    sgemm(w.dg1, 4 * H, 1, p->lstm_weight_ih[1], H, 1, w.dhd0, H, N, H, 4 * H, st);            // [n][256] = dg1 W_ih_l1
    // layer 0
    hipLaunchKernelGGL(lstm_gates_bwd_kernel, dim3(1024), dim3(256), 0, st, w.dhd0, w.gates0, w.mask0, N, w.dg0);
    sgemm(w.dg0, 1, 4 * H, w.pooled, 64, 1, g->lstm_weight_ih[0], 64, 4 * H, 64, N, st);       // [1024][64] = dg0^T pooled
    hipLaunchKernelGGL(colsum_kernel, dim3(16), dim3(1024), 0, st, w.dg0, N, 4 * H, g->lstm_bias[0]);
    sgemm(w.dg0, 4 * H, 1, p->lstm_weight_ih[0], 64, 1, w.dpooled, 64, N, 64, 4 * H, st);      // [n][64] = dg0 W_ih_l0
    hipLaunchKernelGGL(scale_kernel, dim3(256), dim3(256), 0, st, w.dpooled, 1.0f / float(kTH * width), n * 64, w.gp);
    WW_HIP(hipGetLastError());
    // conv stack
    hipLaunchKernelGGL(pack_dgrad_b_dev_kernel, dim3(72), dim3(256), 0, st, p->conv_weight[1], w.dgrad_b_op);
    hipLaunchKernelGGL(conv2_wgrad_kernel, dim3(gw), dim3(512), sizeof(float) * kWgLdsFloats, st, mel, w.mid, w.gp, N, width, p->conv_weight[0],
                       p->conv_bias[0], w.wg_partial);
    hipLaunchKernelGGL(conv2_dgrad_kernel, dim3(gd), dim3(256), sizeof(float) * kDgLdsFloats, st, mel, w.mid, w.gp, N, width, p->conv_weight[0],
                       p->conv_bias[0], w.dgrad_b_op, w.dg_partial);
    WW_HIP(hipGetLastError());
    // partial layout: [64*32*9 dW2][64 db2] and [32*9 dW1][32 db1]: weights and bias are contiguous in the partial, separate in the grads
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(72), dim3(256), 0, st, w.wg_partial, gw, kWgPartial, w.dg1 /*scratch: dg1 is dead now*/);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(2), dim3(256), 0, st, w.dg_partial, gd, kDgPartial, w.dg1 + kWgPartial);
    WW_HIP(hipGetLastError());
    WW_HIP(hipMemcpyAsync(g->conv_weight[1], w.dg1, sizeof(float) * 64 * 32 * 9, hipMemcpyDeviceToDevice, st));
    WW_HIP(hipMemcpyAsync(g->conv_bias[1], w.dg1 + 64 * 32 * 9, sizeof(float) * 64, hipMemcpyDeviceToDevice, st));
    WW_HIP(hipMemcpyAsync(g->conv_weight[0], w.dg1 + kWgPartial, sizeof(float) * 32 * 9, hipMemcpyDeviceToDevice, st));
    WW_HIP(hipMemcpyAsync(g->conv_bias[0], w.dg1 + kWgPartial + 32 * 9, sizeof(float) * 32, hipMemcpyDeviceToDevice, st));
    return WW_OK;
}

}  // namespace ww
