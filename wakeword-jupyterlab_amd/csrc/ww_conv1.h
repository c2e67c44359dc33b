// conv1 (1 -> 32 channels) on the matrix cores in split precision, shared by the forward kernels (ww_cnn.hip) and the
// split-precision backward kernels (ww_train_h.hip): operand layouts, per-clip scaling, the f16 hi/lo split.
#pragma once
#include "ww_internal.h"

namespace ww {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

// 2*relu(v) = v + |v| : one VALU op, exact, NaN-propagating (the factor 2 is folded into the pool scale)
__device__ __forceinline__ float relu2(float v) { return v + __builtin_fabsf(v); }

constexpr int kC1H = WW_N_MELS;    // image rows

// conv1 on the matrix cores for the producers: one v_mfma_f32_32x32x16_f16 triple per image row,
//   D[ci][x] = sum_k W1'[ci][k] * P'[k][x],  k = 3*dy + dx (9 taps; the other 7 of the 16 are zero), split precision as conv2.
// The log-mel tile is kept as two f16 planes (hi, lo) so the patch operand needs no conversions.  The result lands
// with the column on the lane and 16 channels in registers: descale, 2*relu, split, 8-byte stores
// into the position records.  The bias enters as the MFMA's C operand (16 VGPRs per producer lane, rebuilt per clip).
//
// Dynamic range (so that the split precision holds for ANY finite weights and inputs, not only log-mel in [-80, 0] dB):
//   weights   conv1: one power-of-two scale 2^S1 for the tensor (a wave-uniform descale); conv2 / conv3 / LSTM: every
//             output channel has its own (a per-lane constant of the D layout) -- host, ww_tables.cpp;
//   inputs    P' = x * 2^-e with ONE exponent e per clip, chosen from the clip's max |x| so that max |P'| is in [2^14, 2^15);
//   outputs   the tile holds 2 relu(conv1) * 2^-a with one exponent a per clip chosen from the bound
//             max|x| * max_c sum_k |w1[c][k]| + max |b1|, so that it stays below 2^15 (f16 overflows at 65504) and
//             small activations keep both halves normal;  conv2's descale carries 2^a.
// Powers of two commute with fp32 rounding, so for log-mel inputs the result is what the unscaled arithmetic gives.
constexpr int kMelHRS = 36;                              // f16 plane row stride (columns -1..34)
constexpr int kMelHPlane = (kC1H + 2) * kMelHRS;           // halfs per plane
// One row of conv1 on the matrix cores, split into its three stages so that a producer can run the stages of its
// (up to three) rows side by side: the LDS, MFMA and VALU latencies of one row hide under the other rows' work.
struct Conv1Row {
    half8 ph, pl;       // patch operand B[k = 8h + j][x], hi and lo halves
    f32x16 acc;
};

// The per-clip constants of a producer lane: the accumulator of channel c = 16h + j starts at binit[j] = b1[c] * 2^(S1 - e)
// (the bias in the accumulator's scale) and the finished row is multiplied by sc = 2^(e - a - S1) (wave-uniform).
struct Conv1Scale { f32x16 binit; float sc; };

// floor(log2 |v|) of a normal float (-127 for zero / subnormals, 128 for inf / NaN)
__device__ __forceinline__ int exp_of(float v) { return int((__float_as_uint(v) >> 23) & 0xffu) - 127; }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ float pow2i(int e) { return __uint_as_float(uint32_t(127 + e) << 23); }   // -126 <= e <= 127

// Patch operand of a row: lane (x, h) needs B[k = 8h + j][x], j = 0..7 -- taps 0..7 for the lower half-wave; tap 8 and
// seven zeros for the upper one.  Columns 34 and 35 of every plane row are zero, so the upper half-wave differs from the
// lower one only in its ADDRESSES: eight per-lane offsets, computed once, replace 32 selects per row.  (ds_read_u16_d16
// pairs would also save the packing, but with SRAM ECC on a d16 load clears the other half of its register.)
struct GatherLanes { int o[8]; };      // offsets in halfs relative to the first plane row of the patch, column 0

__device__ __forceinline__ GatherLanes gather_lanes(int x, int h) {
    GatherLanes g;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int lower = (j / 3) * kMelHRS + j % 3 + x;                             // tile coords: row y+dy, col x+dx
        const int upper = j == 0 ? 2 * kMelHRS + 2 + x : 34 + (j & 1);
        g.o[j] = h ? upper : lower;
    }
    return g;
}

__device__ __forceinline__ void conv1_row_gather(Conv1Row& r, const _Float16* __restrict__ mh, const GatherLanes& gl, int y) {
    const _Float16* __restrict__ row = mh + y * kMelHRS;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        r.ph[j] = row[gl.o[j]];
        r.pl[j] = row[gl.o[j] + kMelHPlane];             // the lo plane follows the hi plane
    }
}

__device__ __forceinline__ void conv1_row_mfma(Conv1Row& r, half8 a1h, half8 a1l, const Conv1Scale& cs) {
    r.acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, r.ph, cs.binit, 0, 0, 0);
    r.acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, r.pl, r.acc, 0, 0, 0);
    r.acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, r.ph, r.acc, 0, 0, 0);
}

#ifndef WW_SPLIT_MIX
#define WW_SPLIT_MIX 1
#endif
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

// x ~= hi + lo for two values at once: v_cvt_pk_f16_f32, two v_fma_mix_f32, v_cvt_pk_f16_f32
// (2 VALU instructions per value instead of 6 for the scalar form; same round-to-nearest result)
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    const float2_t v = {a, b};
    const half2_t h = __builtin_convertvector(v, half2_t);
    // a - float(hi) as one mixed-precision fma per value (v_fma_mix_f32 reads the f16 half directly, exact like the subtraction):
    // 4 instructions per pair instead of 5 with a half-rate packed subtract (bit-identical, -3 % on the conv kernel)
    hi = __builtin_bit_cast(uint32_t, h);
#if WW_SPLIT_MIX
    // round 4: hipcc lowers the two fmas above to v_cvt_f32_f16 + v_sub_f32 each (6 instructions per pair); written out, v_fma_mix_f32 reads
    // the f16 half straight from the packed register (op_sel picks the half, op_sel_hi marks it as f16): 4 per pair, same bits
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(b));
    const float2_t r = {r0, r1};
#else
    const float2_t r = {__builtin_fmaf(static_cast<float>(h[0]), -1.0f, a), __builtin_fmaf(static_cast<float>(h[1]), -1.0f, b)};
#endif
    const half2_t l = __builtin_convertvector(r, half2_t);
    lo = __builtin_bit_cast(uint32_t, l);
}

}  // namespace ww
