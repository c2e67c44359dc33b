// K1: 16 kHz PCM -> log-mel dB [80][32] per clip, one persistent 4-wave workgroup per clip stream.
//
// Replaces normalize_audio + zero-pad + librosa.feature.melspectrogram + power_to_db(ref=np.max)
// (/root/reference/wakeword_training_script.py:73-101).  Per clip:
//
//   32 frames of 2048 samples (hop 512, centred: 1024 zeros either side), one frame per WAVE at a time:
//     z[n] = w[2n] x[2n] + i w[2n+1] x[2n+1]                      (1024 complex points, 16 per lane)
//     1024-point complex FFT = radix 8 x 8 x 16 in registers, two exchanges through the wave's
//       private 8 KiB LDS slab (no workgroup barrier: LDS is in-order per wave)
//     real-input split  X[k] = E[k] + W_2048^k O[k],  |X[k]|^2 and |X[1024-k]|^2 from the pair (Z[k], Z[1024-k])
//   every 4 frames: sparse mel (2004 non-zeros as 8-bin pieces) over the 4 power spectra in LDS
//   after 32 frames: per-clip max, 10 log10, clamp at -80 dB, coalesced store.
//
// Peak normalisation is deferred: the FFT is linear, so |X(x/p)|^2 = |X(x)|^2 / p^2 and the gain is
// applied to the 2560 mel powers instead of the 16000 samples (the peak falls out of the frame loads).
//
// Algorithmic HBM bytes per clip: 64,000 read + 10,240 written = 74,240.
#include <mutex>

#include "ww_internal.h"

namespace ww {

// wave slab: 1024 complex = 2048 floats (+4 keeps 16-byte alignment and staggers the slabs over the banks)
constexpr int kSlab = 2048 + 4;
constexpr int kMelStride = kFrames + 1;
// The per-wave piece sums live in the upper half of the wave's slab (the power spectrum only needs floats 0..1024).
constexpr int kPartialInSlab = 1032;
static_assert(kPartialInSlab + kPieces <= kSlab, "piece sums must fit behind the power spectrum");

// LDS map (floats): slabs | mel | reduce scratch | piece index | filter index | pass-2 twiddles | pair twiddles.
// Window, pass-1 twiddles and piece weights are read through L1 from the global table (25 KB less LDS per workgroup).
// WAVES = 4: the throughput form, three workgroups per CU, persistent over clips.
// WAVES = 8: the latency form for batches of at most one clip per CU (streaming): a clip's 32 frames take 4 rounds
//            instead of 8, one 83 KB workgroup per CU.
// WW_K1_RESIDENT 1: the lane's 32 window values and 28 pass-1 twiddles live in registers for the life of the workgroup
// (60 VGPRs) instead of being re-read through L1 for every frame; two waves per SIMD instead of three.
// WW_K1_RESIDENT 2: also, a wave takes CONSECUTIVE frames of a clip and keeps the raw samples: frame t + 1 is frame t moved on
// by 512 samples = two of the eight 256-sample groups, so six groups are register moves and only two are loaded.
#ifndef WW_K1_RESIDENT
#define WW_K1_RESIDENT 0
#endif
template <int WAVES>
struct K1Layout {
    static constexpr int kWaves = WAVES;
    static constexpr int kThreads = WAVES * 64;
    static constexpr int kOffMel = WAVES * kSlab;
    static constexpr int kOffRed = kOffMel + kMels * kMelStride;
    static constexpr int kOffFrm = kOffRed + 16;            // auto mode: [WAVES][32] per-frame energy partials, [80] 1 / wmax_b, [4] redo flag
    static constexpr int kOffInvw = kOffFrm + WAVES * 32;
    static constexpr int kOffFlag = kOffInvw + kMels;
    static constexpr int kOffDcNy = kOffFlag + 4;           // auto mode: [32] |X_0|^2 + |X_1024|^2 per frame (the two bins no mel band sees)
    static constexpr int kOffPinfo = kOffDcNy + 32;         // [kPieces] ints
    static constexpr int kOffFp0 = kOffPinfo + kPieces;     // [80] ints
    static constexpr int kOffFcnt = kOffFp0 + kMels;        // [80] ints
    static constexpr int kOffTw2 = kOffFcnt + kMels;        // [7][16] float2
    static constexpr int kOffTwp = kOffTw2 + 7 * 16 * 2;    // [512] float2
    static constexpr int kLdsFloats = kOffTwp + 512 * 2;
    static constexpr int kWavesPerSimd = WAVES == 4 ? (WW_K1_RESIDENT ? 2 : 3) : 2;   // launch bound: 3 (2) x 4 waves or 1 x 8 waves per CU
    static constexpr int kBlocksPerCu = WAVES == 4 ? (WW_K1_RESIDENT ? 2 : 3) : 1;
    static_assert(kOffPinfo % 4 == 0 && kOffTw2 % 4 == 0 && kOffTwp % 2 == 0, "LDS table alignment");
    static_assert(kFrames % WAVES == 0 && WAVES <= 8, "frames are dealt to the waves in whole rounds; red[] holds 16 floats");
};

__device__ __forceinline__ float2 operator+(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 operator-(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 mul_neg_i(float2 a) { return make_float2(a.y, -a.x); }   // a * (-i)

// forward DFTs (e^{-2 pi i nk/N}), natural order in and out, all indices static -> registers
__device__ __forceinline__ void dft4(float2& a0, float2& a1, float2& a2, float2& a3) {
    const float2 t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = mul_neg_i(a1 - a3);
    a0 = t0 + t2; a2 = t0 - t2; a1 = t1 + t3; a3 = t1 - t3;
}

__device__ __forceinline__ void dft8(float2 (&v)[8]) {
    constexpr float c = 0.70710678118654752440f;
    float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    float2 o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    dft4(e0, e1, e2, e3);
    dft4(o0, o1, o2, o3);
    o1 = make_float2(c * (o1.x + o1.y), c * (o1.y - o1.x));      // * W8^1 = (1 - i)/sqrt2
    o2 = mul_neg_i(o2);                                           // * W8^2 = -i
    o3 = make_float2(c * (o3.y - o3.x), -c * (o3.x + o3.y));      // * W8^3 = (-1 - i)/sqrt2
    v[0] = e0 + o0; v[4] = e0 - o0;
    v[1] = e1 + o1; v[5] = e1 - o1;
    v[2] = e2 + o2; v[6] = e2 - o2;
    v[3] = e3 + o3; v[7] = e3 - o3;
}

__device__ __forceinline__ void dft16(float2 (&v)[16]) {
    float2 e[8], o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { e[i] = v[2 * i]; o[i] = v[2 * i + 1]; }
    dft8(e);
    dft8(o);
    constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;   // cos, sin(pi/8)
    constexpr float c2 = 0.70710678118654752440f;
    const float2 w[8] = {{1.f, 0.f}, {c1, -s1}, {c2, -c2}, {s1, -c1}, {0.f, -1.f}, {-s1, -c1}, {-c2, -c2}, {-c1, -s1}};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float2 t = (k == 0) ? o[0] : (k == 4 ? mul_neg_i(o[4]) : cmul(o[k], w[k]));
        v[k] = e[k] + t;
        v[k + 8] = e[k] - t;
    }
}

// 10*log10(x) as its own rounded product (no fma contraction with the following subtract), so that the
// per-clip maximum maps to exactly 0 dB like librosa's `log_spec -= 10*log10(ref)`.
__device__ __forceinline__ float db10(float x) {
#pragma clang fp contract(off)
    return 10.0f * log10f(x);
}

__device__ __forceinline__ void lds_order() {
    // LDS traffic of one wave is in order; this only stops the compiler from moving accesses across phases.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// max(|a|, |b|, m) in one v_max3_f32 with |.| source modifiers (fmaxf semantics: a NaN operand is ignored; m >= 0).
// Written as fmaxf(fabsf ..) the compiler adds a canonicalising v_max per operand: 43 instructions per frame for 8 samples.
__device__ __forceinline__ float absmax3(float a, float b, float m) {
    float r;
    asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(r) : "v"(a), "v"(b), "v"(m));
    return r;
}

// Eight 16-byte loads of one frame (samples base + 256*n1 .. +3) through a buffer descriptor whose range is exactly the
// clip: the hardware returns 0 for every dword outside [0, clip_len) -- the STFT's centre padding, the right zero-pad
// of short clips and "no next clip" (a zero-record descriptor) cost no compare, select or branch, and the eight loads
// stay in flight together.  Everything goes into the per-lane offset: the scalar offset of a buffer instruction is not
// range-checked.  Ring mode (streaming): sample i of the window lives at (pos + i) mod len.
using u32x4_t = __attribute__((ext_vector_type(4))) unsigned int;

template <bool RING, int N0 = 0, int N1 = 8>
__device__ __forceinline__ void load_frame(float4 (&sn)[8], __amdgpu_buffer_rsrc_t rsrc, int base, int ring_pos, int ring_len) {
#pragma unroll
    for (int n1 = N0; n1 < N1; ++n1) {
        const int idx = base + 256 * n1;              // multiple of 4; may be negative or past the end
        int off = idx * 4;
        if constexpr (RING) {
            int at = idx + ring_pos;
            at = at >= ring_len ? at - ring_len : at;
            off = (idx >= 0 && idx < ring_len) ? at * 4 : -1;
        }
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        sn[n1] = __builtin_bit_cast(float4, v);
    }
}

// NaN with a payload no arithmetic produces: "finish this clip in float64".  Auto mode, round 4: the float kernel's floor test is
// evaluated per FRAME, and the redo is per frame too: for a marked clip the float kernel leaves
//     o[0] = kRedoMark,  o[1] = the frames to redo as a bit mask (bits 0 and 1 always set: those two words sit in frames 0 and 1 of band 0),
//     o[band * 32 + frame] = the mel POWER (after the peak gain, before amin / log) everywhere else,
// and logmel64_kernel<.., true> recomputes only the masked frames in float64, takes the other frames' powers as they are and does the
// clip's dB epilogue (round 3 redid all 32 frames of a marked clip).
constexpr uint32_t kRedoMark = 0x7fc5a11eu;
// Auto mode's test.  The float FFT leaves rounding noise of about 2.4 eps^2 E / 1024 per bin under a frame of energy
// E = sum_k |X_k|^2; a band's power P_b = sum_k M_bk |X_k|^2 then carries a relative error of ~2 sigma sqrt(wmax_b / P_b),
// which exceeds 1e-4 dB below P_b / wmax_b ~ 4e-6 E (measured on MI355X: scripts/diag_floor.py).  A frame with a live band
// below kFloorRatio * E sends its clip to the float64 kernel.  E is estimated from the mel tile itself: the triangles
// M_bk / wmax_b are a partition of unity over the bins 1..1023, so sum_b P_b / wmax_b ~ E without the DC and Nyquist bins, which carry
// zero mel weight; those two are added from Z[0] (round 3: a pure Nyquist tone was missed without them, 1.07e-4 dB).
#ifndef WW_FLOOR_RATIO
#define WW_FLOOR_RATIO 1.0e-5f
#endif
constexpr float kFloorRatio = WW_FLOOR_RATIO;

#ifdef WW_STAMPS
__device__ unsigned long long g_stamps[16];
#define STAMP(i) do { unsigned long long t__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__) :: "memory"); \
    if (stamp_on) { acc_st[i] += t__ - last_st; } last_st = t__; } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

// mark != 0 (auto mode): a clip with a live mel band on the float FFT's rounding floor gets kRedoMark in the first word of its
// output and is recomputed by logmel64_kernel<RING, true>, which runs behind this kernel on the same stream.
template <bool RING, int WAVES>
__global__ __launch_bounds__(WAVES * 64, K1Layout<WAVES>::kWavesPerSimd) void logmel_kernel(const float* __restrict__ pcm, int64_t clip_stride,
                                                             int clip_len, int n_clips, int normalize,
                                                             const int32_t* __restrict__ ring_pos_p, int ring_len,
                                                             const LogmelTables* __restrict__ tb,
                                                             float* __restrict__ out, int mark) {
    using L = K1Layout<WAVES>;
    constexpr int kWavesPerBlock = WAVES, kThreads = L::kThreads;
    constexpr int kOffMel = L::kOffMel, kOffRed = L::kOffRed, kOffFrm = L::kOffFrm, kOffInvw = L::kOffInvw, kOffFlag = L::kOffFlag,
                  kOffDcNy = L::kOffDcNy, kOffPinfo = L::kOffPinfo, kOffFp0 = L::kOffFp0,
                  kOffFcnt = L::kOffFcnt, kOffTw2 = L::kOffTw2, kOffTwp = L::kOffTwp;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* mel = lds + kOffMel;                                 // [80][33]
    float* red = lds + kOffRed;                                 // [16]
    const float4* pw4 = reinterpret_cast<const float4*>(&tb->piece_w[0][0][0]);
    const int* pinfo = reinterpret_cast<const int*>(lds + kOffPinfo);
    const int* fp0 = reinterpret_cast<const int*>(lds + kOffFp0);
    const int* fcnt = reinterpret_cast<const int*>(lds + kOffFcnt);
    const float4* tw2_4 = reinterpret_cast<const float4*>(lds + kOffTw2);   // [7][8] float4 = two twiddles each
    const float2* twp_2 = reinterpret_cast<const float2*>(lds + kOffTwp);

    const int tid = threadIdx.x;
    const int lane_id = tid & 63;
    const int lane = lane_id;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* slab = lds + wave * kSlab;
    float2* slab2 = reinterpret_cast<float2*>(slab);
    float4* slab4 = reinterpret_cast<float4*>(slab);
    float* partial = slab + kPartialInSlab;                     // this wave's piece sums, filter-major
    const float4* tw1_4 = reinterpret_cast<const float4*>(&tb->tw1[0][0]);
    const float4* win4 = reinterpret_cast<const float4*>(&tb->window[0]);

    // ---- all tables into LDS once per workgroup (conflict-free, lane-contiguous reads) ----
    for (int i = tid; i < kPieces; i += kThreads) reinterpret_cast<int*>(lds)[kOffPinfo + i] = tb->piece_info[i];
    if (tid < kMels) {
        reinterpret_cast<int*>(lds)[kOffFp0 + tid] = tb->filt_p0[tid];
        reinterpret_cast<int*>(lds)[kOffFcnt + tid] = tb->filt_cnt[tid];
        lds[kOffInvw + tid] = tb->band_bins[tid];                        // 1 / wmax_b
    }
    if (tid < 4) reinterpret_cast<uint32_t*>(lds)[kOffFlag + tid] = 0u;      // [2] frame masks (auto mode), by clip parity
    for (int i = tid; i < 7 * 16 * 2; i += kThreads) lds[kOffTw2 + i] = (&tb->tw2[0][0].x)[i];
    for (int i = tid; i < 512 * 2; i += kThreads) lds[kOffTwp + i] = (&tb->twp[0].x)[i];
    const int ring_pos = RING ? *ring_pos_p : 0;
    __syncthreads();
    // this lane's filters in the per-frame combine: f = lane and f = lane + 64 (< 80)
    const int my_p0a = fp0[lane], my_cnta = fcnt[lane];
    const int my_p0b = lane + 64 < kMels ? fp0[lane + 64] : 0, my_cntb = lane + 64 < kMels ? fcnt[lane + 64] : 0;

    // exchange layouts (float4 units inside the wave slab), chosen so that every ds_read/write_b128 below is
    // bank-conflict free for the hardware's 16-lane groups:
    //   X1: y[k1][n'/2]       at k1*64 + ((n'/2) ^ (8 * ((k1>>1)&1)))
    //   X2: u[reader lane][m] at reader*8 + (m ^ ((reader>>1)&7)),  reader = 8*k1 + k2, m = n''/2
    //   Z : Z[k]              at float2 index k ^ (((k>>4)&3) << 1)
#ifdef WW_STAMPS
    const bool stamp_on = true;
    unsigned long long acc_st[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, last_st = 0;
    STAMP(9);
#endif

    // the samples of a frame are fetched one frame ahead (8 x dwordx4 per lane in flight under the FFT); the chain runs
    // across clip boundaries, so only the very first frame of a workgroup is loaded synchronously
    const unsigned clip_bytes = unsigned(clip_len) * 4u;
    auto clip_rsrc = [&](int c) {        // descriptor of clip c, or an empty one past the end: all-zero loads
        const bool ok = c < n_clips;
        const float* base = pcm + int64_t(ok ? c : 0) * clip_stride;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, ok ? clip_bytes : 0u, 0x00020000);
    };
    constexpr int kRounds = kFrames / kWavesPerBlock;
#if WW_K1_RESIDENT == 2
    const int frame0 = wave * kRounds;                 // this wave's frames: frame0 .. frame0 + kRounds - 1
#else
    const int frame0 = wave;                           // frames wave, wave + WAVES, ...
#endif
    float4 sn[8];
    load_frame<RING>(sn, clip_rsrc(blockIdx.x), frame0 * kHop - kNfft / 2 + 4 * lane, ring_pos, ring_len);
#if WW_K1_RESIDENT
    float4 wres[8], t1res[7];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) wres[n1] = win4[64 * n1 + lane_id];
#pragma unroll
    for (int k1 = 1; k1 < 8; ++k1) t1res[k1 - 1] = tw1_4[(k1 - 1) * 64 + lane_id];
#endif
    int clip_it = 0;
#pragma unroll 1
    for (int clip = blockIdx.x; clip < n_clips; clip += gridDim.x, ++clip_it) {
        const __amdgpu_buffer_rsrc_t rs_cur = clip_rsrc(clip), rs_next = clip_rsrc(clip + int(gridDim.x));
        float peak = 0.f;
        // `sn` was prefetched: by the prologue for the first clip, by the previous clip's last frame otherwise

#pragma unroll 1
        for (int round = 0; round < kRounds; ++round) {
#if WW_K1_RESIDENT == 2
            const int frame = frame0 + round;
#else
            const int frame = round * kWavesPerBlock + wave;
#endif
            // Opaque copy of the lane id: every swizzled LDS address below is a function of it.  Without this the
            // compiler hoists ~100 loop-invariant address VGPRs out of the frame loop and spills the prefetched samples.
            int lane = lane_id;
            asm volatile("" : "+v"(lane));
            const int k1r = lane >> 3, jr = lane & 7;                   // pass-2 role: (k1, j)
#if WW_K1_RESIDENT == 2
            const int base_next = (frame + 1) * kHop - kNfft / 2 + 4 * lane;
#else
            const int base_next = (frame + kWavesPerBlock) * kHop - kNfft / 2 + 4 * lane;
#endif

            STAMP(8);
            // ---- load + window: lane holds z[128 n1 + 2 lane + q], q = 0,1, n1 = 0..7 ----
            float2 za[8], zb[8];
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) {
                const float4 s = sn[n1];
#if WW_K1_RESIDENT
                const float4 w = wres[n1];
#elif defined(WW_K1_ABL_NOTAB)      // timing-only ablation: no window / pass-1 twiddle loads (results are garbage)
                const float4 w = make_float4(0.5f, 0.25f, 0.75f, 1.0f);
#else
                const float4 w = win4[64 * n1 + lane];
#endif
                // every sample sits in 4 frames; samples [512 t, 512 t + 512) = loads n1 4 and 5 of frame t tile the clip once
                if (n1 == 4 || n1 == 5) peak = absmax3(s.z, s.w, absmax3(s.x, s.y, peak));
                za[n1] = make_float2(s.x * w.x, s.y * w.y);
                zb[n1] = make_float2(s.z * w.z, s.w * w.w);
            }
            // ---- pass 1: radix 8 over n1 (stride 128), twiddle W_1024^{n' k1}, store X1 ----
            // Table operands are fetched as a group BEFORE the arithmetic they follow: written at their use, each load
            // is issued right there and the wave pays one L1 / LDS round trip per twiddle.
            float4 t1[7];
#pragma unroll
#if WW_K1_RESIDENT
            for (int k1 = 1; k1 < 8; ++k1) t1[k1 - 1] = t1res[k1 - 1];
#elif defined(WW_K1_ABL_NOTAB)
            for (int k1 = 1; k1 < 8; ++k1) t1[k1 - 1] = make_float4(0.7f, -0.7f, 0.6f, -0.8f);
#else
            for (int k1 = 1; k1 < 8; ++k1) t1[k1 - 1] = tw1_4[(k1 - 1) * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);
#endif
            dft8(za);
            dft8(zb);
#ifdef WW_K1_ABL_NOX1                  // timing-only ablation: exchange 1 without its LDS round trip (results are garbage)
#pragma unroll
            for (int k1 = 1; k1 < 8; ++k1) {
                const float4 t = t1[k1 - 1];
                za[k1] = cmul(za[k1], make_float2(t.x, t.y));
                zb[k1] = cmul(zb[k1], make_float2(t.z, t.w));
            }
#else
            slab4[lane] = make_float4(za[0].x, za[0].y, zb[0].x, zb[0].y);
#pragma unroll
            for (int k1 = 1; k1 < 8; ++k1) {
                const float4 t = t1[k1 - 1];
                const float2 a = cmul(za[k1], make_float2(t.x, t.y));
                const float2 b = cmul(zb[k1], make_float2(t.z, t.w));
                slab4[k1 * 64 + (lane ^ (8 * ((k1 >> 1) & 1)))] = make_float4(a.x, a.y, b.x, b.y);
            }
#endif
            lds_order();
            STAMP(0);
            // ---- pass 2: lane = (k1, j): radix 8 over n2 of y[k1][16 n2 + 2j + q]; twiddle W_128; store X2 ----
            {
                // (n2*8 + j) ^ 8s = (n2 ^ s)*8 + j: even n2 read from base + 8s, odd n2 from base - 8s, at immediate n2*8
                const int s8 = 8 * ((k1r >> 1) & 1);
                const float4* x1e = slab4 + k1r * 64 + jr + s8;
                const float4* x1o = slab4 + k1r * 64 + jr - s8;
#ifndef WW_K1_ABL_NOX1
#pragma unroll
                for (int n2 = 0; n2 < 8; ++n2) {
                    const float4 v = (n2 & 1) ? x1o[n2 * 8] : x1e[n2 * 8];
                    za[n2] = make_float2(v.x, v.y);
                    zb[n2] = make_float2(v.z, v.w);
                }
#else
                asm volatile("" :: "v"(x1e), "v"(x1o));
#endif
                float4 t2[7];
#pragma unroll
                for (int k2 = 1; k2 < 8; ++k2) t2[k2 - 1] = tw2_4[(k2 - 1) * 8 + jr];
                lds_order();
                __builtin_amdgcn_sched_barrier(0);
#ifndef WW_K1_ABL_NODFT8P2             // timing-only ablation: pass 2's butterflies
                dft8(za);
                dft8(zb);
#endif
                // writer (k1, j), reader lane 8 k1 + k2, slot j ^ ((reader >> 1) & 7) = j ^ (4 (k1 & 1) + (k2 >> 1)):
                // four lane bases (one per k2 >> 1), everything else is an immediate offset
                float4* x2w[4];
#pragma unroll
                for (int hk = 0; hk < 4; ++hk) x2w[hk] = slab4 + 64 * k1r + (jr ^ (4 * (k1r & 1) + hk));
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) {
                    float2 a = za[k2], b = zb[k2];
                    if (k2 > 0) {
                        const float4 t = t2[k2 - 1];
                        a = cmul(a, make_float2(t.x, t.y));
                        b = cmul(b, make_float2(t.z, t.w));
                    }
#ifdef WW_K1_ABL_NOX2                  // timing-only ablation: exchange 2 without its LDS round trip
                    za[k2] = a; zb[k2] = b;
#else
                    x2w[k2 >> 1][8 * k2] = make_float4(a.x, a.y, b.x, b.y);
#endif
                }
            }
            lds_order();
            STAMP(1);
            // ---- pass 3: lane = (k1, k2): radix 16 over n'' -> Z[k1 + 8 k2 + 64 k''] ----
            {
                float2 u[16];
                const int sw2 = (lane >> 1) & 7;
#ifdef WW_K1_ABL_NOX2
#pragma unroll
                for (int m = 0; m < 8; ++m) { u[2 * m] = za[m]; u[2 * m + 1] = zb[m]; }
                asm volatile("" :: "v"(sw2));
#else
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const float4 v = slab4[lane * 8 + (m ^ sw2)];
                    u[2 * m] = make_float2(v.x, v.y);
                    u[2 * m + 1] = make_float2(v.z, v.w);
                }
#endif
                lds_order();
#ifndef WW_K1_ABL_NODFT16              // timing-only ablation: what pass 3's butterflies cost the vector ALU (results are garbage)
                dft16(u);
#endif
                const int lp = (lane >> 3) + 8 * (lane & 7);
                float2* zw = slab2 + (lp ^ (((lp >> 4) & 3) << 1));     // bits 4-5 of k = lp + 64 kk are lp's: one base
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) zw[64 * kk] = u[kk];
            }
            lds_order();
            STAMP(2);
            // mel-piece weights of this lane (L1 hits): issued ahead of the frame prefetch because vmcnt retires in
            // order -- the mel stage then waits for them with the eight sample loads still in flight
            float4 pw0[kPieceRounds], pw1[kPieceRounds];
#pragma unroll
            for (int c = 0; c < kPieceRounds; ++c) {
                pw0[c] = pw4[c * kPieceSlots + lane];
                pw1[c] = pw4[kPieces + c * kPieceSlots + lane];
            }
            // next frame's samples: issued here, after the register-hungry FFT passes, and in flight under the
            // power / mel stages (about a third of the frame time, several times the HBM latency)
            {
                const bool last = round + 1 == kRounds;            // uniform: descriptor select, no branch
#if WW_K1_RESIDENT == 2
                // the next frame's groups 0..5 are this frame's 2..7; groups 6, 7 are new.  On the clip's last round these are
                // the next clip's first frame's 6, 7 and its groups 0..5 are loaded behind the round loop.
#pragma unroll
                for (int n1 = 0; n1 < 6; ++n1) sn[n1] = sn[n1 + 2];
                load_frame<RING, 6, 8>(sn, last ? rs_next : rs_cur, last ? frame0 * kHop - kNfft / 2 + 4 * lane : base_next, ring_pos, ring_len);
#else
                load_frame<RING>(sn, last ? rs_next : rs_cur, last ? frame0 * kHop - kNfft / 2 + 4 * lane : base_next, ring_pos,
                                 ring_len);
#endif
            }
            // ---- real-input split + power: bins k = lane + 64 j and 1024 - k ----
            {
                float2 a[8], b[8];
                // k = lane + 64 j and 1024 - k = (64 - lane) + 64 (15 - j): the swizzle only involves the low 6 bits, so
                // both streams are one lane base plus immediates (lane 0 pairs bin 512 with itself at j = 0, and its other
                // partners 1024 - 64 j sit at 64 (16 - j) with a zero swizzle)
                const int lowb = (64 - lane) & 63;
                const float2* za_p = slab2 + (lane ^ (((lane >> 4) & 3) << 1));
                const float2* zb_p = slab2 + (lowb ^ (((lowb >> 4) & 3) << 1)) + (lane == 0 ? 64 : 0);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    a[j] = za_p[64 * j];
                    b[j] = zb_p[64 * (15 - j)];
                }
                if (lane == 0) {
                    // lane 0's first read is Z[0] = (sum of even samples, sum of odd samples): X_0 = Zr + Zi, X_1024 = Zr - Zi.  No mel band
                    // sees these two bins, but their energy feeds the float FFT's rounding floor like any other bin's: auto mode's
                    // frame-energy estimate needs it (a Nyquist- or DC-dominated frame otherwise looks 60 dB quieter than it is)
                    if (mark) lds[kOffDcNy + frame] = 2.0f * fmaf(a[0].x, a[0].x, a[0].y * a[0].y);   // (Zr+Zi)^2 + (Zr-Zi)^2
                    a[0] = slab2[512];
                    b[0] = a[0];
                }
                float2 twj[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) twj[j] = twp_2[lane + 64 * j];
                lds_order();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = (j == 0 && lane == 0) ? 512 : lane + 64 * j;
                    const float2 tw = twj[j];
                    const float2 e = make_float2(0.5f * (a[j].x + b[j].x), 0.5f * (a[j].y - b[j].y));
                    const float2 o = make_float2(0.5f * (a[j].y + b[j].y), 0.5f * (b[j].x - a[j].x));   // (a - conj b)/(2i)
                    const float2 t = cmul(o, tw);
                    const float2 p = e + t, m = e - t;
                    slab[k] = fmaf(p.x, p.x, p.y * p.y);
                    slab[1024 - k] = fmaf(m.x, m.x, m.y * m.y);
                }
                if (lane == 0) slab[0] = 0.f;     // bin 0 carries zero mel weight but is read by window 0
            }
            lds_order();
            STAMP(3);
            // ---- sparse mel of THIS wave's frame: lane = slot, 5 pieces each; then per-filter sums ----
            {
                int info[kPieceRounds];
                float4 s0[kPieceRounds], s1[kPieceRounds];
#pragma unroll
                for (int c = 0; c < kPieceRounds; ++c) info[c] = pinfo[c * kPieceSlots + lane];
#pragma unroll
                for (int c = 0; c < kPieceRounds; ++c) {
                    const float4* s4 = reinterpret_cast<const float4*>(slab + (info[c] & 0xffff));
                    s0[c] = s4[0];
                    s1[c] = s4[1];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < kPieceRounds; ++c) {
                    const float4 w0 = pw0[c], w1 = pw1[c];
                    float acc = w0.x * s0[c].x;
                    acc = fmaf(w0.y, s0[c].y, acc);
                    acc = fmaf(w0.z, s0[c].z, acc);
                    acc = fmaf(w0.w, s0[c].w, acc);
                    acc = fmaf(w1.x, s1[c].x, acc);
                    acc = fmaf(w1.y, s1[c].y, acc);
                    acc = fmaf(w1.z, s1[c].z, acc);
                    acc = fmaf(w1.w, s1[c].w, acc);
                    partial[info[c] >> 16] = acc;
                }
                lds_order();
                STAMP(4);
                // per-filter sums in fixed order; all (<= 10) partials are fetched before the first add so that the
                // LDS latency is paid once, not per term
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    const int f = lane + 64 * pass;
                    const int p0 = pass ? my_p0b : my_p0a, cnt = pass ? my_cntb : my_cnta;
                    if (f < kMels) {
                        float v[10];
#pragma unroll
                        for (int q = 0; q < 10; ++q) v[q] = partial[p0 + (q < cnt ? q : 0)];
                        float acc = 0.f;
#pragma unroll
                        for (int q = 0; q < 10; ++q) acc += (q < cnt) ? v[q] : 0.f;
                        mel[f * kMelStride + frame] = acc;
                    }
                }
            }
            lds_order();   // the slab and `partial` are rewritten by the next frame
            STAMP(5);
        }
#if WW_K1_RESIDENT == 2
        load_frame<RING, 0, 6>(sn, rs_next, frame0 * kHop - kNfft / 2 + 4 * lane_id, ring_pos, ring_len);   // in flight under the epilogue
#endif
        __syncthreads();   // all 32 frames' mel bands are in LDS
        STAMP(6);
        // auto mode's frame mask lives in flag[clip_it & 1]; the other word (the previous clip's, whose readers are all past the barrier
        // above) is cleared here for the next clip
        uint32_t* flag = reinterpret_cast<uint32_t*>(lds) + kOffFlag;
        if (mark && tid == 0) flag[(clip_it + 1) & 1] = 0u;

        // ---- per-clip peak and mel max ----
        // auto mode rides on this pass: every idx of a thread belongs to frame tid & 31, so the thread also sums its share of
        // the frame's energy estimate  E = sum_b P_b / wmax_b  (the triangles M_bk / wmax_b are a partition of unity over the bins)
        float mmax = 0.f, e_part = 0.f;
        const float* invw = lds + kOffInvw;
        float* frm = lds + kOffFrm;
        for (int idx = tid; idx < kMels * kFrames; idx += kThreads) {
            const float p = mel[(idx >> 5) * kMelStride + (idx & 31)];
            mmax = fmaxf(mmax, p);
            if (mark) e_part = fmaf(p, invw[idx >> 5], e_part);
        }
        if (mark) {
            e_part += __shfl_xor(e_part, 32);
            if (lane < 32) frm[wave * 32 + lane] = e_part;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mmax = fmaxf(mmax, __shfl_xor(mmax, off));
            peak = fmaxf(peak, __shfl_xor(peak, off));
        }
        if (lane == 0) { red[wave] = mmax; red[kWavesPerBlock + wave] = peak; }
        __syncthreads();
        mmax = red[0];
        peak = red[kWavesPerBlock];
#pragma unroll
        for (int w = 1; w < kWavesPerBlock; ++w) { mmax = fmaxf(mmax, red[w]); peak = fmaxf(peak, red[kWavesPerBlock + w]); }

        // power_to_db(S, ref=np.max, amin=1e-10, top_db=80): NaN must propagate (silent clip, 0/0)
        const float amin = 1e-10f;
        float g2 = 1.f;
        if (normalize) { const float g = 1.0f / peak; g2 = g * g; }
        float ref = mmax * g2;
        ref = ref < amin ? amin : ref;
        const float ref_db = db10(ref);
        float* __restrict__ o = out + int64_t(clip) * (kMels * kFrames);
        // auto mode: a LIVE band (not clamped by top_db -- with a 1 % margin -- nor by amin) whose P_b / wmax_b lies below
        // kFloorRatio * E of its frame sits on the float FFT's rounding floor: the clip is marked for the float64 kernel
        float floor_e = 0.f;
        if (mark) {
#pragma unroll
            for (int w = 0; w < kWavesPerBlock; ++w) floor_e += frm[w * 32 + (tid & 31)];
            floor_e += lds[kOffDcNy + (tid & 31)];
            floor_e *= kFloorRatio;
        }
        const float live_thr = fmaxf(mmax * 0.99e-8f, amin / g2);
        bool redo = false;
        for (int idx = tid; idx < kMels * kFrames; idx += kThreads) {
            const float p = mel[(idx >> 5) * kMelStride + (idx & 31)];
            float v = p * g2;
            v = v < amin ? amin : v;
            float db = db10(v) - ref_db;
            db = db < -80.0f ? -80.0f : db;
            o[idx] = db;
            if (mark) redo |= p > live_thr && p * invw[idx >> 5] < floor_e;     // false for NaN
        }
        if (mark) {
            // every idx of a thread belongs to frame tid & 31 = lane & 31: lanes l and l + 32 of a wave vote for the same frame
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(redo);
            const uint32_t fm = uint32_t(bal) | uint32_t(bal >> 32);
            if (fm != 0u && lane == 0) __hip_atomic_fetch_or(flag + (clip_it & 1), fm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();   // the frame mask is complete (mel / red / frm are rewritten only behind the next clip's barriers)
        if (mark) {
            const uint32_t fm = flag[clip_it & 1];                           // uniform
            if (fm != 0u) {
                // a marked clip: powers instead of dB (the float64 kernel finishes it); words 0 and 1 carry the mark and the mask
                for (int idx = tid; idx < kMels * kFrames; idx += kThreads)
                    if (idx >= 2) o[idx] = mel[(idx >> 5) * kMelStride + (idx & 31)] * g2;
                if (tid == 0) {
                    reinterpret_cast<uint32_t*>(o)[0] = kRedoMark;
                    reinterpret_cast<uint32_t*>(o)[1] = fm | 3u;
                }
            }
        }
        STAMP(7);
    }
#ifdef WW_STAMPS
    if (lane == 0 && wave == 1 && blockIdx.x == 7)
        for (int i = 0; i < 10; ++i) atomicAdd(&g_stamps[i], acc_st[i]);
#endif
}
#ifdef WW_STAMPS
extern "C" __attribute__((visibility("default"))) int ww_debug_stamps(unsigned long long* out) {
    unsigned long long z[16] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(z)) != hipSuccess) return -1;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z));
    return 0;
}
#endif

// ------------------------------------------------------------------------------------------------
// Precise form: the SAME transform with the window product, the 1024-point FFT and the real-input split in float64 --
// what the reference computes (librosa forms window * frame in float64 and calls numpy.fft.rfft, which runs in double,
// then rounds the spectrum to complex64: /root/reference/wakeword_training_script.py:89-98).  A float32 FFT leaves a
// rounding floor ~165 dB under the frame's energy; mel bands of noise-free signals (pure tones, clean speech with digital
// silence) that lie 60-80 dB under the clip's peak sit on it and come out up to 3.4e-4 dB off.  This kernel is selected
// by ww_set_logmel_math(WW_LOGMEL_MATH_F64), or per clip by the auto mode (below).
// Same radix 8 x 8 x 16 structure and the same exchange index maps as the float kernel.  The float kernel's exchange unit is one float4 =
// (a, b), two complex floats of the lane's two interleaved sub-transforms; here a and b are 16-byte complex doubles and live in TWO PLANES
// of the slab (a at unit index i, b at 512 + i): every ds_read/write_b128 then sees exactly the float kernel's conflict-free index maps.
// (Round 3 kept (a, b) side by side as one 32-byte unit: the maps were tuned for 16-byte units, at 32 bytes two lanes of every 16-lane
// group fell on one bank quad -- 41 % of the kernel's LDS cycles were bank conflicts.)  Power spectrum, sparse mel and the dB epilogue are
// the float kernel's.  4 waves, 80 KB of LDS: two workgroups per CU.  ONLY_FLAGGED: redo only the clips the float kernel marked (auto mode).
// ------------------------------------------------------------------------------------------------
struct cd { double x, y; };
struct alignas(16) cd2 { cd a, b; };      // two complex doubles = one exchange unit
__device__ __forceinline__ cd operator+(cd a, cd b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cd operator-(cd a, cd b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cd cmul(cd a, cd b) { return {fma(a.x, b.x, -a.y * b.y), fma(a.x, b.y, a.y * b.x)}; }
__device__ __forceinline__ cd mul_neg_i(cd a) { return {a.y, -a.x}; }
__device__ __forceinline__ void dft4(cd& a0, cd& a1, cd& a2, cd& a3) {
    const cd t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = mul_neg_i(a1 - a3);
    a0 = t0 + t2; a2 = t0 - t2; a1 = t1 + t3; a3 = t1 - t3;
}
__device__ __forceinline__ void dft8(cd (&v)[8]) {
    constexpr double c = 0.70710678118654752440;
    cd e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    cd o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    dft4(e0, e1, e2, e3);
    dft4(o0, o1, o2, o3);
    o1 = {c * (o1.x + o1.y), c * (o1.y - o1.x)};
    o2 = mul_neg_i(o2);
    o3 = {c * (o3.y - o3.x), -c * (o3.x + o3.y)};
    v[0] = e0 + o0; v[4] = e0 - o0;
    v[1] = e1 + o1; v[5] = e1 - o1;
    v[2] = e2 + o2; v[6] = e2 - o2;
    v[3] = e3 + o3; v[7] = e3 - o3;
}
__device__ __forceinline__ void dft16(cd (&v)[16]) {
    cd e[8], o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { e[i] = v[2 * i]; o[i] = v[2 * i + 1]; }
    dft8(e);
    dft8(o);
    constexpr double c1 = 0.92387953251128675613, s1 = 0.38268343236508977173, c2 = 0.70710678118654752440;
    const cd w[8] = {{1., 0.}, {c1, -s1}, {c2, -c2}, {s1, -c1}, {0., -1.}, {-s1, -c1}, {-c2, -c2}, {-c1, -s1}};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const cd t = (k == 0) ? o[0] : (k == 4 ? mul_neg_i(o[4]) : cmul(o[k], w[k]));
        v[k] = e[k] + t;
        v[k + 8] = e[k] - t;
    }
}

constexpr int kSlab64 = 2 * 2048 + 8;                       // floats per wave: 1024 complex doubles (+ pad)
constexpr int k64OffMel = 4 * kSlab64;
constexpr int k64OffRed = k64OffMel + kMels * kMelStride;
constexpr int k64OffPinfo = k64OffRed + 16;
constexpr int k64OffFp0 = k64OffPinfo + kPieces;
constexpr int k64OffFcnt = k64OffFp0 + kMels;
constexpr int k64OffTw2 = (k64OffFcnt + kMels + 3) & ~3;     // [7][16] complex doubles
constexpr int k64LdsFloats = k64OffTw2 + 7 * 16 * 4;
static_assert(k64OffMel % 4 == 0 && k64OffTw2 % 4 == 0, "16-byte alignment of the double tables");
// Round 4: TWO workgroups per CU (two waves per SIMD; round 3 ran one 93 KB workgroup = one wave per SIMD, every LDS and L2 round trip
// exposed): the 8 KB pair-twiddle table is read through L1 like the window and the pass-1 twiddles, which brings a workgroup under 80 KB.
static_assert(sizeof(float) * k64LdsFloats + 512 <= 80 * 1024, "two logmel64 workgroups per CU");

template <bool RING, bool ONLY_FLAGGED>
__global__ __launch_bounds__(256, 2) void logmel64_kernel(const float* __restrict__ pcm, int64_t clip_stride, int clip_len, int n_clips,
                                                          int normalize, const int32_t* __restrict__ ring_pos_p, int ring_len,
                                                          const LogmelTables* __restrict__ tb, float* __restrict__ out) {
    constexpr int kWavesPerBlock = 4, kThreads = 256;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* mel = lds + k64OffMel;
    float* red = lds + k64OffRed;
    const int* pinfo = reinterpret_cast<const int*>(lds + k64OffPinfo);
    const int* fp0 = reinterpret_cast<const int*>(lds + k64OffFp0);
    const int* fcnt = reinterpret_cast<const int*>(lds + k64OffFcnt);
    const cd2* tw2_u = reinterpret_cast<const cd2*>(lds + k64OffTw2);      // [7][8] units = two twiddles each
    const cd* twp_c = reinterpret_cast<const cd*>(&tb->twp_d[0]);          // through L1 (8 KB, every frame)
    const float4* pw4 = reinterpret_cast<const float4*>(&tb->piece_w[0][0][0]);
    const cd2* tw1_u = reinterpret_cast<const cd2*>(&tb->tw1_d[0][0]);     // [7][64] units
    const double* win = &tb->window_d[0];

    const int tid = threadIdx.x, lane_id = tid & 63, lane = lane_id;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if constexpr (ONLY_FLAGGED) {
        // auto mode, usually nothing to do: look at the marks of all this workgroup's clips at once and leave before the
        // tables are staged (the launch then costs a few microseconds)
        int any = 0;
        for (int c = int(blockIdx.x) + tid * int(gridDim.x); c < n_clips; c += kThreads * int(gridDim.x))
            any |= __float_as_uint(__builtin_nontemporal_load(out + int64_t(c) * (kMels * kFrames))) == kRedoMark;
        if (!__syncthreads_or(any)) return;
    }
    float* slabf = lds + wave * kSlab64;                        // power spectrum / piece sums (floats) reuse the slab
    cd* slabc = reinterpret_cast<cd*>(slabf);                   // single complex values (the float kernel's float2 indices)
    cd* slabA = slabc;                                          // exchange planes: unit i of the float kernel's float4 map = (slabA[i], slabB[i])
    cd* slabB = slabc + 512;
    float* partial = slabf + kPartialInSlab;

    for (int i = tid; i < kPieces; i += kThreads) reinterpret_cast<int*>(lds)[k64OffPinfo + i] = tb->piece_info[i];
    if (tid < kMels) {
        reinterpret_cast<int*>(lds)[k64OffFp0 + tid] = tb->filt_p0[tid];
        reinterpret_cast<int*>(lds)[k64OffFcnt + tid] = tb->filt_cnt[tid];
    }
    for (int i = tid; i < 7 * 16 * 2; i += kThreads) reinterpret_cast<double*>(lds + k64OffTw2)[i] = (&tb->tw2_d[0][0].x)[i];
    const int ring_pos = RING ? *ring_pos_p : 0;
    __syncthreads();
    const int my_p0a = fp0[lane], my_cnta = fcnt[lane];
    const int my_p0b = lane + 64 < kMels ? fp0[lane + 64] : 0, my_cntb = lane + 64 < kMels ? fcnt[lane + 64] : 0;
    const unsigned clip_bytes = unsigned(clip_len) * 4u;

#pragma unroll 1
    for (int clip = blockIdx.x; clip < n_clips; clip += gridDim.x) {
        float* __restrict__ o = out + int64_t(clip) * (kMels * kFrames);
        uint32_t fmask = 0xffffffffu;                  // the frames this launch computes; all of them unless the float kernel said which
        if constexpr (ONLY_FLAGGED) {
            if (__builtin_amdgcn_readfirstlane(__float_as_uint(__builtin_nontemporal_load(o))) != kRedoMark) continue;   // uniform
            fmask = __builtin_amdgcn_readfirstlane(__float_as_uint(__builtin_nontemporal_load(o + 1)));
#ifdef WW_ABL_WHOLE_CLIP          // A/B: redo every frame of a marked clip (round 3's behaviour)
            fmask = 0xffffffffu;
#endif
            // the other frames' mel powers as the float kernel left them (already carrying the peak gain)
            for (int idx = tid; idx < kMels * kFrames; idx += kThreads)
                if (!((fmask >> (idx & 31)) & 1u)) mel[(idx >> 5) * kMelStride + (idx & 31)] = __builtin_nontemporal_load(o + idx);
        }
        const int n_mine = (__builtin_popcount(fmask) - wave + kWavesPerBlock - 1) / kWavesPerBlock;     // this wave takes the masked frames number wave, wave + 4, ...
        auto nth_frame = [&](int n) -> int {           // index of the n-th set bit of fmask (scalar); past the last one: a frame outside the clip (zero loads)
            uint32_t m = fmask;
            for (int i = 0; i < n && m; ++i) m &= m - 1u;
            return m ? __builtin_ctz(m) : 64;
        };
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pcm + int64_t(clip) * clip_stride), 0,
                                                                             clip_bytes, 0x00020000);
        // The reference normalises BEFORE the transform, in float32 (audio / np.max(np.abs(audio)), :73-76): every sample is
        // rounded once more, and that rounding is white noise ~177 dB under a tone's peak bin -- visible (2e-4 dB) in bands 85 dB
        // under the frame's energy.  So this kernel does the same: peak first (loads n1 = 4, 5 of the 32 frames tile the clip once),
        // then x / peak as a correctly rounded float division.
        float peak = 0.f;
        if (normalize) {
#pragma unroll 1
            for (int round = 0; round < kFrames / kWavesPerBlock; ++round) {
                float4 sn[8];
                load_frame<RING>(sn, rs, (round * kWavesPerBlock + wave) * kHop - kNfft / 2 + 4 * lane, ring_pos, ring_len);
                peak = absmax3(sn[4].z, sn[4].w, absmax3(sn[4].x, sn[4].y, peak));
                peak = absmax3(sn[5].z, sn[5].w, absmax3(sn[5].x, sn[5].y, peak));
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) peak = fmaxf(peak, __shfl_xor(peak, off));
            if (lane == 0) red[wave] = peak;
            __syncthreads();
            peak = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            __syncthreads();
        }
        // a frame's samples are fetched one frame ahead (in flight under the previous frame's passes 2 and 3)
        float4 sn[8];
        int frame = nth_frame(wave);
        load_frame<RING>(sn, rs, frame * kHop - kNfft / 2 + 4 * lane, ring_pos, ring_len);
#pragma unroll 1
        for (int round = 0; round < n_mine; ++round) {
            const int frame_next = nth_frame((round + 1) * kWavesPerBlock + wave);
            // opaque copy of the lane id (as in the float kernel): the swizzled LDS addresses below are functions of it, and hoisted out
            // of the frame loop they cost more VGPRs than the two-waves-per-SIMD budget of 256 has
            int lane = lane_id;
            asm volatile("" : "+v"(lane));
            const int k1r = lane >> 3, jr = lane & 7;
            cd za[8], zb[8];
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) {
                float4 s = sn[n1];
                if (normalize) { s.x = __fdiv_rn(s.x, peak); s.y = __fdiv_rn(s.y, peak); s.z = __fdiv_rn(s.z, peak); s.w = __fdiv_rn(s.w, peak); }
                const double* w = win + 4 * (64 * n1 + lane);
                za[n1] = {double(s.x) * w[0], double(s.y) * w[1]};
                zb[n1] = {double(s.z) * w[2], double(s.w) * w[3]};
            }
            // ---- pass 1 ----
            dft8(za);
            dft8(zb);
            slabA[lane] = za[0];
            slabB[lane] = zb[0];
#pragma unroll
            for (int k1 = 1; k1 < 8; ++k1) {
                const cd2 t = tw1_u[(k1 - 1) * 64 + lane];
                const int at = k1 * 64 + (lane ^ (8 * ((k1 >> 1) & 1)));
                slabA[at] = cmul(za[k1], t.a);
                slabB[at] = cmul(zb[k1], t.b);
            }
            lds_order();
            // next frame's samples (the last round fetches past the clip: the descriptor returns zeros, nobody reads them)
            load_frame<RING>(sn, rs, frame_next * kHop - kNfft / 2 + 4 * lane, ring_pos, ring_len);
            // ---- pass 2 ----
            {
                const int s8 = 8 * ((k1r >> 1) & 1);
                const cd* x1e = slabA + k1r * 64 + jr + s8;
                const cd* x1o = slabA + k1r * 64 + jr - s8;
#pragma unroll
                for (int n2 = 0; n2 < 8; ++n2) {
                    const cd* src = (n2 & 1) ? x1o + n2 * 8 : x1e + n2 * 8;
                    za[n2] = src[0];
                    zb[n2] = src[512];
                }
                lds_order();
                dft8(za);
                dft8(zb);
                cd* x2w[4];
#pragma unroll
                for (int hk = 0; hk < 4; ++hk) x2w[hk] = slabA + 64 * k1r + (jr ^ (4 * (k1r & 1) + hk));
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) {
                    cd a = za[k2], b = zb[k2];
                    if (k2 > 0) {
                        const cd2 t = tw2_u[(k2 - 1) * 8 + jr];
                        a = cmul(a, t.a);
                        b = cmul(b, t.b);
                    }
                    x2w[k2 >> 1][8 * k2] = a;
                    x2w[k2 >> 1][8 * k2 + 512] = b;
                }
            }
            lds_order();
            // ---- pass 3 ----
            {
                cd u[16];
                const int sw2 = (lane >> 1) & 7;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const cd* src = slabA + lane * 8 + (m ^ sw2);
                    u[2 * m] = src[0];
                    u[2 * m + 1] = src[512];
                }
                lds_order();
#ifndef WW_K1_ABL_NODFT16              // timing-only ablation: what pass 3's butterflies cost the vector ALU (results are garbage)
                dft16(u);
#endif
                const int lp = (lane >> 3) + 8 * (lane & 7);
                cd* zw = slabc + (lp ^ ((lp >> 3) & 7));      // Z[k] at k ^ ((k >> 3) & 7): 16-byte elements, conflict-free stores and `a` reads
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) zw[64 * kk] = u[kk];
            }
            lds_order();
            // ---- real-input split (float64), rounded to complex64 like the reference's spectrum, power in float32 ----
            {
                cd a[8], b[8];
                const int lowb = (64 - lane) & 63;
                const cd* za_p = slabc + (lane ^ ((lane >> 3) & 7));
                const cd* zb_p = slabc + (lowb ^ ((lowb >> 3) & 7)) + (lane == 0 ? 64 : 0);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    a[j] = za_p[64 * j];
                    b[j] = zb_p[64 * (15 - j)];
                }
                if (lane == 0) { a[0] = slabc[512]; b[0] = a[0]; }
                lds_order();
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = (j == 0 && lane == 0) ? 512 : lane + 64 * j;
                    const cd tw = twp_c[lane + 64 * j];
                    const cd e = {0.5 * (a[j].x + b[j].x), 0.5 * (a[j].y - b[j].y)};
                    const cd od = {0.5 * (a[j].y + b[j].y), 0.5 * (b[j].x - a[j].x)};
                    const cd t = cmul(od, tw);
                    const float px = float(e.x + t.x), py = float(e.y + t.y), mx = float(e.x - t.x), my = float(e.y - t.y);
                    slabf[k] = fmaf(px, px, py * py);
                    slabf[1024 - k] = fmaf(mx, mx, my * my);
                }
                if (lane == 0) slabf[0] = 0.f;
            }
            lds_order();
            // ---- sparse mel (as the float kernel) ----
            {
                int info[kPieceRounds];
                float4 s0[kPieceRounds], s1[kPieceRounds], pw0[kPieceRounds], pw1[kPieceRounds];
#pragma unroll
                for (int c = 0; c < kPieceRounds; ++c) {
                    info[c] = pinfo[c * kPieceSlots + lane];
                    pw0[c] = pw4[c * kPieceSlots + lane];
                    pw1[c] = pw4[kPieces + c * kPieceSlots + lane];
                }
#pragma unroll
                for (int c = 0; c < kPieceRounds; ++c) {
                    const float4* s4 = reinterpret_cast<const float4*>(slabf + (info[c] & 0xffff));
                    s0[c] = s4[0];
                    s1[c] = s4[1];
                }
#pragma unroll
                for (int c = 0; c < kPieceRounds; ++c) {
                    const float4 w0 = pw0[c], w1 = pw1[c];
                    float acc = w0.x * s0[c].x;
                    acc = fmaf(w0.y, s0[c].y, acc);
                    acc = fmaf(w0.z, s0[c].z, acc);
                    acc = fmaf(w0.w, s0[c].w, acc);
                    acc = fmaf(w1.x, s1[c].x, acc);
                    acc = fmaf(w1.y, s1[c].y, acc);
                    acc = fmaf(w1.z, s1[c].z, acc);
                    acc = fmaf(w1.w, s1[c].w, acc);
                    partial[info[c] >> 16] = acc;
                }
                lds_order();
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    const int f = lane + 64 * pass;
                    const int p0 = pass ? my_p0b : my_p0a, cnt = pass ? my_cntb : my_cnta;
                    if (f < kMels) {
                        float acc = 0.f;
                        for (int q = 0; q < cnt; ++q) acc += partial[p0 + q];
                        mel[f * kMelStride + frame] = acc;
                    }
                }
            }
            lds_order();
            frame = frame_next;
        }
        __syncthreads();
        // a NaN anywhere (silent clip, 0/0) must reach every output like in the reference: fmaxf would drop it
        float mmax = 0.f;
        bool any_nan = false;
        for (int idx = tid; idx < kMels * kFrames; idx += kThreads) {
            const float p = mel[(idx >> 5) * kMelStride + (idx & 31)];
            any_nan |= p != p;
            mmax = fmaxf(mmax, p);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mmax = fmaxf(mmax, __shfl_xor(mmax, off));
        any_nan = __builtin_amdgcn_ballot_w64(any_nan) != 0ull;
        if (lane == 0) red[wave] = any_nan ? __uint_as_float(0x7fc00000u) : mmax;
        __syncthreads();
        mmax = red[0];
#pragma unroll
        for (int w = 1; w < kWavesPerBlock; ++w) mmax = (red[w] != red[w] || mmax != mmax) ? __uint_as_float(0x7fc00000u) : fmaxf(mmax, red[w]);
        const float amin = 1e-10f;
        float ref = mmax;
        ref = ref < amin ? amin : ref;
        const float ref_db = db10(ref);
        for (int idx = tid; idx < kMels * kFrames; idx += kThreads) {
            float v = mel[(idx >> 5) * kMelStride + (idx & 31)];
            v = v < amin ? amin : v;
            float db = db10(v) - ref_db;
            db = db < -80.0f ? -80.0f : db;
            o[idx] = db;
        }
        __syncthreads();
    }
}

template <int WAVES>
static int launch_logmel_w(const float* pcm, int64_t n_clips, int64_t clip_stride, int64_t clip_len, int normalize,
                           const int32_t* ring_pos, int64_t ring_len, float* logmel, const LogmelTables* tb, int mark, hipStream_t stream) {
    using L = K1Layout<WAVES>;
    const int64_t resident = int64_t(device_cu_count()) * L::kBlocksPerCu;   // what LDS and VGPRs admit per CU
    const int grid = int(n_clips < resident ? n_clips : resident);
    const size_t lds_bytes = sizeof(float) * L::kLdsFloats;
    if (ring_pos)
        hipLaunchKernelGGL((logmel_kernel<true, WAVES>), dim3(grid), dim3(L::kThreads), lds_bytes, stream, pcm, clip_stride, int(clip_len),
                           int(n_clips), normalize, ring_pos, int(ring_len), tb, logmel, mark);
    else
        hipLaunchKernelGGL((logmel_kernel<false, WAVES>), dim3(grid), dim3(L::kThreads), lds_bytes, stream, pcm, clip_stride, int(clip_len),
                           int(n_clips), normalize, ring_pos, int(ring_len), tb, logmel, mark);
    WW_HIP(hipGetLastError());
    return WW_OK;
}

template <bool ONLY_FLAGGED>
static int launch_logmel64(const float* pcm, int64_t n_clips, int64_t clip_stride, int64_t clip_len, int normalize,
                           const int32_t* ring_pos, int64_t ring_len, float* logmel, const LogmelTables* tb, hipStream_t stream) {
    const int64_t resident = 2 * int64_t(device_cu_count());      // two 80 KB workgroups per CU
    const int grid = int(n_clips < resident ? n_clips : resident);
    const size_t lds_bytes = sizeof(float) * k64LdsFloats;
    if (ring_pos)
        hipLaunchKernelGGL((logmel64_kernel<true, ONLY_FLAGGED>), dim3(grid), dim3(256), lds_bytes, stream, pcm, clip_stride, int(clip_len),
                           int(n_clips), normalize, ring_pos, int(ring_len), tb, logmel);
    else
        hipLaunchKernelGGL((logmel64_kernel<false, ONLY_FLAGGED>), dim3(grid), dim3(256), lds_bytes, stream, pcm, clip_stride, int(clip_len),
                           int(n_clips), normalize, ring_pos, int(ring_len), tb, logmel);
    WW_HIP(hipGetLastError());
    return WW_OK;
}

// > 64 KiB of dynamic LDS needs an opt-in per kernel, once per device
static int logmel_opt_in_lds() {
    static std::mutex mu;
    static bool done[64] = {};
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    WW_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(WW_EINVAL, "device ordinal out of range");
    if (done[dev]) return WW_OK;
    const int b4 = int(sizeof(float) * K1Layout<4>::kLdsFloats), b8 = int(sizeof(float) * K1Layout<8>::kLdsFloats);
    const int b64 = int(sizeof(float) * k64LdsFloats);
    WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(logmel_kernel<true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, b4));
    WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(logmel_kernel<false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, b4));
    WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(logmel_kernel<true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, b8));
    WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(logmel_kernel<false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, b8));
    WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(logmel64_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, b64));
    WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(logmel64_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, b64));
    WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(logmel64_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, b64));
    WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(logmel64_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, b64));
    done[dev] = true;
    return WW_OK;
}

int launch_logmel(const float* pcm, int64_t n_clips, int64_t clip_stride, int64_t clip_len, int normalize,
                  const int32_t* ring_pos, int64_t ring_len, float* logmel, hipStream_t stream) {
    if (n_clips == 0) return WW_OK;
    const LogmelTables* tb = device_tables();
    if (!tb) return WW_EHIP;
    if (int rc = logmel_opt_in_lds()) return rc;
    const int mode = logmel_math_mode();
    if (mode == WW_LOGMEL_MATH_F64)
        return launch_logmel64<false>(pcm, n_clips, clip_stride, clip_len, normalize, ring_pos, ring_len, logmel, tb, stream);
    const int mark = mode == WW_LOGMEL_MATH_AUTO;
    // at most one clip per CU (streaming, small batches): the 8-wave latency form; otherwise the 4-wave throughput form
    int rc;
    if (n_clips <= device_cu_count())
        rc = launch_logmel_w<8>(pcm, n_clips, clip_stride, clip_len, normalize, ring_pos, ring_len, logmel, tb, mark, stream);
    else
        rc = launch_logmel_w<4>(pcm, n_clips, clip_stride, clip_len, normalize, ring_pos, ring_len, logmel, tb, mark, stream);
    if (rc != WW_OK || !mark) return rc;
    return launch_logmel64<true>(pcm, n_clips, clip_stride, clip_len, normalize, ring_pos, ring_len, logmel, tb, stream);
}

}  // namespace ww
