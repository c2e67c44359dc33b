// KA: AudioProcessor.augment_audio on the GPU (training-side, feeds K1).
//
// Replaces /root/reference/wakeword_training_script.py:103-123, one plan (the caller's random draws) per clip:
//     np.roll -> librosa.effects.pitch_shift -> librosa.effects.time_stretch + pad_or_truncate -> + N(0, sigma)
// librosa.effects.time_stretch = istft(phase_vocoder(stft(y), rate), length=round(len/rate)), n_fft 2048 / hop 512 / Hann;
// pitch_shift = resample(time_stretch(y, 2^(-n/12)), ratio) cut or zero-padded to the input length.
//
//   roll_kernel       out[i] = in[(i - shift) mod L]                        (also the copy into the work buffer)
//   stft_pv_kernel    one workgroup per clip: four frames per round (one per wave: window, 1024-point complex FFT of the packed
//                     real frame (ww_fft.h), real-input split in place) into a ring of eight LDS slabs, then the phase vocoder
//                     over the output steps whose two columns are there: thread = bin; librosa's arithmetic types are kept
//                     (float32 magnitudes and phase accumulator, float64 phase advance) so that the accumulator rounds the
//                     same way; angle / magnitude / phasor by short in-kernel forms instead of libm's atan2f / hypotf / sincosf.
//                     (stft_kernel + pv_kernel, the two-launch form with the columns in HBM, are kept for the
//                     WW_AUG_SPLIT_STFT_PV timing build: same bits.)
//   istft_kernel      Hermitian spectrum -> conj(Z) -> the same forward FFT -> frame, four frames per round into a ring
//                     of eight LDS slabs; the hop segments a round completes are summed straight from the slabs in
//                     frame order with the window and its sum-square, centre-trimmed, cropped / zero-padded
//   resample_kernel   windowed-sinc interpolation at t = i / ratio (resampy 'kaiser_best' table in LDS, float32 MACs): the stand-in
//                     for librosa's soxr_hq (absent third-party library; parity unpinned)
//   noise_kernel      + sigma * normal(seed, i), the build's counter-based generator (splitmix64 -> Box-Muller)
// Clips whose plan switches a transform off skip its kernels (their blocks copy the data through).
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

#include "ww_fft.h"

namespace ww {

constexpr int kAugFrames = kFrames;                 // STFT frames of a 16000-sample clip (1 + 16000/512 = 32)
constexpr int kAugMaxOut = 46;                      // phase-vocoder output steps: ceil(32 / rate), rate >= 32/46
constexpr int kAugYStride = 25600;                  // stretched-clip scratch row
constexpr int kSpec = kBins;                        // 1025 complex bins per spectrum row
#ifdef WW_AUG_SPLIT_STFT_PV                         // timing-only build: stft_kernel and pv_kernel as two launches with the columns in HBM
constexpr int kAugDFrames = kAugFrames;
#else
constexpr int kAugDFrames = 0;                      // the STFT columns live in LDS (stft_pv_kernel): no spectrum buffer in the workspace
#endif

struct AugDev {            // one per clip, derived on the host from ww_augment_plan
    int32_t shift;         // np.roll shift reduced to [0, L)
    int32_t crop;          // crop start after time_stretch
    int32_t p_out, p_len, p_res;   // pitch: PV steps, istft length, resampled length (0 = pitch off)
    int32_t s_out, s_len;          // stretch: PV steps, istft length (0 = stretch off)
    uint32_t seed;
    double p_rate, p_ratio, s_rate;
    float sigma;
    float pad_;
};

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void roll_kernel(const float* __restrict__ in, int64_t stride, const AugDev* __restrict__ plan,
                                                   float* __restrict__ out) {
    const int clip = blockIdx.y;
    const int i = 4 * (blockIdx.x * 256 + threadIdx.x);          // four outputs per thread: one 16-byte store (kClip % 4 == 0)
    if (i >= kClip) return;
    const float* __restrict__ x = in + int64_t(clip) * stride;
    int src = i - plan[clip].shift;
    src += src < 0 ? kClip : 0;
    float4 v;
    if (src + 3 < kClip) { v.x = x[src]; v.y = x[src + 1]; v.z = x[src + 2]; v.w = x[src + 3]; }
    else {                                                       // the wrap falls inside this group of four
        v.x = x[src]; v.y = x[src + 1 < kClip ? src + 1 : src + 1 - kClip];
        v.z = x[src + 2 < kClip ? src + 2 : src + 2 - kClip]; v.w = x[src + 3 - kClip];
    }
    *reinterpret_cast<float4*>(out + int64_t(clip) * kClip + i) = v;
}

#ifdef WW_AUG_SPLIT_STFT_PV     // timing-only build: the two-launch form (stft_kernel, pv_kernel below) with the columns in HBM
// ------------------------------------------------------------------------------------------------
// which = 0: pitch stage, 1: stretch stage (selects the on/off flag)
__global__ __launch_bounds__(256) void stft_kernel(const float* __restrict__ x, const AugDev* __restrict__ plan, int which,
                                                   const LogmelTables* __restrict__ tb, float2* __restrict__ D) {
    __shared__ __attribute__((aligned(16))) float slabs[4 * fft::kSlabFloats];
    const int clip = blockIdx.x;
    if ((which == 0 ? plan[clip].p_out : plan[clip].s_out) == 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* slab = slabs + wave * fft::kSlabFloats;
    const float2* slab2 = reinterpret_cast<const float2*>(slab);
    const float* xc = x + int64_t(clip) * kClip;
    const float4* win4 = reinterpret_cast<const float4*>(&tb->window[0]);
    for (int frame = wave; frame < kAugFrames; frame += 4) {
        float2 za[8], zb[8];
        const int base = frame * kHop - kNfft / 2 + 4 * lane;
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) {
            const int idx = base + 256 * n1;                     // multiple of 4: the float4 is all inside or all outside
            const float4 s = (idx >= 0 && idx < kClip) ? *reinterpret_cast<const float4*>(xc + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 w = win4[64 * n1 + lane];
            za[n1] = make_float2(s.x * w.x, s.y * w.y);
            zb[n1] = make_float2(s.z * w.z, s.w * w.w);
        }
        fft::wave_fft1024(za, zb, slab, tb, lane);
        // real-input split: X[k] = E + W^k O, X[1024-k] = conj(E - W^k O), E = (Z[k] + conj Z[1024-k])/2, O = (Z[k] - conj Z[1024-k])/(2i)
        float2* Dr = D + (int64_t(clip) * kAugFrames + frame) * kSpec;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = lane + 64 * j;                         // 0..511
            const float2 a = slab2[fft::zpos(k)], b = slab2[fft::zpos((1024 - k) & 1023)];
            const float2 tw = tb->twr[k];
            const float2 e = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
            const float2 o = make_float2(0.5f * (a.y + b.y), 0.5f * (b.x - a.x));
            const float2 t = fft::cmul(o, tw);
            Dr[k] = make_float2(e.x + t.x, e.y + t.y);
            Dr[1024 - k] = make_float2(e.x - t.x, -(e.y - t.y));
        }
        if (lane == 0) { const float2 z = slab2[fft::zpos(512)]; Dr[512] = make_float2(z.x, -z.y); }
        fft::lds_order();                                        // the slab is rewritten by the next frame
    }
}

#endif

#ifdef WW_AUG_SPLIT_STFT_PV
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float2 pv_col(const float2* __restrict__ Dc, int f, int k) {
    return f < kAugFrames ? Dc[int64_t(f) * kSpec + k] : make_float2(0.f, 0.f);     // librosa pads two zero columns
}
#endif

// atan2 for the vocoder: a = min/max in [0, 1], atan(a) = a P(a^2) (degree 8 in a^2, |err| <= 1.2e-7 in float32 evaluation -- the size of
// libm's own last-place error at these magnitudes), octant and quadrant folded back; signed zeros and (0, 0) as atan2f has them.
__device__ __forceinline__ float pv_atan2(float y, float x) {
#ifdef WW_ABL_PV_LIBM_ATAN
    return atan2f(y, x);
#endif
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
#ifndef WW_ABL_PV_PRECISE_DIV
    // min / max by v_rcp_f32 (1 ulp; the polynomial's own error is of that size).  The hardware reciprocal takes denormals for zero, so
    // tiny pairs are scaled up first (exact, the ratio is unchanged)
    const float sc = mx < 5.4210109e-20f ? 1.8446744e19f : 1.0f;                      // 2^-64, 2^64
    mx *= sc; mn *= sc;
    const float a = mx > 0.f ? mn * __builtin_amdgcn_rcpf(mx) : 0.f;
#else
    const float a = mx > 0.f ? mn / mx : 0.f;
#endif
    const float q = a * a;
    float p = 2.399661113e-03f;
    p = fmaf(p, q, -1.415910292e-02f);
    p = fmaf(p, q, 3.935785964e-02f);
    p = fmaf(p, q, -7.195615768e-02f);
    p = fmaf(p, q, 1.047824398e-01f);
    p = fmaf(p, q, -1.415504068e-01f);
    p = fmaf(p, q, 1.998492777e-01f);
    p = fmaf(p, q, -3.333252668e-01f);
    p = fmaf(p, q, 9.999998808e-01f);
    float r = a * p;
    r = ay > ax ? 1.57079632679489662f - r : r;
    r = __builtin_signbitf(x) ? 3.14159265358979324f - r : r;
    return __builtin_copysignf(r, y);
}

// cos / sin of the float32 phase accumulator.  The accumulator reaches 10^4 .. 10^5 radians (phi = pi k / 2 per step): libm's sincosf takes
// its large-argument reduction path there (round 2: the bulk of this kernel's time).  Here the turn count acc / 2 pi is formed in double, its
// fraction goes to the hardware's v_sin_f32 / v_cos_f32 (arguments in revolutions; absolute error ~1e-6, five hundred times under the tests' bound
// on the output samples).
// |c| by v_sqrt_f32 (1 ulp) instead of the correctly rounded expansion
__device__ __forceinline__ float pv_abs(float2 c) {
#ifndef WW_ABL_PV_PRECISE_DIV
    return __builtin_amdgcn_sqrtf(fmaf(c.x, c.x, c.y * c.y));
#else
    return sqrtf(fmaf(c.x, c.x, c.y * c.y));
#endif
}

// round(dphase / 2 pi) with the quotient formed by the reciprocal: differs from the division only when the quotient is within an ulp of a
// half-integer, where either neighbour wraps the phase to the same angle (the accumulator then differs by its own rounding of +-2 pi)
__device__ __forceinline__ double pv_wrap(double dphase) {
    const double two_pi = 6.283185307179586476925286766559;
#ifndef WW_ABL_PV_PRECISE_DIV
    return dphase - two_pi * __builtin_rint(dphase * 0.15915494309189533576888);
#else
    return dphase - two_pi * __builtin_rint(dphase / two_pi);
#endif
}

__device__ __forceinline__ void pv_sincos(float acc, float& sn, float& cs) {
#ifdef WW_ABL_PV_LIBM_SINCOS
    sincosf(acc, &sn, &cs);
    return;
#endif
    const double turns = double(acc) * 0.15915494309189533576888;       // 1 / (2 pi)
    const float fr = float(turns - __builtin_rint(turns));                // [-0.5, 0.5]
    sn = __builtin_amdgcn_sinf(fr);
    cs = __builtin_amdgcn_cosf(fr);
}

#ifdef WW_AUG_SPLIT_STFT_PV
__global__ __launch_bounds__(256) void pv_kernel(const float2* __restrict__ D, const AugDev* __restrict__ plan, int which,
                                                 float2* __restrict__ S) {
#pragma clang fp contract(off)
    const int clip = blockIdx.x;
    const int n_out = which == 0 ? plan[clip].p_out : plan[clip].s_out;
    if (n_out == 0) return;
    const double rate = which == 0 ? plan[clip].p_rate : plan[clip].s_rate;
    const float2* Dc = D + int64_t(clip) * kAugFrames * kSpec;
    float2* Sc = S + int64_t(clip) * kAugMaxOut * kSpec;
    const double two_pi = 6.283185307179586476925286766559;
    for (int k = threadIdx.x; k < kSpec; k += 256) {
        const double phi = double(k) * two_pi * 0.25;            // hop * 2 pi k / n_fft
        const float2 d0 = Dc[k];
        float acc = pv_atan2(d0.y, d0.x);                        // np.angle(D[:, 0]): float32 accumulator
        // |c| and angle(c) of the two columns a step reads are kept from the previous step: the column index moves on by 0, 1 or 2 per
        // step (uniform over the workgroup), so most steps compute one new column instead of two (same values, a third less work)
        int have = -2;                                           // columns `have`, `have + 1` are in (m0, a0), (m1, a1)
        float m0 = 0.f, a0 = 0.f, m1 = 0.f, a1 = 0.f;
        for (int t = 0; t < n_out; ++t) {
            const double step = double(t) * rate;                // np.arange(0, n, rate)[t]
            const int i0 = int(step);
            const double alpha = step - double(i0);
            if (i0 != have) {
                if (i0 == have + 1) { m0 = m1; a0 = a1; }
                else { const float2 c0 = pv_col(Dc, i0, k); m0 = pv_abs(c0); a0 = pv_atan2(c0.y, c0.x); }
                const float2 c1 = pv_col(Dc, i0 + 1, k);
                m1 = pv_abs(c1);                                 // |c|: STFT magnitudes of unit-peak clips stay far inside float range
                a1 = pv_atan2(c1.y, c1.x);
                have = i0;
            }
            const float mag = float(1.0 - alpha) * m0 + float(alpha) * m1;
            float sn, cs;
            pv_sincos(acc, sn, cs);
            Sc[int64_t(t) * kSpec + k] = make_float2(cs * mag, sn * mag);
            const float da = a1 - a0;
            double dphase = double(da) - phi;
            dphase = pv_wrap(dphase);
            acc = float(double(acc) + (phi + dphase));
        }
    }
}
#endif

// ------------------------------------------------------------------------------------------------
// stft_kernel + pv_kernel in one pass over the clip: the STFT columns never leave the CU.  Four frames per round (one per wave) are
// transformed into a ring of eight LDS slabs and split IN PLACE into their spectrum column (a lane reads Z[k], Z[1024-k] and writes
// D[k], D[1024-k] back to the same two slots; D[1024] takes the slab's spare slot), then every thread advances its bins' vocoder state
// over the output steps whose two columns are there.  A step reads columns i0, i0 + 1 with i0 non-decreasing, so round r + 1 may
// overwrite the columns of round r - 1 (two barriers per round).  Same arithmetic as the two kernels, bit-identical S.
constexpr int kStftPvLds = 8 * fft::kSlabFloats * int(sizeof(float));                    // 65,664 B: two workgroups per CU

__global__ __launch_bounds__(256) void stft_pv_kernel(const float* __restrict__ x, const AugDev* __restrict__ plan, int which,
                                                      const LogmelTables* __restrict__ tb, float2* __restrict__ S) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int clip = blockIdx.x, tid = threadIdx.x;
    const int n_out = which == 0 ? plan[clip].p_out : plan[clip].s_out;
    if (n_out == 0) return;
    const double rate = which == 0 ? plan[clip].p_rate : plan[clip].s_rate;
    const int lane = tid & 63, wave = tid >> 6;
    const float* xc = x + int64_t(clip) * kClip;
    const float4* win4 = reinterpret_cast<const float4*>(&tb->window[0]);
    float2* Sc = S + int64_t(clip) * kAugMaxOut * kSpec;
    constexpr int kB = 5;                                        // bins tid + 256 b; b = 4 is bin 1024 (thread 0 only)
    float acc[kB], m0[kB], a0[kB], m1[kB], a1[kB];
#pragma unroll
    for (int b = 0; b < kB; ++b) { acc[b] = 0.f; m0[b] = 0.f; a0[b] = 0.f; m1[b] = 0.f; a1[b] = 0.f; }
    int have = -2, t = 0;
    for (int r = 0; r < kAugFrames / 4; ++r) {
        {
            const int frame = 4 * r + wave;
            float* slab = lds + (frame & 7) * fft::kSlabFloats;
            float2* slab2 = reinterpret_cast<float2*>(slab);
            float2 za[8], zb[8];
            const int base = frame * kHop - kNfft / 2 + 4 * lane;
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) {
                const int idx = base + 256 * n1;
                const float4 s = (idx >= 0 && idx < kClip) ? *reinterpret_cast<const float4*>(xc + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 w = win4[64 * n1 + lane];
                za[n1] = make_float2(s.x * w.x, s.y * w.y);
                zb[n1] = make_float2(s.z * w.z, s.w * w.w);
            }
            fft::wave_fft1024(za, zb, slab, tb, lane);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = lane + 64 * j;                     // 0..511
                const int pa = fft::zpos(k), pb = fft::zpos((1024 - k) & 1023);
                const float2 a = slab2[pa], b = slab2[pb];
                const float2 tw = tb->twr[k];
                const float2 e = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
                const float2 o = make_float2(0.5f * (a.y + b.y), 0.5f * (b.x - a.x));
                const float2 tt = fft::cmul(o, tw);
                slab2[pa] = make_float2(e.x + tt.x, e.y + tt.y);
                slab2[k == 0 ? 1024 : pb] = make_float2(e.x - tt.x, -(e.y - tt.y));
            }
            if (lane == 0) { const int pz = fft::zpos(512); const float2 z = slab2[pz]; slab2[pz] = make_float2(z.x, -z.y); }
        }
        __syncthreads();                                         // columns <= 4r + 3 are in their slabs
        {
#pragma clang fp contract(off)
            const double two_pi = 6.283185307179586476925286766559;
            if (r == 0) {
                const float2* c = reinterpret_cast<const float2*>(lds);
#pragma unroll
                for (int b = 0; b < kB; ++b) {
                    const int k = tid + 256 * b;
                    if (k < kSpec) { const float2 d0 = c[fft::zpos(k)]; acc[b] = pv_atan2(d0.y, d0.x); }
                }
            }
            const int last = 4 * r + 3;
            for (; t < n_out; ++t) {
                const double step = double(t) * rate;
                const int i0 = int(step);
                if ((i0 + 1 < kAugFrames ? i0 + 1 : kAugFrames - 1) > last) break;      // this step's columns come with a later round
                const double alpha = step - double(i0);
                const bool fresh0 = i0 != have && i0 != have + 1, fresh1 = i0 != have;
                const float2* c0 = reinterpret_cast<const float2*>(lds + (i0 & 7) * fft::kSlabFloats);
                const float2* c1 = reinterpret_cast<const float2*>(lds + ((i0 + 1) & 7) * fft::kSlabFloats);
#pragma unroll
                for (int b = 0; b < kB; ++b) {
                    const int k = tid + 256 * b;
                    if (k < kSpec) {
                        const double phi = double(k) * two_pi * 0.25;
                        if (fresh1) {
                            if (!fresh0) { m0[b] = m1[b]; a0[b] = a1[b]; }
                            else {
                                const float2 v = i0 < kAugFrames ? c0[fft::zpos(k)] : make_float2(0.f, 0.f);
                                m0[b] = pv_abs(v);
                                a0[b] = pv_atan2(v.y, v.x);
                            }
                            const float2 v = i0 + 1 < kAugFrames ? c1[fft::zpos(k)] : make_float2(0.f, 0.f);
                            m1[b] = pv_abs(v);
                            a1[b] = pv_atan2(v.y, v.x);
                        }
                        const float mag = float(1.0 - alpha) * m0[b] + float(alpha) * m1[b];
                        float sn, cs;
                        pv_sincos(acc[b], sn, cs);
                        Sc[int64_t(t) * kSpec + k] = make_float2(cs * mag, sn * mag);
                        const float da = a1[b] - a0[b];
                        double dphase = double(da) - phi;
                        dphase = pv_wrap(dphase);
                        acc[b] = float(double(acc[b]) + (phi + dphase));
                    }
                }
                have = i0;
            }
        }
        __syncthreads();                                         // the next round overwrites the columns of round r - 1
    }
}

// ------------------------------------------------------------------------------------------------
constexpr int kIstftSlabs = 8;                                   // two rounds of four frames stay resident
constexpr int kIstftLds = kIstftSlabs * fft::kSlabFloats * int(sizeof(float));          // 65,664 B: two workgroups per CU

// dst[i], i < dst_len, = y[i + crop] (0 past the stretched length); clips with the stage off copy `passthru` instead.
// Four frames per round (one per wave) into a ring of eight slabs.  A sample of the padded signal in hop segment s is
// covered by frames s-3..s, so after round r the segments 4r..4r+3 are complete: they are summed straight from the
// slabs in frame order (librosa's order), normalised and stored -- no overlap-add buffer, two barriers per round.
__global__ __launch_bounds__(256) void istft_kernel(const float2* __restrict__ S, const AugDev* __restrict__ plan, int which,
                                                    const LogmelTables* __restrict__ tb, const float* __restrict__ passthru,
                                                    float* __restrict__ dst, int64_t dst_stride) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int clip = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int n_out = which == 0 ? plan[clip].p_out : plan[clip].s_out;
    float* out = dst + int64_t(clip) * dst_stride;
    if (n_out == 0) {
        if (passthru)
            for (int i = tid; i < kClip; i += 256) out[i] = passthru[int64_t(clip) * kClip + i];
        return;
    }
    const int length = which == 0 ? plan[clip].p_len : plan[clip].s_len;
    const int crop = which == 0 ? 0 : plan[clip].crop;
    const int dst_len = which == 0 ? length : kClip;
    int n_frames = (length + kNfft + kHop - 1) / kHop;           // ceil((length + n_fft) / hop)
    n_frames = n_frames < n_out ? n_frames : n_out;
    const float2* Sc = S + int64_t(clip) * kAugMaxOut * kSpec;
    const int rounds = (n_frames + 3) / 4 + 1;                   // + a flush round for the tails of the last frames
    for (int r = 0; r < rounds; ++r) {
        const int frame = 4 * r + wave;
        if (frame < n_frames) {
            float* slab = lds + (frame % kIstftSlabs) * fft::kSlabFloats;
            const float2* X = Sc + int64_t(frame) * kSpec;
            float2 za[8], zb[8];
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int m = 128 * n1 + 2 * lane + q;       // 0..1023
                    float2 xm = X[m], xp = X[1024 - m];
                    if (m == 0) { xm.y = 0.f; xp.y = 0.f; }      // irfft ignores the imaginary parts of DC and Nyquist
                    // E = (X[m] + conj X[M-m])/2, O = (X[m] - conj X[M-m])/2 * conj(W^m), Z = E + iO; FFT input conj(Z)
                    const float2 e = make_float2(0.5f * (xm.x + xp.x), 0.5f * (xm.y - xp.y));
                    const float2 d = make_float2(0.5f * (xm.x - xp.x), 0.5f * (xm.y + xp.y));
                    const float2 w = tb->twr[m];
                    const float2 o = fft::cmul(d, make_float2(w.x, -w.y));
                    const float2 zc = make_float2(e.x - o.y, -(e.y + o.x));
                    if (q == 0) za[n1] = zc; else zb[n1] = zc;
                }
            fft::wave_fft1024(za, zb, slab, tb, lane);          // F = FFT(conj Z); z[n] = conj(F[n]) / 1024
        }
        __syncthreads();                                         // frames <= 4r + 3 are in their slabs
        // segments 4r .. 4r+3 of the padded signal: 1024 sample PAIRS (2p, 2p+1), four per thread
        for (int pp = tid; pp < 4 * kHop / 2; pp += 256) {
            const int n = 4 * r * kHop + 2 * pp;                 // even sample of the pair
            const int seg = n / kHop;
            int f0 = seg - 3;
            f0 = f0 < 0 ? 0 : f0;
            const int f1 = seg < n_frames - 1 ? seg : n_frames - 1;
            float vx = 0.f, vy = 0.f, wsx = 0.f, wsy = 0.f;      // window sum-square in float32, frame order, as librosa
            for (int f = f0; f <= f1; ++f) {
                const int j = n - f * kHop;                      // even, 0..2046
                const float2 F = reinterpret_cast<const float2*>(lds + (f % kIstftSlabs) * fft::kSlabFloats)[fft::zpos(j >> 1)];
                const float2 w = *reinterpret_cast<const float2*>(&tb->window[j]);
                vx += w.x * (F.x * (1.0f / 1024.0f));
                vy += w.y * (-F.y * (1.0f / 1024.0f));
                wsx += w.x * w.x;
                wsy += w.y * w.y;
            }
            if (wsx > 1.17549435e-38f) vx = vx / wsx;
            if (wsy > 1.17549435e-38f) vy = vy / wsy;
            const int src = n - kNfft / 2;                       // centre trim; n even, so src and src + 1 share their fate below
            const int i = src - crop;
            if (i >= 0 && i < dst_len) out[i] = src < length ? vx : 0.f;
            if (i + 1 >= 0 && i + 1 < dst_len) out[i + 1] = src + 1 < length ? vy : 0.f;
        }
        __syncthreads();                                         // the next round overwrites the slabs of round r - 1
    }
    // samples past the reach of the last frame (only when the spectrogram is shorter than `length` asks for)
    for (int i = 4 * rounds * kHop - kNfft / 2 - crop + tid; i < dst_len; i += 256)
        if (i >= 0) out[i] = 0.f;
}

// ------------------------------------------------------------------------------------------------
constexpr int kKbZeros = 64, kKbTable = 512;
constexpr int kKbLen = kKbZeros * kKbTable + 1;                  // 32,769 table entries
// transposed table of resample_kernel: (step + 2) rows of R floats; step <= 512 -> R = 68; the pitch range's smallest step (scale 32/46) is 356 -> R = 96
constexpr int kResampleLds = 150 * 1024;

// One workgroup of 1024 threads per clip.  Pitch-off clips copy through.
// The taps of one output are the table entries offset + i * step, i = 0, 1, ... -- `step` = int(scale * 512) is the same for every
// output of a clip, `offset` is a pseudo-random phase.  Round 2 kept the half-window in LDS in its natural order: one ds_read2_b32 per
// tap at a random address per lane (3.5-way bank conflicts on average, 3.3 ms per 4096 clips, half of the augmentation).  Round 3 loads
// it TRANSPOSED by phase, P[ph][i] = tab[ph + i * step] (rows of R floats, step + 2 rows: 137-145 KB): a lane's taps are consecutive in
// its row, so four taps are one ds_read_b128 (two rows: the entry and its upper neighbour for the interpolation), and the four samples
// one 16-byte global load.  Same weights, same fused multiply-adds in the same order: the results are bit-identical to the round-2 kernel.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));       // four consecutive samples at a 4-byte aligned address

__global__ __launch_bounds__(1024) void resample_kernel(const float* __restrict__ Y, const AugDev* __restrict__ plan,
                                                        const LogmelTables* __restrict__ tb, const float* __restrict__ passthru,
                                                        float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float P[];   // [step + 2][R]
    const int clip = blockIdx.x, tid = threadIdx.x;
    float* o = out + int64_t(clip) * kClip;
    if (plan[clip].p_out == 0) {
        for (int i = tid; i < kClip; i += 1024) o[i] = passthru[int64_t(clip) * kClip + i];
        return;
    }
    const double ratio = plan[clip].p_ratio;
    const double scale = ratio < 1.0 ? ratio : 1.0;
    const int index_step = int(scale * kKbTable);
    const int cnt = kKbLen / index_step + 1;                      // entries of row 0: i * step <= kKbLen
    int R = (cnt + 3) & ~3;
    if (((R >> 2) & 1) == 0) R += 4;                              // an odd number of 16-byte slots per row spreads the rows over the banks
    // index kKbLen is the pad np.diff(win) gets (the last value again): tab[kKbLen] = tab[kKbLen - 1]
    {   // every thread takes (step + 2) * R / 1024 entries, phase-fastest (coalesced table reads), four loads in flight
        const int rows = index_step + 2, total = rows * R;
#pragma unroll 4
        for (int e = tid; e < total; e += 1024) {
            const int i = e / rows, ph = e - i * rows;
            const int j = ph + i * index_step;
            P[ph * R + i] = (i < cnt && j <= kKbLen) ? tb->kaiser_best[j < kKbLen ? j : kKbLen - 1] : 0.f;
        }
    }
    __syncthreads();
    const int n_orig = plan[clip].p_len, n_res = plan[clip].p_res;
    const float* y = Y + int64_t(clip) * kAugYStride;
    const double inv = 1.0 / ratio;
    for (int t = tid; t < kClip; t += 1024) {
        // positions and table fractions in float64 (t / ratio needs ~15 integer + 9 fraction bits); the ~140 products per
        // output are float32 FMAs in two independent chains (left wing, right wing), each in tap order.  The two wings advance in
        // lock-step while both have taps left (round 3: twice the loads in flight per wave -- the kernel waits on its table and sample
        // reads, four waves per SIMD being all the LDS-resident table leaves room for)
        float accl = 0.f, accr = 0.f;
        const double time_register = double(t) * inv;
        const int n = int(time_register);
        if (t < n_res && n < n_orig) {
            const double frac_l = scale * (time_register - double(n));
            const double if_l = frac_l * kKbTable;
            const int off_l = int(if_l);
            const float eta_l = float(if_l - double(off_l));
            int i_max = (kKbLen - off_l) / index_step;
            i_max = i_max < n + 1 ? i_max : n + 1;
            const double frac_r = scale - frac_l;
            const double if_r = frac_r * kKbTable;
            const int off_r = int(if_r);
            const float eta_r = float(if_r - double(off_r));
            int k_max = (kKbLen - off_r) / index_step;
            k_max = k_max < n_orig - n - 1 ? k_max : n_orig - n - 1;
#ifdef WW_ABL_RS_SAMEROW          // timing-only ablation: every lane of a wave reads the same table rows (broadcast reads; results are garbage)
            const float* l0 = P + __builtin_amdgcn_readfirstlane(off_l) * R;
            const float* r0 = P + __builtin_amdgcn_readfirstlane(off_r) * R;
#else
            const float* l0 = P + off_l * R;
            const float* r0 = P + off_r * R;
#endif
            const float* l1 = l0 + R;
            const float* r1 = r0 + R;
            auto left4 = [&](int i) {
                const float4 t0 = *reinterpret_cast<const float4*>(l0 + i), t1 = *reinterpret_cast<const float4*>(l1 + i);
#ifdef WW_ABL_RS_NOY              // timing-only ablation: no sample loads
                const f4u yy = {eta_l, t0.x, t1.y, 1.0f};
#else
                const f4u yy = *reinterpret_cast<const f4u*>(y + n - i - 3);              // y[n-i-3 .. n-i]
#endif
                accl = fmaf(fmaf(eta_l, t1.x - t0.x, t0.x), yy.w, accl);
                accl = fmaf(fmaf(eta_l, t1.y - t0.y, t0.y), yy.z, accl);
                accl = fmaf(fmaf(eta_l, t1.z - t0.z, t0.z), yy.y, accl);
                accl = fmaf(fmaf(eta_l, t1.w - t0.w, t0.w), yy.x, accl);
            };
            auto right4 = [&](int k) {
                const float4 t0 = *reinterpret_cast<const float4*>(r0 + k), t1 = *reinterpret_cast<const float4*>(r1 + k);
#ifdef WW_ABL_RS_NOY
                const f4u yy = {eta_r, t0.x, t1.y, 1.0f};
#else
                const f4u yy = *reinterpret_cast<const f4u*>(y + n + 1 + k);              // y[n+1+k .. n+4+k]
#endif
                accr = fmaf(fmaf(eta_r, t1.x - t0.x, t0.x), yy.x, accr);
                accr = fmaf(fmaf(eta_r, t1.y - t0.y, t0.y), yy.y, accr);
                accr = fmaf(fmaf(eta_r, t1.z - t0.z, t0.z), yy.z, accr);
                accr = fmaf(fmaf(eta_r, t1.w - t0.w, t0.w), yy.w, accr);
            };
            const int both = (i_max < k_max ? i_max : k_max) & ~3;
            int i = 0;
#ifndef WW_ABL_RS_SEQUENTIAL
            for (; i < both; i += 4) { left4(i); right4(i); }
#endif
            int k = i;
            for (; i + 4 <= i_max; i += 4) left4(i);
            for (; k + 4 <= k_max; k += 4) right4(k);
            // the last one to three taps of a wing: one more block of four with the weights past the end set to zero (fma(0, y, acc) = acc:
            // the same sum) when its four samples are inside the signal, tap by tap at the signal's edges -- a tap-by-tap tail is a
            // dependent load round trip per tap
#ifndef WW_ABL_RS_SCALAR_TAIL
            if (i < i_max && n - i - 3 >= 0) {
                const float4 t0 = *reinterpret_cast<const float4*>(l0 + i), t1 = *reinterpret_cast<const float4*>(l1 + i);
                const f4u yy = *reinterpret_cast<const f4u*>(y + n - i - 3);
                accl = fmaf(fmaf(eta_l, t1.x - t0.x, t0.x), yy.w, accl);
                accl = fmaf(i + 1 < i_max ? fmaf(eta_l, t1.y - t0.y, t0.y) : 0.f, yy.z, accl);
                accl = fmaf(i + 2 < i_max ? fmaf(eta_l, t1.z - t0.z, t0.z) : 0.f, yy.y, accl);
                i = i_max;
            }
            if (k < k_max && n + 4 + k < n_orig) {
                const float4 t0 = *reinterpret_cast<const float4*>(r0 + k), t1 = *reinterpret_cast<const float4*>(r1 + k);
                const f4u yy = *reinterpret_cast<const f4u*>(y + n + 1 + k);
                accr = fmaf(fmaf(eta_r, t1.x - t0.x, t0.x), yy.x, accr);
                accr = fmaf(k + 1 < k_max ? fmaf(eta_r, t1.y - t0.y, t0.y) : 0.f, yy.y, accr);
                accr = fmaf(k + 2 < k_max ? fmaf(eta_r, t1.z - t0.z, t0.z) : 0.f, yy.z, accr);
                k = k_max;
            }
#endif
            for (; i < i_max; ++i) {
                const float w0 = l0[i], w1 = l1[i];
                accl = fmaf(fmaf(eta_l, w1 - w0, w0), y[n - i], accl);
            }
            for (; k < k_max; ++k) {
                const float w0 = r0[k], w1 = r1[k];
                accr = fmaf(fmaf(eta_r, w1 - w0, w0), y[n + k + 1], accr);
            }
        }
        float acc = accl + accr;
        if (ratio < 1.0) acc *= float(ratio);
        o[t] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// Four samples per thread (float4 in, float4 out).  The uniforms are the hash generator's exact 32-bit words; Box-Muller runs in float32
// (round 2: float64 log / sqrt / cos, 0.35 ms per 4096 clips for a 0.5 GB stream; the float32 form is bound by that stream).  Against the
// float64 evaluation the normal deviate moves by <= 3e-7 of sigma, four orders under the augmentation tests' tolerance.
__global__ __launch_bounds__(256) void noise_kernel(const float* __restrict__ in, const AugDev* __restrict__ plan,
                                                    float* __restrict__ out, int64_t out_stride) {
    const int clip = blockIdx.y;
    const int i = 4 * (blockIdx.x * 256 + threadIdx.x);
    if (i >= kClip) return;
    float4 v = *reinterpret_cast<const float4*>(in + int64_t(clip) * kClip + i);
    const float sigma = plan[clip].sigma;
    if (sigma != 0.f) {
        const uint64_t seed = plan[clip].seed;
        const uint64_t k1 = mix64(seed * 0x9E3779B97F4A7C15ull + 1ull * 0xD1B54A32D192ED03ull + 0x2545F4914F6CDD1Dull);
        const uint64_t k2 = mix64(seed * 0x9E3779B97F4A7C15ull + 2ull * 0xD1B54A32D192ED03ull + 0x2545F4914F6CDD1Dull);
        float* pv = &v.x;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint64_t j = uint64_t(i + q) + 1ull;
            const uint32_t w1 = uint32_t(mix64(k1 + j * 0x9E3779B97F4A7C15ull) >> 32), w2 = uint32_t(mix64(k2 + j * 0x9E3779B97F4A7C15ull) >> 32);
            const float u1 = (float(w1 >> 8) + (float(w1 & 0xffu) + 0.5f) * (1.0f / 256.0f)) * (1.0f / 16777216.0f);   // (w1 + 0.5) / 2^32 to float32
            const float u2 = (float(w2 >> 8) + float(w2 & 0xffu) * (1.0f / 256.0f)) * (1.0f / 16777216.0f) + (0.5f / 4294967296.0f);
            const float nrm = sqrtf(-2.0f * logf(u1)) * cospif(2.0f * u2);
            pv[q] = fmaf(sigma, nrm, pv[q]);
        }
    }
    float* o = out + int64_t(clip) * out_stride + i;
    if ((out_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) *reinterpret_cast<float4*>(o) = v;
    else { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
}

// ------------------------------------------------------------------------------------------------
static int64_t up256(int64_t b) { return (b + 255) & ~int64_t(255); }

int64_t augment_workspace_bytes(int64_t n) {
    return up256(n * int64_t(sizeof(AugDev))) + 2 * up256(n * int64_t(kClip) * 4) + up256(n * int64_t(kAugDFrames) * kSpec * 8) +
           up256(n * int64_t(kAugMaxOut) * kSpec * 8) + up256(n * int64_t(kAugYStride) * 4);
}

// Pinned staging for the per-clip records: two slots per device, each guarded by an event, so the call can return as soon as the
// copy and the kernels are enqueued (round 1 synchronised the stream because the records lived in a pageable vector).
struct PlanStage {
    AugDev* host = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    bool in_use = false;
};
static std::mutex g_stage_mu;
static PlanStage g_stage[16][2];
static int g_stage_next[16] = {};

// plans -> the per-clip records the kernels read (librosa's lengths are host arithmetic: len(np.arange), round, ceil)
int augment_prepare(const ww_augment_plan* plans_host, int64_t n, void* records_host, int* any_pitch_out, int* any_stretch_out) {
    AugDev* host = static_cast<AugDev*>(records_host);
    bool any_pitch = false, any_stretch = false;
    for (int64_t c = 0; c < n; ++c) {
        const ww_augment_plan& p = plans_host[c];
        AugDev d = {};
        int64_t sh = int64_t(p.shift) % kClip;
        d.shift = int32_t(sh < 0 ? sh + kClip : sh);
        d.sigma = p.noise_sigma;
        d.seed = p.noise_seed;
        if (!(p.noise_sigma >= 0.f)) return fail(WW_EINVAL, "plan %lld: noise_sigma must be >= 0", (long long)c);
        auto steps = [](double rate) { return int(std::ceil(double(kAugFrames) / rate)); };
        auto check = [&](double rate, const char* what) {
            if (!(rate > 0.0) || steps(rate) > kAugMaxOut || steps(rate) < 2)
                return fail(WW_EUNSUPPORTED, "plan %lld: %s rate %g outside [%g, 32)", (long long)c, what, rate, double(kAugFrames) / kAugMaxOut);
            return int(WW_OK);
        };
        if (p.pitch_rate != 0.0) {
            if (int rc = check(p.pitch_rate, "pitch")) return rc;
            d.p_rate = p.pitch_rate;
            d.p_out = steps(p.pitch_rate);
            d.p_len = int32_t(std::nearbyint(double(kClip) / p.pitch_rate));      // Python round(): half to even
            d.p_ratio = double(WW_SAMPLE_RATE) / (double(WW_SAMPLE_RATE) / p.pitch_rate);
            d.p_res = int32_t(std::ceil(double(d.p_len) * d.p_ratio));
            any_pitch = true;
        }
        if (p.stretch_rate != 0.0) {
            if (int rc = check(p.stretch_rate, "stretch")) return rc;
            d.s_rate = p.stretch_rate;
            d.s_out = steps(p.stretch_rate);
            d.s_len = int32_t(std::nearbyint(double(kClip) / p.stretch_rate));
            const int over = d.s_len > kClip ? d.s_len - kClip : 0;
            if (p.crop_start < 0 || p.crop_start > over)
                return fail(WW_EINVAL, "plan %lld: crop_start %d outside [0, %d]", (long long)c, p.crop_start, over);
            d.crop = p.crop_start;
            any_stretch = true;
        }
        host[c] = d;
    }
    if (any_pitch_out) *any_pitch_out = any_pitch;
    if (any_stretch_out) *any_stretch_out = any_stretch;
    return WW_OK;
}
int64_t augment_record_bytes() { return int64_t(sizeof(AugDev)); }

// The kernels alone, on records already in device memory: nothing but launches on `stream` (capturable into a hipGraph).  A stage whose
// flag is off for a clip copies that clip through, so both stages may always be launched (what a captured graph must do).
int launch_augment_records(const float* pcm, int64_t n, int64_t stride, const void* records_dev, bool any_pitch, bool any_stretch, float* out,
                           int64_t out_stride, void* workspace, hipStream_t stream) {
    if (n == 0) return WW_OK;
    const LogmelTables* tb = device_tables();
    if (!tb) return WW_EHIP;
    const AugDev* plan = static_cast<const AugDev*>(records_dev);
    char* w = static_cast<char*>(workspace);
    w += up256(n * int64_t(sizeof(AugDev)));                        // (the slot ww_augment_f32 copies its records into)
    float* bufA = reinterpret_cast<float*>(w); w += up256(n * int64_t(kClip) * 4);
    float* bufB = reinterpret_cast<float*>(w); w += up256(n * int64_t(kClip) * 4);
    [[maybe_unused]] float2* D = reinterpret_cast<float2*>(w); w += up256(n * int64_t(kAugDFrames) * kSpec * 8);
    float2* S = reinterpret_cast<float2*>(w); w += up256(n * int64_t(kAugMaxOut) * kSpec * 8);
    float* Y = reinterpret_cast<float*>(w);
    {
        static std::mutex mu;
        static bool attr[64] = {};
        std::lock_guard<std::mutex> lock(mu);
        int dev = 0;
        WW_HIP(hipGetDevice(&dev));
        if (dev >= 0 && dev < 64 && !attr[dev]) {
            WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(istft_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kIstftLds));
            WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(stft_pv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kStftPvLds));
            WW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(resample_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kResampleLds));
            attr[dev] = true;
        }
    }
    const dim3 egrid((kClip / 4 + 255) / 256, unsigned(n));
    hipLaunchKernelGGL(roll_kernel, egrid, dim3(256), 0, stream, pcm, stride, plan, bufA);
    float* cur = bufA;
    float* other = bufB;
    if (any_pitch) {
#ifdef WW_AUG_SPLIT_STFT_PV
        hipLaunchKernelGGL(stft_kernel, dim3(unsigned(n)), dim3(256), 0, stream, cur, plan, 0, tb, D);
        hipLaunchKernelGGL(pv_kernel, dim3(unsigned(n)), dim3(256), 0, stream, D, plan, 0, S);
#else
        hipLaunchKernelGGL(stft_pv_kernel, dim3(unsigned(n)), dim3(256), kStftPvLds, stream, cur, plan, 0, tb, S);
#endif
        hipLaunchKernelGGL(istft_kernel, dim3(unsigned(n)), dim3(256), kIstftLds, stream, S, plan, 0, tb,
                           static_cast<const float*>(nullptr), Y, int64_t(kAugYStride));
        hipLaunchKernelGGL(resample_kernel, dim3(unsigned(n)), dim3(1024), kResampleLds, stream, Y, plan, tb, cur, other);
        float* t = cur; cur = other; other = t;
    }
    if (any_stretch) {
#ifdef WW_AUG_SPLIT_STFT_PV
        hipLaunchKernelGGL(stft_kernel, dim3(unsigned(n)), dim3(256), 0, stream, cur, plan, 1, tb, D);
        hipLaunchKernelGGL(pv_kernel, dim3(unsigned(n)), dim3(256), 0, stream, D, plan, 1, S);
#else
        hipLaunchKernelGGL(stft_pv_kernel, dim3(unsigned(n)), dim3(256), kStftPvLds, stream, cur, plan, 1, tb, S);
#endif
        hipLaunchKernelGGL(istft_kernel, dim3(unsigned(n)), dim3(256), kIstftLds, stream, S, plan, 1, tb, cur, other, int64_t(kClip));
        float* t = cur; cur = other; other = t;
    }
    hipLaunchKernelGGL(noise_kernel, dim3((kClip / 4 + 255) / 256, unsigned(n)), dim3(256), 0, stream, cur, plan, out, out_stride);
    WW_HIP(hipGetLastError());
    return WW_OK;
}

int launch_augment(const float* pcm, int64_t n, int64_t stride, const ww_augment_plan* plans_host, float* out,
                   int64_t out_stride, void* workspace, hipStream_t stream) {
    if (n == 0) return WW_OK;
    std::vector<AugDev> host(static_cast<size_t>(n));
    int any_pitch = 0, any_stretch = 0;
    if (int rc = augment_prepare(plans_host, n, host.data(), &any_pitch, &any_stretch)) return rc;
    AugDev* plan = reinterpret_cast<AugDev*>(workspace);
    {
        int dev = 0;
        WW_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= 16) return fail(WW_EUNSUPPORTED, "device ordinal %d out of range", dev);
        std::lock_guard<std::mutex> lock(g_stage_mu);
        PlanStage& st = g_stage[dev][g_stage_next[dev]];
        g_stage_next[dev] ^= 1;
        if (st.in_use) WW_HIP(hipEventSynchronize(st.ev));          // the copy that last read this slot (two calls ago) must be done
        if (st.cap < size_t(n)) {
            if (st.host) WW_HIP(hipHostFree(st.host));
            st.host = nullptr;
            st.cap = 0;
            WW_HIP(hipHostMalloc(reinterpret_cast<void**>(&st.host), size_t(n) * sizeof(AugDev), hipHostMallocDefault));
            st.cap = size_t(n);
        }
        if (!st.ev) WW_HIP(hipEventCreateWithFlags(&st.ev, hipEventDisableTiming));
        std::memcpy(st.host, host.data(), size_t(n) * sizeof(AugDev));
        WW_HIP(hipMemcpyAsync(plan, st.host, size_t(n) * sizeof(AugDev), hipMemcpyHostToDevice, stream));
        WW_HIP(hipEventRecord(st.ev, stream));
        st.in_use = true;
    }
    return launch_augment_records(pcm, n, stride, plan, any_pitch != 0, any_stretch != 0, out, out_stride, workspace, stream);
}

}  // namespace ww
