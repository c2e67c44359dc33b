// K0: PCM decode + mono mix + polyphase resample to 16 kHz + peak-normalise + crop/pad to one second.
//
// Replaces the numeric part of AudioProcessor.load_audio = librosa.load(path, sr=16000)
// (/root/reference/wakeword_training_script.py:65-71) followed by normalize_audio (:73-76) and pad_or_truncate (:78-83),
// i.e. process_audio_file :125-133 up to the mel call.  File reading and RIFF parsing stay on the host; the sample
// conversion (soundfile's int -> float scaling), channel mean (librosa.to_mono) and the resampler run here.
//
// The resampler is a Kaiser-windowed-sinc polyphase filter with scipy.signal.resample_poly's exact design
// (firwin(20*max(up,down)+1, 1/max(up,down), ('kaiser', 5.0)) * up, centred as upfirdn does).  librosa's own resampler
// (soxr_hq, a third-party library that is not installed) uses a different filter: parity with it is UNPINNED; parity
// with the scipy design is what tests/ checks.  Files already at 16 kHz take no filter at all (exact).
//
// One workgroup per clip: every output sample of the WHOLE file is computed once (the peak is taken over the whole
// file, as the reference normalises before cropping), samples inside the 1 s window are stored, then rescaled.
#include <cmath>
#include <map>
#include <mutex>
#include <vector>

#include "ww_internal.h"

namespace ww {

struct ResampleFilter {
    float* taps_dev;   // [2*half_len + 1] = firwin(...) * up
    int up, down, half_len;
};
static std::mutex g_rs_mu;
static std::map<long long, ResampleFilter> g_filters;   // key = device << 32 | sample_rate

static long long gcdll(long long a, long long b) { return b ? gcdll(b, a % b) : a; }

// scipy.signal.firwin(numtaps, cutoff, window=('kaiser', 5.0)) with pass_zero=True, fs=2 -- host, double precision
static void firwin_kaiser(int numtaps, double cutoff, std::vector<double>& h) {
    const double alpha = 0.5 * (numtaps - 1), beta = 5.0;
    h.resize(numtaps);
    double sum = 0.0;
    for (int n = 0; n < numtaps; ++n) {
        const double m = n - alpha;
        const double xs = M_PI * cutoff * m;
        const double sinc = xs == 0.0 ? 1.0 : std::sin(xs) / xs;
        const double r = (n - alpha) / alpha;
        const double w = std::cyl_bessel_i(0.0, beta * std::sqrt(std::fmax(0.0, 1.0 - r * r))) / std::cyl_bessel_i(0.0, beta);
        h[n] = cutoff * sinc * w;
        sum += h[n];
    }
    for (double& v : h) v /= sum;
}

int resample_taps_host(int sample_rate, float* out, int max_taps, int* up_o, int* down_o, int* half_len_o) {
    if (sample_rate < 1000 || sample_rate > 384000) return fail(WW_EINVAL, "sample rate %d out of range", sample_rate);
    const long long g = gcdll(WW_SAMPLE_RATE, sample_rate);
    const int up = int(WW_SAMPLE_RATE / g), down = int(sample_rate / g);
    const int max_rate = up > down ? up : down, half_len = 10 * max_rate, n = 2 * half_len + 1;
    if (up_o) *up_o = up;
    if (down_o) *down_o = down;
    if (half_len_o) *half_len_o = half_len;
    if (up == 1 && down == 1) return 0;
    if (!out) return n;
    if (n > max_taps) return fail(WW_EINVAL, "filter for %d Hz needs %d taps", sample_rate, n);
    std::vector<double> h;
    firwin_kaiser(n, 1.0 / max_rate, h);
    for (int i = 0; i < n; ++i) out[i] = float(h[i] * up);
    return n;
}

static int get_filter(int sample_rate, ResampleFilter* f) {
    int dev = 0;
    WW_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_rs_mu);
    const long long key = (static_cast<long long>(dev) << 32) | static_cast<unsigned>(sample_rate);
    auto it = g_filters.find(key);
    if (it != g_filters.end()) { *f = it->second; return WW_OK; }
    ResampleFilter nf{nullptr, 1, 1, 0};
    const int n = resample_taps_host(sample_rate, nullptr, 0, &nf.up, &nf.down, &nf.half_len);
    if (n < 0) return n;
    if (n > 0) {
        std::vector<float> taps(n);
        resample_taps_host(sample_rate, taps.data(), n, nullptr, nullptr, nullptr);
        WW_HIP(hipMalloc(reinterpret_cast<void**>(&nf.taps_dev), sizeof(float) * n));
        WW_HIP(hipMemcpy(nf.taps_dev, taps.data(), sizeof(float) * n, hipMemcpyHostToDevice));
    }
    g_filters[key] = nf;
    *f = nf;
    return WW_OK;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sample_mono(const uint8_t* __restrict__ p, int64_t frame, int channels, int fmt) {
    // soundfile's conversion to float32 followed by librosa.to_mono (mean over channels)
    float s = 0.f;
    for (int c = 0; c < channels; ++c) {
        const int64_t i = frame * channels + c;
        float v;
        switch (fmt) {
            case WW_FMT_S16: v = float(reinterpret_cast<const int16_t*>(p)[i]) * (1.0f / 32768.0f); break;
            case WW_FMT_U8:  v = (float(p[i]) - 128.0f) * (1.0f / 128.0f); break;
            case WW_FMT_S24: {
                const uint8_t* b = p + 3 * i;
                int32_t x = int32_t(b[0]) | (int32_t(b[1]) << 8) | (int32_t(int8_t(b[2])) << 16);
                v = float(x) * (1.0f / 8388608.0f);
                break;
            }
            case WW_FMT_S32: v = float(reinterpret_cast<const int32_t*>(p)[i]) * (1.0f / 2147483648.0f); break;
            case WW_FMT_F64: v = float(reinterpret_cast<const double*>(p)[i]); break;
            default:         v = reinterpret_cast<const float*>(p)[i]; break;
        }
        s += v;
    }
    return channels > 1 ? s / float(channels) : s;
}

// Files that need the filter and whose filter fits go to resample_lds_kernel; decode_resample_kernel takes the rest (16 kHz files: no
// filter at all; rates whose reduced up / down leave a filter too long for LDS: the taps from global memory).  Both kernels are
// launched over all clips and each skips the other's: the split needs nothing from the host (the descriptors live in device memory).
constexpr int kRsTaps = 8960;                       // filter taps kept in LDS (44.1 kHz family: 2 * 10 * 441 + 1 = 8821)
constexpr int kRsSpan = 7168;                       // input frames of one output block, converted to mono float32 once
constexpr int kRsThreads = 512;
constexpr int kRsLds = (kRsTaps + kRsSpan) * int(sizeof(float));            // 64,512 B: two workgroups per CU

__device__ __forceinline__ bool resample_in_lds(int up, int down, int half_len) {
    const int lh = 2 * half_len + 1;
    return !(up == 1 && down == 1) && lh <= kRsTaps && lh / up + 8 < kRsSpan / 2;
}

// One workgroup per file.  Output blocks of up to 2048 samples: the block's input span is decoded (sample format, channel mean) into LDS
// once -- the direct form decodes every input frame once per tap that touches it, ~20 times at 48 kHz --, the filter sits in LDS beside it,
// and each output is the same fused multiply-add chain in the same order as decode_resample_kernel's (bit-identical results).
__global__ __launch_bounds__(kRsThreads) void resample_lds_kernel(const uint8_t* __restrict__ raw, const ww_clip_desc* __restrict__ descs,
                                                                  int n_clips, int normalize, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float rs_lds[];
    float* tapsL = rs_lds;
    float* xs = rs_lds + kRsTaps;
    __shared__ float red[kRsThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const void* taps_loaded = nullptr;
    for (int clip = blockIdx.x; clip < n_clips; clip += gridDim.x) {
        const ww_clip_desc d = descs[clip];
        const int up = d.up, down = d.down, half_len = d.half_len;
        if (!resample_in_lds(up, down, half_len)) continue;
        const uint8_t* __restrict__ p = raw + d.byte_offset;
        const int64_t n_in = d.n_frames;
        const int lh = 2 * half_len + 1;
        if (d.taps_dev != taps_loaded) {                          // consecutive files usually share their rate
            __syncthreads();
            const float* __restrict__ taps = reinterpret_cast<const float*>(d.taps_dev);
            for (int i = tid; i < lh; i += kRsThreads) tapsL[i] = taps[i];
            taps_loaded = d.taps_dev;
        }
        int64_t n_out = n_in * up;
        n_out = n_out / down + (n_out % down ? 1 : 0);
        const int n_pre_pad = down - half_len % down;
        const int n_pre_remove = (half_len + n_pre_pad) / down;
        float* __restrict__ o = out + int64_t(clip) * kClip;
        float peak = 0.f;
        const int64_t total = n_out > d.crop_start + kClip ? n_out : d.crop_start + kClip;   // also writes the zero pad
        int64_t blk64 = int64_t(kRsSpan - lh / up - 8) * up / down;
        const int blk = int(blk64 > 2048 ? 2048 : (blk64 < 1 ? 1 : blk64));
        auto first_in = [&](int64_t c) -> int64_t { return c - lh + 1 <= 0 ? 0 : (c - lh + 1 + up - 1) / up; };
        auto last_in = [&](int64_t c) -> int64_t {
            int64_t i = c < 0 ? -1 : c / up;
            return i > n_in - 1 ? n_in - 1 : i;
        };
        // without normalisation nothing outside the 1 s window is needed (no whole-file peak): load_audio's windows then cost one window each,
        // not one file each (round-3 review: a long recording was resampled once per second of its length); per output the same fma chain
        const int64_t j_begin = normalize ? 0 : d.crop_start;
        const int64_t j_end = normalize || total < d.crop_start + kClip ? total : d.crop_start + kClip;
        for (int64_t j0 = j_begin; j0 < j_end; j0 += blk) {
            const int64_t j1 = (j0 + blk < n_out ? j0 + blk : n_out) - 1;       // last filtered output of the block (j1 < j0: none)
            const int64_t i_min = first_in((j0 + n_pre_remove) * int64_t(down) - n_pre_pad);
            const int64_t i_max = j1 >= j0 ? last_in((j1 + n_pre_remove) * int64_t(down) - n_pre_pad) : i_min - 1;
            __syncthreads();                                      // the previous block's outputs are done with xs (and the taps are in place)
            for (int64_t i = i_min + tid; i <= i_max; i += kRsThreads) xs[i - i_min] = sample_mono(p, i, d.channels, d.format);
            __syncthreads();
            // inside a block everything is 32-bit and relative to the block's first input frame: c - i_min * up fits easily (span * up),
            // and since i_min * up is a multiple of up the floor / ceiling divisions carry over (64-bit divides per output cost as much as its taps)
            const int crel0 = int((j0 + n_pre_remove) * int64_t(down) - n_pre_pad - i_min * up);
            const int64_t room = n_in - 1 - i_min;
            const int i_cap = room < kRsSpan ? int(room) : kRsSpan;
            const int jn = int((j0 + blk < j_end ? j0 + blk : j_end) - j0);
            for (int jj = tid; jj < jn; jj += kRsThreads) {
                const int64_t j = j0 + jj;
                float y = 0.f;
                if (j < n_out) {
                    const int crel = crel0 + jj * down;                       // = c - i_min * up, c the output's centre tap position
                    const int x = crel - lh + 1;
                    const int i_lo = x <= 0 ? 0 : (x + up - 1) / up;
                    int i_hi = crel < 0 ? -1 : crel / up;
                    i_hi = i_hi < i_cap ? i_hi : i_cap;
                    int t = crel - i_lo * up;
#pragma unroll 8
                    for (int i = i_lo; i <= i_hi; ++i, t -= up) y = fmaf(xs[i], tapsL[t], y);
                    peak = fmaxf(peak, fabsf(y));
                }
                const int64_t w = j - d.crop_start;
                if (w >= 0 && w < kClip) o[w] = y;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) peak = fmaxf(peak, __shfl_xor(peak, off));
        __syncthreads();
        if (lane == 0) red[wave] = peak;
        __syncthreads();
        peak = red[0];
#pragma unroll
        for (int w = 1; w < kRsThreads / 64; ++w) peak = fmaxf(peak, red[w]);
        if (normalize) {
            const int64_t valid = n_out - d.crop_start < kClip ? n_out - d.crop_start : kClip;
            for (int w = tid; w < valid; w += kRsThreads) o[w] = o[w] / peak;
        }
    }
}

__global__ __launch_bounds__(256) void decode_resample_kernel(const uint8_t* __restrict__ raw,
                                                             const ww_clip_desc* __restrict__ descs, int n_clips,
                                                             int normalize, float* __restrict__ out) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int clip = blockIdx.x; clip < n_clips; clip += gridDim.x) {
        const ww_clip_desc d = descs[clip];
#ifndef WW_ABL_K0_DIRECT
        if (resample_in_lds(d.up, d.down, d.half_len)) continue;      // resample_lds_kernel's
#endif
        const uint8_t* __restrict__ p = raw + d.byte_offset;
        const float* __restrict__ taps = reinterpret_cast<const float*>(d.taps_dev);
        const int64_t n_in = d.n_frames;
        const int up = d.up, down = d.down, half_len = d.half_len;
        int64_t n_out = n_in * up;
        n_out = n_out / down + (n_out % down ? 1 : 0);
        const int n_pre_pad = (up == 1 && down == 1) ? 0 : down - half_len % down;
        const int n_pre_remove = (up == 1 && down == 1) ? 0 : (half_len + n_pre_pad) / down;
        const int lh = 2 * half_len + 1;
        float* __restrict__ o = out + int64_t(clip) * kClip;
        float peak = 0.f;
#ifndef WW_ABL_K0_NOFAST
        // the data set's usual file -- 16 kHz, mono, PCM-16, at most one second (create_sample_data's format) -- in one pass: eight samples
        // per 16-byte load, the clip held in registers between the peak reduction and the scaled store (the general loop below stores,
        // reduces, then reads and rewrites).  Same conversion, same division: bit-identical.
        if (up == 1 && down == 1 && d.channels == 1 && d.format == WW_FMT_S16 && n_in <= kClip && d.crop_start == 0 &&
            (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (reinterpret_cast<uintptr_t>(o) & 15) == 0) {
            constexpr int kV = (kClip / 8 + 255) / 256;                  // 8 vectors of 8 samples per thread cover 16,384 >= 16,000
            float v[kV][8];
#pragma unroll
            for (int k = 0; k < kV; ++k) {
                const int base = 8 * (tid + 256 * k);
                u32x4 w = {0u, 0u, 0u, 0u};
                if (base + 8 <= n_in) w = *reinterpret_cast<const u32x4*>(p + 2 * base);
                else if (base < n_in) {                                  // the file's last, partial vector: sample by sample, nothing read past its end
                    for (int q = 0; q < int(n_in) - base; ++q)
                        w[q >> 1] |= uint32_t(reinterpret_cast<const uint16_t*>(p)[base + q]) << (16 * (q & 1));
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int16_t sx = int16_t((w[q >> 1] >> (16 * (q & 1))) & 0xffffu);
                    v[k][q] = base + q < n_in ? float(sx) * (1.0f / 32768.0f) : 0.f;
                    peak = fmaxf(peak, fabsf(v[k][q]));
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) peak = fmaxf(peak, __shfl_xor(peak, off));
            __syncthreads();
            if (lane == 0) red[wave] = peak;
            __syncthreads();
            peak = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
#pragma unroll
            for (int k = 0; k < kV; ++k) {
                const int base = 8 * (tid + 256 * k);
                if (base < kClip) {
                    float r[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) r[q] = (normalize && base + q < n_in) ? v[k][q] / peak : v[k][q];
                    *reinterpret_cast<float4*>(o + base) = make_float4(r[0], r[1], r[2], r[3]);
                    *reinterpret_cast<float4*>(o + base + 4) = make_float4(r[4], r[5], r[6], r[7]);
                }
            }
            continue;
        }
#endif
        const int64_t total = n_out > d.crop_start + kClip ? n_out : d.crop_start + kClip;   // also writes the zero pad
        const int64_t j_begin = normalize ? 0 : d.crop_start;          // see resample_lds_kernel
        const int64_t j_end = normalize || total < d.crop_start + kClip ? total : d.crop_start + kClip;
        for (int64_t j = j_begin + tid; j < j_end; j += 256) {
            float y = 0.f;
            if (j < n_out) {
                if (up == 1 && down == 1) {
                    y = sample_mono(p, j, d.channels, d.format);
                } else {
                    const int64_t c = (j + n_pre_remove) * int64_t(down) - n_pre_pad;   // tap index t = c - i*up
                    int64_t i_hi = c / up;
                    if (c < 0) i_hi = -1;
                    if (i_hi > n_in - 1) i_hi = n_in - 1;
                    int64_t i_lo = (c - lh + 1 + up - 1) / up;      // ceil((c - lh + 1) / up) for the positive case
                    if (c - lh + 1 <= 0) i_lo = 0;
                    for (int64_t i = i_lo; i <= i_hi; ++i) y = fmaf(sample_mono(p, i, d.channels, d.format), taps[c - i * up], y);
                }
                peak = fmaxf(peak, fabsf(y));
            }
            const int64_t w = j - d.crop_start;
            if (w >= 0 && w < kClip) o[w] = y;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) peak = fmaxf(peak, __shfl_xor(peak, off));
        __syncthreads();
        if (lane == 0) red[wave] = peak;
        __syncthreads();
        peak = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (normalize) {
            // x / max|x| over the WHOLE file (normalize_audio precedes pad_or_truncate); 0/0 = NaN like the reference,
            // except for the zero padding, which the reference appends after normalising
            const int64_t valid = n_out - d.crop_start < kClip ? n_out - d.crop_start : kClip;
            for (int w = tid; w < valid; w += 256) o[w] = o[w] / peak;
        }
    }
}

}  // namespace ww

using namespace ww;

extern "C" {

int ww_resample_taps_host(int32_t sample_rate, float* taps_host, int32_t max_taps, int32_t* up, int32_t* down, int32_t* half_len) {
    int u = 1, dn = 1, hl = 0;
    const int n = resample_taps_host(sample_rate, taps_host, max_taps, &u, &dn, &hl);
    if (up) *up = u;
    if (down) *down = dn;
    if (half_len) *half_len = hl;
    return n;
}

int ww_resampler_prepare(int32_t sample_rate, ww_clip_desc* desc_host) {
    if (!desc_host) return fail(WW_EINVAL, "null descriptor");
    if (int rc = require_gfx950()) return rc;
    ResampleFilter f;
    if (int rc = get_filter(sample_rate, &f)) return rc;
    desc_host->sample_rate = sample_rate;
    desc_host->up = f.up;
    desc_host->down = f.down;
    desc_host->half_len = f.half_len;
    desc_host->taps_dev = f.taps_dev;
    return WW_OK;
}

int ww_decode_resample(const uint8_t* raw_dev, const ww_clip_desc* descs_dev, int64_t n_clips, int normalize,
                       float* pcm_out_dev, ww_stream_t stream) {
    if (n_clips < 0 || n_clips > (int64_t(1) << 24)) return fail(WW_EINVAL, "n_clips %lld out of range", (long long)n_clips);
    if (n_clips == 0) return WW_OK;
    if (!raw_dev || !descs_dev || !pcm_out_dev) return fail(WW_EINVAL, "null pointer");
    if (int rc = require_gfx950()) return rc;
    const int64_t resident = int64_t(device_cu_count()) * 8;
    const int grid = int(n_clips < resident ? n_clips : resident);
    hipLaunchKernelGGL(decode_resample_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), raw_dev,
                       descs_dev, int(n_clips), normalize, pcm_out_dev);
#ifndef WW_ABL_K0_DIRECT
    const int64_t resident2 = int64_t(device_cu_count()) * 2;
    hipLaunchKernelGGL(resample_lds_kernel, dim3(int(n_clips < resident2 ? n_clips : resident2)), dim3(kRsThreads), kRsLds,
                       static_cast<hipStream_t>(stream), raw_dev, descs_dev, int(n_clips), normalize, pcm_out_dev);
#endif
    WW_HIP(hipGetLastError());
    return WW_OK;
}

}  // extern "C"
