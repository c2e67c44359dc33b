// C ABI entry points of libwakeword_amd.so: argument checking, per-device table cache, the composed
// PCM -> logits path and the hipGraph-captured streaming step.  Declared in include/wakeword_amd.h.
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>

#include <atomic>

#include "ww_internal.h"

namespace ww {

constexpr int kMaxDevices = 16;
static std::mutex g_mu;
static LogmelTables* g_tables[kMaxDevices] = {};
static int g_cu[kMaxDevices] = {};
static int g_checked[kMaxDevices] = {};   // 0 unknown, 1 gfx950, -1 refused

static int current_device(int* dev) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return fail(WW_ENODEVICE, "no HIP device visible: the HIP kernels are the only implementation of this path");
    }
    WW_HIP(hipGetDevice(dev));
    if (*dev < 0 || *dev >= kMaxDevices) return fail(WW_EUNSUPPORTED, "device ordinal %d out of range", *dev);
    return WW_OK;
}

int require_gfx950() {
    int dev = 0;
    if (int rc = current_device(&dev)) return rc;
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_checked[dev] == 0) {
        hipDeviceProp_t prop;
        WW_HIP(hipGetDeviceProperties(&prop, dev));
        g_cu[dev] = prop.multiProcessorCount;
        g_checked[dev] = std::strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : -1;
    }
    if (g_checked[dev] < 0) return fail(WW_ENODEVICE, "device %d is not gfx950 (MI355X): this library carries gfx950 code only", dev);
    return WW_OK;
}

int device_cu_count() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 256;
    return g_cu[dev] > 0 ? g_cu[dev] : 256;
}

const LogmelTables* device_tables() {
    int dev = 0;
    if (require_gfx950() != WW_OK) return nullptr;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_tables[dev]) return g_tables[dev];
    LogmelTables* host = new (std::nothrow) LogmelTables;
    if (!host) { fail(WW_EHIP, "out of host memory"); return nullptr; }
    const int pieces = build_logmel_tables(host);
    if (pieces < 0) { delete host; return nullptr; }
    LogmelTables* devp = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&devp), sizeof(LogmelTables));
    if (e == hipSuccess) e = hipMemcpy(devp, host, sizeof(LogmelTables), hipMemcpyHostToDevice);
    delete host;
    if (e != hipSuccess) {
        fail(WW_EHIP, "uploading the log-mel tables failed: %s", hipGetErrorString(e));
        return nullptr;
    }
    g_tables[dev] = devp;
    return devp;
}

// Process-wide defaults (atomics: set from any thread) and per-thread overrides (WW_MATH_INHERIT = none): a launch reads its mode ONCE,
// the calling thread's override first -- two host threads (a streamer and a batch job) can use different arithmetics safely.
static std::atomic<int> g_conv_math{WW_CONV_MATH_F16X3};
static thread_local int t_conv_math = WW_MATH_INHERIT, t_logmel_math = WW_MATH_INHERIT;
static int g_train_math = WW_TRAIN_MATH_F16X3;
int train_math_mode() { return g_train_math; }
int conv_math_mode() { return t_conv_math != WW_MATH_INHERIT ? t_conv_math : g_conv_math.load(std::memory_order_relaxed); }
void set_conv_math_mode(int mode) { g_conv_math.store(mode, std::memory_order_relaxed); }

static std::atomic<int> g_logmel_math{WW_LOGMEL_MATH_AUTO};
int logmel_math_mode() { return t_logmel_math != WW_MATH_INHERIT ? t_logmel_math : g_logmel_math.load(std::memory_order_relaxed); }
void set_logmel_math_mode(int mode) { g_logmel_math.store(mode, std::memory_order_relaxed); }

static int64_t align256(int64_t b) { return (b + 255) & ~int64_t(255); }
int64_t cnn_scratch_bytes(int64_t n, int n_conv);

struct Workspace {
    float* logmel;
    float* pooled;
    void* scratch;
    int64_t total;
};
static Workspace carve(void* base, int64_t n, int n_conv) {
    Workspace w{};
    char* p = static_cast<char*>(base);
    int64_t o = 0;
    w.logmel = reinterpret_cast<float*>(p + o); o += align256(n * kMels * kFrames * int64_t(sizeof(float)));
    w.pooled = reinterpret_cast<float*>(p + o); o += align256(n * 128 * int64_t(sizeof(float)));
    w.scratch = p + o; o += align256(cnn_scratch_bytes(n, n_conv));
    w.total = o;
    return w;
}

static int check_pcm(const float* pcm, int64_t n_clips, int64_t clip_stride, int64_t clip_len) {
    if (n_clips < 0 || n_clips > (int64_t(1) << 30)) return fail(WW_EINVAL, "n_clips %lld out of range", (long long)n_clips);
    if (n_clips == 0) return WW_OK;
    if (!pcm) return fail(WW_EINVAL, "null pcm pointer");
    if (clip_len <= 0 || clip_len > kClip)
        return fail(WW_EINVAL, "clip_len %lld: expected 1..%d samples (longer clips are cropped by the host, "
                    "pad_or_truncate wakeword_training_script.py:78-83)", (long long)clip_len, kClip);
    if (clip_stride < clip_len && n_clips > 1) return fail(WW_EINVAL, "clip_stride %lld < clip_len %lld", (long long)clip_stride, (long long)clip_len);
    // a single clip has no second row: its stride is never used (any clip_len, e.g. 15999, is fine)
    if ((reinterpret_cast<uintptr_t>(pcm) & 15) || (n_clips > 1 && (clip_stride & 3)))
        return fail(WW_EINVAL, "pcm must be 16-byte aligned with clip_stride %% 4 == 0 (got %p, %lld)", (const void*)pcm, (long long)clip_stride);
    return WW_OK;
}

static int check_model(int64_t n, int32_t width, const float* packed, int32_t n_conv) {
    if (n < 0 || n > (int64_t(1) << 30)) return fail(WW_EINVAL, "batch %lld out of range", (long long)n);
    if (n_conv != 2 && n_conv != 3) return fail(WW_EINVAL, "n_conv must be 2 or 3, got %d", n_conv);
    if (width < 1 || width > WW_MAX_WIDTH) return fail(WW_EUNSUPPORTED, "mel width %d: the conv kernels take 1..%d frames", width, WW_MAX_WIDTH);
    if (n > 0 && (!packed || (reinterpret_cast<uintptr_t>(packed) & 15))) return fail(WW_EINVAL, "packed weights must be a 16-byte aligned device pointer");
    return WW_OK;
}

}  // namespace ww

using namespace ww;

// --------------------------------------------------------------------------------------------------
// streaming
// --------------------------------------------------------------------------------------------------
struct ww_streamer {
    int n_mics, hop, n_conv;
    const float* packed;
    hipStream_t stream;
    float* ring;         // [n_mics][16000]
    int32_t* pos;        // device: [0] index of the oldest sample (== next write position), [1] append-kernel ticket
    void* workspace;
    hipGraph_t graph;
    hipGraphExec_t exec;
    const float* cap_hop;
    float* cap_prob;
    float* cap_logits;
    float* own_logits;   // used when the caller passes logits_dev == NULL
};

// Append one hop to every microphone's ring and THEN advance the shared ring position, in one launch: every workgroup
// reads `pos` before it writes samples and takes a ticket after; the workgroup that draws the last ticket knows all
// others are past their read of `pos` and publishes pos + hop for the kernels that follow in the graph.
__global__ void ring_append_kernel(float* __restrict__ ring, int32_t* __restrict__ pos_p, uint32_t* __restrict__ ticket,
                                   const float* __restrict__ hop, int n_mics, int hop_len) {
    const int pos = *reinterpret_cast<volatile int32_t*>(pos_p);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_mics * hop_len; i += gridDim.x * blockDim.x) {
        const int m = i / hop_len, k = i - m * hop_len;
        int at = pos + k;
        if (at >= kClip) at -= kClip;
        ring[int64_t(m) * kClip + at] = hop[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicInc(ticket, gridDim.x - 1) == gridDim.x - 1) {     // wraps to 0: ready for the next replay
            const int p = pos + hop_len;
            *pos_p = p >= kClip ? p - kClip : p;
        }
    }
}
__global__ void ring_unroll_kernel(const float* __restrict__ ring, const int32_t* __restrict__ pos_p,
                                   float* __restrict__ out, int n_mics) {
    const int pos = *pos_p;
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < int64_t(n_mics) * kClip;
         i += int64_t(gridDim.x) * blockDim.x) {
        const int m = int(i / kClip), k = int(i - int64_t(m) * kClip);
        int at = pos + k;
        if (at >= kClip) at -= kClip;
        out[i] = ring[int64_t(m) * kClip + at];
    }
}

static int streamer_enqueue(ww_streamer* s, const float* hop_dev, float* prob_dev, float* logits_dev) {
    const int threads = 256;
    const int blocks = (s->n_mics * s->hop + threads - 1) / threads;
    hipLaunchKernelGGL(ring_append_kernel, dim3(blocks), dim3(threads), 0, s->stream, s->ring, s->pos,
                       reinterpret_cast<uint32_t*>(s->pos + 1), hop_dev, s->n_mics, s->hop);
    WW_HIP(hipGetLastError());
    Workspace w = carve(s->workspace, s->n_mics, s->n_conv);
    if (int rc = launch_logmel(s->ring, s->n_mics, kClip, kClip, 1, s->pos, kClip, w.logmel, s->stream)) return rc;
    if (int rc = launch_cnn_pool(w.logmel, s->n_mics, kFrames, s->packed, s->n_conv, w.scratch, w.pooled, s->stream)) return rc;
    return launch_lstm_fc(w.pooled, s->n_mics, s->packed, s->n_conv, logits_dev, prob_dev, s->stream);
}

extern "C" {

int ww_set_conv_math(int mode) {
    if (mode != WW_CONV_MATH_F32 && mode != WW_CONV_MATH_F16X3 && mode != WW_CONV_MATH_F16X3_DIRECT)
        return fail(WW_EINVAL, "unknown conv math mode %d", mode);
    set_conv_math_mode(mode);
    return WW_OK;
}
int ww_get_conv_math(void) { return conv_math_mode(); }
int ww_set_conv_math_thread(int mode) {
    if (mode != WW_MATH_INHERIT && mode != WW_CONV_MATH_F32 && mode != WW_CONV_MATH_F16X3 && mode != WW_CONV_MATH_F16X3_DIRECT)
        return fail(WW_EINVAL, "unknown conv math mode %d", mode);
    t_conv_math = mode;
    return WW_OK;
}
int ww_set_train_math(int mode) {
    if (mode != WW_TRAIN_MATH_F32 && mode != WW_TRAIN_MATH_F16X3) return fail(WW_EINVAL, "train math %d: expected WW_TRAIN_MATH_F32 or WW_TRAIN_MATH_F16X3", mode);
    g_train_math = mode;
    return WW_OK;
}
int ww_get_train_math(void) { return g_train_math; }

int ww_set_logmel_math(int mode) {
    if (mode != WW_LOGMEL_MATH_F32 && mode != WW_LOGMEL_MATH_F64 && mode != WW_LOGMEL_MATH_AUTO)
        return fail(WW_EINVAL, "unknown log-mel math mode %d", mode);
    set_logmel_math_mode(mode);
    return WW_OK;
}
int ww_get_logmel_math(void) { return logmel_math_mode(); }
int ww_set_logmel_math_thread(int mode) {
    if (mode != WW_MATH_INHERIT && mode != WW_LOGMEL_MATH_F32 && mode != WW_LOGMEL_MATH_F64 && mode != WW_LOGMEL_MATH_AUTO)
        return fail(WW_EINVAL, "unknown log-mel math mode %d", mode);
    t_logmel_math = mode;
    return WW_OK;
}

int ww_init(void) {
    if (int rc = require_gfx950()) return rc;
    return device_tables() ? WW_OK : WW_EHIP;
}

int ww_sync_timeouts(void) {
    if (int rc = require_gfx950()) return rc;
    unsigned int c = 0;
    if (int rc = sync_timeouts(&c)) return rc;
    return int(c > 0x7fffffffu ? 0x7fffffffu : c);
}

int ww_device_info(int* n_cu, int* clock_khz, char* name, int name_len) {
    if (int rc = require_gfx950()) return rc;
    int dev = 0;
    WW_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    WW_HIP(hipGetDeviceProperties(&prop, dev));
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (clock_khz) *clock_khz = prop.clockRate;
    if (name && name_len > 0) std::snprintf(name, size_t(name_len), "%s (%s)", prop.name, prop.gcnArchName);
    return WW_OK;
}

int ww_logmel_f32(const float* pcm_dev, int64_t n_clips, int64_t clip_stride, int64_t clip_len, int normalize,
                  float* logmel_dev, ww_stream_t stream) {
    if (int rc = check_pcm(pcm_dev, n_clips, clip_stride, clip_len)) return rc;
    if (n_clips > 0 && !logmel_dev) return fail(WW_EINVAL, "null output pointer");
    if (int rc = require_gfx950()) return rc;
    return launch_logmel(pcm_dev, n_clips, clip_stride, clip_len, normalize, nullptr, 0, logmel_dev,
                         static_cast<hipStream_t>(stream));
}

int64_t ww_augment_workspace_bytes(int64_t n_clips) {
    if (n_clips < 0 || n_clips > (int64_t(1) << 24)) return fail(WW_EINVAL, "n_clips %lld out of range", (long long)n_clips);
    return augment_workspace_bytes(n_clips);
}

int ww_augment_f32(const float* pcm_dev, int64_t n_clips, int64_t clip_stride, const ww_augment_plan* plans_host,
                   float* out_dev, void* workspace_dev, ww_stream_t stream) {
    if (n_clips > (int64_t(1) << 24)) return fail(WW_EINVAL, "n_clips %lld out of range", (long long)n_clips);
    if (int rc = check_pcm(pcm_dev, n_clips, clip_stride, kClip)) return rc;
    if (n_clips == 0) return WW_OK;
    if (!plans_host || !out_dev || !workspace_dev) return fail(WW_EINVAL, "null plan / output / workspace pointer");
    if ((reinterpret_cast<uintptr_t>(out_dev) & 15) || (reinterpret_cast<uintptr_t>(workspace_dev) & 255))
        return fail(WW_EINVAL, "out_dev must be 16-byte and workspace_dev 256-byte aligned");
    if (int rc = require_gfx950()) return rc;
    return launch_augment(pcm_dev, n_clips, clip_stride, plans_host, out_dev, kClip, workspace_dev, static_cast<hipStream_t>(stream));
}

int64_t ww_augment_record_bytes(void) { return augment_record_bytes(); }

int ww_augment_plans_prepare(const ww_augment_plan* plans_host, int64_t n_clips, void* records_host) {
    if (n_clips < 0 || n_clips > (int64_t(1) << 24)) return fail(WW_EINVAL, "n_clips %lld out of range", (long long)n_clips);
    if (n_clips == 0) return WW_OK;
    if (!plans_host || !records_host) return fail(WW_EINVAL, "null plan / record pointer");
    return augment_prepare(plans_host, n_clips, records_host, nullptr, nullptr);
}

int ww_augment_records_f32(const float* pcm_dev, int64_t n_clips, int64_t clip_stride, const void* records_dev, float* out_dev,
                           void* workspace_dev, ww_stream_t stream) {
    if (n_clips > (int64_t(1) << 24)) return fail(WW_EINVAL, "n_clips %lld out of range", (long long)n_clips);
    if (int rc = check_pcm(pcm_dev, n_clips, clip_stride, kClip)) return rc;
    if (n_clips == 0) return WW_OK;
    if (!records_dev || !out_dev || !workspace_dev) return fail(WW_EINVAL, "null record / output / workspace pointer");
    if ((reinterpret_cast<uintptr_t>(out_dev) & 15) || (reinterpret_cast<uintptr_t>(workspace_dev) & 255) || (reinterpret_cast<uintptr_t>(records_dev) & 7))
        return fail(WW_EINVAL, "out_dev must be 16-byte, workspace_dev 256-byte and records_dev 8-byte aligned");
    if (int rc = require_gfx950()) return rc;
    return launch_augment_records(pcm_dev, n_clips, clip_stride, records_dev, true, true, out_dev, kClip, workspace_dev, static_cast<hipStream_t>(stream));
}

int64_t ww_cnn_scratch_bytes(int64_t n, int32_t n_conv) { return cnn_scratch_bytes(n, n_conv); }

int ww_cnn_pool_f32(const float* mel_dev, int64_t n, int32_t width, const float* packed_dev, int32_t n_conv,
                    void* scratch_dev, float* pooled_dev, ww_stream_t stream) {
    if (int rc = check_model(n, width, packed_dev, n_conv)) return rc;
    if (n > 0 && (!mel_dev || !pooled_dev)) return fail(WW_EINVAL, "null tensor pointer");
    if (int rc = require_gfx950()) return rc;
    return launch_cnn_pool(mel_dev, n, width, packed_dev, n_conv, scratch_dev, pooled_dev, static_cast<hipStream_t>(stream));
}

int ww_lstm_fc_f32(const float* pooled_dev, int64_t n, const float* packed_dev, int32_t n_conv, float* logits_dev,
                   ww_stream_t stream) {
    if (int rc = check_model(n, 1, packed_dev, n_conv)) return rc;
    if (n > 0 && (!pooled_dev || !logits_dev)) return fail(WW_EINVAL, "null tensor pointer");
    if (int rc = require_gfx950()) return rc;
    return launch_lstm_fc(pooled_dev, n, packed_dev, n_conv, logits_dev, nullptr, static_cast<hipStream_t>(stream));
}

int64_t ww_workspace_bytes(int64_t n, int32_t n_conv) {
    if (n < 0 || (n_conv != 2 && n_conv != 3)) return fail(WW_EINVAL, "bad workspace query");
    return carve(nullptr, n, n_conv).total;
}

int ww_model_forward_f32(const float* mel_dev, int64_t n, int32_t width, const float* packed_dev, int32_t n_conv,
                         void* workspace_dev, float* logits_dev, ww_stream_t stream) {
    if (int rc = check_model(n, width, packed_dev, n_conv)) return rc;
    if (n == 0) return WW_OK;
    if (!mel_dev || !logits_dev || !workspace_dev) return fail(WW_EINVAL, "null tensor / workspace pointer");
    if (int rc = require_gfx950()) return rc;
    Workspace w = carve(workspace_dev, n, n_conv);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (int rc = launch_cnn_pool(mel_dev, n, width, packed_dev, n_conv, w.scratch, w.pooled, st)) return rc;
    return launch_lstm_fc(w.pooled, n, packed_dev, n_conv, logits_dev, nullptr, st);
}

int ww_forward_pcm_f32(const float* pcm_dev, int64_t n_clips, int64_t clip_stride, int64_t clip_len, int normalize,
                       const float* packed_dev, int32_t n_conv, void* workspace_dev, float* logits_dev,
                       ww_stream_t stream) {
    if (int rc = check_pcm(pcm_dev, n_clips, clip_stride, clip_len)) return rc;
    if (int rc = check_model(n_clips, kFrames, packed_dev, n_conv)) return rc;
    if (n_clips == 0) return WW_OK;
    if (!logits_dev || !workspace_dev) return fail(WW_EINVAL, "null logits / workspace pointer");
    if (int rc = require_gfx950()) return rc;
    Workspace w = carve(workspace_dev, n_clips, n_conv);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (int rc = launch_logmel(pcm_dev, n_clips, clip_stride, clip_len, normalize, nullptr, 0, w.logmel, st)) return rc;
    if (int rc = launch_cnn_pool(w.logmel, n_clips, kFrames, packed_dev, n_conv, w.scratch, w.pooled, st)) return rc;
    return launch_lstm_fc(w.pooled, n_clips, packed_dev, n_conv, logits_dev, nullptr, st);
}

static int check_train(const float* mel, int64_t n, int32_t width, const ww_train_params* p, const void* ws) {
    if (n < 1 || n > (int64_t(1) << 24)) return fail(WW_EINVAL, "training batch %lld out of range", (long long)n);
    if (width < 1 || width > WW_MAX_WIDTH) return fail(WW_EUNSUPPORTED, "mel width %d: the conv kernels take 1..%d frames", width, WW_MAX_WIDTH);
    if (!p || !mel || !ws) return fail(WW_EINVAL, "null argument");
    if ((p->n_conv != 2 && p->n_conv != 3) || p->hidden != kHidden) return fail(WW_EUNSUPPORTED, "the training step is built for 2 or 3 convs and hidden 256");
    for (int i = 0; i < p->n_conv; ++i)
        if (!p->conv_weight[i] || !p->conv_bias[i]) return fail(WW_EINVAL, "null parameter pointer");
    for (int i = 0; i < 2; ++i)
        if (!p->lstm_weight_ih[i] || !p->lstm_bias_ih[i] || !p->lstm_bias_hh[i]) return fail(WW_EINVAL, "null parameter pointer");
    if (!p->fc_weight || !p->fc_bias) return fail(WW_EINVAL, "null parameter pointer");
    if (reinterpret_cast<uintptr_t>(ws) & 255) return fail(WW_EINVAL, "workspace must be 256-byte aligned");
    return WW_OK;
}

// WW_TRAIN_MATH_DEFAULT -> the process-wide setting, read ONCE per call (the caller should then hand the resolved value to the other half)
static int resolve_train_math(int32_t mode, int* out) {
    if (mode == WW_TRAIN_MATH_DEFAULT) mode = g_train_math;
    if (mode != WW_TRAIN_MATH_F32 && mode != WW_TRAIN_MATH_F16X3)
        return fail(WW_EINVAL, "train math %d: expected WW_TRAIN_MATH_F32, WW_TRAIN_MATH_F16X3 or WW_TRAIN_MATH_DEFAULT", mode);
    *out = mode;
    return WW_OK;
}

int64_t ww_train_workspace_bytes(int64_t n, int32_t n_conv, int32_t train_math) {
    if (n < 0 || n > (int64_t(1) << 24) || (n_conv != 2 && n_conv != 3)) return fail(WW_EINVAL, "bad training workspace query");
    int mode;
    if (int rc = resolve_train_math(train_math, &mode)) return rc;
    return train_workspace_bytes(n, n_conv, mode);
}

int ww_train_forward_f32(const float* mel_dev, int64_t n, int32_t width, const ww_train_params* params, float p_lstm, float p_fc,
                         uint64_t seed, int32_t train_math, void* workspace_dev, int64_t workspace_bytes, float* logits_dev, ww_stream_t stream) {
    int mode;
    if (int rc = resolve_train_math(train_math, &mode)) return rc;
    if (int rc = check_train(mel_dev, n, width, params, workspace_dev)) return rc;
    if (workspace_bytes < train_workspace_bytes(n, params->n_conv, mode))
        return fail(WW_EINVAL, "training workspace of %lld bytes: %lld clips of the %d-conv model under train math %d need %lld",
                    (long long)workspace_bytes, (long long)n, params->n_conv, mode, (long long)train_workspace_bytes(n, params->n_conv, mode));
    if (!logits_dev) return fail(WW_EINVAL, "null logits pointer");
    if (!(p_lstm >= 0.f && p_lstm < 1.f) || !(p_fc >= 0.f && p_fc < 1.f)) return fail(WW_EINVAL, "drop probabilities must lie in [0, 1)");
    if (int rc = require_gfx950()) return rc;
    return train_forward(mel_dev, n, width, params, p_lstm, p_fc, seed, mode, workspace_dev, workspace_bytes, logits_dev, static_cast<hipStream_t>(stream));
}

int ww_train_masks(const void* workspace_dev, int64_t n, int32_t n_conv, float* mask0_dev, float* mask1_dev, ww_stream_t stream) {
    if (!workspace_dev || !mask0_dev || !mask1_dev || n < 1 || (n_conv != 2 && n_conv != 3)) return fail(WW_EINVAL, "bad argument");
    if (int rc = require_gfx950()) return rc;
    return train_masks(workspace_dev, n, n_conv, mask0_dev, mask1_dev, static_cast<hipStream_t>(stream));
}

int ww_train_packed_image(const void* workspace_dev, int64_t n, int32_t n_conv, float* img_dev, ww_stream_t stream) {
    if (!workspace_dev || !img_dev || n < 1 || (n_conv != 2 && n_conv != 3)) return fail(WW_EINVAL, "ww_train_packed_image: bad arguments");
    return train_packed_image(workspace_dev, n, n_conv, img_dev, static_cast<hipStream_t>(stream));
}

int ww_train_bit_images(const void* workspace_dev, int64_t n, int32_t n_conv, uint8_t* mask_last_dev, uint32_t* sign1_dev, ww_stream_t stream) {
    if (!workspace_dev || !mask_last_dev || !sign1_dev || n < 1 || (n_conv != 2 && n_conv != 3)) return fail(WW_EINVAL, "ww_train_bit_images: bad arguments");
    return train_bit_images(workspace_dev, n, n_conv, mask_last_dev, sign1_dev, static_cast<hipStream_t>(stream));
}

int ww_train_backward_f32(const float* mel_dev, int64_t n, int32_t width, const ww_train_params* params, const float* dlogits_dev,
                          int32_t train_math, void* workspace_dev, int64_t workspace_bytes, const ww_train_grads* grads, ww_stream_t stream) {
    int mode;
    if (int rc = resolve_train_math(train_math, &mode)) return rc;
    if (int rc = check_train(mel_dev, n, width, params, workspace_dev)) return rc;
    if (workspace_bytes < train_workspace_bytes(n, params->n_conv, mode))
        return fail(WW_EINVAL, "training workspace of %lld bytes: %lld clips of the %d-conv model under train math %d need %lld",
                    (long long)workspace_bytes, (long long)n, params->n_conv, mode, (long long)train_workspace_bytes(n, params->n_conv, mode));
    if (!dlogits_dev || !grads) return fail(WW_EINVAL, "null argument");
    for (int i = 0; i < params->n_conv; ++i)
        if (!grads->conv_weight[i] || !grads->conv_bias[i]) return fail(WW_EINVAL, "null gradient pointer");
    for (int i = 0; i < 2; ++i)
        if (!grads->lstm_weight_ih[i] || !grads->lstm_bias[i]) return fail(WW_EINVAL, "null gradient pointer");
    if (!grads->fc_weight || !grads->fc_bias) return fail(WW_EINVAL, "null gradient pointer");
    if (int rc = require_gfx950()) return rc;
    return train_backward(mel_dev, n, width, params, dlogits_dev, mode, workspace_dev, workspace_bytes, grads, static_cast<hipStream_t>(stream));
}

int ww_streamer_create(int32_t n_mics, int32_t hop_samples, const float* packed_dev, int32_t n_conv,
                       ww_stream_t stream, ww_streamer** out) {
    if (!out) return fail(WW_EINVAL, "null out pointer");
    *out = nullptr;
    if (n_mics < 1 || n_mics > (1 << 20)) return fail(WW_EINVAL, "n_mics %d out of range", n_mics);
    if (hop_samples < 4 || hop_samples > kClip || (hop_samples & 3) || (kClip % hop_samples))
        return fail(WW_EINVAL, "hop_samples %d: must be a multiple of 4 that divides %d", hop_samples, kClip);
    if (int rc = check_model(n_mics, kFrames, packed_dev, n_conv)) return rc;
    if (!device_tables()) return WW_EHIP;   // also the gfx950 check; must precede graph capture
    ww_streamer* s = new (std::nothrow) ww_streamer();
    if (!s) return fail(WW_EHIP, "out of host memory");
    s->n_mics = n_mics; s->hop = hop_samples; s->n_conv = n_conv; s->packed = packed_dev;
    s->stream = static_cast<hipStream_t>(stream);
    const int64_t ws = carve(nullptr, n_mics, n_conv).total;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&s->ring), sizeof(float) * int64_t(n_mics) * kClip);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->pos), 16);
    if (e == hipSuccess) e = hipMalloc(&s->workspace, size_t(ws));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->own_logits), sizeof(float) * 2 * n_mics);
    if (e == hipSuccess) e = hipMemsetAsync(s->ring, 0, sizeof(float) * int64_t(n_mics) * kClip, s->stream);
    if (e == hipSuccess) e = hipMemsetAsync(s->pos, 0, 16, s->stream);
    if (e != hipSuccess) {
        ww_streamer_destroy(s);
        return fail(WW_EHIP, "streamer allocation failed: %s", hipGetErrorString(e));
    }
    *out = s;
    return WW_OK;
}

int ww_streamer_step(ww_streamer* s, const float* hop_dev, float* prob_dev, float* logits_dev) {
    if (!s || !hop_dev || !prob_dev) return fail(WW_EINVAL, "null argument");
    if (reinterpret_cast<uintptr_t>(hop_dev) & 3) return fail(WW_EINVAL, "hop_dev must be float-aligned");
    float* lg = logits_dev ? logits_dev : s->own_logits;
    if (!s->exec || s->cap_hop != hop_dev || s->cap_prob != prob_dev || s->cap_logits != lg) {
        // (re)capture: the I/O pointers are baked into the graph's kernel nodes
        if (s->exec) { (void)hipGraphExecDestroy(s->exec); s->exec = nullptr; }
        if (s->graph) { (void)hipGraphDestroy(s->graph); s->graph = nullptr; }
        WW_HIP(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
        const int rc = streamer_enqueue(s, hop_dev, prob_dev, lg);
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(s->stream, &g);
        if (rc != WW_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
        if (e != hipSuccess) return fail(WW_EHIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
        s->graph = g;
        WW_HIP(hipGraphInstantiate(&s->exec, s->graph, nullptr, nullptr, 0));
        s->cap_hop = hop_dev; s->cap_prob = prob_dev; s->cap_logits = lg;
    }
    WW_HIP(hipGraphLaunch(s->exec, s->stream));
    return WW_OK;
}

int ww_streamer_window(ww_streamer* s, float* window_dev) {
    if (!s || !window_dev) return fail(WW_EINVAL, "null argument");
    hipLaunchKernelGGL(ring_unroll_kernel, dim3(1024), dim3(256), 0, s->stream, s->ring, s->pos, window_dev, s->n_mics);
    WW_HIP(hipGetLastError());
    return WW_OK;
}

int ww_streamer_destroy(ww_streamer* s) {
    if (!s) return WW_OK;
    if (s->exec) (void)hipGraphExecDestroy(s->exec);
    if (s->graph) (void)hipGraphDestroy(s->graph);
    if (s->ring) (void)hipFree(s->ring);
    if (s->pos) (void)hipFree(s->pos);
    if (s->workspace) (void)hipFree(s->workspace);
    if (s->own_logits) (void)hipFree(s->own_logits);
    delete s;
    return WW_OK;
}

}  // extern "C"
