// Internal definitions shared by the translation units of libwakeword_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wakeword_amd.h"

namespace ww {

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
int fail(int code, const char* fmt, ...);
#define WW_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess)                                                               \
            return ::ww::fail(WW_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                              __FILE__, __LINE__);                                           \
    } while (0)

// ---------------------------------------------------------------------------------------------
// K1 tables (one image per device, built on the host in double precision)
// ---------------------------------------------------------------------------------------------
constexpr int kClip = WW_CLIP_SAMPLES;
constexpr int kNfft = WW_N_FFT;
constexpr int kHop = WW_HOP;
constexpr int kMels = WW_N_MELS;
constexpr int kFrames = WW_N_FRAMES;
constexpr int kBins = WW_N_BINS;

// The mel matrix (2004 non-zeros, <=2 filters per bin, every filter one contiguous run of 9..75 bins) is cut
// into PIECES: (filter f, aligned window of 8 bins [8m, 8m+8)) with 8 weights (zero outside the filter).
// 317 real pieces -> 5 rounds of 64 lane-slots; every lane of the mel stage runs the same trip count and reads
// its 8 power bins with two aligned ds_read_b128.  Slot order = sorted by window, dealt out so that the 16 lanes
// of one ds_read_b128 hardware group read neighbouring windows (conflict-free, mostly broadcast).
constexpr int kPieceLen = 8;
constexpr int kPieceRounds = 5;             // pieces per lane-slot
constexpr int kPieceSlots = 64;             // lane-slots
constexpr int kPieces = kPieceRounds * kPieceSlots;  // 320 >= number of real pieces (checked on host)

struct LogmelTables {
    float window[kNfft];          // periodic Hann
    float2 tw1[7][128];           // W_1024^(n' * k1),  k1 = 1..7, n' = 0..127
    float2 tw2[7][16];            // W_128^(n'' * k2),  k2 = 1..7, n'' = 0..15
    float2 twp[512];              // W_2048^k, k = 0..511, except twp[0] = W_2048^512 (lane 0 takes bin 512)
    float piece_w[2][kPieces][4]; // [half][slot index c*64+lane][4]: weights of bins 8m+4*half .. +3
    int32_t piece_info[kPieces];  // (8m) | (output position << 16); output positions are filter-major
    int32_t filt_p0[kMels];       // first output position of filter f
    int32_t filt_cnt[kMels];      // number of pieces of filter f
    // augmentation kernels (ww_augment.hip)
    float2 twr[1024];             // W_2048^k, k = 0..1023
    float kaiser_best[32772];     // resampler half-window: 64 zero crossings x 512 + 1 entries (+3 pad)
    // float64 forms for the precise log-mel kernel (logmel64_kernel): the same entries, not rounded to float
    double window_d[kNfft];
    double2 tw1_d[7][128];
    double2 tw2_d[7][16];
    double2 twp_d[512];
    // per mel band: the largest weight of the triangle and its reciprocal (the auto mode's rounding-floor test)
    float band_wmax[kMels];
    float band_bins[kMels];
};

void build_mel_filterbank(float* out /*[80][1025]*/);
void build_hann(double* out /*[2048]*/);
int build_logmel_tables(LogmelTables* t);   // host; returns number of real pieces or <0
const LogmelTables* device_tables();        // lazily uploaded for the current device (nullptr on error)

// ---------------------------------------------------------------------------------------------
// packed weight image (floats).  Offsets for n_conv = 2 | 3.
// ---------------------------------------------------------------------------------------------
constexpr int kHidden = WW_HIDDEN;
constexpr int kGateCols = 3 * kHidden;     // i, g, o (the forget gate is dead with zero state)

struct PackedLayout {
    int n_conv;
    int c_last;        // 64 | 128
    int64_t conv1_w;   // [32][9]
    int64_t conv1_b;   // [32]
    int64_t conv2_w;   // [2 ntile][144][64 lanes]  B-operand order
    int64_t conv2_b;   // [64]
    int64_t conv3_w;   // [4 ntile][288][64 lanes]  (n_conv 3)
    int64_t conv3_b;   // [128]
    int64_t l0_w;      // [c_last][768]  k-major, column = (hb*3 + gate)*32 + u
    int64_t l0_b;      // [768]
    int64_t l1_w;      // [256][768]
    int64_t l1_b;      // [768]
    int64_t fc_w;      // [2][256]
    int64_t fc_b;      // [2] (+2 pad)
    // split-precision images for the f16x3 kernels: W * 2^S = hi + lo (two f16) in MFMA operand order
    int64_t conv2_hs;  // [64]: 2^-S per output channel (descale applied to the f32 accumulator)
    int64_t conv1_h;   // conv1 as a 32x32x16 f16 MFMA A operand: [hi,lo][64 lanes][4 dwords]; k = tap 0..8, rest 0
    int64_t conv1_hs;  // [4]: 2^-S of conv1 (one scale for the tensor), 0, 0, 0
    int64_t conv3_h;   // (n_conv 3) conv3 split-precision B operands for 16x16x32: [8 ntile][18 kstep = (cb*3+dx)*3+dy][hi,lo][64][4]
    int64_t conv3_hs;  // [128]: 2^-S per output channel
    int64_t conv2_h16; // same weights for v_mfma_f32_16x16x32_f16: [4 ntile][9 kstep = dx*3+dy][hi,lo][64 lanes][4 dwords]
    int64_t l0_h;      // W_ih l0 split for v_mfma_f32_16x16x32_f16: [K/32][48 ntile][hi,lo][64 lanes][4 dwords], columns as l0_w
    int64_t l1_h;      // same for layer 1 (K = 256)
    int64_t lstm_hs;   // [2][768]: 2^-S per packed gate column, layer 0 then layer 1
    int64_t conv2_hw;  // conv2 as 1-D Winograd F(2,3) along rows, split precision: [4 ntile][12 kstep = xi*3+dx][hi,lo][64 lanes][4 dwords]
    int64_t conv2_hws; // [64]: 2^-S per output channel of the transformed weights
    int64_t conv3_hw;  // (n_conv 3) conv3 in the same Winograd form: [8 ntile][24 kstep = (xi*3+dx)*2 + cb][hi,lo][64 lanes][4 dwords]
    int64_t conv3_hws; // [128]
    int64_t conv2_hx;  // conv2 in the same Winograd form as B operands of v_mfma_f32_32x32x16_f16 (cnn2x_kernel): [2 ntile][4 xi][6 step = dx*2 + c][hi,lo][64 lanes][4 dwords]; scales = conv2_hws
    int64_t range;     // [8]: l1 bound of conv1 (max over channels of sum |w|), max |b1|, the same for conv2, 0...
    int64_t total;
};
PackedLayout packed_layout(int n_conv);

// ---------------------------------------------------------------------------------------------
// kernel launchers (defined in the .hip files)
// ---------------------------------------------------------------------------------------------
int launch_logmel(const float* pcm, int64_t n_clips, int64_t clip_stride, int64_t clip_len, int normalize,
                  const int32_t* ring_pos /*nullable: streaming ring start per launch*/, int64_t ring_len,
                  float* logmel, hipStream_t stream);
// log-mel arithmetic: 0 = f32 FFT, 1 = f64 FFT (what the reference's numpy.fft.rfft is), 2 = auto (f32, then the clips whose
// quiet bands sit on the f32 FFT's rounding floor are redone in f64)
int logmel_math_mode();
void set_logmel_math_mode(int mode);
int launch_augment(const float* pcm, int64_t n, int64_t stride, const ww_augment_plan* plans_host, float* out,
                   int64_t out_stride, void* workspace, hipStream_t stream);
int64_t augment_workspace_bytes(int64_t n);
int augment_prepare(const ww_augment_plan* plans_host, int64_t n, void* records_host, int* any_pitch_out, int* any_stretch_out);
int64_t augment_record_bytes();
int launch_augment_records(const float* pcm, int64_t n, int64_t stride, const void* records_dev, bool any_pitch, bool any_stretch, float* out,
                           int64_t out_stride, void* workspace, hipStream_t stream);
void build_kaiser_best(float* out /*[32769]*/);
int sync_timeouts(unsigned int* count);   // bounded LDS-counter waits that expired (must be 0)
int launch_cnn_pool(const float* mel, int64_t n, int width, const float* packed, int n_conv, void* scratch,
                    float* pooled, hipStream_t stream);
// conv math: 0 = exact f32 MFMA (v_mfma_f32_32x32x2_f32), 1 = f16x3 split (3 x v_mfma_f32_16x16x32_f16 per product block; conv2 of
// the 2-conv model as 1-D Winograd), 2 = f16x3 split with every conv in its direct form
int conv_math_mode();
void set_conv_math_mode(int mode);
int launch_lstm_fc(const float* pooled, int64_t n, const float* packed, int n_conv, float* logits,
                   float* prob /*nullable*/, hipStream_t stream);

// training step (ww_train.hip)
int64_t train_workspace_bytes(int64_t n, int n_conv, int mode);
int train_forward(const float* mel, int64_t n, int width, const ww_train_params* p, float p_lstm, float p_fc, uint64_t seed, int mode,
                  void* workspace, int64_t workspace_bytes, float* logits, hipStream_t st);
int train_masks(const void* workspace, int64_t n, int n_conv, float* mask0, float* mask1, hipStream_t st);
int train_packed_image(const void* workspace, int64_t n, int n_conv, float* img, hipStream_t st);
int train_bit_images(const void* workspace, int64_t n, int n_conv, uint8_t* mask_last, uint32_t* sign1, hipStream_t st);
int launch_decode_mask_image(const uint32_t* img, int64_t n, int C, uint8_t* out, hipStream_t st);
int train_backward(const float* mel, int64_t n, int width, const ww_train_params* p, const float* dlogits, int mode, void* workspace,
                   int64_t workspace_bytes, const ww_train_grads* g, hipStream_t st);

int train_math_mode();   // 0 exact fp32, 1 split-precision conv backward where a kernel exists (ww_train_h.hip)
int launch_pack_conv_h_dev(const ww_train_params* p, float* img, hipStream_t st);
int launch_cnn3w_pool_bits(const float* mel, int64_t n, int width, const float* packed, float* mid2, float* apow2, float* pooled,
                           uint32_t* bits3, uint32_t* bits1, hipStream_t stream);
int launch_cnn2w_pool_bits(const float* mel, int64_t n, int width, const float* packed, float* pooled, uint32_t* bits, uint32_t* bits1,
                           hipStream_t stream);
int launch_conv2_wgrad_h(const float* mel, const uint32_t* maskbits, const float* gp, int64_t n, int width, const float* packed,
                         float* partial, int grid, hipStream_t st);

int launch_conv3_wgrad_h(const float* act2, const float* apow2, const uint32_t* maskbits, const float* gp, int64_t n, float* partial,
                         float* dw, float* db, int grid, hipStream_t st);
int64_t dgrad_h_scratch_floats(int64_t n, int n_conv);
float* dgrad_h_scratch2(float* scratch, int64_t n);
float* dgrad_h_dzs(float* scratch, int64_t n);
int launch_conv2_wgrad_h_dense(const float* mel, const float* dz2h, const float* dzs, int64_t n, int width, const float* packed, float* partial,
                               int grid, hipStream_t st);
int launch_conv2_dgrad_h_dense(const float* mel, const float* dz2h, const float* dzs, const uint32_t* bits1, const float* w2, float* scratch2,
                               int64_t n, int width, float* partial, int grid, hipStream_t st);
int launch_gp_max(const float* dpooled, float s, int64_t n, int n_conv, float* gp, float* scratch, hipStream_t st);
int launch_conv3_dgrad_h(const float* act2, const uint32_t* maskbits, const float* gp, const float* w3, float* scratch, int64_t n, float* dz2h,
                         float* dzs, int grid, hipStream_t st);
int launch_conv2_dgrad_h(const float* mel, const uint32_t* maskbits, const uint32_t* bits1, const float* gp, const float* w2, float* scratch,
                         int64_t n, int width, float* partial, int grid, hipStream_t st);

int require_gfx950();
int device_cu_count();   // CUs of the current device (256 on MI355X); cached

}  // namespace ww
