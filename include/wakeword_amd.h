/*
 * wakeword_amd.h -- C ABI of libwakeword_amd.so (MI355X / gfx950 native).
 *
 * The reference (sarpel/wakeword-jupyterlab) is pure Python and has no FFI: its boundary for the
 * hot path is three Python call signatures.  Each entry point below names the reference code it
 * replaces (paths relative to the reference repository).  The Python host layer
 * (the .py files of wakeword-jupyterlab_amd) binds these with ctypes and re-exposes the reference signatures;
 * INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every `*_dev` pointer is a DEVICE pointer (HBM) on the current HIP device; every `*_host`
 *     pointer is ordinary host memory.  No torch / C++ types cross this boundary.
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).  All launch
 *     functions are asynchronous on that stream, allocate nothing, never synchronise and are
 *     therefore capturable into a hipGraph once ww_init() has run on the device (exception, stated
 *     at the function: ww_augment_f32 reads its plans from host memory and is not capturable; its two-call form
 *     ww_augment_plans_prepare + ww_augment_records_f32 is).
 *   - return value: WW_OK (0) or a negative WW_E* code; ww_last_error() gives the message of the
 *     calling thread's most recent failure.  Nothing falls back to a CPU path: without a usable
 *     gfx950 device every launch function fails with WW_ENODEVICE.
 *   - all tensors are float32, C-contiguous in the stated shape unless a stride is a parameter.
 */
#ifndef WAKEWORD_AMD_H
#define WAKEWORD_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WW_ABI_VERSION 4 /* 2: ww_set_logmel_math, ww_train_*, third conv-math mode; 3: ww_set_train_math (round 2); 4: ww_wav_reader_*, ww_read_wav_batch_host (round 3) */

#if defined(WW_BUILD)
#define WW_API __attribute__((visibility("default")))
#else
#define WW_API
#endif

#define WW_OK 0
#define WW_EINVAL (-1)    /* bad shape / alignment / argument */
#define WW_ENODEVICE (-2) /* no HIP device, or not gfx950 */
#define WW_EHIP (-3)      /* a HIP runtime call failed */
#define WW_EUNSUPPORTED (-4)
#define WW_ENOSPACE (-5)  /* a caller-sized buffer (the WAV reader's staging) is too small for this request */

/* AudioConfig, wakeword_training_script.py:29-37 (dup wakeword_training.ipynb cell 3);
 * Config, wakeword_training/train_wakeword.py:16-25; ModelConfig, wakeword_training_script.py:39-43 */
#define WW_SAMPLE_RATE 16000
#define WW_CLIP_SAMPLES 16000
#define WW_N_FFT 2048
#define WW_HOP 512
#define WW_N_MELS 80
#define WW_N_FRAMES 32 /* 1 + 16000/512, librosa center=True */
#define WW_N_BINS 1025
#define WW_HIDDEN 256
#define WW_N_CLASSES 2
#define WW_MAX_WIDTH 32 /* widest mel image (frames) the conv kernels take: T in [1, 32] */

typedef void* ww_stream_t;

/* ---- library / device ------------------------------------------------------------------- */
WW_API int ww_abi_version(void);
WW_API const char* ww_last_error(void);
/* Build and upload the front-end tables (window, twiddles, sparse mel pieces) for the current
 * device.  Idempotent.  Must have run once on a device before any launch below is captured into
 * a hipGraph (it allocates and synchronises); the launch functions call it lazily otherwise. */
WW_API int ww_init(void);
/* Device facts used by bench.py for the roofline denominator: number of CUs, max clock (kHz). */
WW_API int ww_device_info(int* n_cu, int* clock_khz, char* name, int name_len);
/* Health check (synchronises the device): how many bounded in-kernel waits of the conv kernel's producer/consumer
 * protocol have expired since the library was loaded.  Always 0 unless the protocol is broken; negative = error code.
 * A workgroup whose wait expired overwrites every output it produced with NaN before it exits, so a broken launch
 * cannot be consumed silently even without this call. */
WW_API int ww_sync_timeouts(void);

/* Arithmetic of the implicit GEMMs (conv1/conv2/conv3 and the LSTM gate GEMMs: all but ~1 % of the model's flops);
 * process-wide, default F16X3.
 *   WW_CONV_MATH_F32    v_mfma_f32_32x32x2_f32 / 16x16x4_f32: exact fp32 products and accumulation (an fmaf chain)
 *   WW_CONV_MATH_F16X3  each fp32 operand carried as two f16 halves (22 significant bits), every product block as
 *                       three f16 MFMAs (v_mfma_f32_16x16x32_f16) with fp32 accumulation: ~2^-21 relative error per
 *                       product (PyTorch's own default for convolutions on the reference's GPU is TF32, 2^-11), 3/16
 *                       of the matrix-core cycles.  Holds for any finite weights and inputs: weights carry a
 *                       power-of-two scale per output channel, inputs and activations one exponent per clip chosen
 *                       from the clip's max |x| and the layers' l1 bounds, so no f16 half overflows or goes subnormal. */
#define WW_CONV_MATH_F32 0
#define WW_CONV_MATH_F16X3 1
/*   WW_CONV_MATH_F16X3_DIRECT  F16X3 with every convolution in its direct (implicit-GEMM) form.  Under F16X3 conv2 of BOTH models and
 *                       conv3 of the 3-conv model run as a one-dimensional Winograd F(2,3) along the image rows (1.5x fewer
 *                       matrix instructions; the transforms are fp32 adds on the activations and exact-in-double combinations
 *                       of the weights); under F16X3_DIRECT every convolution of both models is direct. */
#define WW_CONV_MATH_F16X3_DIRECT 2
WW_API int ww_set_conv_math(int mode);
WW_API int ww_get_conv_math(void);   /* what a launch from the calling thread would use now (its override, else the process default) */
/* Per-thread override of the process-wide setting: launches issued by the CALLING THREAD use `mode` until it is cleared with
 * WW_MATH_INHERIT; other threads are unaffected.  A launch reads its mode once.  (ww_set_logmel_math_thread: the same for K1.) */
#define WW_MATH_INHERIT (-1)
WW_API int ww_set_conv_math_thread(int mode);

/* Arithmetic of the log-mel front end (K1), process-wide, default AUTO.  The reference's STFT is float64 (librosa forms
 * window * frame in float64 and numpy.fft.rfft runs in double; wakeword_training_script.py:89-98), then everything is
 * float32.  A float32 FFT leaves a rounding floor ~165 dB under a frame's energy: mel bands of noise-free signals that lie
 * 60-80 dB under the clip's peak come out up to 3.4e-4 dB off (clips with a broadband floor above about -60 dB, like the
 * benchmark's sine + noise clips, are unaffected: <= 1.5e-5 dB).
 *   WW_LOGMEL_MATH_F32   float32 FFT for every clip (the throughput kernel)
 *   WW_LOGMEL_MATH_F64   window product, FFT and real-input split in float64 for every clip (~4x the time)
 *   WW_LOGMEL_MATH_AUTO  the float32 kernel, which marks the clips that have a live (unclamped) mel band on its rounding
 *                        floor; a second launch redoes exactly those clips in float64.  MEASURED, not proven: <= 1.5e-5 dB
 *                        against the float64 reference on every noise-free test signal of tests/test_gpu_parity.py (the marking
 *                        threshold, band/frame energy ratio 1e-5, sits 2.7x above the boundary 10^-5.43 found on that sample,
 *                        profiles/r02_diag_floor.log).  Cost: one near-empty launch when no clip is marked; a marked clip costs
 *                        the float64 kernel (~3x): clean-speech / digitally silent data pays up to K1 x3 (bench line:
 *                        stages.K1_logmel.noise_free_batch). */
#define WW_LOGMEL_MATH_F32 0
#define WW_LOGMEL_MATH_F64 1
#define WW_LOGMEL_MATH_AUTO 2
WW_API int ww_set_logmel_math(int mode);
WW_API int ww_get_logmel_math(void);
WW_API int ww_set_logmel_math_thread(int mode);

/* ---- front-end tables, host side (no GPU needed; lets CPU tests check them) ------------------ */
/* librosa.filters.mel(sr=16000, n_fft=2048, n_mels=80, fmin=0, fmax=8000, htk=False,
 * norm='slaney') as used by wakeword_training_script.py:89-98 -> [80][1025] float32. */
WW_API int ww_mel_filterbank_host(float* out_host);
/* periodic Hann window of 2048 points (librosa.stft window='hann') -> [2048] float32. */
WW_API int ww_hann_window_host(float* out_host);

/* ---- K0: decode + mono + resample + normalise + crop/pad ------------------- */
/* Replaces the numeric part of AudioProcessor.load_audio = librosa.load(path, sr=16000)
 * (wakeword_training_script.py:65-71), normalize_audio (:73-76) and pad_or_truncate (:78-83): process_audio_file
 * :125-133 up to the mel call.  The host reads the file and parses the RIFF header; everything on samples runs here.
 * The resampler follows scipy.signal.resample_poly's design; librosa's (soxr_hq) differs: unpinned, see DESIGN.md. */
#define WW_FMT_S16 1
#define WW_FMT_S24 2
#define WW_FMT_S32 3
#define WW_FMT_F32 4
#define WW_FMT_U8 5
#define WW_FMT_F64 6 /* IEEE double samples (WAVE_FORMAT_IEEE_FLOAT, 64 bits): rounded to float32 as soundfile does for dtype float32 */
typedef struct ww_clip_desc {
    int64_t byte_offset;  /* start of the interleaved sample data of this file inside raw_dev */
    int64_t n_frames;     /* sample frames in the file */
    int32_t channels;
    int32_t sample_rate;
    int32_t format;       /* WW_FMT_* */
    int32_t crop_start;   /* first 16 kHz output sample of the 1 s window (host draws it: pad_or_truncate's random crop) */
    int32_t up, down, half_len, _pad; /* filled by ww_resampler_prepare */
    const void* taps_dev; /* filled by ww_resampler_prepare (NULL when the file is already 16 kHz) */
} ww_clip_desc;
/* The polyphase taps for `sample_rate` -> 16 kHz on the host (no GPU needed); returns the tap count (0 = no filter),
 * or the count needed when taps_host is NULL. */
WW_API int ww_resample_taps_host(int32_t sample_rate, float* taps_host, int32_t max_taps, int32_t* up, int32_t* down, int32_t* half_len);
/* Fill up/down/half_len/taps_dev of one descriptor; uploads the filter for this rate once per device (allocates). */
WW_API int ww_resampler_prepare(int32_t sample_rate, ww_clip_desc* desc_host);
/* raw_dev: the files' sample bytes; descs_dev: [n_clips] descriptors in device memory; pcm_out_dev [n_clips][16000]. */
WW_API int ww_decode_resample(const uint8_t* raw_dev, const ww_clip_desc* descs_dev, int64_t n_clips, int normalize,
                              float* pcm_out_dev, ww_stream_t stream);

/* ---- file-fed batches: the host half of load_audio for many files at once ------------------------------------- */
/* Replaces, for a batch of paths, the file access of AudioProcessor.load_audio = librosa.load(path, sr=16000)
 * (wakeword_training_script.py:65-71) as it is driven by WakewordDataset.__getitem__ (:204-216) under
 * DataLoader(batch_size=16, num_workers=2) (:461-463) -- the loop that bounds the reference at 453 clips/s
 * (wakeword_training.ipynb:742).  A reader owns `n_threads` host threads, `n_slots` pinned staging buffers with their
 * device twins, and a copy stream.  Per batch:
 *   ww_read_wav_batch_host  the threads open the files, walk the RIFF chunks and pread the sample bytes straight into
 *                           the slot's pinned staging; one ww_clip_desc per file is filled in (host, pinned; returned so
 *                           that the caller can set crop_start -- pad_or_truncate's random crop (:78-83) stays the
 *                           caller's draw: n_out = ceil(n_frames * up / down) is known now).  status_host[i] = 1, or a
 *                           WW_WAV_E* code for a file that could not be used (the reference prints and substitutes
 *                           zeros, :66-71, :210-211: such a file decodes to a zero clip here).  Blocks until the slot's
 *                           previous upload has left the staging buffer.
 *   ww_wav_batch_decode     H2D copy of the slot on the reader's copy stream, then K0 (ww_decode_resample) on `stream`
 *                           behind it: pcm_out_dev [n][16000].  Asynchronous; the next ww_read_wav_batch_host on ANOTHER
 *                           slot overlaps with it.  The upload of a slot waits for the K0 that last read its device twin.
 * A reader serves ONE caller at a time (its thread pool runs one batch); use one reader per consumer thread. */
#define WW_WAV_EOPEN (-1)    /* cannot open */
#define WW_WAV_ENOTRIFF (-2) /* not a RIFF/WAVE file */
#define WW_WAV_ECHUNK (-3)   /* fmt or data chunk missing / truncated */
#define WW_WAV_EFORMAT (-4)  /* encoding K0 does not take (it takes PCM u8/s16/s24/s32 and float32/float64), or an absurd rate */
#define WW_WAV_EIO (-5)      /* read error */
#define WW_WAV_ESPACE (-6)   /* the staging buffer was full (the call then returns WW_ENOSPACE with the size needed) */
typedef struct ww_wav_reader ww_wav_reader;
/* flags: WW_READER_HOST_ONLY = staging in ordinary host memory, no device twin, no GPU needed (ww_wav_batch_decode then returns
 * WW_EUNSUPPORTED): the RIFF walk and the threaded reads can be checked -- and run under a sanitizer -- on a CPU-only machine. */
#define WW_READER_HOST_ONLY 1
WW_API int ww_wav_reader_create(int32_t n_threads, int32_t n_slots, int64_t max_clips, int64_t max_raw_bytes, int32_t flags, ww_wav_reader** out);
WW_API int ww_wav_reader_destroy(ww_wav_reader* r);
/* The slot's staging buffer as the last ww_read_wav_batch_host left it (descs[i].byte_offset points into it); for tests. */
WW_API int ww_wav_reader_staging(ww_wav_reader* r, int32_t slot, const uint8_t** raw_host_out, int64_t* raw_bytes_out);
/* One file's header only (no GPU needed): returns 1 and fills n_frames / channels / sample_rate / format / up / down / half_len, with
 * byte_offset = the position of the sample data INSIDE THE FILE; or a WW_WAV_E* code. */
WW_API int ww_wav_probe_host(const char* path, ww_clip_desc* desc_host);
/* raw_bytes_out (may be NULL): sample bytes of the batch, 16-byte aligned per file; on WW_ENOSPACE the size to create a reader with. */
WW_API int ww_read_wav_batch_host(ww_wav_reader* r, const char* const* paths, int64_t n, int32_t slot, ww_clip_desc** descs_host_out,
                                  int8_t* status_host, int64_t* raw_bytes_out);
WW_API int ww_wav_batch_decode(ww_wav_reader* r, int32_t slot, int normalize, float* pcm_out_dev, ww_stream_t stream);

/* ---- KA: training-time augmentation ------------------------------ */
/* Replaces AudioProcessor.augment_audio (wakeword_training_script.py:103-123): np.roll time shift ->
 * librosa.effects.pitch_shift -> librosa.effects.time_stretch + pad_or_truncate -> additive Gaussian noise.
 * The random draws stay with the caller (the reference uses Python's `random`): one plan per clip.
 * librosa's phase vocoder (n_fft 2048, hop 512, Hann) is restated; its resampler (soxr_hq, third party)
 * is replaced by a Kaiser-windowed-sinc interpolator (resampy 'kaiser_best' design) and np.random.normal by
 * the build's counter-based generator: parity against librosa itself is unpinned (oracle/augment_oracle.py). */
typedef struct ww_augment_plan {
    int32_t shift;        /* np.roll shift in samples, any sign; 0 = no shift (:106-108) */
    int32_t crop_start;   /* pad_or_truncate's random crop start after time_stretch, in [0, round(16000/rate) - 16000] */
    double pitch_rate;    /* 2^(-n_steps/12) of pitch_shift (:110-112); 0 = off */
    double stretch_rate;  /* time_stretch rate (:114-117); 0 = off.  Both rates: 32/46 <= rate < 32 */
    float noise_sigma;    /* NOISE_FACTOR (:119-121); 0 = off */
    uint32_t noise_seed;
} ww_augment_plan;
WW_API int64_t ww_augment_workspace_bytes(int64_t n_clips);
/* pcm_dev [n_clips] rows of 16000 samples at pcm_dev + i*clip_stride (16-byte aligned, clip_stride % 4 == 0);
 * plans_host [n_clips] in HOST memory: read before the call returns (staged through pinned memory owned by the library, so
 * the call is asynchronous on `stream`; it is not graph-capturable); out_dev [n_clips][16000], may alias pcm_dev;
 * workspace_dev >= ww_augment_workspace_bytes(n_clips), 256-byte aligned. */
WW_API int ww_augment_f32(const float* pcm_dev, int64_t n_clips, int64_t clip_stride, const ww_augment_plan* plans_host,
                   float* out_dev, void* workspace_dev, ww_stream_t stream);
/* The same in two halves, for hipGraph capture: ww_augment_plans_prepare turns the plans into the fixed-size records the kernels read
 * (host arithmetic only: [n_clips] x ww_augment_record_bytes() bytes, e.g. into pinned memory), ww_augment_records_f32 is nothing but
 * kernel launches on `stream` reading the records from DEVICE memory -- capturable; both transform stages are always launched (a clip
 * whose plan switches one off is copied through it).  A captured step = {copy records host -> device, ww_augment_records_f32};
 * before every replay: ww_augment_plans_prepare into the host buffer.  Results equal ww_augment_f32's bit for bit. */
WW_API int64_t ww_augment_record_bytes(void);
WW_API int ww_augment_plans_prepare(const ww_augment_plan* plans_host, int64_t n_clips, void* records_host);
WW_API int ww_augment_records_f32(const float* pcm_dev, int64_t n_clips, int64_t clip_stride, const void* records_dev, float* out_dev,
                                  void* workspace_dev, ww_stream_t stream);
/* The resampler's half-window (32769 floats: 64 zero crossings x 512 + 1) on the host, for checking on a CPU. */
WW_API int ww_kaiser_best_host(float* out_host);

/* ---- K1: log-mel front-end --------------------------------------------------------------- */
/* Replaces AudioProcessor.normalize_audio (:73-76), the zero-pad branch of pad_or_truncate
 * (:78-83) and AudioProcessor.audio_to_mel (:85-101) =
 * librosa.feature.melspectrogram + librosa.power_to_db(ref=np.max), batched.
 *   pcm_dev      [n_clips] rows of `clip_len` valid samples, row i at pcm_dev + i*clip_stride
 *                (floats; 16-byte aligned base, clip_stride % 4 == 0 -- ignored for a single clip --, 0 < clip_len <= 16000;
 *                shorter rows are right-zero-padded like pad_or_truncate does)
 *   normalize    != 0: divide the clip by max|x| first (process_audio_file order, :131-133);
 *                a silent clip then yields NaN, as the reference does
 *   logmel_dev   [n_clips][80][32]  (== the [B,1,80,32] tensor WakewordDataset.__getitem__ :204-216
 *                returns, batched), dB in [-80, 0] with per-clip max exactly 0 */
WW_API int ww_logmel_f32(const float* pcm_dev, int64_t n_clips, int64_t clip_stride, int64_t clip_len,
                  int normalize, float* logmel_dev, ww_stream_t stream);

/* ---- weights ------------------------------------------------------------------------------ */
/* The reference state_dict (train_wakeword.py:28-36 / wakeword_training_script.py:141-165) as
 * host pointers in torch layout.  conv3_* are NULL for SimpleWakewordModel.  weight_hh_l* are not
 * part of the struct: with seq_len 1 and zero initial state they never reach the output. */
typedef struct ww_state_dict {
    int32_t n_conv;            /* 2 (SimpleWakewordModel) or 3 (WakewordModel) */
    int32_t hidden;            /* 256 */
    const float* conv_weight[3]; /* [32,1,3,3], [64,32,3,3], [128,64,3,3] */
    const float* conv_bias[3];   /* [32], [64], [128] */
    const float* lstm_weight_ih[2]; /* [4*hidden, C_last], [4*hidden, hidden]; gate order i,f,g,o */
    const float* lstm_bias_ih[2];   /* [4*hidden] */
    const float* lstm_bias_hh[2];   /* [4*hidden] */
    const float* fc_weight;         /* [2, hidden] */
    const float* fc_bias;           /* [2] */
} ww_state_dict;

/* Number of floats of the packed (kernel-layout) weight image for a model with n_conv convs. */
WW_API int64_t ww_packed_weights_floats(int32_t n_conv);
/* Pack on the host (pure CPU): MFMA B-operand order for the convs, transposed W_ih with the dead
 * forget-gate rows dropped, b_ih + b_hh pre-summed.  The caller uploads `packed_host` to HBM once
 * (model.load_state_dict / .to(device)) and passes the device copy to the functions below. */
WW_API int ww_pack_weights_host(const ww_state_dict* sd, float* packed_host);

/* ---- K2: conv stack + global average pool --------------------------------------------------- */
/* Replaces `x = F.relu(self.conv1(x)); x = F.relu(self.conv2(x)); [conv3]; x = self.pool(x)`
 * (train_wakeword.py:39-41 / wakeword_training_script.py:170-173).
 *   mel_dev    [n][80][width]   (the [B,1,80,T] model input), 1 <= width <= 32
 *   pooled_dev [n][C_last]      C_last = 64 (n_conv 2) or 128 (n_conv 3)
 *   scratch_dev  n_conv 3 only: ww_cnn_scratch_bytes(n, 3) bytes; may be NULL for n_conv 2 */
WW_API int64_t ww_cnn_scratch_bytes(int64_t n, int32_t n_conv);
WW_API int ww_cnn_pool_f32(const float* mel_dev, int64_t n, int32_t width, const float* packed_dev,
                    int32_t n_conv, void* scratch_dev, float* pooled_dev, ww_stream_t stream);

/* ---- K3: 2-layer LSTM (one step, zero state) + Linear ---------------------------------------- */
/* Replaces `lstm_out, _ = self.lstm(x.unsqueeze(1)); x = lstm_out[:, -1, :]; x = self.fc(x)`
 * (train_wakeword.py:42-48 / wakeword_training_script.py:175-182), dropout = identity (eval).
 *   pooled_dev [n][C_last] -> logits_dev [n][2] */
WW_API int ww_lstm_fc_f32(const float* pooled_dev, int64_t n, const float* packed_dev, int32_t n_conv,
                   float* logits_dev, ww_stream_t stream);

/* ---- composed entry points ----------------------------------------------------------------- */
/* model.forward(x): SimpleWakewordModel.forward train_wakeword.py:38-49 /
 * WakewordModel.forward wakeword_training_script.py:167-184.  workspace >= ww_workspace_bytes. */
WW_API int64_t ww_workspace_bytes(int64_t n, int32_t n_conv);
WW_API int ww_model_forward_f32(const float* mel_dev, int64_t n, int32_t width, const float* packed_dev,
                         int32_t n_conv, void* workspace_dev, float* logits_dev, ww_stream_t stream);
/* PCM -> logits: the eval loop body of wakeword_training.ipynb cell 17 / validate()
 * wakeword_training_script.py:269-289 with the Dataset's mel path folded in (K1 -> K2 -> K3). */
WW_API int ww_forward_pcm_f32(const float* pcm_dev, int64_t n_clips, int64_t clip_stride, int64_t clip_len,
                       int normalize, const float* packed_dev, int32_t n_conv, void* workspace_dev,
                       float* logits_dev, ww_stream_t stream);

/* ---- training step ----------------------------------------------------- */
/* Replaces, for one batch, `output = model(data)` in train mode and `loss.backward()` of the reference's training loops
 * (wakeword_training/train_wakeword.py:109-115; WakewordTrainer.train_epoch, wakeword_training_script.py:241-267) for
 * SimpleWakewordModel and the 3-conv WakewordModel: the forward with nn.LSTM's inter-layer dropout and nn.Dropout before fc (train_wakeword.py:34-35,
 * 46-47), and d loss / d parameter given d loss / d logits.  CrossEntropyLoss and the optimiser stay with the caller.
 * Arithmetic: ww_set_train_math below (exact fp32, or the conv stack of the 2-conv model in split precision on the f16 / f64 matrix cores).
 * All pointers in the two structs are DEVICE pointers in torch layout (the live parameters
 * and their .grad buffers): nothing is packed on the host, the weights may change between calls.
 * Dropout factors come from a counter-based generator keyed by (seed, layer, clip, unit): the same seed reproduces the
 * step; torch's own random stream cannot be matched (documented, tests/test_gpu_train.py). */
typedef struct ww_train_params {
    int32_t n_conv;              /* 2 (SimpleWakewordModel) or 3 (WakewordModel) */
    int32_t hidden;              /* 256 */
    const float* conv_weight[3];
    const float* conv_bias[3];
    const float* lstm_weight_ih[2];
    const float* lstm_bias_ih[2];
    const float* lstm_bias_hh[2];
    const float* fc_weight;
    const float* fc_bias;
} ww_train_params;
typedef struct ww_train_grads {  /* outputs, overwritten (not accumulated) */
    float* conv_weight[3];       /* [32,1,3,3], [64,32,3,3], [128,64,3,3] (n_conv 3) */
    float* conv_bias[3];
    float* lstm_weight_ih[2];    /* [1024, 64 | 128], [1024, 256]; forget-gate rows come out zero like autograd's */
    float* lstm_bias[2];         /* [1024]: d/d bias_ih == d/d bias_hh; weight_hh gradients are exactly zero (h0 = 0): caller zero-fills */
    float* fc_weight;            /* [2, 256] */
    float* fc_bias;              /* [2] */
} ww_train_grads;
/* The arithmetic is part of every call of a step (ABI 4): train_math = WW_TRAIN_MATH_F32 | WW_TRAIN_MATH_F16X3 (below), or
 * WW_TRAIN_MATH_DEFAULT = the process-wide ww_set_train_math value read at that call.  The workspace LAYOUT depends on it (0.23 GB under
 * F16X3, 2.7 GB under F32 for 4096 clips of the 2-conv model), so forward and backward also take the size of the buffer they were given
 * and return WW_EINVAL when it is smaller than ww_train_workspace_bytes(n, n_conv, train_math) -- a mode switch between the size query and
 * the launch cannot write past the buffer; a backward under another mode than its forward is WW_EINVAL too. */
#define WW_TRAIN_MATH_DEFAULT (-1)
WW_API int64_t ww_train_workspace_bytes(int64_t n, int32_t n_conv, int32_t train_math);
/* mel_dev [n][80][width]; p_lstm / p_fc: drop probabilities (0 = eval-mode arithmetic); workspace_dev: workspace_bytes >=
 * ww_train_workspace_bytes bytes, 256-byte aligned, must stay untouched until the matching ww_train_backward_f32 has run; logits_dev [n][2]. */
WW_API int ww_train_forward_f32(const float* mel_dev, int64_t n, int32_t width, const ww_train_params* params, float p_lstm,
                                float p_fc, uint64_t seed, int32_t train_math, void* workspace_dev, int64_t workspace_bytes,
                                float* logits_dev, ww_stream_t stream);
WW_API int ww_train_backward_f32(const float* mel_dev, int64_t n, int32_t width, const ww_train_params* params,
                                 const float* dlogits_dev, int32_t train_math, void* workspace_dev, int64_t workspace_bytes,
                                 const ww_train_grads* grads, ww_stream_t stream);
/* Diagnostic: the dropout factors (0 or 1 / (1 - p)) the last forward on this workspace applied to the layer-0 output and to
 * fc's input, [n][256] each -- lets a test replay the step in another framework with the same masks. */
WW_API int ww_train_masks(const void* workspace_dev, int64_t n, int32_t n_conv, float* mask0_dev, float* mask1_dev, ww_stream_t stream);
/* Diagnostic: the packed image (ww_packed_weights_floats(2) floats) that the last WW_TRAIN_MATH_F16X3 forward of the 2-conv model wrote on
 * the device from the live parameters -- its conv1 / conv2-Winograd / range entries equal ww_pack_weights_host's bit for bit; the rest is 0. */
WW_API int ww_train_packed_image(const void* workspace_dev, int64_t n, int32_t n_conv, float* img_dev, ww_stream_t stream);
/* Diagnostic: what the last WW_TRAIN_MATH_F16X3 forward on this workspace kept of the activations instead of the activations themselves --
 * mask_last_dev [n][80][32][C/8] bytes, bit c of the position = [relu(last conv)[c] > 0] (C = 64 | 128, byte cb = channels 8 cb ..), in canonical
 * order (the kernels' own image holds the accumulator ballots); sign1_dev [n][80][32] words, bit c = [relu(conv1)[c] > 0]. */
WW_API int ww_train_bit_images(const void* workspace_dev, int64_t n, int32_t n_conv, uint8_t* mask_last_dev, uint32_t* sign1_dev, ww_stream_t stream);
/* Arithmetic of the training step's convolution kernels (the head is always exact fp32); ww_set_train_math sets the process-wide DEFAULT that
 * WW_TRAIN_MATH_DEFAULT resolves to:
 *   WW_TRAIN_MATH_F32    exact fp32 matrix instructions throughout (v_mfma_f32_32x32x2_f32)
 *   WW_TRAIN_MATH_F16X3  (default) the conv stack in split precision on the f16 matrix instructions, for both models: forward = the
 *                        inference kernels with the ReLU masks as extra outputs (bit images); backward of the last conv: one operand
 *                        is the 0/1 mask, exact in f16, the other is carried as two f16 halves; below it both operands as two halves;
 *                        conv1's weight gradient in double on the f64 matrix instructions.  The head runs as under F32.
 *                        Gradients agree with F32 to the 2^-22 of the split. */
#define WW_TRAIN_MATH_F32 0
#define WW_TRAIN_MATH_F16X3 1
WW_API int ww_set_train_math(int mode);
WW_API int ww_get_train_math(void);

/* ---- streaming: sliding 1 s window, one hop per step, many microphones ------------------------ */
/* Semantics per window = predict_wakeword (wakeword_training.ipynb cell 19): normalise the last
 * 16000 samples, log-mel, forward, softmax, p[wakeword].  The reference has no streaming code;
 * this is that function applied to every hop of every microphone.
 * The per-hop step (ring append -> K1 -> K2 -> K3 -> softmax) is captured once into a hipGraph and
 * replayed by ww_streamer_step. */
typedef struct ww_streamer ww_streamer;
WW_API int ww_streamer_create(int32_t n_mics, int32_t hop_samples, const float* packed_dev, int32_t n_conv,
                       ww_stream_t stream, ww_streamer** out);
/* hop_dev [n_mics][hop_samples] new samples; prob_dev [n_mics] softmax p(wakeword) of the window
 * ending at this hop; logits_dev (may be NULL) [n_mics][2]. */
WW_API int ww_streamer_step(ww_streamer* s, const float* hop_dev, float* prob_dev, float* logits_dev);
/* Copy of the current 1 s window of every mic, oldest sample first: [n_mics][16000] (for tests). */
WW_API int ww_streamer_window(ww_streamer* s, float* window_dev);
WW_API int ww_streamer_destroy(ww_streamer* s);

#ifdef __cplusplus
}
#endif
#endif /* WAKEWORD_AMD_H */
