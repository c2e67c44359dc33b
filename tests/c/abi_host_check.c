/* Plain-C consumer of include/wakeword_amd.h: proves that the boundary is a C ABI (the header compiles as C99 with
 * -Wall -Werror -pedantic, the library links without any C++/HIP/torch symbol on the caller's side) and exercises the
 * host-only entry points plus the no-GPU error behaviour.  Built and run by tests/test_host_native.py. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "wakeword_amd.h"

#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            fprintf(stderr, "FAILED %s:%d: %s (last error: %s)\n", __FILE__, __LINE__, #cond, ww_last_error()); \
            return 1;                                                      \
        }                                                                  \
    } while (0)

int main(int argc, char** argv) {
    int have_gpu = argc > 1 && strcmp(argv[1], "gpu") == 0;
    static float mel[WW_N_MELS * WW_N_BINS];
    static float win[WW_N_FFT];
    static float kb[32769];
    float taps[64];
    int32_t up = 0, down = 0, half = 0;
    double s;
    int k, n;

    CHECK(ww_abi_version() == WW_ABI_VERSION);
    CHECK(ww_mel_filterbank_host(mel) == WW_OK);
    s = 0.0;
    for (k = 0; k < WW_N_BINS; ++k) s += mel[40 * WW_N_BINS + k];
    CHECK(s > 0.0 && mel[0] == 0.0f);                        /* Slaney filters: bin 0 carries no weight */
    CHECK(ww_hann_window_host(win) == WW_OK);
    CHECK(win[0] == 0.0f && fabs(win[WW_N_FFT / 2] - 1.0) < 1e-7);
    CHECK(ww_kaiser_best_host(kb) == WW_OK);
    CHECK(fabs(kb[0] - 0.9475937167399596) < 1e-7 && fabs(kb[32768]) < 1e-6);
    n = ww_resample_taps_host(48000, NULL, 0, &up, &down, &half);
    CHECK(n == 61 && up == 1 && down == 3 && half == 30);
    CHECK(ww_resample_taps_host(48000, taps, 64, NULL, NULL, NULL) == 61);
    CHECK(ww_resample_taps_host(7, NULL, 0, NULL, NULL, NULL) == WW_EINVAL);
    CHECK(ww_packed_weights_floats(2) > 0 && ww_packed_weights_floats(3) > ww_packed_weights_floats(2));
    CHECK(ww_packed_weights_floats(4) == WW_EINVAL && strstr(ww_last_error(), "n_conv") != NULL);
    CHECK(ww_workspace_bytes(16, 2) > 0 && ww_cnn_scratch_bytes(16, 2) == 0 && ww_augment_workspace_bytes(16) > 0);
    /* argument errors are reported before the device is touched */
    CHECK(ww_logmel_f32(NULL, 4, 16000, 16000, 1, NULL, NULL) == WW_EINVAL);
    CHECK(ww_logmel_f32((const float*)16, 4, 16000, 20000, 1, (float*)16, NULL) == WW_EINVAL);
    CHECK(ww_set_conv_math(7) == WW_EINVAL && ww_get_conv_math() == WW_CONV_MATH_F16X3);
    CHECK(ww_train_workspace_bytes(64, 2, WW_TRAIN_MATH_F16X3) < ww_train_workspace_bytes(64, 2, WW_TRAIN_MATH_F32));
    CHECK(ww_set_conv_math_thread(WW_CONV_MATH_F32) == WW_OK && ww_get_conv_math() == WW_CONV_MATH_F32);
    CHECK(ww_set_conv_math_thread(WW_MATH_INHERIT) == WW_OK && ww_get_conv_math() == WW_CONV_MATH_F16X3);
    /* the batch WAV reader from C, host-only mode (no GPU): one file written here, read back through the thread pool */
    {
        unsigned char hdr[44] = {'R', 'I', 'F', 'F', 0, 0, 0, 0, 'W', 'A', 'V', 'E', 'f', 'm', 't', ' ', 16, 0, 0, 0, 1, 0, 1, 0,
                                 0x80, 0x3e, 0, 0, 0, 0x7d, 0, 0, 2, 0, 16, 0, 'd', 'a', 't', 'a', 0, 0, 0, 0};
        const char* path = argc > 2 ? argv[2] : "/tmp/ww_abi_host_check.wav";
        const char* paths[2];
        short samples[100];
        ww_wav_reader* rd = NULL;
        ww_clip_desc* descs = NULL;
        ww_clip_desc probe;
        const uint8_t* stage = NULL;
        int8_t status[2];
        int64_t need = 0, staged = 0;
        FILE* f;
        for (k = 0; k < 100; ++k) samples[k] = (short)(37 * k - 1000);
        hdr[4] = (unsigned char)(36 + 200); hdr[40] = 200;
        f = fopen(path, "wb");
        CHECK(f != NULL);
        CHECK(fwrite(hdr, 1, 44, f) == 44 && fwrite(samples, 2, 100, f) == 100);
        fclose(f);
        paths[0] = path; paths[1] = "/nonexistent/ww.wav";
        CHECK(ww_wav_probe_host(path, &probe) == 1 && probe.n_frames == 100 && probe.sample_rate == 16000 && probe.byte_offset == 44);
        CHECK(ww_wav_reader_create(2, 2, 8, 4096, WW_READER_HOST_ONLY, &rd) == WW_OK);
        CHECK(ww_read_wav_batch_host(rd, paths, 2, 1, &descs, status, &need) == WW_OK);
        CHECK(status[0] == 1 && status[1] == WW_WAV_EOPEN && need == 208 && descs[0].n_frames == 100 && descs[0].format == WW_FMT_S16);
        CHECK(descs[1].n_frames == 0 && descs[0].up == 1 && descs[0].down == 1);
        CHECK(ww_wav_reader_staging(rd, 1, &stage, &staged) == WW_OK && staged == 208);
        CHECK(memcmp(stage + descs[0].byte_offset, samples, 200) == 0);
        CHECK(ww_wav_batch_decode(rd, 1, 1, (float*)16, NULL) == WW_EUNSUPPORTED);       /* host-only: nothing to decode on */
        CHECK(ww_wav_reader_create(2, 2, 8, 64, WW_READER_HOST_ONLY, NULL) == WW_EINVAL);
        CHECK(ww_wav_reader_destroy(rd) == WW_OK);
        remove(path);
    }
    if (!have_gpu) {
        /* no CPU fallback: every launch fails loudly without a gfx950 device */
        CHECK(ww_init() == WW_ENODEVICE);
        CHECK(ww_sync_timeouts() == WW_ENODEVICE);
        CHECK(ww_logmel_f32((const float*)16, 1, 16000, 16000, 1, (float*)16, NULL) == WW_ENODEVICE);
    }
    printf("abi_host_check OK\n");
    return 0;
}
