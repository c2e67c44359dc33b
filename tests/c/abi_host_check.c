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
    if (!have_gpu) {
        /* no CPU fallback: every launch fails loudly without a gfx950 device */
        CHECK(ww_init() == WW_ENODEVICE);
        CHECK(ww_sync_timeouts() == WW_ENODEVICE);
        CHECK(ww_logmel_f32((const float*)16, 1, 16000, 16000, 1, (float*)16, NULL) == WW_ENODEVICE);
    }
    printf("abi_host_check OK\n");
    return 0;
}
