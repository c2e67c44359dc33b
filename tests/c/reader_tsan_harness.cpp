// ThreadSanitizer harness for the batch WAV reader (csrc/ww_files.cpp) in its host-only mode: 8 threads, 3 slots, 30 rounds over the
// files given on the command line.  Built and run by tests/test_host_files.py::test_reader_is_clean_under_thread_sanitizer with the
// reader's own source compiled -fsanitize=thread as plain C++ (sanitizers run on the CPU build only: the GPU pool refuses them).
// The internal helpers the reader takes from other translation units are stubbed here; the exported C entry points it calls
// (ww_resample_taps_host, ww_resampler_prepare, ww_decode_resample) come from libwakeword_amd.so.
#include <cstdarg>
#include <cstdio>
#include <vector>

#include "wakeword_amd.h"

namespace ww {
int fail(int code, const char*, ...) { return code; }
int require_gfx950() { return WW_ENODEVICE; }
}  // namespace ww

int main(int argc, char** argv) {
    const int n = argc - 1;
    std::vector<const char*> paths(argv + 1, argv + argc);
    ww_wav_reader* rd = nullptr;
    if (ww_wav_reader_create(8, 3, n, 64 << 20, WW_READER_HOST_ONLY, &rd) != WW_OK) { std::printf("create failed\n"); return 1; }
    std::vector<int8_t> st(n);
    long ok = 0;
    for (int round = 0; round < 30; ++round) {
        ww_clip_desc* d = nullptr;
        int64_t need = 0;
        if (ww_read_wav_batch_host(rd, paths.data(), n, round % 3, &d, st.data(), &need) != WW_OK) { std::printf("read failed\n"); return 2; }
        for (int i = 0; i < n; ++i) ok += st[i] == 1;
    }
    ww_wav_reader_destroy(rd);
    std::printf("READER_TSAN_OK %ld\n", ok);
    return 0;
}
