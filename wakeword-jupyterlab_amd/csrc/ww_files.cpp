// File-fed batches: the host half of AudioProcessor.load_audio = librosa.load(path, sr=16000)
// (/root/reference/wakeword_training_script.py:65-71) for a whole batch of files, feeding K0 (ww_decode.hip).
//
// The reference reads one file per DataLoader worker call (2 workers, :461-463) and decodes it on the CPU; that loop is
// what bounds it at 453 clips/s (wakeword_training.ipynb:742).  Here a batch of paths is read by a pool of host threads --
// open, RIFF chunk walk, pread of the sample bytes STRAIGHT into pinned staging owned by the library (no intermediate copy,
// no Python in the loop) -- then one H2D copy on the library's copy stream and K0 on the caller's stream.  Staging is
// multi-buffered ("slots"): while the GPU decodes / runs the model on slot s, the host threads fill slot s+1.
//
// Order inside the staging buffer is first-come (an atomic bump allocator): descriptors carry the byte offsets, so the
// layout does not matter and no second pass over the files is needed.
#include <fcntl.h>
#include <sys/prctl.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "ww_internal.h"

#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
#include <emmintrin.h>
#define WW_STREAMING_COPY 1
#endif

namespace ww {

struct WavInfo {
    int tag = 0, channels = 0, sample_rate = 0, bits = 0;
    int64_t data_start = -1, data_len = 0;
};

static inline uint32_t le32(const uint8_t* p) { return uint32_t(p[0]) | uint32_t(p[1]) << 8 | uint32_t(p[2]) << 16 | uint32_t(p[3]) << 24; }
static inline uint32_t le16(const uint8_t* p) { return uint32_t(p[0]) | uint32_t(p[1]) << 8; }

// bytes [pos, pos+len) of the file: from the header window when it covers them, else a pread
static bool fetch(int fd, const uint8_t* win, int64_t win_len, int64_t pos, int64_t len, uint8_t* dst) {
    if (pos + len <= win_len) { std::memcpy(dst, win + pos, size_t(len)); return true; }
    int64_t got = 0;
    while (got < len) {
        const ssize_t r = pread(fd, dst + got, size_t(len - got), off_t(pos + got));
        if (r <= 0) return false;
        got += r;
    }
    return true;
}

// RIFF/WAVE chunk walk: the last `fmt ` and the last `data` chunk win, odd chunk sizes are padded, a data chunk longer
// than the file is cut at the end of the file (the host reader of audio.py walks the same way).  `win` holds the first
// win_len bytes of the file; anything beyond is fetched with pread.
static int parse_wav(int fd, int64_t fsize, const uint8_t* win, int64_t win_len, WavInfo* w) {
    if (win_len < 12 || std::memcmp(win, "RIFF", 4) || std::memcmp(win + 8, "WAVE", 4)) return WW_WAV_ENOTRIFF;
    int64_t pos = 12;
    bool have_fmt = false;
    while (pos + 8 <= fsize) {
        uint8_t h[8];
        if (!fetch(fd, win, win_len, pos, 8, h)) return WW_WAV_EIO;
        const int64_t size = le32(h + 4);
        if (!std::memcmp(h, "fmt ", 4)) {
            uint8_t b[40] = {0};
            const int64_t avail = fsize - pos - 8;
            const int64_t want = size < 40 ? size : 40;
            const int64_t blen = want < avail ? want : avail;
            if (blen < 16 || !fetch(fd, win, win_len, pos + 8, blen, b)) return WW_WAV_ECHUNK;
            w->tag = int(le16(b));
            w->channels = int(le16(b + 2));
            w->sample_rate = int(le32(b + 4));
            w->bits = int(le16(b + 14));
            if (w->tag == 0xFFFE && blen >= 26) w->tag = int(le16(b + 24));   // WAVE_FORMAT_EXTENSIBLE: the real tag is in the GUID
            have_fmt = true;
        } else if (!std::memcmp(h, "data", 4)) {
            w->data_start = pos + 8;
            const int64_t rest = fsize - pos - 8;
            // a size of 0xFFFFFFFF (what ffmpeg / sox write into a pipe) or 0 (a header that was never finalised) means "to the end of
            // the file", as libsndfile reads it; nothing can follow such a chunk
            if (size == 0 || size == 0xFFFFFFFFll) { w->data_len = rest; break; }
            w->data_len = size < rest ? size : rest;
        }
        pos += 8 + size + (size & 1);
    }
    if (!have_fmt || w->data_start < 0) return WW_WAV_ECHUNK;
    return 1;
}

// The head of the file in ONE read, and the file's size from that read (a short read of a regular file is its end) or, for files larger
// than the window, from fstat (one more syscall, only for files over 68 KB).  The size fields INSIDE the file are never trusted for how
// much staging a file gets: a streaming header (RIFF / data size 0xFFFFFFFF) or a truncated file would otherwise reserve gigabytes and
// fail the whole batch with WW_ENOSPACE (round-3 review); every length is clamped to the bytes that exist.  Whatever of the sample data
// the window does not hold is read straight into the pinned staging buffer.
#ifndef WW_HEAD_WINDOW
#define WW_HEAD_WINDOW 69632
#endif
// 68 KB: a 1 s / 16 kHz PCM-16 clip (32,044 bytes, the reference's data format) arrives whole in the head read: open + ONE pread + close
// and a 32 KB copy into the staging buffer.  Measured against a 512-byte head (two preads, samples straight into staging, no copy) on
// the same MI355X host, interleaved (scripts/ab_reader.py, 16 threads): 1.06 M vs 0.96 M files/s.
constexpr int64_t kHeadWindow = WW_HEAD_WINDOW;
static int64_t read_head(int fd, uint8_t* win, int64_t* fsize_out) {
    const ssize_t r = pread(fd, win, size_t(kHeadWindow), 0);
    if (r < 0) return -1;
    const int64_t got = r;
    int64_t fsize = got;                                          // a short read of a regular file is its end
    if (got == kHeadWindow) {
        struct stat sb;
        if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) return -1;
        fsize = int64_t(sb.st_size);
        if (fsize < got) fsize = got;                             // the file grew or shrank under us: what was read is there
    }
    *fsize_out = fsize;
    return got;
}

static int format_of(const WavInfo& w) {
    if (w.tag == 1 && w.bits == 16) return WW_FMT_S16;
    if (w.tag == 1 && w.bits == 24) return WW_FMT_S24;
    if (w.tag == 1 && w.bits == 32) return WW_FMT_S32;
    if (w.tag == 3 && w.bits == 32) return WW_FMT_F32;
    if (w.tag == 3 && w.bits == 64) return WW_FMT_F64;
    if (w.tag == 1 && w.bits == 8) return WW_FMT_U8;
    return 0;
}

struct Slot {
    uint8_t* raw_host = nullptr;          // pinned
    ww_clip_desc* descs_host = nullptr;   // pinned
    uint8_t* raw_dev = nullptr;
    ww_clip_desc* descs_dev = nullptr;
    hipEvent_t copied = nullptr, decoded = nullptr;
    bool copy_inflight = false, decode_inflight = false;
    int64_t n = 0, raw_bytes = 0;
};

struct Job {
    const char* const* paths = nullptr;
    int64_t n = 0;
    Slot* slot = nullptr;
    int8_t* status = nullptr;
    std::atomic<int64_t> next{0}, cursor{0};
    int64_t capacity = 0;
};

}  // namespace ww

using namespace ww;

struct ww_wav_reader {
    int device = 0, n_threads = 1, n_slots = 2;
    bool host_only = false;
    int64_t max_clips = 0, max_raw = 0;
    hipStream_t copy_stream = nullptr;
    std::vector<Slot> slots;
    std::map<int, ww_clip_desc> filters;      // sample rate -> prototype with up/down/half_len/taps_dev
    // worker pool
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    Job* job = nullptr;
    uint64_t generation = 0;
    int active = 0;
    bool stop = false;
};

namespace ww {

// Payload -> pinned staging.  The staging slots (3 x 33 MB at the bench's size) live in DRAM and the next reader of these bytes is the
// copy engine, not a CPU: streaming stores write the lines without first reading them for ownership and without evicting the page
// cache's lines the next read() wants (same host, 16 threads, open + read + copy + close: 3.1-3.3 -> 2.65-2.70 ms per 4096 files on
// the container's overlay, 2.30 -> 1.70 on tmpfs; scripts/proto/reader_micro.c modes 2 / 4).  dst is 16-byte aligned (the allocator).
static void copy_to_staging(uint8_t* dst, const uint8_t* src, size_t n) {
#if !defined(__HIP_DEVICE_COMPILE__)
    size_t i = 0;
#ifdef WW_STREAMING_COPY
    if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0 && n >= 256) {
        for (; i + 64 <= n; i += 64) {
            const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i));
            const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 16));
            const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 32));
            const __m128i d = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 48));
            _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i), a);
            _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i + 16), b);
            _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i + 32), c);
            _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i + 48), d);
        }
        _mm_sfence();                                          // before the batch is handed on (the caller's thread starts the upload)
    }
#endif
    std::memcpy(dst + i, src + i, n - i);
#endif
}

static void read_one(Job* j, int64_t i, uint8_t* win) {
    ww_clip_desc d;
    std::memset(&d, 0, sizeof d);
    d.channels = 1; d.format = WW_FMT_S16; d.up = 1; d.down = 1; d.sample_rate = WW_SAMPLE_RATE;   // what K0 sees for an unreadable file: nothing
    int st = WW_WAV_EOPEN;
    const int fd = j->paths[i] ? open(j->paths[i], O_RDONLY | O_CLOEXEC) : -1;
    if (fd >= 0) {
        WavInfo w;
        int64_t fsize = 0;
        const int64_t win_len = read_head(fd, win, &fsize);
        st = win_len < 0 ? WW_WAV_EIO : parse_wav(fd, fsize, win, win_len, &w);
        if (st == 1) {
            const int fmt = format_of(w);
            if (!fmt || w.channels < 1 || w.sample_rate < 1000 || w.sample_rate > 384000) st = WW_WAV_EFORMAT;
            else {
                const int64_t frame_bytes = int64_t(w.channels) * (w.bits / 8);
                const int64_t frames = w.data_len / frame_bytes, bytes = frames * frame_bytes;
                const int64_t aligned = (bytes + 15) & ~int64_t(15);                          // every file starts 16-byte aligned
                const int64_t off = j->cursor.fetch_add(aligned, std::memory_order_relaxed);
                if (off + aligned > j->capacity) st = WW_WAV_ESPACE;
                else {
                    uint8_t* dst = j->slot->raw_host + off;
                    // what the head window already holds is copied (tiny files); the rest is read straight into the staging buffer
                    int64_t got = 0;
                    if (w.data_start < win_len) {
                        got = win_len - w.data_start < bytes ? win_len - w.data_start : bytes;
                        copy_to_staging(dst, win + w.data_start, size_t(got));
                    }
                    while (got < bytes) {
                        const ssize_t r = pread(fd, dst + got, size_t(bytes - got), off_t(w.data_start + got));
                        if (r <= 0) break;
                        got += r;
                    }
                    // fewer bytes than the headers promised: the file is cut -- keep the whole sample frames that are there
                    const int64_t have_frames = got / frame_bytes, have_bytes = have_frames * frame_bytes;
                    std::memset(dst + have_bytes, 0, size_t(aligned - have_bytes));
                    d.byte_offset = off; d.n_frames = have_frames; d.channels = w.channels; d.sample_rate = w.sample_rate; d.format = fmt;
                }
            }
        }
        close(fd);
    }
    j->slot->descs_host[i] = d;
    j->status[i] = int8_t(st);
}

static void read_range(Job* j) {
    alignas(16) uint8_t win[kHeadWindow];                     // on the thread's stack (68 KB)
    for (int64_t i; (i = j->next.fetch_add(1, std::memory_order_relaxed)) < j->n;) read_one(j, i, win);
}

// What a pool thread shares with the rest of the process decides the reader's rate: with 16+ threads opening ~1 M files/s, every
// open() / close() takes the lock of the process's descriptor table and bumps the reference count of the process's credentials --
// two cache lines that all threads fight over (MI355X host, 16-CPU quota, scripts/proto/reader_micro.c, open + read + copy + close
// from 16 threads: 0.81 M files/s as the process comes, 1.02 M on private descriptor tables, 1.33 M with private credentials too).
// Each pool thread therefore leaves both behind:
//  * close_range(3, ~0, CLOSE_RANGE_UNSHARE): an empty descriptor table of its own (0-2 are copied, nothing else: the thread holds
//    no reference to the process's sockets or device nodes -- closing one elsewhere still closes it -- and what it opens is its
//    own).  A pool thread never uses a descriptor it did not open itself.
//  * prctl(PR_SET_KEEPCAPS, <the value it has>): changes nothing, but the call commits a copy of the credentials to this thread --
//    the same identity in an object of its own, so open()'s get_cred() no longer bounces a line between CPUs.
// Kernels without close_range (< 5.9) or a seccomp filter that denies either call: the thread keeps what it shares, nothing else
// changes.  WW_READER_SHARED_FDS=1 keeps both shared (ThreadSanitizer tracks descriptors by number for the whole process).
static void leave_shared_process_state() {
    static const bool keep = getenv("WW_READER_SHARED_FDS") != nullptr;
    if (keep) return;
#ifdef SYS_close_range
    (void)syscall(SYS_close_range, 3u, ~0u, 2u /* CLOSE_RANGE_UNSHARE */);
#endif
    const int keepcaps = prctl(PR_GET_KEEPCAPS, 0, 0, 0, 0);
    if (keepcaps >= 0) (void)prctl(PR_SET_KEEPCAPS, keepcaps, 0, 0, 0);
}

static void worker_main(ww_wav_reader* r) {
    leave_shared_process_state();
    uint64_t seen = 0;
    for (;;) {
        Job* j;
        {
            std::unique_lock<std::mutex> lk(r->mu);
            r->cv_work.wait(lk, [&] { return r->stop || r->generation != seen; });
            if (r->stop) return;
            seen = r->generation;
            j = r->job;
        }
        read_range(j);
        {
            std::lock_guard<std::mutex> lk(r->mu);
            if (--r->active == 0) r->cv_done.notify_all();
        }
    }
}

static void run_job(ww_wav_reader* r, Job* j) {
    if (r->workers.empty() || j->n < 4) {                     // tiny batches: not worth waking the pool
        read_range(j);
        return;
    }
    {
        std::lock_guard<std::mutex> lk(r->mu);
        r->job = j;
        r->active = int(r->workers.size());
        ++r->generation;
    }
    r->cv_work.notify_all();
    read_range(j);                                             // the caller helps
    std::unique_lock<std::mutex> lk(r->mu);
    r->cv_done.wait(lk, [&] { return r->active == 0; });
    r->job = nullptr;
}

static void free_slots(ww_wav_reader* r) {
    for (Slot& s : r->slots) {
        if (r->host_only) {
            std::free(s.raw_host);
            std::free(s.descs_host);
            continue;
        }
        if (s.copied) { (void)hipEventSynchronize(s.copied); (void)hipEventDestroy(s.copied); }
        if (s.decoded) { (void)hipEventSynchronize(s.decoded); (void)hipEventDestroy(s.decoded); }
        if (s.raw_host) (void)hipHostFree(s.raw_host);
        if (s.descs_host) (void)hipHostFree(s.descs_host);
        if (s.raw_dev) (void)hipFree(s.raw_dev);
        if (s.descs_dev) (void)hipFree(s.descs_dev);
    }
    r->slots.clear();
}

}  // namespace ww

extern "C" {

int ww_wav_probe_host(const char* path, ww_clip_desc* desc_host) {
    if (!path || !desc_host) return fail(WW_EINVAL, "ww_wav_probe_host: null argument");
    std::memset(desc_host, 0, sizeof *desc_host);
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return WW_WAV_EOPEN;
    WavInfo w;
    alignas(16) uint8_t win[kHeadWindow];
    int64_t fsize = 0;
    const int64_t win_len = read_head(fd, win, &fsize);
    int st = win_len < 0 ? WW_WAV_EIO : parse_wav(fd, fsize, win, win_len, &w);
    close(fd);
    if (st != 1) return st;
    const int fmt = format_of(w);
    if (!fmt || w.channels < 1 || w.sample_rate < 1000 || w.sample_rate > 384000) return WW_WAV_EFORMAT;
    desc_host->byte_offset = w.data_start;
    desc_host->n_frames = w.data_len / (int64_t(w.channels) * (w.bits / 8));
    desc_host->channels = w.channels;
    desc_host->sample_rate = w.sample_rate;
    desc_host->format = fmt;
    (void)ww_resample_taps_host(w.sample_rate, nullptr, 0, &desc_host->up, &desc_host->down, &desc_host->half_len);
    return 1;
}

int ww_wav_reader_staging(ww_wav_reader* r, int32_t slot, const uint8_t** raw_host_out, int64_t* raw_bytes_out) {
    if (!r || slot < 0 || slot >= r->n_slots) return fail(WW_EINVAL, "ww_wav_reader_staging: bad reader / slot");
    if (raw_host_out) *raw_host_out = r->slots[size_t(slot)].raw_host;
    if (raw_bytes_out) *raw_bytes_out = r->slots[size_t(slot)].raw_bytes;
    return WW_OK;
}

int ww_wav_reader_create(int32_t n_threads, int32_t n_slots, int64_t max_clips, int64_t max_raw_bytes, int32_t flags, ww_wav_reader** out) {
    if (!out) return fail(WW_EINVAL, "null out pointer");
    *out = nullptr;
    if (n_threads < 1 || n_threads > 256 || n_slots < 1 || n_slots > 8 || max_clips < 1 || max_clips > (int64_t(1) << 24) ||
        max_raw_bytes < 16 || max_raw_bytes > (int64_t(1) << 36))
        return fail(WW_EINVAL, "ww_wav_reader_create: threads %d slots %d clips %lld bytes %lld out of range", n_threads, n_slots,
                    (long long)max_clips, (long long)max_raw_bytes);
    const bool host_only = (flags & WW_READER_HOST_ONLY) != 0;
    if (!host_only)
        if (int rc = require_gfx950()) return rc;
    ww_wav_reader* r = new ww_wav_reader;
    r->n_threads = n_threads; r->n_slots = n_slots; r->max_clips = max_clips; r->host_only = host_only;
    r->max_raw = (max_raw_bytes + 15) & ~int64_t(15);
    if (host_only) {
        r->slots.resize(size_t(n_slots));
        for (Slot& s : r->slots) {
            s.raw_host = static_cast<uint8_t*>(std::malloc(size_t(r->max_raw)));
            s.descs_host = static_cast<ww_clip_desc*>(std::malloc(sizeof(ww_clip_desc) * size_t(max_clips)));
            if (!s.raw_host || !s.descs_host) { free_slots(r); delete r; return fail(WW_EINVAL, "ww_wav_reader_create: out of host memory"); }
        }
        for (int t = 1; t < n_threads; ++t) r->workers.emplace_back(worker_main, r);
        *out = r;
        return WW_OK;
    }
    auto bail = [&](hipError_t e, const char* what) {
        free_slots(r);
        if (r->copy_stream) (void)hipStreamDestroy(r->copy_stream);
        delete r;
        return fail(WW_EHIP, "ww_wav_reader_create: %s failed: %s", what, hipGetErrorString(e));
    };
    hipError_t e;
    if ((e = hipGetDevice(&r->device)) != hipSuccess) return bail(e, "hipGetDevice");
    if ((e = hipStreamCreateWithFlags(&r->copy_stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    r->slots.resize(size_t(n_slots));
    for (Slot& s : r->slots) {
        if ((e = hipHostMalloc(reinterpret_cast<void**>(&s.raw_host), size_t(r->max_raw), hipHostMallocDefault)) != hipSuccess) return bail(e, "hipHostMalloc(raw)");
        if ((e = hipHostMalloc(reinterpret_cast<void**>(&s.descs_host), sizeof(ww_clip_desc) * size_t(max_clips), hipHostMallocDefault)) != hipSuccess) return bail(e, "hipHostMalloc(descs)");
        if ((e = hipMalloc(reinterpret_cast<void**>(&s.raw_dev), size_t(r->max_raw))) != hipSuccess) return bail(e, "hipMalloc(raw)");
        if ((e = hipMalloc(reinterpret_cast<void**>(&s.descs_dev), sizeof(ww_clip_desc) * size_t(max_clips))) != hipSuccess) return bail(e, "hipMalloc(descs)");
        if ((e = hipEventCreateWithFlags(&s.copied, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
        if ((e = hipEventCreateWithFlags(&s.decoded, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    }
    for (int t = 1; t < n_threads; ++t) r->workers.emplace_back(worker_main, r);     // the calling thread is worker 0
    *out = r;
    return WW_OK;
}

int ww_wav_reader_destroy(ww_wav_reader* r) {
    if (!r) return WW_OK;
    {
        std::lock_guard<std::mutex> lk(r->mu);
        r->stop = true;
    }
    r->cv_work.notify_all();
    for (std::thread& t : r->workers) t.join();
    free_slots(r);
    if (r->copy_stream) { (void)hipStreamSynchronize(r->copy_stream); (void)hipStreamDestroy(r->copy_stream); }
    delete r;
    return WW_OK;
}

int ww_read_wav_batch_host(ww_wav_reader* r, const char* const* paths, int64_t n, int32_t slot, ww_clip_desc** descs_host_out,
                           int8_t* status_host, int64_t* raw_bytes_out) {
    if (!r || !paths || !status_host || n < 0) return fail(WW_EINVAL, "ww_read_wav_batch_host: null argument");
    if (slot < 0 || slot >= r->n_slots) return fail(WW_EINVAL, "slot %d of %d", slot, r->n_slots);
    if (n > r->max_clips) return fail(WW_EINVAL, "%lld files but the reader was created for %lld per batch", (long long)n, (long long)r->max_clips);
    Slot& s = r->slots[size_t(slot)];
    if (s.copy_inflight) {                     // the previous upload of this slot still reads the pinned staging
        WW_HIP(hipEventSynchronize(s.copied));
        s.copy_inflight = false;
    }
    Job j;
    j.paths = paths; j.n = n; j.slot = &s; j.status = status_host; j.capacity = r->max_raw;
    static const bool trace = getenv("WW_READER_TRACE") != nullptr;     // diagnostics: time of the threaded part of every call, on stderr
    const auto t0 = std::chrono::steady_clock::now();
    run_job(r, &j);
    if (trace)
        std::fprintf(stderr, "[ww reader] %lld files, %d threads: %.3f ms\n", (long long)n, r->n_threads,
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    s.n = n;
    const int64_t need = j.cursor.load();                      // every usable file reserved its aligned size, whether it fitted or not
    s.raw_bytes = need < r->max_raw ? need : r->max_raw;
    if (raw_bytes_out) *raw_bytes_out = need;
    if (descs_host_out) *descs_host_out = s.descs_host;
    if (need > r->max_raw)
        return fail(WW_ENOSPACE, "staging too small: this batch holds %lld bytes of samples, the reader was created with %lld", (long long)need,
                    (long long)r->max_raw);
    if (r->host_only) {                        // no device: the resampler geometry only (taps_dev stays NULL)
        for (int64_t i = 0; i < n; ++i) {
            ww_clip_desc& d = s.descs_host[i];
            if (status_host[i] == 1) (void)ww_resample_taps_host(d.sample_rate, nullptr, 0, &d.up, &d.down, &d.half_len);
        }
        return WW_OK;
    }
    // resampler prototypes per distinct sample rate (device upload once per rate; HIP calls stay on the caller's thread)
    int dev = 0;
    WW_HIP(hipGetDevice(&dev));
    if (dev != r->device) return fail(WW_EINVAL, "reader belongs to device %d, current device is %d", r->device, dev);
    for (int64_t i = 0; i < n; ++i) {
        if (status_host[i] != 1) continue;
        ww_clip_desc& d = s.descs_host[i];
        auto it = r->filters.find(d.sample_rate);
        if (it == r->filters.end()) {
            ww_clip_desc proto;
            std::memset(&proto, 0, sizeof proto);
            if (int rc = ww_resampler_prepare(d.sample_rate, &proto)) return rc;
            it = r->filters.emplace(d.sample_rate, proto).first;
        }
        d.up = it->second.up; d.down = it->second.down; d.half_len = it->second.half_len; d.taps_dev = it->second.taps_dev;
    }
    return WW_OK;
}

int ww_wav_batch_decode(ww_wav_reader* r, int32_t slot, int normalize, float* pcm_out_dev, ww_stream_t stream) {
    if (!r || !pcm_out_dev) return fail(WW_EINVAL, "ww_wav_batch_decode: null argument");
    if (slot < 0 || slot >= r->n_slots) return fail(WW_EINVAL, "slot %d of %d", slot, r->n_slots);
    if (r->host_only) return fail(WW_EUNSUPPORTED, "this reader was created WW_READER_HOST_ONLY (no device twin to decode from)");
    Slot& s = r->slots[size_t(slot)];
    if (s.n == 0) return WW_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (s.decode_inflight) WW_HIP(hipStreamWaitEvent(r->copy_stream, s.decoded, 0));   // K0 of the previous batch still reads raw_dev
    if (s.raw_bytes) WW_HIP(hipMemcpyAsync(s.raw_dev, s.raw_host, size_t(s.raw_bytes), hipMemcpyHostToDevice, r->copy_stream));
    WW_HIP(hipMemcpyAsync(s.descs_dev, s.descs_host, sizeof(ww_clip_desc) * size_t(s.n), hipMemcpyHostToDevice, r->copy_stream));
    WW_HIP(hipEventRecord(s.copied, r->copy_stream));
    s.copy_inflight = true;
    WW_HIP(hipStreamWaitEvent(st, s.copied, 0));
    if (int rc = ww_decode_resample(s.raw_dev, s.descs_dev, s.n, normalize, pcm_out_dev, stream)) return rc;
    WW_HIP(hipEventRecord(s.decoded, st));
    s.decode_inflight = true;
    return WW_OK;
}

}  // extern "C"
