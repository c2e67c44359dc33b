#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <stdatomic.h>
#include <sched.h>
#include <emmintrin.h>
#include <sys/prctl.h>
enum { N = 4096, SZ = 32044, REPS = 48 };   /* one job = REPS passes over the files: ~0.2 s, several quota periods */
static char paths[N][64];
static atomic_int next_i;
static int mode, nthreads, private_fds, private_cred, ndirs = 1, use_dirfd;
static char dirs[64][64], names[N][32];
static char* slot;
static double now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static void copy_nt(char* dst, const char* src, size_t n) {      /* dst 16-byte aligned; streaming stores: no read-for-ownership, no cache fill */
  size_t i = 0;
  for (; i + 64 <= n; i += 64) {
    __m128i a = _mm_loadu_si128((const __m128i*)(src + i)), b = _mm_loadu_si128((const __m128i*)(src + i + 16));
    __m128i c = _mm_loadu_si128((const __m128i*)(src + i + 32)), d = _mm_loadu_si128((const __m128i*)(src + i + 48));
    _mm_stream_si128((__m128i*)(dst + i), a); _mm_stream_si128((__m128i*)(dst + i + 16), b);
    _mm_stream_si128((__m128i*)(dst + i + 32), c); _mm_stream_si128((__m128i*)(dst + i + 48), d);
  }
  memcpy(dst + i, src + i, n - i);
  _mm_sfence();
}
static void* work(void* a) {
  char* buf = aligned_alloc(4096, 69632);
  if (private_fds && unshare(CLONE_FILES) != 0) { perror("unshare"); exit(1); }
  if (private_cred && prctl(PR_SET_KEEPCAPS, 0, 0, 0, 0) != 0) { perror("prctl"); exit(1); }   /* commit_creds(): a cred struct of this thread's own */
  int dfd[64];
  if (use_dirfd) for (int d = 0; d < ndirs; ++d) dfd[d] = open(dirs[d], O_RDONLY | O_DIRECTORY);
  for (;;) {
    int i = atomic_fetch_add(&next_i, 1);
    if (i >= N * REPS) break;
    i %= N;
    if (mode < 0) continue;
    int fd = use_dirfd ? openat(dfd[i % ndirs], names[i], O_RDONLY | O_CLOEXEC) : open(paths[i], O_RDONLY | O_CLOEXEC);
    if (fd < 0) { perror("open"); exit(1); }
    if (mode >= 1) {
      if (mode == 3) { struct stat st; fstat(fd, &st); ssize_t g = pread(fd, slot + (size_t)i * 32768, st.st_size, 0); (void)g; }
      else { ssize_t g = pread(fd, buf, 69632, 0); if (mode == 2) memcpy(slot + (size_t)i * 32768, buf + 44, g - 44); if (mode == 4) copy_nt(slot + (size_t)i * 32768, buf + 44, g - 44); }
    }
    close(fd);
  }
  free(buf);
  if (use_dirfd) for (int d = 0; d < ndirs; ++d) close(dfd[d]);
  return 0;
}
int main(int argc, char** argv) {
  nthreads = atoi(argv[1]);
  const char* dir = argv[2];
  private_fds = argc > 3 ? atoi(argv[3]) : 0;
  private_cred = argc > 4 ? atoi(argv[4]) : 0;
  ndirs = argc > 5 ? atoi(argv[5]) : 1;
  use_dirfd = argc > 6 ? atoi(argv[6]) : 0;
  for (int d = 0; d < ndirs; ++d) { snprintf(dirs[d], 64, "%s/d%02d", dir, d); mkdir(dirs[d], 0755); }
  slot = aligned_alloc(4096, (size_t)N * 32768);
  memset(slot, 1, (size_t)N * 32768);
  char* data = malloc(SZ); memset(data, 7, SZ);
  for (int i = 0; i < N; ++i) {
    snprintf(names[i], 32, "f%05d.wav", i);
    snprintf(paths[i], 64, "%s/%s", dirs[i % ndirs], names[i]);
    int fd = open(paths[i], O_WRONLY | O_CREAT | O_TRUNC, 0644); if (write(fd, data, SZ) != SZ) return 1; close(fd);
  }
  for (mode = -1; mode < 5; ++mode) {
    double best = 1e9;
    for (int rep = 0; rep < 2; ++rep) {
      atomic_store(&next_i, 0);
      pthread_t th[64];
      double t0 = now();
      for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], 0, work, 0);
      for (int t = 0; t < nthreads; ++t) pthread_join(th[t], 0);
      double dt = now() - t0; if (dt < best) best = dt;
    }
    best /= REPS;
    printf("mode %d (%s): %.3f ms per 4096 files = %.2f M files/s, %.2f us/file/thread\n", mode,
           mode < 0 ? "threads only" : mode == 0 ? "open+close" : mode == 1 ? "open+pread68K+close" : mode == 2 ? "+memcpy to slot" : mode == 4 ? "+streaming-store copy to slot" : "open+fstat+pread direct+close",
           best * 1e3, N / best * 1e-6, best * 1e6 * nthreads / N);
  }
  return 0;
}
