// io_probe.cpp — how fast can T threads of this box (a) create one large file and (b) stream one back, by which system
// interface?  Decides the strategy of gnx_io.cpp's writers (write_blocks) and reader (load_text).
//   g++ -O2 -std=c++17 -pthread io_probe.cpp -o io_probe && ./io_probe /dev/shm 2   (directory, GiB)
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class F>
static void par(int T, F&& f) {
  std::vector<std::thread> th;
  for (int t = 1; t < T; ++t) th.emplace_back([&, t] { f(t); });
  f(0);
  for (auto& x : th) x.join();
}

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "/dev/shm";
  const size_t S = (size_t)(argc > 2 ? atof(argv[2]) * (1 << 30) : (size_t)1 << 30);
  const std::string path = dir + "/gnx_io_probe.bin";
  const size_t CH = (size_t)1 << 20;  // 1 MiB blocks, claimed dynamically
  std::vector<char> src(CH);
  for (size_t i = 0; i < CH; ++i) src[i] = (char)('0' + (i * 7) % 10);
  const int Ts[] = {1, 4, 16, 32, 64, 128, 256};
  const size_t nblk = S / CH;
  // ---- writes ----------------------------------------------------------------------------------------------------------
  for (int mode = 0; mode < 3; ++mode) {  // 0 pwrite, 1 mmap shared, 2 mmap shared + MADV_POPULATE_WRITE per block
    for (int T : Ts) {
      unlink(path.c_str());
      const int fd = open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
      if (fd < 0) return 1;
      const double t0 = now();
      char* m = nullptr;
      if (mode >= 1) {
        if (ftruncate(fd, (off_t)S) != 0) return 2;
        m = (char*)mmap(nullptr, S, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) return 3;
      }
      std::atomic<size_t> next{0};
      par(T, [&](int) {
        for (;;) {
          const size_t b = next.fetch_add(1);
          if (b >= nblk) break;
          if (mode == 0) {
            if (pwrite(fd, src.data(), CH, (off_t)(b * CH)) != (ssize_t)CH) abort();
          } else {
            if (mode == 2) madvise(m + b * CH, CH, 23 /* MADV_POPULATE_WRITE */);
            memcpy(m + b * CH, src.data(), CH);
          }
        }
      });
      const double t1 = now();
      if (m) munmap(m, S);
      close(fd);
      const double t2 = now();
      printf("write mode=%s T=%3d  %.3f s (+%.3f s unmap/close)  %.2f GB/s\n", mode == 0 ? "pwrite" : mode == 1 ? "mmap" : "mmap+populate", T,
             t1 - t0, t2 - t1, S / (t2 - t0) / 1e9);
      fflush(stdout);
    }
  }
  // ---- reads (file left by the last write) ---------------------------------------------------------------------------------
  for (int mode = 0; mode < 4; ++mode) {  // 0 mmap private, 1 mmap + MADV_POPULATE_READ per block, 2 pread into a thread buffer, 3 mmap + hugepage advice
    for (int T : Ts) {
      const int fd = open(path.c_str(), O_RDONLY);
      const double t0 = now();
      char* m = nullptr;
      if (mode != 2) {
        m = (char*)mmap(nullptr, S, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return 4;
        if (mode == 3) madvise(m, S, MADV_HUGEPAGE);
      }
      std::atomic<size_t> next{0};
      std::atomic<size_t> total{0};
      par(T, [&](int) {
        std::vector<char> buf(mode == 2 ? CH : 0);
        size_t cnt = 0;
        for (;;) {
          const size_t b = next.fetch_add(1);
          if (b >= nblk) break;
          const char* p;
          if (mode == 2) {
            if (pread(fd, buf.data(), CH, (off_t)(b * CH)) != (ssize_t)CH) abort();
            p = buf.data();
          } else {
            p = m + b * CH;
            if (mode == 1) madvise((void*)p, CH, 22 /* MADV_POPULATE_READ */);
          }
          const char* q = p;
          const char* e = p + CH;
          while ((q = (const char*)memchr(q, '7', (size_t)(e - q)))) {
            ++cnt;
            ++q;
          }
        }
        total += cnt;
      });
      const double t1 = now();
      if (m) munmap(m, S);
      close(fd);
      const double t2 = now();
      printf("read  mode=%s T=%3d  %.3f s (+%.3f s unmap)  %.2f GB/s  (%zu)\n",
             mode == 0 ? "mmap" : mode == 1 ? "mmap+populate" : mode == 2 ? "pread" : "mmap+thp", T, t1 - t0, t2 - t1, S / (t2 - t0) / 1e9, total.load());
      fflush(stdout);
    }
  }
  unlink(path.c_str());
  return 0;
}
