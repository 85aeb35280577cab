// GPU box: how fast do N threads bring a file that sits in the page cache into PINNED memory -- by pread (what the read feed does), by
// memcpy out of a mapping of the file, by a copy with non-temporal stores out of the mapping?   (files -> table is bound by this copy:
// DESIGN.md 6, tools/probes/ingest_stages.py)
//   hipcc -O3 -o /tmp/read_probe tools/probes/read_probe.hip -lpthread && /tmp/read_probe [GB = 6] [dir = /tmp]
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <immintrin.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const size_t gb = argc > 1 ? (size_t)atoi(argv[1]) : 6;
    const std::string dir = argc > 2 ? argv[2] : "/tmp";
    const size_t total = gb << 30, BUF = (size_t)128 << 20;
    const std::string path = dir + "/read_probe.bin";
    {   // distinct bytes, written by 8 threads
        const int fd = open(path.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)total) != 0) { perror("create"); return 1; }
        const double t0 = now();
        std::vector<std::thread> th;
        for (int i = 0; i < 8; ++i) th.emplace_back([&, i]() {
            std::vector<uint64_t> b((size_t)(8 << 20) / 8);
            uint64_t x = 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1);
            for (size_t off = (size_t)i * (8 << 20); off < total; off += (size_t)8 * (8 << 20)) {
                for (auto &w : b) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; w = x; }
                if (pwrite(fd, b.data(), 8 << 20, (off_t)off) != (8 << 20)) { perror("pwrite"); exit(1); }
            }
        });
        for (auto &t : th) t.join();
        close(fd);
        printf("wrote %zu GB in %.1f s\n", gb, now() - t0);
    }
    char *pin[2];
    if (hipHostMalloc((void **)&pin[0], BUF, hipHostMallocDefault) != hipSuccess || hipHostMalloc((void **)&pin[1], BUF, hipHostMallocDefault) != hipSuccess) { printf("hipHostMalloc failed\n"); return 1; }
    memset(pin[0], 1, BUF); memset(pin[1], 1, BUF);
    const int fd = open(path.c_str(), O_RDONLY);
    const char *map = (const char *)mmap(nullptr, total, PROT_READ, MAP_SHARED, fd, 0);
    if (map == MAP_FAILED) { perror("mmap"); return 1; }
    madvise((void *)map, total, MADV_SEQUENTIAL);
    auto run = [&](const char *label, int nth, int mode) {
        const double t0 = now();
        uint64_t sink = 0;
        for (size_t off = 0, it = 0; off < total; off += BUF, ++it) {
            char *dst = pin[it & 1];
            const size_t todo = std::min(BUF, total - off), part = (todo + nth - 1) / nth;
            std::vector<std::thread> th;
            for (int i = 0; i < nth; ++i) th.emplace_back([&, i]() {
                size_t lo = (size_t)i * part, hi = std::min(todo, lo + part);
                if (mode == 0) { while (lo < hi) { const ssize_t k = pread(fd, dst + lo, hi - lo, (off_t)(off + lo)); if (k <= 0) exit(2); lo += (size_t)k; } }
                else if (mode == 1) memcpy(dst + lo, map + off + lo, hi - lo);
                else {      // 64-byte non-temporal stores, the source touched a few pages ahead
                    const char *s = map + off;
                    for (size_t p = lo; p < hi; p += 64) {
                        if ((p & 4095) == 0 && p + 16384 < hi) __builtin_prefetch(s + p + 16384);
                        const __m256i a = _mm256_loadu_si256((const __m256i *)(s + p)), b = _mm256_loadu_si256((const __m256i *)(s + p + 32));
                        _mm256_stream_si256((__m256i *)(dst + p), a);
                        _mm256_stream_si256((__m256i *)(dst + p + 32), b);
                    }
                    _mm_sfence();
                }
            });
            for (auto &t : th) t.join();
            sink += (unsigned char)dst[todo / 2];
        }
        const double dt = now() - t0;
        printf("%-44s %2d threads: %.2f s = %.1f GB/s  (%llu)\n", label, nth, dt, (double)total / dt / 1e9, (unsigned long long)sink);
        fflush(stdout);
    };
    for (int rep = 0; rep < 2; ++rep) {
        for (int nth : {8, 16, 32}) {
            run("pread -> pinned", nth, 0);
            run("memcpy from the mapping -> pinned", nth, 1);
            run("non-temporal copy from the mapping -> pinned", nth, 2);
        }
    }
    munmap((void *)map, total);
    close(fd);
    unlink(path.c_str());
    return 0;
}
