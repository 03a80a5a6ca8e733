// tools/tmpfs_write_probe.cpp — how fast N threads fill ONE fresh file on /dev/shm: 0 pwrite, 1 stores through a shared mapping, 2 MADV_POPULATE_WRITE + stores, 3 fallocate + pwrite
// (16 MB chunks; g++ -O2 -pthread tools/tmpfs_write_probe.cpp -o /tmp/wprobe && /tmp/wprobe GB THREADS MODE).  Measurement aid for index_on_disk_inclusive.
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>
#include <chrono>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
int main(int argc, char **argv) {
    const size_t GB = argc > 1 ? atol(argv[1]) : 4; const int T = argc > 2 ? atoi(argv[2]) : 8; const int mode = argc > 3 ? atoi(argv[3]) : 0;
    const size_t total = GB << 30, chunk = 16u << 20;
    std::vector<char> src(chunk * T);
    for (size_t i = 0; i < src.size(); i += 4096) src[i] = (char)i;
    const char *path = "/dev/shm/wtest.bin";
    unlink(path);
    int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (mode >= 1) { if (ftruncate(fd, total)) return 1; }
    char *map = nullptr;
    if (mode >= 1) { map = (char *)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); if (map == MAP_FAILED) { perror("mmap"); return 1; } }
    std::atomic<size_t> next(0);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([&, t]() {
        const char *s = src.data() + chunk * t;
        for (;;) {
            const size_t off = next.fetch_add(chunk);
            if (off >= total) break;
            if (mode == 0) { if (pwrite(fd, s, chunk, off) != (ssize_t)chunk) { perror("pwrite"); exit(1); } }
            else if (mode == 1) memcpy(map + off, s, chunk);
            else if (mode == 2) { if (madvise(map + off, chunk, MADV_POPULATE_WRITE)) { perror("madvise"); exit(1); } memcpy(map + off, s, chunk); }
            else if (mode == 3) { if (fallocate(fd, 0, off, chunk)) { perror("fallocate"); exit(1); } if (pwrite(fd, s, chunk, off) != (ssize_t)chunk) exit(1); }
        }
    });
    for (auto &x : th) x.join();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("mode %d, %zu GB, %d threads: %.2f s = %.2f GB/s\n", mode, GB, T, s, GB / s * 1.073741824);
    close(fd); unlink(path);
    return 0;
}
