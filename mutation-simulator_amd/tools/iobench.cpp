// Page-cache output micro-benchmark behind csrc/file_io.hip: how fast can 1.2 GiB land in a fresh file, and by which route?
//   g++ -O2 -pthread -o /tmp/iobench mutation-simulator_amd/tools/iobench.cpp && /tmp/iobench /dev/shm && /tmp/iobench /tmp
// A: map the span, copy into it (first-touch faults), unmap.  B: fallocate + map + populate + copy + unmap.  C: write() in
// 8 MiB pieces.  D: fallocate + pwrite.  E: a thread allocating ahead of the copy.  F: A with 4 threads on 4 slices of ONE
// file.  G: two files, a thread each.  H: MAP_POPULATE.  (profiles/r04_iobench.txt holds the MI355X box's numbers.)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>
using clk = std::chrono::steady_clock;
static double ms(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }
static const size_t N = 1200ull << 20, CH = 8u << 20;
static char *src;
static void fillmap(char *m, size_t n) { for (size_t o = 0; o < n; o += CH) memcpy(m + o, src, n - o < CH ? n - o : CH); }
int main(int argc, char **argv) {
    std::string dir = argc > 1 ? argv[1] : "/dev/shm";
    src = (char *)aligned_alloc(4096, CH); memset(src, 'A', CH);
    auto path = [&](const char *n) { return dir + "/iob_" + n; };
    auto run = [&](const char *name, auto f) {
        std::string p = path(name);
        unlink(p.c_str());
        int fd = open(p.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        auto t0 = clk::now(); f(fd); auto t1 = clk::now();
        close(fd); unlink(p.c_str());
        printf("%-46s %8.1f ms  %6.2f GB/s\n", name, ms(t0, t1), N / 1e6 / ms(t0, t1));
        fflush(stdout);
    };
    for (int rep = 0; rep < 2; rep++) {
    run("A ftruncate+mmap+memcpy(faults)+munmap", [&](int fd) {
        ftruncate(fd, N); char *m = (char *)mmap(0, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        auto a = clk::now(); fillmap(m, N); auto b = clk::now(); munmap(m, N); auto c = clk::now();
        printf("   copy %.1f munmap %.1f\n", ms(a, b), ms(b, c)); });
    run("B fallocate+mmap+populate+memcpy+munmap", [&](int fd) {
        auto a = clk::now(); if (fallocate(fd, 0, 0, N)) perror("fallocate"); auto b = clk::now();
        char *m = (char *)mmap(0, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (madvise(m, N, 23)) perror("populate"); auto c = clk::now(); fillmap(m, N); auto d = clk::now(); munmap(m, N); auto e = clk::now();
        printf("   fallocate %.1f populate %.1f copy %.1f munmap %.1f\n", ms(a, b), ms(b, c), ms(c, d), ms(d, e)); });
    run("C write() 8MB pieces", [&](int fd) { for (size_t o = 0; o < N; o += CH) if (write(fd, src, CH) != (ssize_t)CH) perror("write"); });
    run("D fallocate + pwrite", [&](int fd) {
        auto a = clk::now(); fallocate(fd, 0, 0, N); auto b = clk::now();
        for (size_t o = 0; o < N; o += CH) if (pwrite(fd, src, CH, o) != (ssize_t)CH) perror("pwrite");
        printf("   fallocate %.1f pwrite %.1f\n", ms(a, b), ms(b, clk::now())); });
    run("E fallocate(thread, 64MB steps) || memcpy behind it", [&](int fd) {
        ftruncate(fd, N); char *m = (char *)mmap(0, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        volatile size_t done = 0;
        std::thread th([&] { for (size_t o = 0; o < N; o += (64u << 20)) { fallocate(fd, 0, o, 64u << 20); madvise(m + o, 64u << 20, 23); done = o + (64u << 20); } });
        for (size_t o = 0; o < N; o += CH) { while (done < o + CH) std::this_thread::yield(); memcpy(m + o, src, CH); }
        th.join(); auto b = clk::now(); munmap(m, N); printf("   munmap %.1f\n", ms(b, clk::now())); });
    run("F A with 4 threads on 4 slices (same file)", [&](int fd) {
        ftruncate(fd, N); char *m = (char *)mmap(0, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        std::vector<std::thread> t; for (int i = 0; i < 4; i++) t.emplace_back([&, i] { fillmap(m + i * (N / 4), N / 4); });
        for (auto &x : t) x.join(); munmap(m, N); });
    run("G two files, one thread each (A), 600MB each", [&](int fd) {
        std::string p2 = path("second"); int fd2 = open(p2.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        auto one = [&](int f) { ftruncate(f, N / 2); char *m = (char *)mmap(0, N / 2, PROT_READ | PROT_WRITE, MAP_SHARED, f, 0); fillmap(m, N / 2); munmap(m, N / 2); };
        std::thread a(one, fd), b(one, fd2); a.join(); b.join(); close(fd2); unlink(p2.c_str()); });
    run("H MAP_POPULATE on ftruncated file + memcpy", [&](int fd) {
        ftruncate(fd, N); auto a = clk::now(); char *m = (char *)mmap(0, N, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, fd, 0); auto b = clk::now();
        fillmap(m, N); auto c = clk::now(); munmap(m, N); printf("   mmap+populate %.1f copy %.1f munmap %.1f\n", ms(a, b), ms(b, c), ms(c, clk::now())); });
    }
    return 0;
}
