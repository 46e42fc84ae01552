// Device -> pageable host memory: one hipMemcpy against the same bytes cut into slices copied by several threads on streams of their own.
// (The runtime stages a pageable copy through pinned buffers and a CPU copy; one thread does not fill the link.)
//   hipcc --offload-arch=gfx950 -O3 -o d2h_threads d2h_threads.hip -lpthread && ./d2h_threads
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

int main()
{
    const size_t n = 1ull << 30;
    uint8_t *d; hipMalloc(&d, n); hipMemset(d, 7, n);
    uint8_t *h = static_cast<uint8_t *>(malloc(n)); memset(h, 1, n);
    uint8_t *pin; hipHostMalloc(reinterpret_cast<void **>(&pin), n, hipHostMallocDefault);
    hipStream_t st[16];
    for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    auto now = [] { return std::chrono::steady_clock::now(); };
    for (int dir = 0; dir < 2; dir++) for (int nt : {1, 2, 4, 8, 16}) {
        double best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            const auto t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < nt; t++) th.emplace_back([&, t] {
                const size_t lo = n / nt * t, len = n / nt;
                if (dir == 0) hipMemcpyAsync(h + lo, d + lo, len, hipMemcpyDeviceToHost, st[t]); else hipMemcpyAsync(d + lo, h + lo, len, hipMemcpyHostToDevice, st[t]);
                hipStreamSynchronize(st[t]);
            });
            for (auto &x : th) x.join();
            const double s = std::chrono::duration<double>(now() - t0).count();
            if (s < best) best = s;
        }
        printf("%s pageable, 1 GiB, %2d thread(s): %6.1f ms = %5.1f GB/s\n", dir == 0 ? "D2H" : "H2D", nt, best * 1e3, n / best / 1e9);
    }
    for (int dir = 0; dir < 2; dir++) {
        double best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            const auto t0 = now();
            if (dir == 0) hipMemcpyAsync(pin, d, n, hipMemcpyDeviceToHost, st[0]); else hipMemcpyAsync(d, pin, n, hipMemcpyHostToDevice, st[0]);
            hipStreamSynchronize(st[0]);
            const double s = std::chrono::duration<double>(now() - t0).count();
            if (s < best) best = s;
        }
        printf("%s pinned,   1 GiB:               %6.1f ms = %5.1f GB/s\n", dir == 0 ? "D2H" : "H2D", best * 1e3, n / best / 1e9);
    }
    return 0;
}
