// How many scattered 64-byte sectors per second the memory system of an MI355X delivers, as a function of the size of the region they are
// scattered over and of the number of requests in flight.  The lane-per-chunk LZ77 loop of levels 1-3 (zgpu_lz_serial.hip) is a dependent chain of
// such requests per lane with every chunk of a 4 GiB call in flight (65 536 lanes, ~30 GB touched); it moves ~1.2 TB/s of sectors.  Is that the memory?
//
//   hipcc --offload-arch=gfx950 -O3 -o random_sectors random_sectors.hip && ./random_sectors
//
// (region sizes are powers of two: the offset is a mask.)  Every active lane walks a chain: the next address is a function of the value just loaded (16 bytes at a 64-byte-aligned random offset), `par`
// independent chains per lane.  Grid: `waves` waves of 64 lanes of which `active` take part (the serial kernel runs 16 lanes per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

template <int PAR> __global__ void __launch_bounds__(64) chase(const uint4 *__restrict__ buf, uint64_t sectors, int iters, uint32_t active, uint32_t *out)
{
    if (threadIdx.x >= active) return;
    uint64_t x[PAR];
    for (int k = 0; k < PAR; k++) x[k] = (blockIdx.x * 64ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + k * 0xD1B54A32D192ED03ull + 1;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        uint4 v[PAR];
#pragma unroll
        for (int k = 0; k < PAR; k++) v[k] = buf[((x[k] >> 7) & (sectors - 1)) * 4];
#pragma unroll
        for (int k = 0; k < PAR; k++) { x[k] = x[k] * 6364136223846793005ull + 1442695040888963407ull + v[k].x; x[k] ^= x[k] >> 29; acc += v[k].y; }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main()
{
    const uint64_t max_bytes = 64ull << 30;
    uint4 *buf; uint32_t *out;
    if (hipMalloc(&buf, max_bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, max_bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const uint64_t sizes[] = {64ull << 20, 256ull << 20, 1ull << 30, 4ull << 30, 16ull << 30, 64ull << 30};
    struct { uint32_t waves, active; int par; } cfg[] = {{4096, 16, 1}, {4096, 64, 1}, {16384, 64, 1}, {4096, 64, 4}, {16384, 64, 4}};
    printf("%10s %8s %7s %4s %12s %10s %12s\n", "region", "waves", "lanes", "par", "in flight", "TB/s", "ns/request");
    for (uint64_t sz : sizes) for (auto &c : cfg) {
        const int iters = 400;
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(a);
            if (c.par == 1) hipLaunchKernelGGL(chase<1>, dim3(c.waves), dim3(64), 0, 0, buf, sz / 64, iters, c.active, out);
            else hipLaunchKernelGGL(chase<4>, dim3(c.waves), dim3(64), 0, 0, buf, sz / 64, iters, c.active, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        const double inflight = (double)c.waves * c.active * c.par, reqs = inflight * iters;
        printf("%7llu MiB %8u %7u %4d %12.0f %10.3f %12.0f\n", (unsigned long long)(sz >> 20), c.waves, c.active, c.par, inflight, reqs * 64 / (best * 1e-3) / 1e12, best * 1e6 / iters);
        fflush(stdout);
    }
    return 0;
}
