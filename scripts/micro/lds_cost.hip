// LDS read throughput on gfx950 for the access shapes the match kernel uses: random addresses per lane,
// aligned vs unaligned, 1/2/4/8 bytes.  16 waves per CU, every CU busy; reports cycles per wave-instruction per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE> __global__ void __launch_bounds__(1024) k(uint32_t* out, long long* cyc, uint32_t seed)
{
    extern __shared__ uint8_t s[];
    for (int i = threadIdx.x; i < 65536 + 64; i += blockDim.x) s[i] = (uint8_t)(i * 131 + 7);
    __syncthreads();
    uint32_t x = (threadIdx.x + 1) * 2654435761u + seed + blockIdx.x, acc = 0;
    long long t0 = clock64();
#pragma unroll 4
    for (int it = 0; it < 4096; it++) {
        x = x * 1664525u + 1013904223u;
        uint32_t a = (x >> 8) & 0xFFFF;
        if (MODE == 0) acc += s[a];                                               // u8
        if (MODE == 1) acc += *reinterpret_cast<uint16_t*>(s + (a & ~1u));        // u16 aligned
        if (MODE == 2) acc += *reinterpret_cast<uint32_t*>(s + (a & ~3u));        // b32 aligned
        if (MODE == 3) acc += *reinterpret_cast<uint32_t*>(s + a);                // b32 unaligned
        if (MODE == 4) { uint64_t v = *reinterpret_cast<uint64_t*>(s + (a & ~7u)); acc += (uint32_t)v ^ (uint32_t)(v >> 32); } // b64 aligned
        if (MODE == 5) { uint64_t v = *reinterpret_cast<uint64_t*>(s + a); acc += (uint32_t)v ^ (uint32_t)(v >> 32); }        // b64 unaligned
        if (MODE == 6) { uint32_t i = a >> 2; uint32_t lo = reinterpret_cast<uint32_t*>(s)[i], hi = reinterpret_cast<uint32_t*>(s)[i + 1]; acc += __builtin_amdgcn_alignbyte(hi, lo, a & 3); } // 2 aligned dwords + alignbyte
        if (MODE == 7) acc += *reinterpret_cast<uint16_t*>(s + a);                // u16 unaligned
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name)
{
    uint32_t* out; long long* cyc; hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 64);
    k<MODE><<<256, 1024, 65536 + 64>>>(out, cyc, 1);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); k<MODE><<<256, 1024, 65536 + 64>>>(out, cyc, 2); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long c[256]; hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; i++) avg += c[i]; avg /= 256;
    // per CU: 16 waves x 4096 instructions
    printf("%-28s %8.3f ms  %7.2f clk per wave-instruction per CU (s_memtime %0.f)\n", name, ms, avg / (16.0 * 4096), avg);
    hipFree(out); hipFree(cyc);
}
int main()
{
    run<0>("u8"); run<1>("u16 aligned"); run<7>("u16 unaligned"); run<2>("b32 aligned"); run<3>("b32 unaligned");
    run<4>("b64 aligned"); run<5>("b64 unaligned"); run<6>("2x b32 aligned + alignbyte");
    return 0;
}
