// Does gfx950 LDS serve unaligned 2/4/8-byte reads correctly (and at what cost)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__global__ void k(const uint8_t* in, uint32_t* out32, uint16_t* out16, uint64_t* out64, long long* cyc)
{
    __shared__ __attribute__((aligned(16))) uint8_t s[4096 + 16];
    for (int i = threadIdx.x; i < 4096 + 16; i += blockDim.x) s[i] = in[i];
    __syncthreads();
    uint32_t a = threadIdx.x * 7 + 1; // odd-ish addresses
    long long t0 = clock64();
    uint32_t v32 = *reinterpret_cast<const uint32_t*>(s + a);
    uint16_t v16 = *reinterpret_cast<const uint16_t*>(s + a);
    uint64_t v64 = *reinterpret_cast<const uint64_t*>(s + a);
    long long t1 = clock64();
    out32[threadIdx.x] = v32; out16[threadIdx.x] = v16; out64[threadIdx.x] = v64;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    uint8_t h[4096 + 16]; for (int i = 0; i < 4096 + 16; i++) h[i] = (uint8_t)(i * 131 + 7);
    uint8_t* d; uint32_t* o32; uint16_t* o16; uint64_t* o64; long long* cyc;
    hipMalloc(&d, sizeof h); hipMalloc(&o32, 64 * 4); hipMalloc(&o16, 64 * 2); hipMalloc(&o64, 64 * 8); hipMalloc(&cyc, 8);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o32, o16, o64, cyc);
    uint32_t r32[64]; uint16_t r16[64]; uint64_t r64[64]; long long c;
    hipError_t e = hipDeviceSynchronize();
    printf("sync: %s\n", hipGetErrorString(e));
    hipMemcpy(r32, o32, sizeof r32, hipMemcpyDeviceToHost); hipMemcpy(r16, o16, sizeof r16, hipMemcpyDeviceToHost); hipMemcpy(r64, o64, sizeof r64, hipMemcpyDeviceToHost); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; t++) { uint32_t a = t * 7 + 1, e32; uint16_t e16; uint64_t e64; memcpy(&e32, h + a, 4); memcpy(&e16, h + a, 2); memcpy(&e64, h + a, 8);
        if (e32 != r32[t] || e16 != r16[t] || e64 != r64[t]) { if (bad < 5) printf("lane %d addr %u: 32 %08x/%08x 16 %04x/%04x\n", t, a, r32[t], e32, r16[t], e16); bad++; } }
    printf("unaligned LDS reads: %s (%d bad), cycles %lld\n", bad ? "WRONG" : "correct", bad, c);
    return 0;
}
