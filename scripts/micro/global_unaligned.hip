// Does gfx950 serve 8-byte global loads at 2-byte alignment (one instruction), and correctly?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__global__ void k(const uint16_t* s, uint64_t* out)
{
    const uint16_t* p = s + threadIdx.x * 3 + 1; // 2-byte aligned, mostly not 8-byte aligned
    uint64_t v;
    asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    out[threadIdx.x] = v;
}
int main()
{
    uint16_t h[512]; for (int i = 0; i < 512; i++) h[i] = (uint16_t)(i * 977 + 13);
    uint16_t* d; uint64_t* o; hipMalloc(&d, sizeof h); hipMalloc(&o, 64 * 8);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o);
    uint64_t r[64]; hipError_t e = hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; t++) { uint64_t want; memcpy(&want, h + t * 3 + 1, 8); if (want != r[t]) bad++; }
    printf("%s: unaligned global_load_dwordx2: %d bad of 64\n", hipGetErrorString(e), bad);
    return 0;
}
