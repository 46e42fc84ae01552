// Does one wave's ds_add_rtn_u32 serve lanes that hit the same address in ascending lane order?  (sort kernel, pass A)
// hipcc --offload-arch=gfx950 -O3 -o build/lds_atomic_order scripts/micro/lds_atomic_order.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t *bad, uint32_t rounds, uint32_t seed)
{
    __shared__ uint32_t cnt[4096];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (uint32_t i = tid; i < 4096; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    uint32_t x = seed ^ (blockIdx.x * 0x9E3779B9u) ^ (wave * 0x85EBCA6Bu);
    uint32_t errs = 0;
    for (uint32_t r = 0; r < rounds; r++) {
        x = x * 1664525u + 1013904223u; // same x in all lanes of the wave
        const uint32_t nkeys = 1u + ((x >> 8) % 40u);          // distinct addresses used this round
        uint32_t y = x ^ (lane * 0x27D4EB2Fu); y ^= y >> 15; y *= 0x2C1B3C6Du; y ^= y >> 12;
        const uint32_t key = y % nkeys;
        // each wave owns addresses with (a & 7) == wave & 7 pattern like the sort kernel: a = key*16 + wave (waves interleave in the same dwords' neighbours)
        const uint32_t a = (key * 16u + wave) & 4095u;
        const uint32_t half = (x >> 3) & 1u;
        const uint32_t old = atomicAdd(&cnt[a], half ? 65536u : 1u);
        // expected: lanes of this wave with the same (a) get consecutive values in lane order
        const uint32_t mine = half ? old >> 16 : old & 0xffffu;
        for (uint32_t l = 0; l < 64; l++) {
            const uint32_t al = __shfl(a, l), ml = __shfl(mine, l);
            if (l < lane && al == a && ml >= mine) errs++;
        }
    }
    if (errs) atomicAdd(bad, errs);
}
int main()
{
    uint32_t *bad; hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
    for (int it = 0; it < 8; it++) hipLaunchKernelGGL(k, dim3(1024), dim3(1024), 0, 0, bad, 400u, 12345u + it);
    uint32_t h = 1; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("lds atomic lane-order violations: %u\n", h);
    return 0;
}
