// Dev probe (not a test): do the lanes of ONE ds_add_rtn_u32 instruction that hit the same LDS word receive their pre-op values in
// ascending lane order?  (If so, an LDS atomic with return ranks the items of a wave stably -- 1 LDS instruction per item instead of
// 5 VALU per digit bit of ballot ranking.)  Random address patterns with heavy collisions, millions of trials.
//   hipcc --offload-arch=gfx950 -O2 -o lds_atomic_order.bin lds_atomic_order.hip && ./lds_atomic_order.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(uint32_t seed, int trials, int ndigits, unsigned long long *bad, unsigned long long *total)
{
    __shared__ uint32_t cnt[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = seed ^ (blockIdx.x * 0x9E3779B9u) ^ (threadIdx.x * 0x85EBCA6Bu);
    unsigned long long nb = 0;
    for (int t = 0; t < trials; t++) {
        for (int i = lane; i < 256; i += 64) cnt[wave][i] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        x = x * 1664525u + 1013904223u;
        const uint32_t d = (x >> 16) % (uint32_t)ndigits;
        // expected: rank among the lanes with the same digit = number of lower lanes with that digit
        uint32_t expect = 0;
        for (int l = 0; l < 64; l++) { const uint32_t dl = __shfl(d, l, 64); if (l < lane && dl == d) expect++; }
        const uint32_t got = atomicAdd(&cnt[wave][d], 1u);
        // second instruction on top: continues behind the first one's counts
        const uint32_t got2 = atomicAdd(&cnt[wave][d], 1u);
        uint32_t same = 0;
        for (int l = 0; l < 64; l++) { const uint32_t dl = __shfl(d, l, 64); if (dl == d) same++; }
        if (got != expect || got2 != same + expect) nb++;
    }
    atomicAdd(bad, nb);
    atomicAdd(total, (unsigned long long)trials);
}
int main()
{
    unsigned long long *d; (void)hipMalloc(&d, 16); 
    for (int nd : {1, 2, 3, 7, 16, 64, 128, 256}) {
        (void)hipMemset(d, 0, 16);
        hipLaunchKernelGGL(probe, dim3(1024), dim3(256), 0, 0, 12345u + nd, 2000, nd, d, d + 1);
        unsigned long long h[2]; (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("digits %3d: %llu lane-trials, %llu out of lane order\n", nd, h[1], h[0]);
    }
    return 0;
}
