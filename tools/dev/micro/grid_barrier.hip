// micro-benchmark: cost of a device-wide barrier (atomic counter + spin, agent scope) between phases of a persistent kernel on MI355X
// build + run:  hipcc -O3 --offload-arch=gfx950 -o /tmp/grid_barrier tools/dev/micro/grid_barrier.hip && /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void grid_barrier(unsigned *bar, unsigned target)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}
__global__ void k(unsigned *bar, int nsync, float *data, int touch)
{
    float acc = 0.f;
    for (int i = 0; i < nsync; i++) {
        if (touch) { data[(size_t)blockIdx.x * 256 + threadIdx.x] += 1.0f; }   // a little traffic per phase
        grid_barrier(bar, (unsigned)(i + 1) * gridDim.x);
    }
    if (acc == 123.f) data[0] = acc;
}
int main()
{
    unsigned *bar; float *data;
    hipMalloc(&bar, 4); hipMalloc(&data, 4096 * 256 * 4); hipMemset(data, 0, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, 256, 0);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("CUs %d, max blocks/CU %d\n", p.multiProcessorCount, occ);
    for (int grid : {256, 512, 1024}) for (int touch : {0, 1}) for (int nsync : {1, 11, 41}) {
        hipMemset(bar, 0, 4);
        void *args[] = { &bar, &nsync, &data, &touch };
        hipEventRecord(e0);
        hipError_t err = hipLaunchCooperativeKernel((void *)k, dim3(grid), dim3(256), args, 0, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("grid %4d touch %d nsync %2d: %8.2f us  (%s)\n", grid, touch, nsync, ms * 1e3, hipGetErrorString(err));
    }
    return 0;
}
