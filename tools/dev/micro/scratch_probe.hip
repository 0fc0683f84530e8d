// Developer probe: does a kernel whose vector registers the COMPILER spills to scratch keep its values on this
// box?  Every thread keeps N lane-unique values alive across a long dependent loop under a register cap that forces spills, then checks
// them.  Reports the spilled-register count is in the build remarks; the program prints how many (block, lane) positions came back
// wrong and where.      hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage scratch_probe.hip -o scratch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N 96
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void probe(int P, const float *__restrict__ in, unsigned *__restrict__ bad, int spin)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    float v[N];
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = in[(idx + 7 * i) % P] * (float)(i + 1);
    // something long and dependent in between, so that other waves of the CU run (and spill) meanwhile
    float t = in[idx];
    for (int k = 0; k < spin; k++) t = __builtin_fmaf(t, 1.0000001f, 1e-9f);
    unsigned wrong = 0;
#pragma unroll
    for (int i = 0; i < N; i++) wrong += (v[i] != in[(idx + 7 * i) % P] * (float)(i + 1)) ? 1u : 0u;
    bad[idx] = wrong + (t == 123.f ? 1u : 0u);
}
// the same with a SMALL scratch footprint (16 bytes per lane, what the spilling preprocess_bwd build had): a volatile private array
__global__ __launch_bounds__(256) void probe_small(int P, const float *__restrict__ in, unsigned *__restrict__ bad, int spin)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    volatile float priv[4];
    for (int i = 0; i < 4; i++) priv[i] = in[(idx + 7 * i) % P] * (float)(i + 1);
    float t = in[idx];
    for (int k = 0; k < spin; k++) t = __builtin_fmaf(t, 1.0000001f, 1e-9f);
    unsigned wrong = 0;
    for (int i = 0; i < 4; i++) wrong += (priv[i] != in[(idx + 7 * i) % P] * (float)(i + 1)) ? 1u : 0u;
    bad[idx] = wrong + (t == 123.f ? 1u : 0u);
}
int main()
{
    const int P = 100000;
    std::vector<float> h(P);
    for (int i = 0; i < P; i++) h[i] = 1.0f + (float)(i % 9973) * 1e-3f;
    float *d; unsigned *b;
    hipMalloc(&d, P * 4); hipMalloc(&b, P * 4);
    hipMemcpy(d, h.data(), P * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 6; rep++) {
        hipMemset(b, 0, P * 4);
        if (rep < 3) hipLaunchKernelGGL(probe, dim3((P + 255) / 256), dim3(256), 0, 0, P, d, b, 2000);
        else hipLaunchKernelGGL(probe_small, dim3((P + 255) / 256), dim3(256), 0, 0, P, d, b, 2000);
        hipDeviceSynchronize();
        std::vector<unsigned> hb(P);
        hipMemcpy(hb.data(), b, P * 4, hipMemcpyDeviceToHost);
        long nbad = 0; int first = -1; long lanes[64] = { 0 };
        for (int i = 0; i < P; i++) if (hb[i]) { nbad++; if (first < 0) first = i; lanes[i & 63]++; }
        printf("rep %d: %ld of %d threads read back a wrong spilled value; first %d; per lane:", rep, nbad, P, first);
        for (int l = 0; l < 64; l++) if (lanes[l]) printf(" %d:%ld", l, lanes[l]);
        printf("\n");
    }
    return 0;
}
