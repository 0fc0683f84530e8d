// Dev microbenchmark (not a test): issue cost of VALU instructions the compositing kernels use, on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CHAINS 8
#define OPS(X) \
  X(0, "v_cndmask_b32_e32 %0, %0, %1, vcc", s) \
  X(1, "v_cndmask_b32_e64 %0, %0, %1, vcc", s) \
  X(2, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]", s) \
  X(3, "v_cndmask_b32_e64 %0, 0, %1, s[20:21]", s) \
  X(4, "v_cndmask_b32_e32 %0, 0, %1, vcc", s) \
  X(5, "v_cndmask_b32_e32 %0, %1, %2, vcc", s) \
  X(6, "v_cndmask_b32_e64 %0, %1, %2, s[20:21]", s) \
  X(7, "v_fma_f32 %0, %0, %1, %2", s) \
  X(8, "v_fma_f32 %0, %1, %2, %0", s) \
  X(9, "v_fmac_f32 %0, %1, %2", s) \
  X(10, "v_mul_f32 %0, %1, %2", s) \
  X(11, "v_max_f32 %0, %1, %2", s) \
  X(12, "v_add_f32 %0, %1, %2", s)
template <int MODE> __global__ void k(float *out, int iters, float a, float b)
{
    float s[CHAINS]; v2f p[CHAINS];
    for (int i = 0; i < CHAINS; i++) { s[i] = threadIdx.x * 0.001f + i; p[i] = (v2f){s[i], s[i] + 1.f}; }
    v2f av = {a, a}, bv = {b, b};
    float a2 = a + 1.f;
    asm volatile("s_mov_b64 s[20:21], -1\n s_mov_b32 s22, 0x3f000000\n s_mov_b64 vcc, -1" ::: "s20", "s21", "s22", "vcc");
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) {
#define X(M, STR, KIND) if (MODE == M) { if (#KIND[0] == 's') { if (M == 13) asm volatile(STR : "+v"(s[i]), "+v"(a2)); else asm volatile(STR : "+v"(s[i]) : "v"(a), "v"(b)); } else asm volatile(STR : "+v"(p[i]) : "v"(av), "v"(bv)); }
            OPS(X)
#undef X
        }
    }
    float r = a2; for (int i = 0; i < CHAINS; i++) r += s[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE> void run(const char *name, float *out, int wavesPerSimd)
{
    const int iters = 20000, blocks = 256 * wavesPerSimd;   // 256 threads = 4 waves per block, one block per CU per wave/SIMD
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0001f, 0.5f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * CHAINS * wavesPerSimd;
    printf("%-70s w/SIMD=%d %.3f ms -> %.2f cyc/instr/SIMD @2.4GHz\n", name, wavesPerSimd, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
}
int main()
{
    float *out; (void)hipMalloc(&out, 256 * 4 * 8 * 64 * sizeof(float) * 4);
    for (int w : {1, 4}) {
#define X(M, STR, KIND) run<M>(STR, out, w);
        OPS(X)
#undef X
    }
    return 0;
}
