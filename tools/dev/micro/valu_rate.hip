// Dev microbenchmark (not a test): issue cost of the VALU instruction forms the compositing kernels are made of, on gfx950,
// as a function of the number of waves per SIMD.  Round 6: re-measures round 3's table (pk_rate.hip) with the packed forms,
// the transcendental / plain mixes and the DPP forms, before the compositing steps are re-packed.
//   hipcc --offload-arch=gfx950 -O2 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CH 8
// every body is one "unit" of N instructions applied to chain i: s = scalar chain register, p = register pair, a / b plain inputs
#define OPS(X) \
  X(0, 1, "v_fma_f32 %0, %0, %2, %3") \
  X(1, 1, "v_mul_f32 %0, %0, %2") \
  X(2, 1, "v_add_f32 %0, %0, %2") \
  X(3, 1, "v_pk_fma_f32 %1, %1, %4, %5") \
  X(4, 1, "v_pk_fma_f32 %1, %1, %4, %5 op_sel_hi:[1,0,1]") \
  X(5, 1, "v_pk_mul_f32 %1, %1, %4") \
  X(6, 1, "v_pk_add_f32 %1, %1, %4") \
  X(7, 1, "v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf") \
  X(8, 1, "v_add_f32_dpp %0, %0, %2 row_shr:1 row_mask:0xf bank_mask:0xf") \
  X(9, 1, "v_exp_f32 %0, %0") \
  X(10, 1, "v_rcp_f32 %0, %0") \
  X(11, 1, "v_cmp_le_u32_e64 s[20:21], %0, %2") \
  X(12, 1, "v_cndmask_b32_e64 %0, %0, %2, s[22:23]") \
  X(13, 1, "v_min_f32 %0, %0, %2") \
  X(14, 1, "v_mov_b32 %0, %2") \
  X(15, 2, "v_exp_f32 %0, %0\n v_fma_f32 %2, %2, %3, %3") \
  X(16, 4, "v_exp_f32 %0, %0\n v_fma_f32 %2, %2, %3, %3\n v_fma_f32 %3, %3, %2, %2\n v_mul_f32 %2, %2, %3") \
  X(17, 4, "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %2, %2, %3, %3\n v_fma_f32 %3, %3, %2, %2\n v_mul_f32 %2, %2, %3") \
  X(18, 2, "v_pk_fma_f32 %1, %1, %4, %5\n v_fma_f32 %0, %0, %2, %3") \
  X(19, 3, "v_pk_fma_f32 %1, %1, %4, %5\n v_fma_f32 %0, %0, %2, %3\n v_mul_f32 %2, %2, %3") \
  X(20, 2, "v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_fma_f32 %2, %2, %3, %3") \
  X(21, 2, "v_cmp_le_u32_e64 s[20:21], %0, %2\n v_fma_f32 %2, %2, %3, %3") \
  X(22, 1, "v_lshl_or_b32 %0, %0, 4, %2") \
  X(23, 1, "v_max_u32 %0, %0, %2") \
  X(24, 1, "v_sub_f32 %0, %0, %2") \
  X(25, 1, "v_pk_mul_f32 %1, %1, %4 op_sel_hi:[1,0]") \
  X(26, 1, "v_cmp_le_u32_e32 vcc, %0, %2") \
  X(27, 1, "v_permlane32_swap_b32 %0, %2") \
  X(28, 1, "v_fmac_f32 %0, %2, %3") \
  X(29, 1, "v_fma_f32 %0, %0, s24, %3") \
  X(30, 1, "v_pk_fma_f32 %1, %1, s[24:25], %5") \
  X(31, 1, "v_mul_f32 %0, s24, %0")

template <int MODE> __global__ void k(float *out, int iters, float a, float b)
{
    float s[CH]; v2f p[CH];
    for (int i = 0; i < CH; i++) { s[i] = 0.5f + threadIdx.x * 0.0001f + 0.01f * i; p[i] = (v2f){s[i], s[i] * 0.5f}; }
    v2f av = {a, a}, bv = {b, b};
    float a2 = a, b2 = b;
    asm volatile("s_mov_b64 s[20:21], -1\n s_mov_b64 s[22:23], 0x5555\n s_mov_b32 s24, 0x3f7fff00\n s_mov_b32 s25, 0x3f7fff00\n s_mov_b64 vcc, -1" ::: "s20", "s21", "s22", "s23", "s24", "s25", "vcc");
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < CH; i++) {
#define X(M, N, STR) if (MODE == M) asm volatile(STR : "+v"(s[i]), "+v"(p[i]), "+v"(a2), "+v"(b2) : "v"(av), "v"(bv) : "s20", "s21", "vcc");
            OPS(X)
#undef X
        }
    }
    float r = a2 + b2; for (int i = 0; i < CH; i++) r += s[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
static int g_cus = 256;
template <int MODE> void run(const char *name, int n, float *out, int wavesPerSimd)
{
    const int iters = 8000, blocks = g_cus * wavesPerSimd;   // 256 threads = 4 waves per block: one wave per SIMD and block
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 50, 0.9999f, 0.0001f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.9999f, 0.0001f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double units_per_simd = (double)iters * CH * wavesPerSimd;
    const double cyc = ms * 1e-3 * 2.4e9 / units_per_simd;
    printf("w/SIMD=%d  %7.2f cyc/unit  %6.2f cyc/instr  (%d instr)  %s\n", wavesPerSimd, cyc, cyc / n, n, name);
}
int main()
{
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0); g_cus = prop.multiProcessorCount;
    printf("CUs %d  clock %d kHz\n", g_cus, prop.clockRate);
    float *out; (void)hipMalloc(&out, (size_t)g_cus * 8 * 256 * sizeof(float));
    for (int w : {1, 2, 4, 8}) {
#define X(M, N, STR) run<M>(STR, N, out, w);
        OPS(X)
#undef X
    }
    return 0;
}
