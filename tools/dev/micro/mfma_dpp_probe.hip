// Dev probe (not a test): checks, on a real gfx950, the lane layouts and cross-lane primitives the MFMA compositing
// backward relies on, and prices them.
//   1. v_mfma_f32_16x16x4_f32 operand / result layout (A[i=l&15][k=l>>4], B[k=l>>4][n=l&15], D[row=4(l>>4)+r][col=l&15])
//   2. inclusive prefix product / sum over the 16 lanes of a DPP row (row_shr:1,2,4,8), seeded with a carry in lane 0
//   3. issue cost of the DPP forms, and of 3 MFMAs riding along ~45 VALU instructions
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_mfma_layout(const float *A, const float *B, float *D)   // A[16][4], B[4][16] row-major, D[16][16]
{
    const int l = threadIdx.x;
    const float a = A[(l & 15) * 4 + (l >> 4)];
    const float b = B[(l >> 4) * 16 + (l & 15)];
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

// update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl): lanes whose source is outside the row keep `old`
template <int CTRL> __device__ __forceinline__ float dpp_or(float old, float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
#define ROW_SHR(n) (0x110 + (n))
#define ROW_ROR(n) (0x120 + (n))

__global__ void k_scan(const float *x, const float *carry, float *prod, float *sum, float *ror, float *asmprod)
{
    const int l = threadIdx.x;
    float p = x[l];
    // step 1 seeds lane 0 of every row with the carry (the row-uniform value `carry[row]`)
    p *= dpp_or<ROW_SHR(1)>(carry[l >> 4], p);
    p *= dpp_or<ROW_SHR(2)>(1.f, p);
    p *= dpp_or<ROW_SHR(4)>(1.f, p);
    p *= dpp_or<ROW_SHR(8)>(1.f, p);
    prod[l] = p;
    float s = x[l];
    s += dpp_or<ROW_SHR(1)>(carry[l >> 4], s);
    s += dpp_or<ROW_SHR(2)>(0.f, s);
    s += dpp_or<ROW_SHR(4)>(0.f, s);
    s += dpp_or<ROW_SHR(8)>(0.f, s);
    sum[l] = s;
    ror[l] = dpp_or<ROW_ROR(15)>(-1.f, x[l]);          // lane i <- lane (i+1)&15 ?  (direction check)
    // hand-written single-instruction form: dst = dpp(src0) * src1, lanes without a source keep dst
    float q = x[l];
    asm volatile("s_nop 1\n v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                 "s_nop 1\n v_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                 "s_nop 1\n v_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n"
                 "s_nop 1\n v_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n" : "+v"(q));
    asmprod[l] = q;
}

// ---- timing ------------------------------------------------------------------------------------
template <int MODE> __global__ void k_rate(float *out, int iters, float a, float b)
{
    float s[8];
    for (int i = 0; i < 8; i++) s[i] = threadIdx.x * 0.001f + i + 1.f;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {          // 8 plain multiplies
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(a));
        } else if (MODE == 1) {   // 8 DPP multiplies on independent registers (no back-to-back hazard on the same register)
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_mul_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(s[i]) : "v"(a));
        } else if (MODE == 2) {   // dependent 4-step scan, compiler-scheduled (update_dpp + mul)
#pragma unroll
            for (int i = 0; i < 2; i++) {
                float p = s[i];
                p *= dpp_or<ROW_SHR(1)>(a, p); p *= dpp_or<ROW_SHR(2)>(1.f, p); p *= dpp_or<ROW_SHR(4)>(1.f, p); p *= dpp_or<ROW_SHR(8)>(1.f, p);
                s[i] = p * b;
            }
        } else if (MODE == 3 || MODE == 4) {   // 48 VALU (+ 3 MFMA in mode 4)
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(a), "v"(b));
            if (MODE == 4) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(s[0], s[1], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(s[0], s[2], acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(s[0], s[3], acc2, 0, 0, 0);
            }
        } else if (MODE == 5) {   // ds_swizzle broadcast of lane 15 of every row (LDS pipe, no memory)
#pragma unroll
            for (int i = 0; i < 8; i++) s[i] = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(s[i]), 0x01F0 | (0xF << 5)));
        }
    }
    float r = 0.f;
    for (int i = 0; i < 8; i++) r += s[i];
    for (int i = 0; i < 4; i++) r += acc0[i] + acc1[i] + acc2[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE> void run(const char *name, float *out, int wavesPerSimd, double instr_per_iter)
{
    const int iters = 20000, blocks = 256 * wavesPerSimd;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0001f, 0.5f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s w/SIMD=%d %.3f ms -> %.1f cyc per iteration per SIMD-wave-slot (%.2f cyc/instr)\n", name, wavesPerSimd, ms,
           ms * 1e-3 * 2.4e9 / ((double)iters * wavesPerSimd), ms * 1e-3 * 2.4e9 / ((double)iters * wavesPerSimd * instr_per_iter));
}

int main()
{
    // ---- 1. MFMA layout, asymmetric operands
    std::vector<float> A(64), B(64), D(256), Dref(256, 0.f);
    for (int i = 0; i < 16; i++) for (int k = 0; k < 4; k++) A[i * 4 + k] = 1.f + 0.37f * i - 0.11f * k * k + 0.013f * i * k;
    for (int k = 0; k < 4; k++) for (int n = 0; n < 16; n++) B[k * 16 + n] = -0.5f + 0.21f * n + 0.7f * k - 0.017f * n * n * (k + 1);
    for (int i = 0; i < 16; i++) for (int n = 0; n < 16; n++) { float s = 0.f; for (int k = 0; k < 4; k++) s = fmaf(A[i * 4 + k], B[k * 16 + n], s); Dref[i * 16 + n] = s; }
    float *dA, *dB, *dD;
    (void)hipMalloc(&dA, 256); (void)hipMalloc(&dB, 256); (void)hipMalloc(&dD, 1024);
    (void)hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    (void)hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    double e = 0; for (int i = 0; i < 256; i++) e = fmax(e, fabs(D[i] - Dref[i]));
    printf("[1] mfma_f32_16x16x4 layout A[i=l&15][k=l>>4] B[k=l>>4][n=l&15] D[4(l>>4)+r][l&15]: max err %.3g %s (bitwise fmaf chain: %s)\n",
           e, e < 1e-5 ? "PASS" : "FAIL", e == 0 ? "yes" : "no");

    // ---- 2. DPP scans
    std::vector<float> x(64), carry = {1.5f, 0.25f, 3.f, 0.75f}, prod(64), sum(64), ror(64), asmprod(64);
    for (int l = 0; l < 64; l++) x[l] = 0.9f + 0.01f * l + 0.003f * (l % 7);
    float *dx, *dc, *dp, *ds, *dr, *da;
    (void)hipMalloc(&dx, 256); (void)hipMalloc(&dc, 16); (void)hipMalloc(&dp, 256); (void)hipMalloc(&ds, 256); (void)hipMalloc(&dr, 256); (void)hipMalloc(&da, 256);
    (void)hipMemcpy(dx, x.data(), 256, hipMemcpyHostToDevice); (void)hipMemcpy(dc, carry.data(), 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(64), 0, 0, dx, dc, dp, ds, dr, da);
    (void)hipMemcpy(prod.data(), dp, 256, hipMemcpyDeviceToHost); (void)hipMemcpy(sum.data(), ds, 256, hipMemcpyDeviceToHost);
    (void)hipMemcpy(ror.data(), dr, 256, hipMemcpyDeviceToHost); (void)hipMemcpy(asmprod.data(), da, 256, hipMemcpyDeviceToHost);
    double ep = 0, es = 0, ea = 0;
    for (int row = 0; row < 4; row++) {
        double p = carry[row], s = carry[row], q = 1.0;
        for (int n = 0; n < 16; n++) {
            p *= x[row * 16 + n]; s += x[row * 16 + n]; q *= x[row * 16 + n];
            ep = fmax(ep, fabs(prod[row * 16 + n] - p) / p); es = fmax(es, fabs(sum[row * 16 + n] - s) / s);
            ea = fmax(ea, fabs(asmprod[row * 16 + n] - q) / q);
        }
    }
    printf("[2] row_shr prefix product with carry seed: rel err %.3g %s | prefix sum: %.3g %s | asm v_mul_f32_dpp scan: %.3g %s\n",
           ep, ep < 1e-5 ? "PASS" : "FAIL", es, es < 1e-5 ? "PASS" : "FAIL", ea, ea < 1e-5 ? "PASS" : "FAIL");
    printf("    row_ror:15 : lane0 <- x[%s] , lane15 <- x[%s]\n", ror[0] == x[1] ? "1" : (ror[0] == x[15] ? "15" : "?"),
           ror[15] == x[0] ? "0" : (ror[15] == x[14] ? "14" : "?"));

    // ---- 3. prices
    float *out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int w : {1, 2, 4}) {
        run<0>("8 x v_mul_f32", out, w, 8);
        run<1>("8 x v_mul_f32_dpp row_shr:1 (independent)", out, w, 8);
        run<2>("2 x dependent 4-step DPP product scan", out, w, 2);
        run<3>("48 x v_fma_f32", out, w, 48);
        run<4>("48 x v_fma_f32 + 3 x mfma_16x16x4_f32", out, w, 48);
        run<5>("8 x ds_swizzle row-broadcast", out, w, 8);
    }
    return 0;
}
