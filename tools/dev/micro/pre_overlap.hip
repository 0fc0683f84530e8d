// Dev probe (not a test): do the load phase and the arithmetic phase of a preprocess_fwd-shaped wave overlap ACROSS the waves of a CU?
// Every wave: all loads of preprocess_fwd up front (248 MB per launch at 1.0 M Gaussians), then N VALU instructions that depend on them, then
// (optionally) the kernel's stores; LDS per workgroup as in the kernel (26.6 KB: 6 workgroups per CU) or none.
// If time(N) ~ max(load time, N x issue) the phases overlap and the real kernel's structure is what keeps them apart; if ~ load time + N x issue
// the hardware runs the co-resident waves in lockstep whatever the code does.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N, int LDSB, int ST, int TR, int PRED> __global__ __launch_bounds__(256) void k(int P, const float *m, const float *d, const float *s, const float4 *q, const float *o, const float4 *sh,
                                                                       float *out, float4 *rec, float *ds, uint32_t *w5)
{
    __shared__ float lds[LDSB / 4 + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wc = blockIdx.x * 4 + wave;
    const int idx = wc * 64 + lane;
    if (wc * 64 >= P) return;
    float4 pf[12];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    unsigned long long need = ~0ull;
    if (PRED) {
        // the frustum test in front of the SH loads: one dependent round trip (means3D), ~40 VALU, a ballot
        float mx = 0.f, my = 0.f, mz = 0.f;
        if (idx < P) { mx = m[3 * (size_t)idx]; my = m[3 * (size_t)idx + 1]; mz = m[3 * (size_t)idx + 2]; }
        float t = mx * 0.5f + my * 0.25f + mz * 0.125f;
#pragma unroll
        for (int i = 0; i < 10; i++) t = __builtin_fmaf(t, 1.0000001f, mx);
        need = __ballot(t < 1e30f);
        a0 = t;
    }
    if (need == ~0ull) {
#pragma unroll
        for (int it = 0; it < 12; it++) pf[it] = sh[(size_t)wc * 768 + it * 64 + lane];
    } else {
#pragma unroll
        for (int it = 0; it < 12; it++) pf[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (idx < P) {
        if (!PRED) a0 = m[3 * (size_t)idx] + m[3 * (size_t)idx + 1] + m[3 * (size_t)idx + 2];
        a1 = d[3 * (size_t)idx] + d[3 * (size_t)idx + 1] + d[3 * (size_t)idx + 2];
        a2 = s[3 * (size_t)idx] + s[3 * (size_t)idx + 1] + s[3 * (size_t)idx + 2];
        const float4 r = q[idx]; a3 = r.x + r.y + r.z + r.w + o[idx];
    }
    if (LDSB > 64) { lds[threadIdx.x] = a0; __syncthreads(); a0 += lds[threadIdx.x ^ 1]; }
    if (TR && LDSB > 26000) {
        // the SH block through half a padded slice at a time (rows of 52 floats), like the kernel: commit 6 x float4, the 32 lanes of the half read their 12 x float4
        float *slice = lds + 64 + wave * (32 * 52);
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const int qq = (6 * h + i) * 64 + lane;
                const int g = qq / 12 - 32 * h, v = qq % 12;
                *reinterpret_cast<float4 *>(slice + g * 52 + 4 * v) = pf[6 * h + i];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if ((lane >> 5) == h) {
                const float4 *row = reinterpret_cast<const float4 *>(slice + (lane & 31) * 52);
#pragma unroll
                for (int v = 0; v < 12; v++) { const float4 t4 = row[v]; a0 += t4.x; a1 += t4.y; a2 += t4.z; a3 += t4.w; }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    } else {
#pragma unroll
        for (int it = 0; it < 12; it++) { a0 += pf[it].x; a1 += pf[it].y; a2 += pf[it].z; a3 += pf[it].w; }
    }
    // N wave instructions, four independent chains
#pragma unroll 16
    for (int i = 0; i < N / 4; i++) {
        a0 = __builtin_fmaf(a0, 1.0000001f, a1); a1 = __builtin_fmaf(a1, 0.9999999f, a2);
        a2 = __builtin_fmaf(a2, 1.0000002f, a3); a3 = __builtin_fmaf(a3, 0.9999998f, a0);
    }
    const float acc = a0 + a1 + a2 + a3;
    if (ST == 2 || ST == 3) {
        // the same bytes as whole-wave contiguous streams (what a transposition through LDS would issue): records 4 KB, direction sums 2304 B per wave
        // (ST == 3: only for the 81 % "visible" share, decided per wave, so that the byte count matches the strided form)
        const bool wvis = ST == 2 || ((unsigned)wc * 2654435761u >> 8) % 100u < 81u;
        const float4 x = make_float4(acc, acc, acc, acc);
        if (wvis) {
#pragma unroll
            for (int i = 0; i < 4; i++) rec[(size_t)wc * 256 + i * 64 + lane] = x;
            float4 *o4 = reinterpret_cast<float4 *>(ds) + (size_t)wc * 144;
            o4[lane] = x; o4[64 + lane] = x; if (lane < 16) o4[128 + lane] = x;
        }
        if (idx < P) { w5[idx] = 1u; w5[P + idx] = 2u; w5[2 * (size_t)P + idx] = 3u; w5[3 * (size_t)P + idx] = 4u; w5[4 * (size_t)P + idx] = 5u; }
    } else if (ST == 4 || ST == 5) {
        // strided form, records only (4) / direction sums only (5), + the five word streams
        if (idx < P) {
            const bool vis = (idx * 2654435761u >> 8) % 100u < 81u;
            const float4 x = make_float4(acc, acc, acc, acc);
            if (vis && ST == 4) { float4 *r = rec + 4 * (size_t)idx; r[0] = x; r[1] = x; r[2] = x; r[3] = x; }
            if (vis && ST == 5) { float *oo = ds + 9 * (size_t)idx;
#pragma unroll
                for (int i = 0; i < 9; i++) oo[i] = acc; }
            w5[idx] = 1u; w5[P + idx] = 2u; w5[2 * (size_t)P + idx] = 3u; w5[3 * (size_t)P + idx] = 4u; w5[4 * (size_t)P + idx] = 5u;
        }
    } else if (ST == 6) {
        if (idx < P) { w5[idx] = 1u; w5[P + idx] = 2u; w5[2 * (size_t)P + idx] = 3u; w5[3 * (size_t)P + idx] = 4u; w5[4 * (size_t)P + idx] = 5u; }
    } else if (ST) {
        if (idx < P) {
            const bool vis = (idx * 2654435761u >> 8) % 100u < 81u;
            const float4 x = make_float4(acc, acc, acc, acc);
            if (vis) {
                float4 *r = rec + 4 * (size_t)idx; r[0] = x; r[1] = x; r[2] = x; r[3] = x;
                float *oo = ds + 9 * (size_t)idx;
#pragma unroll
                for (int i = 0; i < 9; i++) oo[i] = acc;
            }
            w5[idx] = 1u; w5[P + idx] = 2u; w5[2 * (size_t)P + idx] = 3u; w5[3 * (size_t)P + idx] = 4u; w5[4 * (size_t)P + idx] = 5u;
        }
    } else if (acc == 12345.678f) out[idx] = acc;
}
template <int N, int LDSB, int ST, int TR = 0, int PRED = 0> void run(int P, float *m, float *d, float *s, float4 *q, float *o, float4 *sh, float *out, float4 *rec, float *ds, uint32_t *w)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<N, LDSB, ST, TR, PRED>), dim3((P + 255) / 256), dim3(256), 0, 0, P, m, d, s, q, o, sh, out, rec, ds, w);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL((k<N, LDSB, ST, TR, PRED>), dim3((P + 255) / 256), dim3(256), 0, 0, P, m, d, s, q, o, sh, out, rec, ds, w);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("N = %5d VALU / wave  LDS %5d B / workgroup  stores %d  transposition %d  predicate %d : %7.1f us   (issue alone: %5.1f us at 4 cycles, 1024 SIMDs, 2.4 GHz)\n", N, LDSB, ST, TR, PRED, ms / 20 * 1e3,
           (double)N * (P / 64) * 4.0 / (1024 * 2.4e9) * 1e6);
}
int main()
{
    const int P = 1000000;
    float *m, *d, *s, *o, *out; float4 *q, *sh;
    (void)hipMalloc(&m, 12 * (size_t)P + 64); (void)hipMalloc(&d, 12 * (size_t)P + 64); (void)hipMalloc(&s, 12 * (size_t)P + 64); (void)hipMalloc(&q, 16 * (size_t)P);
    (void)hipMalloc(&o, 4 * (size_t)P); (void)hipMalloc(&sh, 192 * (size_t)P + 4096); (void)hipMalloc(&out, 4 * (size_t)P);
    (void)hipMemset(m, 0, 12 * (size_t)P); (void)hipMemset(d, 0, 12 * (size_t)P); (void)hipMemset(s, 0, 12 * (size_t)P); (void)hipMemset(q, 0, 16 * (size_t)P);
    (void)hipMemset(o, 0, 4 * (size_t)P); (void)hipMemset(sh, 0, 192 * (size_t)P);
    float4 *rec; float *ds; uint32_t *w;
    (void)hipMalloc(&rec, 64 * (size_t)P + 4096); (void)hipMalloc(&ds, 36 * (size_t)P + 4096); (void)hipMalloc(&w, 20 * (size_t)P);
#define R(N, L, S) run<N, L, S>(P, m, d, s, q, o, sh, out, rec, ds, w)
    for (int rep = 0; rep < 2; rep++) {
        run<1400, 26640, 1, 1, 0>(P, m, d, s, q, o, sh, out, rec, ds, w); run<1400, 26640, 1, 0, 1>(P, m, d, s, q, o, sh, out, rec, ds, w); run<1400, 26640, 1, 1, 1>(P, m, d, s, q, o, sh, out, rec, ds, w);
        run<1400, 26640, 3, 1, 1>(P, m, d, s, q, o, sh, out, rec, ds, w); run<1400, 26640, 0, 1, 1>(P, m, d, s, q, o, sh, out, rec, ds, w);
        R(1400, 26640, 1); R(1400, 26640, 2); R(1400, 26640, 3); R(1400, 26640, 4); R(1400, 26640, 5); R(1400, 26640, 6); R(0, 26640, 2); R(0, 26640, 3);
        R(0, 4, 0); R(400, 4, 0); R(800, 4, 0); R(1400, 4, 0); R(2800, 4, 0);
        R(0, 26640, 0); R(800, 26640, 0); R(1400, 26640, 0); R(2800, 26640, 0);
        R(0, 26640, 1); R(800, 26640, 1); R(1400, 26640, 1); R(2800, 26640, 1);
        R(1400, 4, 1);
    }
    return 0;
}
