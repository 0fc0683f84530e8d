// Dev probe (not a test): the LOAD pattern of preprocess_fwd without its arithmetic -- what does the memory system give these instructions?
//   variant 0: everything (means3D / dir3D / scales as 3 dwords at stride 12, rotations float4, opacity, the wave's 12 KB SH block as 12 x float4)
//   variant 1: the SH block alone        variant 2: everything but the SH block      variant 3: 0 with the 12-byte rows read as whole-wave float4 streams
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int V> __global__ __launch_bounds__(256) void k(int P, const float *m, const float *d, const float *s, const float4 *q, const float *o, const float4 *sh, float *out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wc = blockIdx.x * 4 + wave;
    const int idx = wc * 64 + lane;
    if (wc * 64 >= P) return;
    float acc = 0.f;
    float4 pf[12];
    if (V != 2) {
#pragma unroll
        for (int it = 0; it < 12; it++) pf[it] = sh[(size_t)wc * 768 + it * 64 + lane];
    }
    if (V == 0 || V == 2) {
        if (idx < P) {
            acc += m[3 * (size_t)idx] + m[3 * (size_t)idx + 1] + m[3 * (size_t)idx + 2];
            acc += d[3 * (size_t)idx] + d[3 * (size_t)idx + 1] + d[3 * (size_t)idx + 2];
            acc += s[3 * (size_t)idx] + s[3 * (size_t)idx + 1] + s[3 * (size_t)idx + 2];
            const float4 r = q[idx]; acc += r.x + r.y + r.z + r.w + o[idx];
        }
    }
    if (V == 3) {
        // 64 rows x 12 bytes = 768 bytes = 48 float4 per wave and array: lanes 0..47
        const float4 *m4 = reinterpret_cast<const float4 *>(m) + (size_t)wc * 48, *d4 = reinterpret_cast<const float4 *>(d) + (size_t)wc * 48, *s4 = reinterpret_cast<const float4 *>(s) + (size_t)wc * 48;
        if (lane < 48) { const float4 a = m4[lane], b = d4[lane], c = s4[lane]; acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c.x + c.y + c.z + c.w; }
        if (idx < P) { const float4 r = q[idx]; acc += r.x + r.y + r.z + r.w + o[idx]; }
    }
    if (V != 2) {
#pragma unroll
        for (int it = 0; it < 12; it++) acc += pf[it].x + pf[it].y + pf[it].z + pf[it].w;
    }
    if (acc == 12345.678f) out[idx] = acc;          // (never: keeps the loads alive)
}
// the STORES of preprocess_fwd: records (4 x float4 per lane, 64 bytes apart), direction sums (9 dwords per lane, 36 bytes apart), five 4-byte words;
// W = 0: as the kernel issues them; W = 1: the same bytes as whole-wave contiguous float4 / dword streams (what a transposition through LDS would issue)
template <int W> __global__ __launch_bounds__(256) void kst(int P, float4 *rec, float *ds, uint32_t *w0, uint32_t *w1, uint32_t *w2, uint32_t *w3, uint32_t *w4, float v)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wc = blockIdx.x * 4 + wave;
    const int idx = wc * 64 + lane;
    if (idx >= P) return;
    const bool vis = (idx * 2654435761u >> 8) % 100u < 81u;
    const float4 x = make_float4(v, v, v, v);
    if (W == 0) {
        if (vis) {
            float4 *r = rec + 4 * (size_t)idx; r[0] = x; r[1] = x; r[2] = x; r[3] = x;
            float *o = ds + 9 * (size_t)idx;
#pragma unroll
            for (int i = 0; i < 9; i++) o[i] = v;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) rec[(size_t)wc * 256 + i * 64 + lane] = x;
        float4 *o4 = reinterpret_cast<float4 *>(ds) + (size_t)wc * 144;
        o4[lane] = x; o4[64 + lane] = x; if (lane < 16) o4[128 + lane] = x;
    }
    w0[idx] = 1u; w1[idx] = 2u; w2[idx] = 3u; w3[idx] = 4u; w4[idx] = 5u;
}
template <int W> void runst(const char *name, int P, double mb, float4 *rec, float *ds, uint32_t *w)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kst<W>, dim3((P + 255) / 256), dim3(256), 0, 0, P, rec, ds, w, w + P, w + 2 * P, w + 3 * P, w + 4 * P, 1.f);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(kst<W>, dim3((P + 255) / 256), dim3(256), 0, 0, P, rec, ds, w, w + P, w + 2 * P, w + 3 * P, w + 4 * P, 1.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %7.1f us  %6.2f TB/s\n", name, ms / 20 * 1e3, mb / (ms / 20 * 1e-3) / 1e6);
}
template <int V> void run(const char *name, int P, double mb, float *m, float *d, float *s, float4 *q, float *o, float4 *sh, float *out)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k<V>, dim3((P + 255) / 256), dim3(256), 0, 0, P, m, d, s, q, o, sh, out);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k<V>, dim3((P + 255) / 256), dim3(256), 0, 0, P, m, d, s, q, o, sh, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %7.1f us  %6.2f TB/s\n", name, ms / 20 * 1e3, mb / (ms / 20 * 1e-3) / 1e6);
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
    for (int rep = 0; rep < 2; rep++) {
        runst<0>("stores as the kernel issues them (81 % of 100 MB + 20 MB)", P, 0.81 * 100 + 20, rec, ds, w);
        runst<1>("the same arrays as whole-wave contiguous streams (120 MB)", P, 120, rec, ds, w);
        run<0>("all loads of preprocess_fwd (248 MB)", P, 248, m, d, s, q, o, sh, out);
        run<1>("the SH block alone (192 MB)", P, 192, m, d, s, q, o, sh, out);
        run<2>("the per-Gaussian arrays alone (56 MB)", P, 56, m, d, s, q, o, sh, out);
        run<3>("all loads, the 12-byte rows as whole-wave float4 streams (248 MB)", P, 248, m, d, s, q, o, sh, out);
    }
    return 0;
}
