// Fused multi-tensor RAdam step for gfx950.
//
// torch.optim.RAdam over the reference's 15 parameter groups (scene/c_gaussian_model.py:430-449) runs ~10 element-wise
// passes per tensor: 3.2 ms per step for the 110 M parameters of a 1 M-Gaussian model on MI355X.  This is one launch that
// streams every tensor once: p, m, v read+write and g read = 28 B/element, 3.07 GB -> 0.38 ms at the 8 TB/s roofline.
// HBM-bound streaming: 16-byte loads/stores, 4096-element chunks, a chunk table in kernel arguments maps workgroups to
// tensors.  ffp-contract is off for this file: the update follows torch's op order in float32.
#include "ex4d_internal.h"
#include "../../include/ex4d_optim.h"
#include <cmath>
#include <cstdio>

namespace {

#define RADAM_CHUNK 4096          // elements per workgroup (256 threads x 4 float4)

struct RadamSlot {
    float *p; const float *g; float *m; float *v;
    long long numel;
    float w1;          // 1 - beta1
    float beta2;
    float w2;          // 1 - beta2
    float bc1;         // 1 - beta1^t
    float lr;
    float sqrt_bc2;    // sqrt(1 - beta2^t)
    float rect;        // variance rectification, valid when rectified
    float eps;
    int rectified;     // rho_t > 5
    unsigned first_chunk;   // index of this tensor's first chunk in the launch
    int sanitize;      // gradient read through nan_to_num (train.py:244-247)
};

struct RadamArgs { RadamSlot slot[EX4D_RADAM_MAX_TENSORS]; int count; };

// torch.nan_to_num with its defaults: NaN -> 0, +inf -> FLT_MAX, -inf -> -FLT_MAX
__device__ __forceinline__ float nan_to_num(float g) { return g != g ? 0.f : fminf(fmaxf(g, -3.402823466e+38f), 3.402823466e+38f); }

__device__ __forceinline__ void radam_update(float &p, float g, float &m, float &v, const RadamSlot &s)
{
    if (s.sanitize) g = nan_to_num(g);
    m = m + s.w1 * (g - m);                       // exp_avg.lerp_(grad, 1 - beta1)
    v = v * s.beta2;                              // exp_avg_sq.mul_(beta2)
    v = v + (s.w2 * g) * g;                       //           .addcmul_(grad, grad, value=1 - beta2)
    const float mhat = m / s.bc1;
    float upd = mhat * s.lr;
    if (s.rectified) {
        const float adaptive = s.sqrt_bc2 / (sqrtf(v) + s.eps);
        upd = (upd * adaptive) * s.rect;
    }
    p = p - upd;
}

__global__ __launch_bounds__(256) void radam_kernel(const RadamArgs a)
{
    // which tensor owns this chunk (<= 32 slots, wave-uniform scan)
    int t = 0;
#pragma unroll 1
    for (int i = 1; i < a.count; i++) if (blockIdx.x >= a.slot[i].first_chunk) t = i;
    const RadamSlot &s = a.slot[t];
    const long long base = (long long)(blockIdx.x - s.first_chunk) * RADAM_CHUNK;
    const long long n = s.numel - base < RADAM_CHUNK ? s.numel - base : RADAM_CHUNK;
    float *p = s.p + base; const float *g = s.g + base; float *m = s.m + base; float *v = s.v + base;
    const bool vec = n == RADAM_CHUNK && ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
    if (vec) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = k * 256 + threadIdx.x;
            float4 pp = ((float4 *)p)[i], mm = ((float4 *)m)[i], vv = ((float4 *)v)[i];
            const float4 gg = ((const float4 *)g)[i];
            radam_update(pp.x, gg.x, mm.x, vv.x, s); radam_update(pp.y, gg.y, mm.y, vv.y, s);
            radam_update(pp.z, gg.z, mm.z, vv.z, s); radam_update(pp.w, gg.w, mm.w, vv.w, s);
            ((float4 *)p)[i] = pp; ((float4 *)m)[i] = mm; ((float4 *)v)[i] = vv;
        }
    } else {
        for (int i = threadIdx.x; i < n; i += 256) {
            float pp = p[i], mm = m[i], vv = v[i];
            radam_update(pp, g[i], mm, vv, s);
            p[i] = pp; m[i] = mm; v[i] = vv;
        }
    }
}

// ---- keyframe tensors with windowed gradients: one thread per (row, time slice) = C consecutive floats
struct SlicedSlot {
    RadamSlot s;             // p, m, v, coefficients (g unused)
    long long slices;        // rows * K
    int K, C, nw;
    int first[EX4D_RADAM_MAX_WINDOWS], count[EX4D_RADAM_MAX_WINDOWS];
    const float *grad[EX4D_RADAM_MAX_WINDOWS];
    const int *first_dev;    // optional: window positions in device memory (uniform loads)
    unsigned first_block;
};
struct SlicedArgs { SlicedSlot slot[EX4D_RADAM_MAX_SLICED]; int count; };

// gradient of flat element e = (row, kk, c) of a [rows, K, C] tensor from the windows (zero outside them); windows add in index order
__device__ __forceinline__ float sliced_grad(const SlicedSlot &t, const int (&first)[EX4D_RADAM_MAX_WINDOWS], long long row, int kk, int c)
{
    float g = 0.f;
    for (int w = 0; w < t.nw; w++) {
        const unsigned rel = (unsigned)(kk - first[w]);
        if (rel < (unsigned)t.count[w]) g += t.grad[w][((size_t)row * t.count[w] + rel) * t.C + c];
    }
    return g;
}

// same streaming structure as radam_kernel (4096-element chunks, 16-byte accesses to p, m, v); only the gradient differs: the few
// elements inside a window read it from the compact blocks, all others take 0 without touching memory
__global__ __launch_bounds__(256) void radam_sliced_kernel(const SlicedArgs a)
{
    int t = 0;
#pragma unroll 1
    for (int i = 1; i < a.count; i++) if (blockIdx.x >= a.slot[i].first_block) t = i;
    const SlicedSlot &s = a.slot[t];
    const long long base = (long long)(blockIdx.x - s.first_block) * RADAM_CHUNK;
    const long long n = s.s.numel - base < RADAM_CHUNK ? s.s.numel - base : RADAM_CHUNK;
    float *p = s.s.p + base, *m = s.s.m + base, *v = s.s.v + base;
    const bool vec = n == RADAM_CHUNK && ((((uintptr_t)p | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
    int first[EX4D_RADAM_MAX_WINDOWS];
#pragma unroll
    for (int w = 0; w < EX4D_RADAM_MAX_WINDOWS; w++) first[w] = (s.first_dev && w < s.nw) ? s.first_dev[w] : s.first[w];
    if (vec) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = k * 256 + threadIdx.x;
            const long long e = base + 4 * (long long)i;
            long long sl = e / s.C;
            int c = (int)(e - sl * s.C);
            long long row = sl / s.K;
            int kk = (int)(sl - row * s.K);
            float g[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                g[j] = sliced_grad(s, first, row, kk, c);
                if (++c == s.C) { c = 0; if (++kk == s.K) { kk = 0; row++; } }
            }
            float4 pp = ((float4 *)p)[i], mm = ((float4 *)m)[i], vv = ((float4 *)v)[i];
            radam_update(pp.x, g[0], mm.x, vv.x, s.s); radam_update(pp.y, g[1], mm.y, vv.y, s.s);
            radam_update(pp.z, g[2], mm.z, vv.z, s.s); radam_update(pp.w, g[3], mm.w, vv.w, s.s);
            ((float4 *)p)[i] = pp; ((float4 *)m)[i] = mm; ((float4 *)v)[i] = vv;
        }
    } else {
        for (int i = threadIdx.x; i < n; i += 256) {
            const long long e = base + i;
            const long long sl = e / s.C, row = sl / s.K;
            float pp = p[i], mm = m[i], vv = v[i];
            radam_update(pp, sliced_grad(s, first, row, (int)(sl - row * s.K), (int)(e - sl * s.C)), mm, vv, s.s);
            p[i] = pp; m[i] = mm; v[i] = vv;
        }
    }
}

thread_local char g_optim_err[256] = "";

static bool fill_coefficients(RadamSlot &s, double lr, long long step, double beta1, double beta2, double eps)
{
    // the Python-double scalar arithmetic of torch's RAdam, cast to float32 where it meets a tensor
    const double rho_inf = 2.0 / (1.0 - beta2) - 1.0;
    const double st = (double)step;
    const double bc1 = 1.0 - std::pow(beta1, st), bc2 = 1.0 - std::pow(beta2, st);
    const double rho_t = rho_inf - 2.0 * st * std::pow(beta2, st) / bc2;
    s.w1 = (float)(1.0 - beta1); s.beta2 = (float)beta2; s.w2 = (float)(1.0 - beta2);
    s.bc1 = (float)bc1; s.lr = (float)lr; s.sqrt_bc2 = (float)std::sqrt(bc2); s.eps = (float)eps;
    s.rectified = rho_t > 5.0;
    s.rect = s.rectified ? (float)std::sqrt((rho_t - 4.0) * (rho_t - 2.0) * rho_inf / ((rho_inf - 4.0) * (rho_inf - 2.0) * rho_t)) : 0.f;
    return true;
}

}  // namespace

extern "C" {

const char *ex4d_optim_last_error(void) { return g_optim_err; }

int ex4d_radam_step(const Ex4dRadamTensor *tensors, int32_t count, double beta1, double beta2, double eps, void *stream_)
{
    g_optim_err[0] = 0;
    if (count < 0 || count > EX4D_RADAM_MAX_TENSORS || (count > 0 && !tensors)) {
        snprintf(g_optim_err, sizeof(g_optim_err), "count %d outside [0, %d]", count, EX4D_RADAM_MAX_TENSORS);
        return EX4D_ERR_ARG;
    }
    RadamArgs a;
    a.count = 0;
    unsigned chunks = 0;
    for (int i = 0; i < count; i++) {
        const Ex4dRadamTensor &t = tensors[i];
        if (t.numel == 0) continue;
        if (t.numel < 0 || t.step < 1 || !t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq) {
            snprintf(g_optim_err, sizeof(g_optim_err), "tensor %d: null pointer, negative size or step < 1", i);
            return EX4D_ERR_ARG;
        }
        RadamSlot &s = a.slot[a.count++];
        s.p = t.param; s.g = t.grad; s.m = t.exp_avg; s.v = t.exp_avg_sq; s.numel = t.numel; s.sanitize = t.nan_to_num != 0;
        fill_coefficients(s, t.lr, t.step, beta1, beta2, eps);
        s.first_chunk = chunks;
        const long long c = (t.numel + RADAM_CHUNK - 1) / RADAM_CHUNK;
        if (c + chunks > 0x7fffffffLL) { snprintf(g_optim_err, sizeof(g_optim_err), "too many elements for one launch"); return EX4D_ERR_ARG; }
        chunks += (unsigned)c;
    }
    if (chunks == 0) return EX4D_OK;
    hipLaunchKernelGGL(radam_kernel, dim3(chunks), dim3(256), 0, (hipStream_t)stream_, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_optim_err, sizeof(g_optim_err), "launch failed: %s", hipGetErrorString(e)); return EX4D_ERR_HIP; }
    return EX4D_OK;
}

int ex4d_radam_step_sliced(const Ex4dRadamSlicedTensor *tensors, int32_t count, double beta1, double beta2, double eps, void *stream_)
{
    g_optim_err[0] = 0;
    if (count < 0 || count > EX4D_RADAM_MAX_SLICED || (count > 0 && !tensors)) {
        snprintf(g_optim_err, sizeof(g_optim_err), "count %d outside [0, %d]", count, EX4D_RADAM_MAX_SLICED);
        return EX4D_ERR_ARG;
    }
    SlicedArgs a;
    a.count = 0;
    unsigned blocks = 0;
    for (int i = 0; i < count; i++) {
        const Ex4dRadamSlicedTensor &t = tensors[i];
        if (t.rows == 0) continue;
        if (t.rows < 0 || t.K < 1 || (t.C != 3 && t.C != 4) || t.step < 1 || !t.param || !t.exp_avg || !t.exp_avg_sq ||
            t.n_windows < 0 || t.n_windows > EX4D_RADAM_MAX_WINDOWS) {
            snprintf(g_optim_err, sizeof(g_optim_err), "sliced tensor %d: bad shape, step < 1, null pointer or too many windows", i);
            return EX4D_ERR_ARG;
        }
        SlicedSlot &s = a.slot[a.count++];
        s.s.p = t.param; s.s.g = nullptr; s.s.m = t.exp_avg; s.s.v = t.exp_avg_sq; s.s.numel = t.rows * t.K * t.C; s.s.first_chunk = 0; s.s.sanitize = 0;
        s.first_dev = t.first_dev;
        fill_coefficients(s.s, t.lr, t.step, beta1, beta2, eps);
        s.slices = t.rows * t.K; s.K = t.K; s.C = t.C; s.nw = t.n_windows;
        for (int w = 0; w < EX4D_RADAM_MAX_WINDOWS; w++) {
            const bool live = w < t.n_windows;
            if (live && (((t.first[w] < 0 || t.first[w] + t.count[w] > t.K) && !t.first_dev) || t.count[w] < 1 || t.count[w] > t.K || !t.grad[w])) {
                snprintf(g_optim_err, sizeof(g_optim_err), "sliced tensor %d: window %d outside [0, K) or null", i, w);
                return EX4D_ERR_ARG;
            }
            s.first[w] = live ? t.first[w] : 0; s.count[w] = live ? t.count[w] : 0; s.grad[w] = live ? t.grad[w] : nullptr;
        }
        s.first_block = blocks;
        const long long nb = (s.s.numel + RADAM_CHUNK - 1) / RADAM_CHUNK;
        if (nb + blocks > 0x7fffffffLL) { snprintf(g_optim_err, sizeof(g_optim_err), "too many elements for one launch"); return EX4D_ERR_ARG; }
        blocks += (unsigned)nb;
    }
    if (blocks == 0) return EX4D_OK;
    hipLaunchKernelGGL(radam_sliced_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_optim_err, sizeof(g_optim_err), "launch failed: %s", hipGetErrorString(e)); return EX4D_ERR_HIP; }
    return EX4D_OK;
}

}  // extern "C"
