// Fused per-frame attribute evaluation of the static + keyframe-interpolated dynamic Gaussians, forward and
// backward (the Python caller on the input side of the rasterizer boundary).
//
// Replaces the ~15 small torch kernels + 3 torch.cat copies per frame of (paths under /root/reference)
//   CGaussianModel.get_xyz_at_t / get_rotation_at_t      scene/c_gaussian_model.py:170-215
//   get_scaling / get_features / get_opacity_at_t        scene/c_gaussian_model.py:330-375
//   cube_interpolate, quat_slerp_interp_uniiterval, time_bigaussian   utils/interpolations.py:81-93, :33-52, :55-61
// and their autograd backward, by two streaming kernels each way: one thread per Gaussian for the 11 floats of
// (xyz, rotation, opacity, scale), one grid-stride element-wise kernel for the [N,16,3] SH block (the only large
// stream: 192 B per Gaussian each way).  Static rows first, then dynamic rows (c_gaussian_model.py:193).
// Compiled with -ffp-contract=off: same float32 operation order as the reference's tensor arithmetic.
#include "ex4d_internal.h"
#include <cstdio>

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

struct Slerp {
    float v1[4], v2[4], n1, n2, raw, d, ac, omega, sn, s, p0, p1, ps_raw, psum, p0n, p1n, r[4], nr;
    bool fallback;
};

// utils/interpolations.py:33-52
__device__ __forceinline__ void slerp_forward(const float *q1, const float *q2, float t, Slerp &c, float *out)
{
    c.n1 = sqrtf(q1[0] * q1[0] + q1[1] * q1[1] + q1[2] * q1[2] + q1[3] * q1[3]);
    c.n2 = sqrtf(q2[0] * q2[0] + q2[1] * q2[1] + q2[2] * q2[2] + q2[3] * q2[3]);
#pragma unroll
    for (int i = 0; i < 4; i++) { c.v1[i] = q1[i] / c.n1; c.v2[i] = q2[i] / c.n2; }
    c.raw = c.v1[0] * c.v2[0] + c.v1[1] * c.v2[1] + c.v1[2] * c.v2[2] + c.v1[3] * c.v2[3];
    c.d = fminf(fmaxf(c.raw, -1.0f + 1e-4f), 1.0f - 1e-4f);
    c.ac = acosf(c.d);
    c.omega = fmaxf(c.ac, 1e-4f);
    c.sn = sinf(c.omega);
    c.s = fmaxf(c.sn, 1e-4f);
    c.p0 = sinf((1.0f - t) * c.omega) / c.s;
    c.p1 = sinf(t * c.omega) / c.s;
    c.ps_raw = c.p0 + c.p1;
    c.psum = fmaxf(c.ps_raw, 1e-4f);
    c.p0n = c.p0 / c.psum;
    c.p1n = c.p1 / c.psum;
    float asum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) { c.r[i] = c.v1[i] * c.p0n + c.v2[i] * c.p1n; asum += fabsf(c.r[i]); }
    c.fallback = !(asum > 1e-4f);
    if (c.fallback) {
#pragma unroll
        for (int i = 0; i < 4; i++) c.r[i] = c.v1[i];
    }
    c.nr = sqrtf(c.r[0] * c.r[0] + c.r[1] * c.r[1] + c.r[2] * c.r[2] + c.r[3] * c.r[3]);
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = c.r[i] / c.nr;
}

// adjoint of slerp_forward (what autograd computes for utils/interpolations.py:33-52)
__device__ __forceinline__ void slerp_backward(const Slerp &c, float t, const float *G, float *g_q1, float *g_q2)
{
    float out[4], g_r[4], g_v1[4], g_v2[4];
    float og = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) { out[i] = c.r[i] / c.nr; og += out[i] * G[i]; }
#pragma unroll
    for (int i = 0; i < 4; i++) g_r[i] = (G[i] - out[i] * og) / c.nr;
    float g_p0n = 0.f, g_p1n = 0.f;
    if (c.fallback) {
#pragma unroll
        for (int i = 0; i < 4; i++) { g_v1[i] = g_r[i]; g_v2[i] = 0.f; }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) { g_v1[i] = g_r[i] * c.p0n; g_v2[i] = g_r[i] * c.p1n; g_p0n += g_r[i] * c.v1[i]; g_p1n += g_r[i] * c.v2[i]; }
    }
    float g_p0 = g_p0n / c.psum, g_p1 = g_p1n / c.psum;
    const float g_psum = -(g_p0n * c.p0 + g_p1n * c.p1) / (c.psum * c.psum);
    if (c.ps_raw >= 1e-4f) { g_p0 += g_psum; g_p1 += g_psum; }
    float g_om = g_p0 * (1.0f - t) * cosf((1.0f - t) * c.omega) / c.s + g_p1 * t * cosf(t * c.omega) / c.s;
    const float g_s = -(g_p0 * c.p0 + g_p1 * c.p1) / c.s;
    if (c.sn >= 1e-4f) g_om += g_s * cosf(c.omega);
    const float g_d = (c.ac >= 1e-4f) ? (-g_om / sqrtf(1.0f - c.d * c.d)) : 0.f;
    const float g_raw = (c.raw >= -1.0f + 1e-4f && c.raw <= 1.0f - 1e-4f) ? g_d : 0.f;
    float d1 = 0.f, d2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) { g_v1[i] += g_raw * c.v2[i]; g_v2[i] += g_raw * c.v1[i]; }
#pragma unroll
    for (int i = 0; i < 4; i++) { d1 += c.v1[i] * g_v1[i]; d2 += c.v2[i] * g_v2[i]; }
#pragma unroll
    for (int i = 0; i < 4; i++) { g_q1[i] = (g_v1[i] - c.v1[i] * d1) / c.n1; g_q2[i] = (g_v2[i] - c.v2[i] * d2) / c.n2; }
}

struct BiGauss { float m, v, D, u, o, out; int arg; bool after, inside; };

// utils/interpolations.py:55-61 with mean/var [Nd,2,1] and a scalar time
__device__ __forceinline__ void bigaussian_forward(float c0, float c1, float v0, float v1, float tau, float var_min, BiGauss &b)
{
    const float d0 = tau - c0, d1 = tau - c1;
    b.arg = (d1 < d0) ? 1 : 0;                   // first minimum
    b.m = b.arg ? d1 : d0;
    b.after = (tau > c0) || (tau > c1);
    b.v = b.after ? v1 : v0;
    b.D = expf(b.v) + var_min / 2.36f;
    b.u = (b.m * b.m) / (b.D * b.D);
    b.o = expf(-1.0f * b.u);
    b.inside = (c0 - tau) * (c1 - tau) < 0.f;
    b.out = b.inside ? 1.0f : b.o;
}

__global__ __launch_bounds__(256) void attributes_fwd_kernel(Ex4dAttrParams a,
    const float *__restrict__ xyz, const float *__restrict__ xyz_disp, const float *__restrict__ rotation,
    const float *__restrict__ opacity, const float *__restrict__ scaling,
    const float *__restrict__ xyz_motion, const float *__restrict__ rotation_motion, const float *__restrict__ opacity_motion,
    const float *__restrict__ dur_center, const float *__restrict__ dur_var, const float *__restrict__ scaling_motion,
    float *__restrict__ means3D, float *__restrict__ rotations, float *__restrict__ opacities, float *__restrict__ scales)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int N = a.Ns + a.Nd;
    if (i >= N) return;
    float m[3], q[4], op, sc[3];
    if (i < a.Ns) {
        // c_gaussian_model.py:180, :198, :357, :335
#pragma unroll
        for (int c = 0; c < 3; c++) m[c] = xyz[3 * (size_t)i + c] + (xyz_disp[3 * (size_t)i + c] * a.t) / a.duration;
        const float4 r = reinterpret_cast<const float4 *>(rotation)[i];
        q[0] = r.x; q[1] = r.y; q[2] = r.z; q[3] = r.w;
        op = sigmoidf_(opacity[i]);
#pragma unroll
        for (int c = 0; c < 3; c++) sc[c] = expf(scaling[3 * (size_t)i + c]);
    } else {
        const size_t j = (size_t)(i - a.Ns);
        // Catmull-Rom Hermite over keyframes k-1..k+2 (interpolations.py:81-93, c_gaussian_model.py:118)
        const float *y = xyz_motion + (j * a.K + (a.k - 1)) * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float y0 = y[c], y1 = y[3 + c], y2 = y[6 + c], y3 = y[9 + c];
            const float mk = (y2 - y0) / 2.0f, mk1 = (y3 - y1) / 2.0f;
            m[c] = a.h00 * y1 + a.h10 * mk + a.h01 * y2 + a.h11 * mk1;
        }
        const float4 *rq = reinterpret_cast<const float4 *>(rotation_motion) + j * a.K + a.k;
        const float4 r1 = rq[0], r2 = rq[1];
        const float q1[4] = { r1.x, r1.y, r1.z, r1.w }, q2[4] = { r2.x, r2.y, r2.z, r2.w };
        Slerp c;
        slerp_forward(q1, q2, a.delta, c, q);
        BiGauss b;
        bigaussian_forward(dur_center[2 * j], dur_center[2 * j + 1], dur_var[2 * j], dur_var[2 * j + 1], a.tau, a.var_min, b);
        op = b.out * sigmoidf_(opacity_motion[j]);
#pragma unroll
        for (int c = 0; c < 3; c++) sc[c] = expf(scaling_motion[3 * j + c]);
    }
#pragma unroll
    for (int c = 0; c < 3; c++) { means3D[3 * (size_t)i + c] = m[c]; scales[3 * (size_t)i + c] = sc[c]; }
    reinterpret_cast<float4 *>(rotations)[i] = make_float4(q[0], q[1], q[2], q[3]);
    opacities[i] = op;
}

// shs[N,16,3] <- cat(cat(dc, rest), cat(dc_motion, rest_motion)): element-wise, fully coalesced both ways.
// GATHER = true: forward copy into shs; false: backward split of dL/dshs into the four gradient tensors.
template <bool GATHER>
__global__ __launch_bounds__(256) void features_kernel(int Ns, int Nd, float *__restrict__ dc, float *__restrict__ rest,
    float *__restrict__ dc_m, float *__restrict__ rest_m, float *__restrict__ shs)
{
    // one float4 of the [N,48] block per thread and step (32-bit index math; 12 float4 per row); the dc / rest side is four
    // scalar accesses at consecutive addresses (rows of 3 and 45 floats cannot be 16-byte aligned)
    const unsigned total4 = (unsigned)(Ns + Nd) * 12u;
    for (unsigned q = blockIdx.x * 256u + threadIdx.x; q < total4; q += gridDim.x * 256u) {
        const unsigned row = q / 12u;
        const int c4 = (int)(q - row * 12u) * 4;
        float *d, *r;
        if (row < (unsigned)Ns) { d = dc + (size_t)row * 3; r = rest + (size_t)row * 45; }
        else { const size_t j = row - (unsigned)Ns; d = dc_m + j * 3; r = rest_m + j * 45; }
        float *p0, *p1, *p2, *p3;
        if (c4 == 0) { p0 = d; p1 = d + 1; p2 = d + 2; p3 = r; }
        else { p0 = r + (c4 - 3); p1 = p0 + 1; p2 = p0 + 2; p3 = p0 + 3; }
        float4 *s4 = reinterpret_cast<float4 *>(shs) + q;
        if (GATHER) *s4 = make_float4(*p0, *p1, *p2, *p3);
        else { const float4 v = *s4; *p0 = v.x; *p1 = v.y; *p2 = v.z; *p3 = v.w; }
    }
}

__global__ __launch_bounds__(256) void attributes_bwd_kernel(Ex4dAttrParams a,
    const float *__restrict__ opacity, const float *__restrict__ scaling,
    const float *__restrict__ rotation_motion, const float *__restrict__ opacity_motion,
    const float *__restrict__ dur_center, const float *__restrict__ dur_var, const float *__restrict__ scaling_motion,
    const float *__restrict__ g_means3D, const float *__restrict__ g_rotations, const float *__restrict__ g_opacities,
    const float *__restrict__ g_scales,
    float *__restrict__ g_xyz, float *__restrict__ g_xyz_disp, float *__restrict__ g_rotation, float *__restrict__ g_opacity,
    float *__restrict__ g_scaling, float *__restrict__ g_xyz_motion, float *__restrict__ g_rotation_motion,
    float *__restrict__ g_opacity_motion, float *__restrict__ g_dur_center, float *__restrict__ g_dur_var,
    float *__restrict__ g_scaling_motion, int sliced)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int N = a.Ns + a.Nd;
    if (i >= N) return;
    float gm[3], gs[3];
#pragma unroll
    for (int c = 0; c < 3; c++) { gm[c] = g_means3D[3 * (size_t)i + c]; gs[c] = g_scales[3 * (size_t)i + c]; }
    const float4 gq = reinterpret_cast<const float4 *>(g_rotations)[i];
    const float go = g_opacities[i];
    if (i < a.Ns) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            g_xyz[3 * (size_t)i + c] = gm[c];
            g_xyz_disp[3 * (size_t)i + c] = (gm[c] / a.duration) * a.t;
            g_scaling[3 * (size_t)i + c] = gs[c] * expf(scaling[3 * (size_t)i + c]);
        }
        reinterpret_cast<float4 *>(g_rotation)[i] = gq;
        const float so = sigmoidf_(opacity[i]);
        g_opacity[i] = go * (so * (1.0f - so));
    } else {
        const size_t j = (size_t)(i - a.Ns);
        // Hermite weights back onto the four keyframes (the other K-4 slices were zero-filled by the host memset)
        // sliced: g_xyz_motion is [Nd,4,3] (keyframes k-1..k+2), g_rotation_motion [Nd,2,4] (keyframes k, k+1): nothing else exists
        float *gy = g_xyz_motion + (sliced ? j * 12 : (j * a.K + (a.k - 1)) * 3);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            gy[c] = -(a.h10 * gm[c]) / 2.0f;
            gy[3 + c] = a.h00 * gm[c] - (a.h11 * gm[c]) / 2.0f;
            gy[6 + c] = (a.h10 * gm[c]) / 2.0f + a.h01 * gm[c];
            gy[9 + c] = (a.h11 * gm[c]) / 2.0f;
        }
        const float4 *rq = reinterpret_cast<const float4 *>(rotation_motion) + j * a.K + a.k;
        const float4 r1 = rq[0], r2 = rq[1];
        const float q1[4] = { r1.x, r1.y, r1.z, r1.w }, q2[4] = { r2.x, r2.y, r2.z, r2.w };
        Slerp c;
        float qo[4];
        slerp_forward(q1, q2, a.delta, c, qo);
        const float G[4] = { gq.x, gq.y, gq.z, gq.w };
        float gq1[4], gq2[4];
        slerp_backward(c, a.delta, G, gq1, gq2);
        float4 *grq = reinterpret_cast<float4 *>(g_rotation_motion) + (sliced ? j * 2 : j * a.K + a.k);
        grq[0] = make_float4(gq1[0], gq1[1], gq1[2], gq1[3]);
        grq[1] = make_float4(gq2[0], gq2[1], gq2[2], gq2[3]);
        // opacity = bigaussian(centres, log-widths, tau) * sigmoid(o)   (c_gaussian_model.py:363-366)
        BiGauss b;
        bigaussian_forward(dur_center[2 * j], dur_center[2 * j + 1], dur_var[2 * j], dur_var[2 * j + 1], a.tau, a.var_min, b);
        const float sg = sigmoidf_(opacity_motion[j]);
        g_opacity_motion[j] = go * b.out * (sg * (1.0f - sg));
        const float g_big = b.inside ? 0.f : go * sg;
        const float g_u = -g_big * b.o;
        const float g_m = g_u * 2.0f * b.m / (b.D * b.D);
        const float g_D = -2.0f * b.u / b.D * g_u;
        const float g_v = g_D * expf(b.v);
        g_dur_center[2 * j + b.arg] = -g_m;
        g_dur_center[2 * j + (1 - b.arg)] = 0.f;
        g_dur_var[2 * j + (b.after ? 1 : 0)] = g_v;
        g_dur_var[2 * j + (b.after ? 0 : 1)] = 0.f;
#pragma unroll
        for (int c2 = 0; c2 < 3; c2++) g_scaling_motion[3 * j + c2] = gs[c2] * expf(scaling_motion[3 * j + c2]);
    }
}

thread_local char g_attr_err[256] = "";

}  // namespace

extern "C" {

const char *ex4d_attributes_last_error(void) { return g_attr_err; }

int ex4d_attributes_forward(const Ex4dAttrParams *a,
    const float *xyz, const float *xyz_disp, const float *rotation, const float *opacity, const float *scaling,
    const float *features_dc, const float *features_rest,
    const float *xyz_motion, const float *rotation_motion, const float *opacity_motion, const float *dur_center,
    const float *dur_var, const float *scaling_motion, const float *features_dc_motion, const float *features_rest_motion,
    float *means3D, float *rotations, float *opacities, float *scales, float *shs, void *stream_)
{
    g_attr_err[0] = 0;
    hipStream_t stream = (hipStream_t)stream_;
    if (!a || a->Ns < 0 || a->Nd < 0) { snprintf(g_attr_err, sizeof(g_attr_err), "bad parameters"); return EX4D_ERR_ARG; }
    const int N = a->Ns + a->Nd;
    if (N == 0) return EX4D_OK;
    if (a->Nd > 0 && (a->k < 1 || a->k + 2 >= a->K)) { snprintf(g_attr_err, sizeof(g_attr_err), "keyframe index %d needs k-1..k+2 inside [0,%d)", a->k, a->K); return EX4D_ERR_ARG; }
    hipLaunchKernelGGL(attributes_fwd_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, *a, xyz, xyz_disp, rotation, opacity, scaling,
        xyz_motion, rotation_motion, opacity_motion, dur_center, dur_var, scaling_motion, means3D, rotations, opacities, scales);
    if (shs) {       // NULL: the caller feeds the rasterizer the four feature tensors directly (Ex4dSplitSH), nothing to gather
        const size_t total = (size_t)N * 12;
        const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
        hipLaunchKernelGGL(features_kernel<true>, dim3(blocks), dim3(256), 0, stream, a->Ns, a->Nd, (float *)features_dc, (float *)features_rest,
            (float *)features_dc_motion, (float *)features_rest_motion, shs);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_attr_err, sizeof(g_attr_err), "launch failed: %s", hipGetErrorString(e)); return EX4D_ERR_HIP; }
    return EX4D_OK;
}

static int attributes_backward_impl(const Ex4dAttrParams *a, int sliced,
    const float *opacity, const float *scaling, const float *rotation_motion, const float *opacity_motion,
    const float *dur_center, const float *dur_var, const float *scaling_motion,
    const float *g_means3D, const float *g_rotations, const float *g_opacities, const float *g_scales, const float *g_shs,
    float *g_xyz, float *g_xyz_disp, float *g_rotation, float *g_opacity, float *g_scaling, float *g_features_dc, float *g_features_rest,
    float *g_xyz_motion, float *g_rotation_motion, float *g_opacity_motion, float *g_dur_center, float *g_dur_var,
    float *g_scaling_motion, float *g_features_dc_motion, float *g_features_rest_motion, void *stream_)
{
    g_attr_err[0] = 0;
    hipStream_t stream = (hipStream_t)stream_;
    if (!a || a->Ns < 0 || a->Nd < 0) { snprintf(g_attr_err, sizeof(g_attr_err), "bad parameters"); return EX4D_ERR_ARG; }
    const int N = a->Ns + a->Nd;
    if (N == 0) return EX4D_OK;
    if (a->Nd > 0) {
        if (a->k < 1 || a->k + 2 >= a->K) { snprintf(g_attr_err, sizeof(g_attr_err), "keyframe index out of range"); return EX4D_ERR_ARG; }
        // dense gradients of the keyframe tensors: only 4 (xyz) / 2 (rotation) of the K slices are non-zero
        if (!sliced && ((((uintptr_t)g_xyz_motion | (uintptr_t)g_rotation_motion) & 15) != 0 ||
            ex4d_launch_zero(g_xyz_motion, (size_t)a->Nd * a->K * 3 * sizeof(float), stream) != hipSuccess ||
            ex4d_launch_zero(g_rotation_motion, (size_t)a->Nd * a->K * 4 * sizeof(float), stream) != hipSuccess)) {
            snprintf(g_attr_err, sizeof(g_attr_err), "clearing the dense keyframe gradients failed (they must be 16-byte aligned)"); return EX4D_ERR_HIP;
        }
    }
    hipLaunchKernelGGL(attributes_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, *a, opacity, scaling, rotation_motion, opacity_motion,
        dur_center, dur_var, scaling_motion, g_means3D, g_rotations, g_opacities, g_scales,
        g_xyz, g_xyz_disp, g_rotation, g_opacity, g_scaling, g_xyz_motion, g_rotation_motion, g_opacity_motion, g_dur_center, g_dur_var,
        g_scaling_motion, sliced);
    if (g_shs) {     // NULL: dL/dsh was written into the four gradient tensors by the rasterizer itself (Ex4dSplitSHGrad)
        const size_t total = (size_t)N * 12;
        const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
        hipLaunchKernelGGL(features_kernel<false>, dim3(blocks), dim3(256), 0, stream, a->Ns, a->Nd, g_features_dc, g_features_rest,
            g_features_dc_motion, g_features_rest_motion, (float *)g_shs);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_attr_err, sizeof(g_attr_err), "launch failed: %s", hipGetErrorString(e)); return EX4D_ERR_HIP; }
    return EX4D_OK;
}

int ex4d_attributes_backward(const Ex4dAttrParams *a,
    const float *opacity, const float *scaling, const float *rotation_motion, const float *opacity_motion,
    const float *dur_center, const float *dur_var, const float *scaling_motion,
    const float *g_means3D, const float *g_rotations, const float *g_opacities, const float *g_scales, const float *g_shs,
    float *g_xyz, float *g_xyz_disp, float *g_rotation, float *g_opacity, float *g_scaling, float *g_features_dc, float *g_features_rest,
    float *g_xyz_motion, float *g_rotation_motion, float *g_opacity_motion, float *g_dur_center, float *g_dur_var,
    float *g_scaling_motion, float *g_features_dc_motion, float *g_features_rest_motion, void *stream_)
{
    return attributes_backward_impl(a, 0, opacity, scaling, rotation_motion, opacity_motion, dur_center, dur_var, scaling_motion, g_means3D, g_rotations,
        g_opacities, g_scales, g_shs, g_xyz, g_xyz_disp, g_rotation, g_opacity, g_scaling, g_features_dc, g_features_rest, g_xyz_motion,
        g_rotation_motion, g_opacity_motion, g_dur_center, g_dur_var, g_scaling_motion, g_features_dc_motion, g_features_rest_motion, stream_);
}

int ex4d_attributes_backward_sliced(const Ex4dAttrParams *a,
    const float *opacity, const float *scaling, const float *rotation_motion, const float *opacity_motion,
    const float *dur_center, const float *dur_var, const float *scaling_motion,
    const float *g_means3D, const float *g_rotations, const float *g_opacities, const float *g_scales, const float *g_shs,
    float *g_xyz, float *g_xyz_disp, float *g_rotation, float *g_opacity, float *g_scaling, float *g_features_dc, float *g_features_rest,
    float *g_xyz_motion_slices, float *g_rotation_motion_slices, float *g_opacity_motion, float *g_dur_center, float *g_dur_var,
    float *g_scaling_motion, float *g_features_dc_motion, float *g_features_rest_motion, int32_t *slices, void *stream_)
{
    if (a && slices) { slices[0] = a->k - 1; slices[1] = 4; slices[2] = a->k; slices[3] = 2; }
    return attributes_backward_impl(a, 1, opacity, scaling, rotation_motion, opacity_motion, dur_center, dur_var, scaling_motion, g_means3D, g_rotations,
        g_opacities, g_scales, g_shs, g_xyz, g_xyz_disp, g_rotation, g_opacity, g_scaling, g_features_dc, g_features_rest, g_xyz_motion_slices,
        g_rotation_motion_slices, g_opacity_motion, g_dur_center, g_dur_var, g_scaling_motion, g_features_dc_motion, g_features_rest_motion, stream_);
}

}  // extern "C"
