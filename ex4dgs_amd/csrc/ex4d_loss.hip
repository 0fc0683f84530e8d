// Fused L1 + SSIM loss, forward and backward, for gfx950.
//
// Replaces l1_loss + ssim/_ssim of the reference (utils/loss_utils.py:22-25, :47-81; used at train.py:144-151) and their
// autograd graph: five depthwise 11x11 conv2d forward, their transposes backward, ~25 element-wise kernels.  Measured on
// MI355X at 1352x1014 the torch version costs 11.7 ms per iteration -- ten times the rasterizer it scores.
//
// The 11x11 window is the outer product of a 1-D Gaussian, so every convolution is separable.  Round 4: ROLLING WINDOW.  A workgroup
// owns a strip of 64 output columns and walks down a segment of SEG = 48 output rows, four image rows per iteration: the rows' 11-tap row
// pass (five moment maps x, y, x^2, y^2, xy per channel in the forward, the three derivative maps in the backward) goes into a ring of
// 16 rows in LDS, the column pass of the four output rows whose window is now complete reads 11 ring rows each.  Every input row is
// read ONCE per strip (rounds 2-3: 16x16 output tiles with a 26x26 halo each read 2.6x the image, and their 104-byte halo rows pulled
// whole 128-byte lines: 237 / 364 MB of HBM traffic per launch against 93 / 98 MB of algorithmic bytes); what is left is the 10-column
// overlap of neighbouring strips (74 / 64; neighbouring strips are given to the SAME XCD back to back, so most of it hits in that
// XCD's L2) and the 10 warm-up rows of a segment (74 / 64).  The next iteration's rows are in flight (registers) while the current
// one is convolved; two workgroup barriers per iteration (the row buffers alternate).
// Forward emits the loss partial sums, the two error maps and, per pixel and channel, the three partial derivatives of the SSIM map
// the backward needs; backward convolves those three maps with the same (symmetric) window and combines them with the pixel values:
//     d(sum ssim)/dx_p = conv(A)_p + 2 x_p conv(B)_p + y_p conv(C)_p,   A = dS/dmu1, B = dS/dE[x^2], C = dS/dE[xy].
#include "ex4d_internal.h"
#include "../../include/ex4d_loss.h"
#include <cstdio>

namespace {

#define LH 5                        // window half width
#define SW 64                       // output columns of a strip
#define SIN (SW + 2 * LH)           // 74 input columns
#define SEG 48                      // output rows of a segment (22 x 22 = 484 workgroups at 1352x1014: one round at 2 per CU)
#define RPI 4                       // image rows per iteration (256 threads = RPI rows x SW columns)
#define RING 16                     // rows of row-pass results kept in LDS (>= 11 + RPI - 1, power of two)
#define CG 3                        // channels convolved together (the reference's images are RGB; more channels run in groups)

// static LDS of the two kernels: the input double buffer + the ring of row-pass results.  Both exceed the 64 KB a workgroup may take on
// older parts and are sized for gfx950's 160 KB per CU (two workgroups per CU): this file builds for gfx950-class LDS only
static_assert(sizeof(float) * (2 * 2 * CG * RPI * (SIN + 2) + CG * 5 * RING * SW + 8) <= 80 * 1024, "l1_ssim_fwd_kernel: two workgroups per CU need <= 80 KB of LDS each");
static_assert(sizeof(float) * (2 * 3 * CG * RPI * (SIN + 2) + CG * 3 * RING * SW) <= 80 * 1024, "l1_ssim_bwd_kernel: two workgroups per CU need <= 80 KB of LDS each");

struct Window { float w[EX4D_SSIM_WINDOW]; };

__device__ __forceinline__ float load_or_zero(const float *__restrict__ p, int x, int y, int W, int H)
{
    return (x >= 0 && x < W && y >= 0 && y < H) ? p[(size_t)y * W + x] : 0.f;    // zero padding (conv2d padding=5)
}

// workgroup -> (strip, segment): consecutive workgroup ids round-robin over the 8 XCDs, so XCD x takes the contiguous run of work items
// [x * per, (x + 1) * per) in row-major (segment, strip) order: horizontally neighbouring strips share their halo columns in one L2
__device__ __forceinline__ int work_item_of_block(int nwork)
{
    const int per = (nwork + 7) >> 3;
    return (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
}

__global__ __launch_bounds__(256) void l1_ssim_fwd_kernel(int C, int H, int W, const float *__restrict__ img,
    const float *__restrict__ gt, Window win, float *__restrict__ l1_errors, float *__restrict__ ssim_errors,
    float *__restrict__ dmaps, float *__restrict__ partials, int nsx, int nsy)
{
    __shared__ float s_in[2][2][CG][RPI][SIN + 2];      // [buffer][x | y][channel][row][column]
    __shared__ float s_ring[CG][5][RING][SW];           // row-pass results of the last RING image rows
    __shared__ float s_red[2][4];
    const int nwork = nsx * nsy;
    const int wi = work_item_of_block(nwork);
    const int col = threadIdx.x & (SW - 1), rsub = threadIdx.x >> 6;
    const size_t HW = (size_t)H * W;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    float l1_total = 0.f, ssim_total = 0.f;
    if (wi < nwork) {
        const int x0 = (wi % nsx) * SW, y0 = (wi / nsx) * SEG;
        const int rows_out = (H - y0) < SEG ? (H - y0) : SEG;
        const int n_in = rows_out + 2 * LH;              // image rows y0 - 5 .. y0 + rows_out + 4
        const int n_it = (n_in + RPI - 1) / RPI;
        const int px = x0 + col;
        // an iteration's rows: every thread takes column (tid & 63) of row (tid >> 6) in every plane (image x channel), threads 0 .. 39
        // also one of the 4 x 10 halo columns 64 .. 73 -- two loads per plane with ONE pixel offset each (a flat element index decoded
        // per load kept ~100 registers of hoisted address arithmetic alive)
        const int hrow = (int)threadIdx.x / (2 * LH), hcol = SW + (int)threadIdx.x % (2 * LH);
        const bool has_halo = threadIdx.x < RPI * 2 * LH;
        for (int c0 = 0; c0 < C; c0 += CG) {
            const int nc = (C - c0) < CG ? (C - c0) : CG;
            const bool first_group = c0 == 0, last_group = c0 + CG >= C;
            float rm[2 * CG], rh[2 * CG];
            auto fetch = [&](int it) {
                const int ym = y0 - LH + it * RPI + rsub, xm = x0 - LH + col;
                const int yh = y0 - LH + it * RPI + hrow, xh = x0 - LH + hcol;
                const bool okm = xm >= 0 && xm < W && ym >= 0 && ym < H, okh = has_halo && xh >= 0 && xh < W && yh >= 0 && yh < H;
                const size_t om = (size_t)ym * W + xm, oh = (size_t)yh * W + xh;
#pragma unroll
                for (int pl = 0; pl < 2 * CG; pl++) {
                    const int ch = pl % CG;
                    const float *src = ((pl / CG) ? gt : img) + (size_t)(c0 + ch) * HW;
                    rm[pl] = (okm && ch < nc) ? src[om] : 0.f;       // zero padding (conv2d padding=5)
                    rh[pl] = (okh && ch < nc) ? src[oh] : 0.f;
                }
            };
            auto commit = [&](int buf) {
#pragma unroll
                for (int pl = 0; pl < 2 * CG; pl++) {
                    s_in[buf][pl / CG][pl % CG][rsub][col] = rm[pl];
                    if (has_halo) s_in[buf][pl / CG][pl % CG][hrow][hcol] = rh[pl];
                }
            };
            __syncthreads();                             // (the previous channel group is done with both buffers and the ring)
            fetch(0);
            commit(0);
            __syncthreads();
            for (int it = 0; it < n_it; it++) {
                const int buf = it & 1;
                if (it + 1 < n_it) fetch(it + 1);        // in flight while this iteration's rows are convolved
                // ---- row pass of image row rin (relative to y0 - 5), five moment maps per channel -> ring; the L1 term of the pixels of
                // this row (they are output pixels when the row lies inside the segment)
                const int rin = it * RPI + rsub;
                if (rin < n_in) {
                    float l1 = 0.f;
#pragma unroll 1
                    for (int ch = 0; ch < nc; ch++) {
                        const float *sx = &s_in[buf][0][ch][rsub][col], *sy = &s_in[buf][1][ch][rsub][col];
                        float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
                        for (int k = 0; k < EX4D_SSIM_WINDOW; k++) {
                            const float a = sx[k], b = sy[k], wk = win.w[k];
                            m1 += wk * a; m2 += wk * b; e11 += wk * (a * a); e22 += wk * (b * b); e12 += wk * (a * b);
                        }
                        const int slot = rin & (RING - 1);
                        s_ring[ch][0][slot][col] = m1; s_ring[ch][1][slot][col] = m2; s_ring[ch][2][slot][col] = e11;
                        s_ring[ch][3][slot][col] = e22; s_ring[ch][4][slot][col] = e12;
                        l1 += fabsf(sx[LH] - sy[LH]);
                    }
                    const int ro = rin - LH;             // this image row as an output row of the segment
                    if (ro >= 0 && ro < rows_out && px < W) {
                        l1_total += l1;
                        if (l1_errors) {
                            float *o = l1_errors + (size_t)(y0 + ro) * W + px;
                            const float v = first_group ? l1 : *o + l1;
                            *o = last_group ? v / (float)C : v;
                        }
                    }
                }
                __syncthreads();
                // ---- column pass of output row ro: its window (image rows ro .. ro + 10 relative to y0 - 5) is complete
                const int ro = it * RPI + rsub - 2 * LH;
                if (ro >= 0 && ro < rows_out && px < W) {
                    float ssim = 0.f;
#pragma unroll 1
                    for (int ch = 0; ch < nc; ch++) {
                        float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
                        for (int k = 0; k < EX4D_SSIM_WINDOW; k++) {
                            const float wk = win.w[k];
                            const int slot = (ro + k) & (RING - 1);
                            mu1 += wk * s_ring[ch][0][slot][col]; mu2 += wk * s_ring[ch][1][slot][col];
                            e11 += wk * s_ring[ch][2][slot][col]; e22 += wk * s_ring[ch][3][slot][col]; e12 += wk * s_ring[ch][4][slot][col];
                        }
                        // utils/loss_utils.py:61-74
                        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
                        const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
                        const float a1 = 2.f * mu12 + C1, a2 = 2.f * s12 + C2, b1 = mu1_sq + mu2_sq + C1, b2 = s1 + s2 + C2;
                        const float inv = 1.f / (b1 * b2);
                        const float S = (a1 * a2) * inv;
                        ssim += S;
                        // partial derivatives of S w.r.t. (mu1, E[x^2], E[xy]) with the gt-side moments held fixed
                        const float dSda1 = a2 * inv, dSda2 = a1 * inv, dSdb1 = -S / b1, dSdb2 = -S / b2;
                        const float dA = 2.f * mu2 * (dSda1 - dSda2) + 2.f * mu1 * (dSdb1 - dSdb2);
                        const size_t o = (size_t)(c0 + ch) * HW + (size_t)(y0 + ro) * W + px;
                        dmaps[o] = dA;
                        dmaps[(size_t)C * HW + o] = dSdb2;
                        dmaps[2 * (size_t)C * HW + o] = 2.f * dSda2;
                    }
                    ssim_total += ssim;
                    if (ssim_errors) {
                        float *o = ssim_errors + (size_t)(y0 + ro) * W + px;
                        const float v = first_group ? ssim : *o + ssim;
                        *o = last_group ? v / (float)C : v;
                    }
                }
                if (it + 1 < n_it) commit(buf ^ 1);
                __syncthreads();
            }
        }
    }
    // per-workgroup partial sums (no single-address atomics: they serialise ~12 ns each)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { l1_total += __shfl_xor(l1_total, o, 64); ssim_total += __shfl_xor(ssim_total, o, 64); }
    if (lane == 0) { s_red[0][wave] = l1_total; s_red[1][wave] = ssim_total; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
        partials[2 * blockIdx.x + 1] = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
    }
}

__global__ __launch_bounds__(256) void l1_ssim_finish_kernel(int nblocks, const float *__restrict__ partials, double inv_count,
    float lambda_dssim, float *__restrict__ loss)
{
    __shared__ double s[2][4];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) { a += (double)partials[2 * i]; b += (double)partials[2 * i + 1]; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    if (lane == 0) { s[0][wave] = a; s[1][wave] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float l1 = (float)((s[0][0] + s[0][1] + s[0][2] + s[0][3]) * inv_count);
        const float ss = (float)((s[1][0] + s[1][1] + s[1][2] + s[1][3]) * inv_count);
        loss[0] = (1.0f - lambda_dssim) * l1 + lambda_dssim * (1.0f - ss);      // train.py:145
    }
}

__global__ __launch_bounds__(256) void l1_ssim_bwd_kernel(int C, int H, int W, const float *__restrict__ img,
    const float *__restrict__ gt, Window win, const float *__restrict__ dmaps, const float *__restrict__ grad_loss,
    float lambda_dssim, float inv_count, float *__restrict__ grad_img, int nsx, int nsy)
{
    __shared__ float s_in[2][3][CG][RPI][SIN + 2];      // [buffer][map A | B | C][channel][row][column]
    __shared__ float s_ring[CG][3][RING][SW];
    const int nwork = nsx * nsy;
    const int wi = work_item_of_block(nwork);
    if (wi >= nwork) return;                             // (uniform: the whole workgroup)
    const int col = threadIdx.x & (SW - 1), rsub = threadIdx.x >> 6;
    const size_t HW = (size_t)H * W;
    const float gl = grad_loss[0];
    const int x0 = (wi % nsx) * SW, y0 = (wi / nsx) * SEG;
    const int rows_out = (H - y0) < SEG ? (H - y0) : SEG;
    const int n_in = rows_out + 2 * LH;
    const int n_it = (n_in + RPI - 1) / RPI;
    const int px = x0 + col;
    const int hrow = (int)threadIdx.x / (2 * LH), hcol = SW + (int)threadIdx.x % (2 * LH);
    const bool has_halo = threadIdx.x < RPI * 2 * LH;
    for (int c0 = 0; c0 < C; c0 += CG) {
        const int nc = (C - c0) < CG ? (C - c0) : CG;
        float rm[3 * CG], rh[3 * CG];
        auto fetch = [&](int it) {
            const int ym = y0 - LH + it * RPI + rsub, xm = x0 - LH + col;
            const int yh = y0 - LH + it * RPI + hrow, xh = x0 - LH + hcol;
            const bool okm = xm >= 0 && xm < W && ym >= 0 && ym < H, okh = has_halo && xh >= 0 && xh < W && yh >= 0 && yh < H;
            const size_t om = (size_t)ym * W + xm, oh = (size_t)yh * W + xh;
#pragma unroll
            for (int pl = 0; pl < 3 * CG; pl++) {
                const int ch = pl % CG;
                const float *src = dmaps + ((size_t)(pl / CG) * C + c0 + ch) * HW;
                rm[pl] = (okm && ch < nc) ? src[om] : 0.f;
                rh[pl] = (okh && ch < nc) ? src[oh] : 0.f;
            }
        };
        auto commit = [&](int buf) {
#pragma unroll
            for (int pl = 0; pl < 3 * CG; pl++) {
                s_in[buf][pl / CG][pl % CG][rsub][col] = rm[pl];
                if (has_halo) s_in[buf][pl / CG][pl % CG][hrow][hcol] = rh[pl];
            }
        };
        __syncthreads();
        fetch(0);
        commit(0);
        __syncthreads();
        for (int it = 0; it < n_it; it++) {
            const int buf = it & 1;
            if (it + 1 < n_it) fetch(it + 1);
            const int rin = it * RPI + rsub;
            if (rin < n_in) {
#pragma unroll 1
                for (int ch = 0; ch < nc; ch++) {
                    const float *sa = &s_in[buf][0][ch][rsub][col], *sb = &s_in[buf][1][ch][rsub][col], *sc = &s_in[buf][2][ch][rsub][col];
                    float a = 0.f, b = 0.f, cc = 0.f;
#pragma unroll
                    for (int k = 0; k < EX4D_SSIM_WINDOW; k++) { const float wk = win.w[k]; a += wk * sa[k]; b += wk * sb[k]; cc += wk * sc[k]; }
                    const int slot = rin & (RING - 1);
                    s_ring[ch][0][slot][col] = a; s_ring[ch][1][slot][col] = b; s_ring[ch][2][slot][col] = cc;
                }
            }
            __syncthreads();
            const int ro = it * RPI + rsub - 2 * LH;
            if (ro >= 0 && ro < rows_out && px < W) {
#pragma unroll 1
                for (int ch = 0; ch < nc; ch++) {
                    float ca = 0.f, cb = 0.f, ccv = 0.f;
#pragma unroll
                    for (int k = 0; k < EX4D_SSIM_WINDOW; k++) {
                        const float wk = win.w[k];
                        const int slot = (ro + k) & (RING - 1);
                        ca += wk * s_ring[ch][0][slot][col]; cb += wk * s_ring[ch][1][slot][col]; ccv += wk * s_ring[ch][2][slot][col];
                    }
                    const size_t o = (size_t)(c0 + ch) * HW + (size_t)(y0 + ro) * W + px;
                    const float xv = img[o], yv = gt[o];
                    const float dssim = ca + 2.f * xv * cb + yv * ccv;                   // d(sum of ssim_map)/dx_p
                    const float diff = xv - yv;
                    const float sgn = (diff > 0.f) ? 1.f : ((diff < 0.f) ? -1.f : 0.f);  // d|x - y|/dx
                    grad_img[o] = gl * ((1.0f - lambda_dssim) * sgn * inv_count - lambda_dssim * dssim * inv_count);
                }
            }
            if (it + 1 < n_it) commit(buf ^ 1);
            __syncthreads();
        }
    }
}

thread_local char g_loss_err[256] = "";

bool check_args(int C, int H, int W, const void *a, const void *b, const float *window)
{
    if (C <= 0 || H <= 0 || W <= 0 || !a || !b || !window) { snprintf(g_loss_err, sizeof(g_loss_err), "bad argument"); return false; }
    return true;
}

}  // namespace

extern "C" {

const char *ex4d_loss_last_error(void) { return g_loss_err; }

static inline int strips_of(int W) { return (W + SW - 1) / SW; }
static inline int segments_of(int H) { return (H + SEG - 1) / SEG; }
static inline int blocks_of(int H, int W) { return 8 * ((strips_of(W) * segments_of(H) + 7) / 8); }       // (padded: the XCD-aware work-item map)

size_t ex4d_l1_ssim_scratch_floats(int32_t H, int32_t W) { return 2 * (size_t)blocks_of(H, W) + 64; }

int ex4d_l1_ssim_forward(int32_t C, int32_t H, int32_t W, const float *img, const float *gt, float lambda_dssim,
                         const float *window, float *loss, float *l1_errors, float *ssim_errors, float *dmaps, float *scratch,
                         void *stream_)
{
    g_loss_err[0] = 0;
    if (!check_args(C, H, W, img, gt, window) || !loss || !dmaps || !scratch) { snprintf(g_loss_err, sizeof(g_loss_err), "bad argument"); return EX4D_ERR_ARG; }
    hipStream_t stream = (hipStream_t)stream_;
    Window win;
    for (int i = 0; i < EX4D_SSIM_WINDOW; i++) win.w[i] = window[i];
    const int nblocks = blocks_of(H, W);
    hipLaunchKernelGGL(l1_ssim_fwd_kernel, dim3(nblocks), dim3(256), 0, stream, C, H, W, img, gt, win, l1_errors, ssim_errors, dmaps, scratch,
                       strips_of(W), segments_of(H));
    hipLaunchKernelGGL(l1_ssim_finish_kernel, dim3(1), dim3(256), 0, stream, nblocks, scratch,
                       1.0 / ((double)C * H * W), lambda_dssim, loss);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_loss_err, sizeof(g_loss_err), "launch failed: %s", hipGetErrorString(e)); return EX4D_ERR_HIP; }
    return EX4D_OK;
}

int ex4d_l1_ssim_backward(int32_t C, int32_t H, int32_t W, const float *img, const float *gt, float lambda_dssim,
                          const float *window, const float *dmaps, const float *grad_loss, float *grad_img, void *stream_)
{
    g_loss_err[0] = 0;
    if (!check_args(C, H, W, img, gt, window) || !dmaps || !grad_loss || !grad_img) { snprintf(g_loss_err, sizeof(g_loss_err), "bad argument"); return EX4D_ERR_ARG; }
    hipStream_t stream = (hipStream_t)stream_;
    Window win;
    for (int i = 0; i < EX4D_SSIM_WINDOW; i++) win.w[i] = window[i];
    hipLaunchKernelGGL(l1_ssim_bwd_kernel, dim3(blocks_of(H, W)), dim3(256), 0, stream, C, H, W, img, gt, win, dmaps, grad_loss, lambda_dssim,
                       (float)(1.0 / ((double)C * H * W)), grad_img, strips_of(W), segments_of(H));
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_loss_err, sizeof(g_loss_err), "launch failed: %s", hipGetErrorString(e)); return EX4D_ERR_HIP; }
    return EX4D_OK;
}

}  // extern "C"
