/*
 * ex4d_oracle.c -- TEST INFRASTRUCTURE ONLY (never imported by the product path).
 *
 * Scalar CPU restatement of the Ex4DGS differentiable rasterizer
 * (reference: submodules/diff_gaussian_rasterization_df, abbreviated below as
 *   CR/  = submodules/diff_gaussian_rasterization_df/cuda_rasterizer/
 *   DGR/ = submodules/diff_gaussian_rasterization_df/ ).
 * Every function cites the reference file:line whose expression tree it follows.
 *
 * PARITY STATUS: "parity unpinned" by the reference itself -- the reference ships
 * no tests, golden vectors or known-answer fixtures for this path and its CUDA
 * kernels cannot be compiled or run here (no nvcc / NVIDIA device).  The oracle
 * is pinned instead by (a) an independent vectorised pure-PyTorch formulation
 * (oracle/oracle_torch.py), (b) torch autograd for the true-gradient subset,
 * (c) the reference's own Python (utils/sh_utils.eval_sh, camera helpers, model
 * getters) imported in the build container -> tests/golden/.
 *
 * Floating-point contract: compiled with -ffp-contract=off (no FMA fusion), SSE
 * float arithmetic, correctly rounded / and sqrt, and the reference's double
 * promotions kept (CR/auxiliary.h:43, :284; CR/forward.cu:112-116).  The HIP
 * preprocess kernels follow the same contract, so every integer-deciding value
 * (cull, radii, rect, tiles_touched, depth key bits, sort order, ranges) is
 * compared bit-exact.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BLOCK_X 16           /* CR/config.h:16 */
#define BLOCK_Y 16           /* CR/config.h:17 */
#define NUM_CHANNELS 3       /* CR/config.h:15 */

/* CR/auxiliary.h:22-39 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f };
static const float SH_C3[] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f };

typedef struct { float x, y, z; } v3;
typedef struct { float c[3][3]; } m3;   /* GLM convention: c[col][row] */

/* GLM 1.0.1 type_mat3x3.inl operator*(mat3, mat3): Result[c][r] = sum_k A[k][r]*B[c][k], k = 0,1,2 left to right */
static m3 m3_mul(const m3 *A, const m3 *B)
{
    m3 R;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++)
            R.c[c][r] = A->c[0][r] * B->c[c][0] + A->c[1][r] * B->c[c][1] + A->c[2][r] * B->c[c][2];
    return R;
}
static m3 m3_transpose(const m3 *A)
{
    m3 R;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++)
            R.c[c][r] = A->c[r][c];
    return R;
}
/* glm::mat3(x0,y0,z0, x1,y1,z1, x2,y2,z2): columns */
static m3 m3_cols(float x0, float y0, float z0, float x1, float y1, float z1, float x2, float y2, float z2)
{
    m3 R;
    R.c[0][0] = x0; R.c[0][1] = y0; R.c[0][2] = z0;
    R.c[1][0] = x1; R.c[1][1] = y1; R.c[1][2] = z1;
    R.c[2][0] = x2; R.c[2][1] = y2; R.c[2][2] = z2;
    return R;
}

/* CUDA float->int conversion semantics (cvt.rzi.s32.f32): truncate, saturate, NaN -> 0 */
static int f2i(float f)
{
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
}
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* CR/auxiliary.h:41-44 */
static float ndc2Pix(float v, int S)
{
    return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5);
}

/* CR/auxiliary.h:46-56 */
static void getRect(float px, float py, int max_radius, int gx, int gy, int *rmin, int *rmax)
{
    rmin[0] = imin(gx, imax(0, f2i((px - (float)max_radius) / (float)BLOCK_X)));
    rmin[1] = imin(gy, imax(0, f2i((py - (float)max_radius) / (float)BLOCK_Y)));
    rmax[0] = imin(gx, imax(0, f2i((px + (float)max_radius + (float)BLOCK_X - (float)1) / (float)BLOCK_X)));
    rmax[1] = imin(gy, imax(0, f2i((py + (float)max_radius + (float)BLOCK_Y - (float)1) / (float)BLOCK_Y)));
}

/* CR/auxiliary.h:68-76 */
static v3 transformPoint4x3(v3 p, const float *m)
{
    v3 t;
    t.x = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
    t.y = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
    t.z = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
    return t;
}
/* CR/auxiliary.h:78-87 */
static void transformPoint4x4(v3 p, const float *m, float out[4])
{
    out[0] = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
    out[1] = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
    out[2] = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
    out[3] = m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15];
}
/* CR/auxiliary.h:89-97 */
static v3 transformVec4x3Transpose(v3 p, const float *m)
{
    v3 t;
    t.x = m[0] * p.x + m[1] * p.y + m[2] * p.z;
    t.y = m[4] * p.x + m[5] * p.y + m[6] * p.z;
    t.z = m[8] * p.x + m[9] * p.y + m[10] * p.z;
    return t;
}
/* CR/auxiliary.h:235-245 */
static v3 dnormvdv(v3 v, v3 dv)
{
    float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    v3 r;
    r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
    r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
    r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
    return r;
}

/* CR/auxiliary.h:267-294 (prefiltered trap omitted: returns -1 through the caller instead) */
static int in_frustum(int idx, const float *orig_points, const float *viewmatrix, const float *projmatrix,
                      float min_depth, float max_depth, v3 *p_view)
{
    v3 p_orig = { orig_points[3 * idx], orig_points[3 * idx + 1], orig_points[3 * idx + 2] };
    float p_hom[4];
    transformPoint4x4(p_orig, projmatrix, p_hom);
    float p_w = 1.0f / (p_hom[3] + 0.0000001f);
    float px = p_hom[0] * p_w, py = p_hom[1] * p_w;
    *p_view = transformPoint4x3(p_orig, viewmatrix);
    if ((p_view->z <= min_depth) || (p_view->z > max_depth) ||
        ((double)px < -1.3 || (double)px > 1.3 || (double)py < -1.3 || (double)py > 1.3))
        return 0;
    return 1;
}

/* CR/forward.cu:128-162 */
static void computeCov3D_fwd(const float *scale, float mod, const float *rot, float *cov3D)
{
    m3 S = m3_cols(1.0f, 0, 0, 0, 1.0f, 0, 0, 0, 1.0f);
    S.c[0][0] = mod * scale[0];
    S.c[1][1] = mod * scale[1];
    S.c[2][2] = mod * scale[2];
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];   /* not normalised (:137) */
    m3 R = m3_cols(
        1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
        2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
        2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    m3 M = m3_mul(&S, &R);
    m3 Mt = m3_transpose(&M);
    m3 Sigma = m3_mul(&Mt, &M);
    cov3D[0] = Sigma.c[0][0];
    cov3D[1] = Sigma.c[0][1];
    cov3D[2] = Sigma.c[0][2];
    cov3D[3] = Sigma.c[1][1];
    cov3D[4] = Sigma.c[1][2];
    cov3D[5] = Sigma.c[2][2];
}

/* CR/forward.cu:74-124; returns (a, b, c, coef) */
static void computeCov2D_fwd(v3 mean, float focal_x, float focal_y, float tan_fovx, float tan_fovy,
                             float kernel_size, const float *cov3D, const float *vm, float out[4])
{
    v3 t = transformPoint4x3(mean, vm);
    const float limx = 1.3f * tan_fovx;
    const float limy = 1.3f * tan_fovy;
    const float txtz = t.x / t.z;
    const float tytz = t.y / t.z;
    t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;

    m3 J = m3_cols(focal_x / t.z, 0.0f, -(focal_x * t.x) / (t.z * t.z),
                   0.0f, focal_y / t.z, -(focal_y * t.y) / (t.z * t.z),
                   0, 0, 0);
    m3 W = m3_cols(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
    m3 T = m3_mul(&W, &J);
    m3 Vrk = m3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    m3 Tt = m3_transpose(&T);
    m3 Vt = m3_transpose(&Vrk);
    m3 TV = m3_mul(&Tt, &Vt);
    m3 cov = m3_mul(&TV, &T);

    /* :112-118 -- float products, max/sqrt/compare in double */
    const float det_0 = (float)fmax(1e-6, (double)(cov.c[0][0] * cov.c[1][1] - cov.c[0][1] * cov.c[0][1]));
    const float det_1 = (float)fmax(1e-6, (double)((cov.c[0][0] + kernel_size) * (cov.c[1][1] + kernel_size) - cov.c[0][1] * cov.c[0][1]));
    float coef = (float)sqrt((double)det_0 / ((double)det_1 + 1e-6) + 1e-6);
    if ((double)det_0 <= 1e-6 || (double)det_1 <= 1e-6)
        coef = 0.0f;
    cov.c[0][0] += kernel_size;
    cov.c[1][1] += kernel_size;
    out[0] = cov.c[0][0]; out[1] = cov.c[0][1]; out[2] = cov.c[1][1]; out[3] = coef;
}

/* CR/forward.cu:20-71 */
static void computeColorFromSH_fwd(int idx, int deg, int max_coeffs, const float *means, const float *campos,
                                   const float *shs, uint8_t *clamped, float rgb[3])
{
    float dx = means[3 * idx] - campos[0], dy = means[3 * idx + 1] - campos[1], dz = means[3 * idx + 2] - campos[2];
    float len = sqrtf(dx * dx + dy * dy + dz * dz);     /* glm::length = sqrt(dot) */
    dx = dx / len; dy = dy / len; dz = dz / len;
    const float *sh = shs + (size_t)idx * max_coeffs * 3;
    for (int ch = 0; ch < 3; ch++) {
#define SH(k) sh[3 * (k) + ch]
        float result = SH_C0 * SH(0);
        if (deg > 0) {
            float x = dx, y = dy, z = dz;
            result = result - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z;
                float xy = x * y, yz = y * z, xz = x * z;
                result = result +
                    SH_C2[0] * xy * SH(4) +
                    SH_C2[1] * yz * SH(5) +
                    SH_C2[2] * (2.0f * zz - xx - yy) * SH(6) +
                    SH_C2[3] * xz * SH(7) +
                    SH_C2[4] * (xx - yy) * SH(8);
                if (deg > 2) {
                    result = result +
                        SH_C3[0] * y * (3.0f * xx - yy) * SH(9) +
                        SH_C3[1] * xy * z * SH(10) +
                        SH_C3[2] * y * (4.0f * zz - xx - yy) * SH(11) +
                        SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
                        SH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) +
                        SH_C3[5] * z * (xx - yy) * SH(14) +
                        SH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
                }
            }
        }
#undef SH
        result += 0.5f;
        clamped[3 * idx + ch] = (result < 0);
        rgb[ch] = fmaxf(result, 0.0f);
    }
}

/* CR/rasterizer_impl.cu:35-50 */
uint32_t ex4d_oracle_getHigherMsb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step;
        else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

/* ------------------------------------------------------------------------------------------
 * Forward, stage 1: CR/forward.cu:165-269 (preprocessCUDA) + inclusive scan CR/rasterizer_impl.cu:295.
 * All per-Gaussian arrays are caller-allocated (numpy).  Fields of culled Gaussians are left
 * as the caller initialised them (the reference leaves them uninitialised).
 * Returns num_rendered (= point_offsets[P-1]), or -1 if prefiltered && culled (device trap in the reference).
 * ------------------------------------------------------------------------------------------ */
int64_t ex4d_oracle_preprocess(
    int P, int D, int M,
    const float *means3D, const float *scales, float scale_modifier, const float *rotations,
    const float *opacities, const float *shs, const float *cov3D_precomp, const float *colors_precomp,
    const float *viewmatrix, const float *projmatrix, const float *cam_pos,
    int W, int H, float tan_fovx, float tan_fovy, float kernel_size,
    float min_depth, float max_depth, int prefiltered,
    int32_t *radii, float *means2D /*[P,2]*/, float *depths, float *cov3Ds /*[P,6]*/, float *rgb /*[P,3]*/,
    float *conic_opacity /*[P,4]*/, uint32_t *tiles_touched, uint8_t *clamped /*[P,3]*/,
    uint32_t *point_offsets)
{
    /* CR/rasterizer_impl.cu:237-238 */
    const float focal_y = H / (2.0f * tan_fovy);
    const float focal_x = W / (2.0f * tan_fovx);
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;

    for (int idx = 0; idx < P; idx++) {
        radii[idx] = 0;
        tiles_touched[idx] = 0;
        v3 p_view;
        if (!in_frustum(idx, means3D, viewmatrix, projmatrix, min_depth, max_depth, &p_view)) {
            if (prefiltered) return -1;
            continue;
        }
        v3 p_orig = { means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2] };
        float p_hom[4];
        transformPoint4x4(p_orig, projmatrix, p_hom);
        float p_w = 1.0f / (p_hom[3] + 0.0000001f);
        float p_proj_x = p_hom[0] * p_w, p_proj_y = p_hom[1] * p_w;

        const float *cov3D;
        if (cov3D_precomp) {
            cov3D = cov3D_precomp + (size_t)idx * 6;
        } else {
            computeCov3D_fwd(scales + 3 * (size_t)idx, scale_modifier, rotations + 4 * (size_t)idx, cov3Ds + (size_t)idx * 6);
            cov3D = cov3Ds + (size_t)idx * 6;
        }
        float cov[4];
        computeCov2D_fwd(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, kernel_size, cov3D, viewmatrix, cov);

        float det = (cov[0] * cov[2] - cov[1] * cov[1]);
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float conic[3] = { cov[2] * det_inv, -cov[1] * det_inv, cov[0] * det_inv };

        float mid = 0.5f * (cov[0] + cov[2]);
        float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        float pix_x = ndc2Pix(p_proj_x, W), pix_y = ndc2Pix(p_proj_y, H);
        int rmin[2], rmax[2];
        getRect(pix_x, pix_y, f2i(my_radius), gx, gy, rmin, rmax);
        if ((uint32_t)(rmax[0] - rmin[0]) * (uint32_t)(rmax[1] - rmin[1]) == 0) continue;

        if (!colors_precomp) {
            float c[3];
            computeColorFromSH_fwd(idx, D, M, means3D, cam_pos, shs, clamped, c);
            rgb[idx * 3 + 0] = c[0]; rgb[idx * 3 + 1] = c[1]; rgb[idx * 3 + 2] = c[2];
        }
        depths[idx] = p_view.z;
        radii[idx] = f2i(my_radius);
        means2D[2 * idx] = pix_x; means2D[2 * idx + 1] = pix_y;
        conic_opacity[4 * idx + 0] = conic[0];
        conic_opacity[4 * idx + 1] = conic[1];
        conic_opacity[4 * idx + 2] = conic[2];
        conic_opacity[4 * idx + 3] = opacities[idx] * cov[3];
        tiles_touched[idx] = (uint32_t)(rmax[1] - rmin[1]) * (uint32_t)(rmax[0] - rmin[0]);
    }
    /* cub::DeviceScan::InclusiveSum, CR/rasterizer_impl.cu:295 (uint32 wrap-around semantics) */
    uint32_t run = 0;
    for (int i = 0; i < P; i++) { run += tiles_touched[i]; point_offsets[i] = run; }
    return P > 0 ? (int64_t)(int32_t)run : 0;   /* read back as int, :298-299 */
}

/* CR/rasterizer_impl.cu:54-68 (checkFrustum / markVisible) */
void ex4d_oracle_mark_visible(int P, const float *means3D, const float *viewmatrix, const float *projmatrix,
                              float min_depth, float max_depth, uint8_t *present)
{
    for (int i = 0; i < P; i++) {
        v3 pv;
        present[i] = (uint8_t)in_frustum(i, means3D, viewmatrix, projmatrix, min_depth, max_depth, &pv);
    }
}

/* ------------------------------------------------------------------------------------------
 * Forward, stage 2: duplicateWithKeys (CR/rasterizer_impl.cu:72-113), stable radix sort on bits
 * [0, 32+bit) (cub::DeviceRadixSort::SortPairs, :321-326), identifyTileRanges (:118-140, memset :328).
 * ------------------------------------------------------------------------------------------ */
void ex4d_oracle_binning(
    int P, int W, int H, int64_t R,
    const int32_t *radii, const float *means2D, const float *depths, const uint32_t *point_offsets,
    uint64_t *keys_unsorted, uint32_t *values_unsorted, uint64_t *keys_sorted, uint32_t *point_list,
    uint32_t *ranges /*[T,2]*/)
{
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    for (int idx = 0; idx < P; idx++) {
        if (radii[idx] > 0) {
            uint32_t off = (idx == 0) ? 0 : point_offsets[idx - 1];
            int rmin[2], rmax[2];
            getRect(means2D[2 * idx], means2D[2 * idx + 1], radii[idx], gx, gy, rmin, rmax);
            for (int y = rmin[1]; y < rmax[1]; y++)
                for (int x = rmin[0]; x < rmax[0]; x++) {
                    uint64_t key = (uint64_t)(y * gx + x);
                    key <<= 32;
                    uint32_t dbits;
                    memcpy(&dbits, &depths[idx], 4);
                    key |= dbits;
                    keys_unsorted[off] = key;
                    values_unsorted[off] = (uint32_t)idx;
                    off++;
                }
        }
    }
    /* stable LSD radix sort, 8-bit digits, over end_bit = 32 + getHigherMsb(T) low bits */
    const int end_bit = 32 + (int)ex4d_oracle_getHigherMsb((uint32_t)(gx * gy));
    const int passes = (end_bit + 7) / 8;
    uint64_t *ka = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(R > 0 ? R : 1));
    uint32_t *va = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(R > 0 ? R : 1));
    uint64_t *kb = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(R > 0 ? R : 1));
    uint32_t *vb = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(R > 0 ? R : 1));
    memcpy(ka, keys_unsorted, sizeof(uint64_t) * (size_t)R);
    memcpy(va, values_unsorted, sizeof(uint32_t) * (size_t)R);
    for (int p = 0; p < passes; p++) {
        const int shift = 8 * p;
        int nb = end_bit - shift; if (nb > 8) nb = 8;
        const uint64_t mask = ((uint64_t)1 << nb) - 1;
        size_t hist[257];
        memset(hist, 0, sizeof(hist));
        for (int64_t i = 0; i < R; i++) hist[((ka[i] >> shift) & mask) + 1]++;
        for (int d = 0; d < 256; d++) hist[d + 1] += hist[d];
        for (int64_t i = 0; i < R; i++) {
            size_t d = (size_t)((ka[i] >> shift) & mask);
            kb[hist[d]] = ka[i]; vb[hist[d]] = va[i]; hist[d]++;
        }
        uint64_t *tk = ka; ka = kb; kb = tk;
        uint32_t *tv = va; va = vb; vb = tv;
    }
    memcpy(keys_sorted, ka, sizeof(uint64_t) * (size_t)R);
    memcpy(point_list, va, sizeof(uint32_t) * (size_t)R);
    free(ka); free(va); free(kb); free(vb);

    memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)(gx * gy));
    for (int64_t idx = 0; idx < R; idx++) {
        uint32_t currtile = (uint32_t)(keys_sorted[idx] >> 32);
        if (idx == 0) ranges[2 * currtile] = 0;
        else {
            uint32_t prevtile = (uint32_t)(keys_sorted[idx - 1] >> 32);
            if (currtile != prevtile) {
                ranges[2 * prevtile + 1] = (uint32_t)idx;
                ranges[2 * currtile] = (uint32_t)idx;
            }
        }
        if (idx == R - 1) ranges[2 * currtile + 1] = (uint32_t)R;
    }
}

/* ------------------------------------------------------------------------------------------
 * Forward, stage 3: renderCUDA CR/forward.cu:274-462, one pixel at a time.
 * The block-cooperative fetch only changes WHEN data is loaded, not what a pixel computes.
 * fragile (optional, test instrumentation, not in the reference): per pixel, the smallest relative
 * distance of any evaluated (alpha vs 1/255) / (test_T vs 1e-4) / (power vs 0) decision to its
 * threshold -- lets the parity test separate 1-ulp exp() threshold flips from real errors.
 * ------------------------------------------------------------------------------------------ */
void ex4d_oracle_render_fwd(
    int W, int H,
    const uint32_t *ranges, const uint32_t *point_list,
    const float *subpixel_offset /*[H,W,2]*/, const float *means2D, const float *features /*[P,3]*/,
    const float *conic_opacity, const float *depths, const float *dir3D /*[P,3]*/,
    const float *bg_color, float min_depth, float max_depth,
    float *final_T, uint32_t *n_contrib,
    float *out_color /*[3,H,W]*/, float *out_depth, float *out_acc, float *out_flow /*[3,H,W]*/, int32_t *out_idx,
    float *fragile /* may be NULL */, float *idx_margin /* may be NULL */,
    float frag_eps, float *flip_w /* may be NULL */)
{
    /* flip_w (optional, test instrumentation): per pixel, the summed blending weight of every decision that lies within
     * frag_eps of its threshold -- what a flipped decision can move: a contributor that appears / disappears carries the
     * weight alpha*T it would have had (alpha ~ 1/255 at the alpha threshold, <= min(0.99, opacity) at power ~ 0), a flipped
     * termination test (CR/forward.cu:383-387) frees or drops everything behind it, which carries at most the transmittance T
     * at that point.  |delta colour| <= flip_w * 2 max|c| (the flipped pair itself plus the rescaling of everything behind it). */
    /* idx_margin (optional, test instrumentation): per pixel, the relative gap (w1 - w2) / w1 between the largest blending weight
     * alpha*T and the runner-up.  The dominant index is the FIRST entry attaining the maximum (strict `>` against the running
     * maximum, CR/forward.cu:411-415), so it can only differ between two float evaluations of the weights (expf vs v_exp_f32,
     * T (1 - alpha) vs T - alpha T) where that gap is within their rounding -- a band of ~1e-6, not the 1e-4 of the decisions. */
    (void)min_depth;
    const int gx = (W + BLOCK_X - 1) / BLOCK_X;
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            const uint32_t pix_id = (uint32_t)(W * py + px);
            float pixf_x = (float)px, pixf_y = (float)py;
            pixf_x += subpixel_offset[2 * pix_id];
            pixf_y += subpixel_offset[2 * pix_id + 1];
            const uint32_t tile = (uint32_t)((py / BLOCK_Y) * gx + (px / BLOCK_X));
            const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];

            float T = 1.0f;
            uint32_t contributor = 0, last_contributor = 0;
            float C[3] = { 0, 0, 0 };
            float Dm = 0.0f, acc = 0.0f, max_vis = 0.0f;
            float F[3] = { 0, 0, 0 };
            float frag = 1.0f, flipw = 0.0f, top1 = 0.0f, top2 = 0.0f;
            int done = 0;
            /* toDo is a signed int in the reference; r1 >= r0 always */
            for (uint32_t k = r0; k < r1 && !done; k++) {
                contributor++;
                const uint32_t id = point_list[k];
                const float dx = means2D[2 * id] - pixf_x, dy = means2D[2 * id + 1] - pixf_y;
                const float *con_o = conic_opacity + 4 * (size_t)id;
                const float power = -0.5f * (con_o[0] * dx * dx + con_o[2] * dy * dy) - con_o[1] * dx * dy;
                if (fragile) {
                    float mag = 0.5f * (fabsf(con_o[0] * dx * dx) + fabsf(con_o[2] * dy * dy)) + fabsf(con_o[1] * dx * dy);
                    if (mag > 0.0f && con_o[3] >= 1.0f / 255.0f) {
                        float m = fabsf(power) / mag; if (m < frag) frag = m;
                        if (flip_w && m < frag_eps) flipw += fminf(0.99f, con_o[3]) * T;
                    }
                }
                if (power > 0.0f) continue;
                const float ex = expf(power);
                float alpha = fminf(0.99f, con_o[3] * ex);
                if (fragile) {
                    float m = fabsf(con_o[3] * ex - 1.0f / 255.0f) * 255.0f; if (m < frag) frag = m;
                    if (flip_w && m < frag_eps) flipw += fmaxf(alpha, 1.0f / 255.0f) * T;
                }
                if (alpha < 1.0f / 255.0f) continue;
                float test_T = T * (1 - alpha);
                if (fragile) {
                    float m = fabsf(test_T - 0.0001f) / 0.0001f; if (m < frag) frag = m;
                    if (flip_w && m < frag_eps) flipw += T;
                }
                if (test_T < 0.0001f) { done = 1; continue; }
                for (int ch = 0; ch < 3; ch++) C[ch] += features[id * 3 + ch] * alpha * T;
                float dep = depths[id];
                Dm += dep * alpha * T;
                acc += alpha * T;
                for (int ch = 0; ch < 3; ch++) F[ch] += dir3D[id * 3 + ch] * alpha * T;
                if (idx_margin) {
                    const float wv = alpha * T;
                    if (wv > top1) { top2 = top1; top1 = wv; }
                    else if (wv > top2) top2 = wv;
                }
                if (alpha * T > max_vis) { max_vis = alpha * T; out_idx[pix_id] = (int32_t)id; }
                T = test_T;
                last_contributor = contributor;
            }
            /* :426-446 */
            if (acc == 0.0f) Dm = Dm + (1.0f - acc) * max_depth;
            else Dm /= acc;
            if (acc != 0.0f) for (int ch = 0; ch < 3; ch++) F[ch] /= acc;
            final_T[pix_id] = T;
            n_contrib[pix_id] = last_contributor;
            for (int ch = 0; ch < 3; ch++) out_color[(size_t)ch * H * W + pix_id] = C[ch] + T * bg_color[ch];
            out_depth[pix_id] = Dm;
            out_acc[pix_id] = acc;
            for (int ch = 0; ch < 3; ch++) out_flow[(size_t)ch * H * W + pix_id] = F[ch];
            if (fragile) fragile[pix_id] = frag;
            if (idx_margin) idx_margin[pix_id] = top1 > 0.0f ? (top1 - top2) / top1 : 1.0f;
            if (flip_w) flip_w[pix_id] = flipw;
        }
}

/* ------------------------------------------------------------------------------------------
 * Backward, stage 1: renderCUDA CR/backward.cu:426-682.  The reference accumulates with float
 * atomicAdd in a non-deterministic order; here: float accumulation in pixel order (row-major) into
 * the outputs, plus optional double-precision sums and sums of |terms| per accumulator
 * (sum13 / abs13, [P,13], test instrumentation for tolerance scaling; may be NULL).
 * accumulator index: 0..2 dL_dmean2D.xyz, 3..5 dL_dconic.(x,y,w), 6 dL_dopacity, 7..9 dL_dcolor, 10..12 dL_ddir
 * ------------------------------------------------------------------------------------------ */
void ex4d_oracle_render_bwd(
    int W, int H,
    const uint32_t *ranges, const uint32_t *point_list,
    const float *subpixel_offset, const float *bg_color,
    const float *means2D, const float *conic_opacity, const float *colors, const float *depths,
    const float *depth_acc /* out_depth */, const float *d_weight_acc /* out_acc */,
    float min_depth, float max_depth,
    const float *final_Ts, const uint32_t *n_contrib,
    const float *dL_dpixels /*[3,H,W]*/, const float *dL_ddepths, const float *dL_dflows /*[3,H,W]*/, const float *dL_daccs,
    float *dL_dmean2D /*[P,3]*/, float *dL_dconic2D /*[P,4]*/, float *dL_ddir /*[P,3]*/,
    float *dL_dopacity /*[P]*/, float *dL_dcolors /*[P,3]*/,
    double *sum13, double *abs13,
    const uint32_t *pixel_order /* may be NULL: row-major */,
    const float *dstate /* [3,H,W] |delta out_depth|, |delta out_acc|, |delta final_T| per pixel; may be NULL */,
    double *state13 /* [P,13]; may be NULL */,
    double *cmag13 /* [P,13]; may be NULL */)
{
    /* cmag13 (optional, test instrumentation): like abs13, but for the six accumulators that dL_dalpha feeds the magnitude of a term is
     * taken BEFORE the cancellation inside dL_dalpha = sum_ch (c - accum_rec) dL_dchannel T + (final_depth - depth) dL_ddepth T T +
     * bg term (CR/backward.cu:603-662): |coefficient| x (sum_ch (|c| + |accum_rec|) |dL_dchannel| T + |depth part| + |bg part|).  A
     * Gaussian whose colour nearly equals what lies behind it has a small dL_dalpha made of O(1) parts; two float evaluations of the
     * reference's own formula then differ by ulps of the PARTS, which the coefficient (up to 0.5 W x conic x dx ~ 1e3 for a sharp
     * Gaussian) turns into 1e-4 of the gradient.  The hand-built accumulator bar of the tests scales with max(abs13, cmag13). */
    /* dstate / state13 (optional, test instrumentation): the backward consumes the forward's per-pixel state (out_depth, out_acc,
     * final_T).  Two forward evaluations differ in that state by rounding, and the backward AMPLIFIES the difference where a term is a
     * small difference of large ones -- (final_depth - depth) dL_ddepth / acc at small acc, CR/backward.cu:535-541, :603-613.
     * Given the per-pixel absolute differences of the two states, state13 accumulates the first-order bound
     * sum_pixels |d term / d state| |delta state| per accumulator: what an END-TO-END comparison (own forward -> own backward on both
     * sides) may differ by on top of the shared-state bound. */
    /* pixel_order (optional, test instrumentation): a permutation of the pixel ids.  The reference's threads (one per pixel) add
     * their terms with float atomicAdd (CR/backward.cu:613-679): the order in which the terms of different pixels reach one
     * accumulator is arbitrary and differs from run to run.  Replaying the pixel loop in several random orders measures the
     * reference's own float32 run-to-run spread (the noise floor the parity tolerances are anchored to). */
    (void)max_depth;
    const int gx = (W + BLOCK_X - 1) / BLOCK_X;
    const float ddelx_dx = (float)(0.5 * W);
    const float ddely_dy = (float)(0.5 * H);
    /* abs13 accumulates |term| x cond, cond = 1 + sum|terms of power| (conditioning of exp(power) for elongated
     * conics; for the mean gradient also the cancellation inside -(A dx + B dy)): the magnitude against which any two
     * float evaluations of the same formula can be expected to agree */
#define ACCM(ptr, k, val, mag) do { float v_ = (val); *(ptr) += v_; \
        if (sum13) { sum13[13 * (size_t)global_id + (k)] += (double)v_; abs13[13 * (size_t)global_id + (k)] += (double)(mag) * cond; } } while (0)
#define ACC(ptr, k, val) ACCM(ptr, k, val, fabs((double)v_))
    for (size_t pn = 0; pn < (size_t)W * (size_t)H; pn++) {
            const uint32_t pix_id = pixel_order ? pixel_order[pn] : (uint32_t)pn;
            const int py = (int)(pix_id / (uint32_t)W), px = (int)(pix_id % (uint32_t)W);
            float pixf_x = (float)px, pixf_y = (float)py;
            pixf_x += subpixel_offset[2 * pix_id];
            pixf_y += subpixel_offset[2 * pix_id + 1];
            const uint32_t tile = (uint32_t)((py / BLOCK_Y) * gx + (px / BLOCK_X));
            const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];

            const float T_final = final_Ts[pix_id];
            float T = T_final;
            uint32_t contributor = r1 - r0;
            const uint32_t last_contributor = n_contrib[pix_id];
            const float final_acc = d_weight_acc[pix_id];
            const float final_depth = depth_acc[pix_id];
            float acc = final_acc;

            float dL_ddepth = dL_ddepths[pix_id];
            if (acc > 0.0f) dL_ddepth /= acc;
            float dL_dflow[3] = { 0, 0, 0 };
            if (acc > 0.0f) {
                dL_dflow[0] = dL_dflows[pix_id] / acc;
                dL_dflow[1] = dL_dflows[(size_t)H * W + pix_id] / acc;
                dL_dflow[2] = dL_dflows[(size_t)2 * H * W + pix_id] / acc;
            }
            float dL_dacc = 0;
            if (acc > 0.0f) dL_dacc = dL_daccs[pix_id];
            /* relative state differences of this pixel (first-order sensitivity bookkeeping, see above) */
            double e_fd = 0.0, e_A = 0.0, e_T = 0.0, n_gacc = 0.0;
            if (dstate && state13) {
                e_fd = (double)dstate[pix_id];                                             /* absolute: enters as |delta final_depth| */
                e_A = acc > 0.0f ? (double)dstate[(size_t)H * W + pix_id] / (double)acc : 0.0;
                e_T = T_final > 0.0f ? (double)dstate[(size_t)2 * H * W + pix_id] / (double)T_final : 0.0;
            }
#define STATE(k, mag) do { if (dstate && state13) state13[13 * (size_t)global_id + (k)] += (mag); } while (0)

            float accum_rec[3] = { 0, 0, 0 };
            float dL_dpixel[3];
            for (int i = 0; i < 3; i++) dL_dpixel[i] = dL_dpixels[(size_t)i * H * W + pix_id];
            float last_alpha = 0;
            float last_color[3] = { 0, 0, 0 };

            for (uint32_t kk = r1; kk > r0; kk--) {
                const uint32_t global_id = point_list[kk - 1];
                contributor--;
                if (contributor >= last_contributor) continue;

                const float dx = means2D[2 * global_id] - pixf_x, dy = means2D[2 * global_id + 1] - pixf_y;
                const float *con_o = conic_opacity + 4 * (size_t)global_id;
                const float power = -0.5f * (con_o[0] * dx * dx + con_o[2] * dy * dy) - con_o[1] * dx * dy;
                if (power > 0.0f) continue;
                const float G = expf(power);
                const float alpha = fminf(0.99f, con_o[3] * G);
                if (alpha < 1.0f / 255.0f) continue;
                const double cond = 1.0 + 0.5 * (fabs((double)con_o[0] * dx * dx) + fabs((double)con_o[2] * dy * dy)) + fabs((double)con_o[1] * dx * dy);

                T = T / (1.f - alpha);
                const float dchannel_dcolor = alpha * T;
                float dL_dalpha = 0.0f;

                const float dep = depths[global_id];
                double a_mag = 0.0;          /* magnitude of the parts of dL_dalpha (before the factor T below) */
                double s_alpha = 0.0;        /* bound on |delta dL_dalpha| (before the factor T below) */
                if ((dep > min_depth) & (alpha * T > 0.0f)) {
                    ACC(&dL_dmean2D[3 * (size_t)global_id + 2], 2, dL_ddepth * dchannel_dcolor);
                    dL_dalpha += (final_depth - dep) * dL_ddepth * T;
                    a_mag += fabs(((double)final_depth - dep) * dL_ddepth * T);
                    STATE(2, fabs((double)dL_ddepth * dchannel_dcolor) * (e_A + e_T));
                    s_alpha += (e_fd + fabs((double)final_depth - dep) * (e_A + e_T)) * fabs((double)dL_ddepth) * T;
                }
                for (int ch = 0; ch < 3; ch++) {
                    const float c = colors[global_id * 3 + ch];
                    accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                    last_color[ch] = c;
                    const float dL_dchannel = dL_dpixel[ch];
                    dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
                    a_mag += (fabs((double)c) + fabs((double)accum_rec[ch])) * fabs((double)dL_dchannel);
                    ACC(&dL_dcolors[global_id * 3 + ch], 7 + ch, dchannel_dcolor * dL_dchannel);
                    STATE(7 + ch, fabs((double)dchannel_dcolor * dL_dchannel) * e_T);
                }
                ACC(&dL_ddir[global_id * 3 + 0], 10, dchannel_dcolor * dL_dflow[0]);
                ACC(&dL_ddir[global_id * 3 + 1], 11, dchannel_dcolor * dL_dflow[1]);
                ACC(&dL_ddir[global_id * 3 + 2], 12, dchannel_dcolor * dL_dflow[2]);
                for (int ch = 0; ch < 3; ch++) STATE(10 + ch, fabs((double)dchannel_dcolor * dL_dflow[ch]) * (e_A + e_T));

                /* every part of dL_dalpha carries one more factor T (relative difference e_T) from here on */
                s_alpha = (s_alpha + fabs((double)dL_dalpha) * e_T) * T;
                dL_dalpha *= T;
                dL_dacc *= T;
                n_gacc += 1.0;               /* dL_dacc = upstream x the product of n_gacc transmittances, each proportional to final_T */
                last_alpha = alpha;

                float bg_dot_dpixel = 0;
                for (int i = 0; i < 3; i++) bg_dot_dpixel += bg_color[i] * dL_dpixel[i];
                dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
                s_alpha += fabs((double)(-T_final / (1.f - alpha)) * bg_dot_dpixel) * e_T;
                {
                    double bg_abs = 0.0;
                    for (int i = 0; i < 3; i++) bg_abs += fabs((double)bg_color[i] * dL_dpixel[i]);
                    a_mag = a_mag * T + fabs((double)T_final / (1.0 - alpha)) * bg_abs;
                }

                const float dL_dG = con_o[3] * dL_dalpha;
                const float gdx = G * dx;
                const float gdy = G * dy;
                const float dG_ddelx = -gdx * con_o[0] - gdy * con_o[1];
                const float dG_ddely = -gdy * con_o[2] - gdx * con_o[1];

                ACCM(&dL_dmean2D[3 * (size_t)global_id + 0], 0, dL_dG * dG_ddelx * ddelx_dx,
                     fabs((double)dL_dG) * (fabs((double)gdx * con_o[0]) + fabs((double)gdy * con_o[1])) * ddelx_dx);
                ACCM(&dL_dmean2D[3 * (size_t)global_id + 1], 1, dL_dG * dG_ddely * ddely_dy,
                     fabs((double)dL_dG) * (fabs((double)gdy * con_o[2]) + fabs((double)gdx * con_o[1])) * ddely_dy);
                ACC(&dL_dconic2D[4 * (size_t)global_id + 0], 3, -0.5f * gdx * dx * dL_dG);
                ACC(&dL_dconic2D[4 * (size_t)global_id + 1], 4, -0.5f * gdx * dy * dL_dG);
                ACC(&dL_dconic2D[4 * (size_t)global_id + 3], 5, -0.5f * gdy * dy * dL_dG);
                ACC(&dL_dopacity[global_id], 6, G * dL_dalpha);
                ACC(&dL_dopacity[global_id], 6, G * dL_dacc);
                if (cmag13) {
                    const double m_G = fabs((double)con_o[3]) * a_mag * cond;     /* |dL_dG| with the parts of dL_dalpha un-cancelled */
                    double *cm = cmag13 + 13 * (size_t)global_id;
                    cm[0] += m_G * (fabs((double)gdx * con_o[0]) + fabs((double)gdy * con_o[1])) * ddelx_dx;
                    cm[1] += m_G * (fabs((double)gdy * con_o[2]) + fabs((double)gdx * con_o[1])) * ddely_dy;
                    cm[3] += 0.5 * fabs((double)gdx * dx) * m_G;
                    cm[4] += 0.5 * fabs((double)gdx * dy) * m_G;
                    cm[5] += 0.5 * fabs((double)gdy * dy) * m_G;
                    cm[6] += (double)G * a_mag * cond + fabs((double)G * dL_dacc) * cond;
                }
                if (dstate && state13) {
                    const double s_G = fabs((double)con_o[3]) * s_alpha;          /* bound on |delta dL_dG| */
                    STATE(0, s_G * (fabs((double)gdx * con_o[0]) + fabs((double)gdy * con_o[1])) * ddelx_dx);
                    STATE(1, s_G * (fabs((double)gdy * con_o[2]) + fabs((double)gdx * con_o[1])) * ddely_dy);
                    STATE(3, 0.5 * fabs((double)gdx * dx) * s_G);
                    STATE(4, 0.5 * fabs((double)gdx * dy) * s_G);
                    STATE(5, 0.5 * fabs((double)gdy * dy) * s_G);
                    STATE(6, (double)G * s_alpha + fabs((double)G * dL_dacc) * n_gacc * e_T);
                }
            }
#undef STATE
        }
#undef ACC
#undef ACCM
}

/* CR/backward.cu:20-139 (SH backward): writes dL_dsh[idx,:,:], adds the direction path into dL_dmeans[idx] */
static void computeColorFromSH_bwd(int idx, int deg, int max_coeffs, const float *means, const float *campos,
                                   const float *shs, const uint8_t *clamped, const float *dL_dcolor,
                                   float *dL_dmeans, float *dL_dshs)
{
    v3 dir_orig = { means[3 * idx] - campos[0], means[3 * idx + 1] - campos[1], means[3 * idx + 2] - campos[2] };
    float len = sqrtf(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
    float x = dir_orig.x / len, y = dir_orig.y / len, z = dir_orig.z / len;
    const float *sh = shs + (size_t)idx * max_coeffs * 3;
    float *dL_dsh = dL_dshs + (size_t)idx * max_coeffs * 3;

    float dL_dRGB[3];
    for (int ch = 0; ch < 3; ch++) dL_dRGB[ch] = dL_dcolor[3 * idx + ch] * (clamped[3 * idx + ch] ? 0 : 1);

    float dRGBdx[3] = { 0, 0, 0 }, dRGBdy[3] = { 0, 0, 0 }, dRGBdz[3] = { 0, 0, 0 };
#define SH(k) sh[3 * (k) + ch]
#define DSH(k, coef) do { float c_ = (coef); for (int ch = 0; ch < 3; ch++) dL_dsh[3 * (k) + ch] = c_ * dL_dRGB[ch]; } while (0)
    DSH(0, SH_C0);
    if (deg > 0) {
        DSH(1, -SH_C1 * y);
        DSH(2, SH_C1 * z);
        DSH(3, -SH_C1 * x);
        for (int ch = 0; ch < 3; ch++) {
            dRGBdx[ch] = -SH_C1 * SH(3);
            dRGBdy[ch] = -SH_C1 * SH(1);
            dRGBdz[ch] = SH_C1 * SH(2);
        }
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z;
            float xy = x * y, yz = y * z, xz = x * z;
            DSH(4, SH_C2[0] * xy);
            DSH(5, SH_C2[1] * yz);
            DSH(6, SH_C2[2] * (2.f * zz - xx - yy));
            DSH(7, SH_C2[3] * xz);
            DSH(8, SH_C2[4] * (xx - yy));
            for (int ch = 0; ch < 3; ch++) {
                dRGBdx[ch] += SH_C2[0] * y * SH(4) + SH_C2[2] * 2.f * -x * SH(6) + SH_C2[3] * z * SH(7) + SH_C2[4] * 2.f * x * SH(8);
                dRGBdy[ch] += SH_C2[0] * x * SH(4) + SH_C2[1] * z * SH(5) + SH_C2[2] * 2.f * -y * SH(6) + SH_C2[4] * 2.f * -y * SH(8);
                dRGBdz[ch] += SH_C2[1] * y * SH(5) + SH_C2[2] * 2.f * 2.f * z * SH(6) + SH_C2[3] * x * SH(7);
            }
            if (deg > 2) {
                DSH(9, SH_C3[0] * y * (3.f * xx - yy));
                DSH(10, SH_C3[1] * xy * z);
                DSH(11, SH_C3[2] * y * (4.f * zz - xx - yy));
                DSH(12, SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
                DSH(13, SH_C3[4] * x * (4.f * zz - xx - yy));
                DSH(14, SH_C3[5] * z * (xx - yy));
                DSH(15, SH_C3[6] * x * (xx - 3.f * yy));
                for (int ch = 0; ch < 3; ch++) {
                    dRGBdx[ch] += (
                        SH_C3[0] * SH(9) * 3.f * 2.f * xy +
                        SH_C3[1] * SH(10) * yz +
                        SH_C3[2] * SH(11) * -2.f * xy +
                        SH_C3[3] * SH(12) * -3.f * 2.f * xz +
                        SH_C3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) +
                        SH_C3[5] * SH(14) * 2.f * xz +
                        SH_C3[6] * SH(15) * 3.f * (xx - yy));
                    dRGBdy[ch] += (
                        SH_C3[0] * SH(9) * 3.f * (xx - yy) +
                        SH_C3[1] * SH(10) * xz +
                        SH_C3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) +
                        SH_C3[3] * SH(12) * -3.f * 2.f * yz +
                        SH_C3[4] * SH(13) * -2.f * xy +
                        SH_C3[5] * SH(14) * -2.f * yz +
                        SH_C3[6] * SH(15) * -3.f * 2.f * xy);
                    dRGBdz[ch] += (
                        SH_C3[1] * SH(10) * xy +
                        SH_C3[2] * SH(11) * 4.f * 2.f * yz +
                        SH_C3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) +
                        SH_C3[4] * SH(13) * 4.f * 2.f * xz +
                        SH_C3[5] * SH(14) * (xx - yy));
                }
            }
        }
    }
#undef SH
#undef DSH
    /* glm::dot(a,b) = a.x*b.x + a.y*b.y + a.z*b.z */
    v3 dL_ddir = {
        dRGBdx[0] * dL_dRGB[0] + dRGBdx[1] * dL_dRGB[1] + dRGBdx[2] * dL_dRGB[2],
        dRGBdy[0] * dL_dRGB[0] + dRGBdy[1] * dL_dRGB[1] + dRGBdy[2] * dL_dRGB[2],
        dRGBdz[0] * dL_dRGB[0] + dRGBdz[1] * dL_dRGB[1] + dRGBdz[2] * dL_dRGB[2] };
    v3 dL_dmean = dnormvdv(dir_orig, dL_ddir);
    dL_dmeans[3 * idx + 0] += dL_dmean.x;
    dL_dmeans[3 * idx + 1] += dL_dmean.y;
    dL_dmeans[3 * idx + 2] += dL_dmean.z;
}

/* CR/backward.cu:304-367 */
static void computeCov3D_bwd(int idx, const float *scale, float mod, const float *rot,
                             const float *dL_dcov3Ds, float *dL_dscales, float *dL_drots)
{
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    m3 R = m3_cols(
        1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
        2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
        2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    m3 S = m3_cols(1.0f, 0, 0, 0, 1.0f, 0, 0, 0, 1.0f);
    float s[3] = { mod * scale[0], mod * scale[1], mod * scale[2] };
    S.c[0][0] = s[0]; S.c[1][1] = s[1]; S.c[2][2] = s[2];
    m3 M = m3_mul(&S, &R);
    const float *d = dL_dcov3Ds + 6 * (size_t)idx;
    m3 dL_dSigma = m3_cols(
        d[0], 0.5f * d[1], 0.5f * d[2],
        0.5f * d[1], d[3], 0.5f * d[4],
        0.5f * d[2], 0.5f * d[4], d[5]);
    /* 2.0f * M * dL_dSigma = (2.0f * M) * dL_dSigma */
    m3 M2;
    for (int c = 0; c < 3; c++) for (int rr = 0; rr < 3; rr++) M2.c[c][rr] = M.c[c][rr] * 2.0f;
    m3 dL_dM = m3_mul(&M2, &dL_dSigma);
    m3 Rt = m3_transpose(&R);
    m3 dL_dMt = m3_transpose(&dL_dM);
    for (int k = 0; k < 3; k++)
        dL_dscales[3 * idx + k] = Rt.c[k][0] * dL_dMt.c[k][0] + Rt.c[k][1] * dL_dMt.c[k][1] + Rt.c[k][2] * dL_dMt.c[k][2];
    for (int k = 0; k < 3; k++)
        for (int rr = 0; rr < 3; rr++)
            dL_dMt.c[k][rr] *= s[k];
#define D(a, b) dL_dMt.c[a][b]
    float qx = 2 * z * (D(0,1) - D(1,0)) + 2 * y * (D(2,0) - D(0,2)) + 2 * x * (D(1,2) - D(2,1));
    float qy = 2 * y * (D(1,0) + D(0,1)) + 2 * z * (D(2,0) + D(0,2)) + 2 * r * (D(1,2) - D(2,1)) - 4 * x * (D(2,2) + D(1,1));
    float qz = 2 * x * (D(1,0) + D(0,1)) + 2 * r * (D(2,0) - D(0,2)) + 2 * z * (D(1,2) + D(2,1)) - 4 * y * (D(2,2) + D(0,0));
    float qw = 2 * r * (D(0,1) - D(1,0)) + 2 * x * (D(2,0) + D(0,2)) + 2 * y * (D(1,2) + D(2,1)) - 4 * z * (D(1,1) + D(0,0));
#undef D
    dL_drots[4 * idx + 0] = qx; dL_drots[4 * idx + 1] = qy; dL_drots[4 * idx + 2] = qz; dL_drots[4 * idx + 3] = qw;
}

/* ------------------------------------------------------------------------------------------
 * Backward, stage 2: computeCov2DCUDA (CR/backward.cu:144-300) then preprocessCUDA (:372-423),
 * in that order over all Gaussians (two kernel launches in the reference, :715 then :735).
 * All outputs are expected zero-initialised by the caller (DGR/rasterize_points.cu:178-187).
 * ------------------------------------------------------------------------------------------ */
void ex4d_oracle_preprocess_bwd(
    int P, int D, int M,
    const float *means3D, const int32_t *radii, const float *shs /* may be NULL */, const uint8_t *clamped,
    const float *scales /* may be NULL */, const float *rotations, float scale_modifier,
    const float *cov3Ds, const float *viewmatrix, const float *projmatrix,
    int W, int H, float tan_fovx, float tan_fovy, float kernel_size, const float *campos,
    const float *dL_dmean2D /*[P,3]*/, const float *dL_dconics /*[P,4]*/,
    float *dL_dmeans /*[P,3]*/, float *dL_dcolor /*[P,3]*/, float *dL_dcov /*[P,6]*/, float *dL_dsh,
    float *dL_dscale, float *dL_drot)
{
    const float h_y = H / (2.0f * tan_fovy);   /* CR/rasterizer_impl.cu:417-418 */
    const float h_x = W / (2.0f * tan_fovx);
    const float *vm = viewmatrix;
    /* kernel 1: computeCov2DCUDA */
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        const float *cov3D = cov3Ds + 6 * (size_t)idx;
        v3 mean = { means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2] };
        float gA = dL_dconics[4 * idx], gB = dL_dconics[4 * idx + 1], gC = dL_dconics[4 * idx + 3];
        v3 t = transformPoint4x3(mean, vm);
        const float limx = 1.3f * tan_fovx;
        const float limy = 1.3f * tan_fovy;
        const float txtz = t.x / t.z;
        const float tytz = t.y / t.z;
        t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
        t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
        const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0 : 1;
        const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0 : 1;

        m3 J = m3_cols(h_x / t.z, 0.0f, -(h_x * t.x) / (t.z * t.z),
                       0.0f, h_y / t.z, -(h_y * t.y) / (t.z * t.z),
                       0, 0, 0);
        m3 Wm = m3_cols(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
        m3 Vrk = m3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
        m3 T = m3_mul(&Wm, &J);
        m3 Tt = m3_transpose(&T);
        m3 Vt = m3_transpose(&Vrk);
        m3 TV = m3_mul(&Tt, &Vt);
        m3 cov2D = m3_mul(&TV, &T);
        /* :201-218: the coef gradient block has no effect on any output -- omitted */
        float a = cov2D.c[0][0] += kernel_size;
        float b = cov2D.c[0][1];
        float c = cov2D.c[1][1] += kernel_size;
        float denom = a * c - b * b;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float *dc = dL_dcov + 6 * (size_t)idx;
#define Tm(i, j) T.c[i][j]
        if (denom2inv != 0) {
            dL_da = denom2inv * (-c * c * gA + 2 * b * c * gB + (denom - a * c) * gC);
            dL_dc = denom2inv * (-a * a * gC + 2 * a * b * gB + (denom - a * c) * gA);
            dL_db = denom2inv * 2 * (b * c * gA - (denom + 2 * b * b) * gB + a * b * gC);
            dc[0] = (Tm(0,0) * Tm(0,0) * dL_da + Tm(0,0) * Tm(1,0) * dL_db + Tm(1,0) * Tm(1,0) * dL_dc);
            dc[3] = (Tm(0,1) * Tm(0,1) * dL_da + Tm(0,1) * Tm(1,1) * dL_db + Tm(1,1) * Tm(1,1) * dL_dc);
            dc[5] = (Tm(0,2) * Tm(0,2) * dL_da + Tm(0,2) * Tm(1,2) * dL_db + Tm(1,2) * Tm(1,2) * dL_dc);
            dc[1] = 2 * Tm(0,0) * Tm(0,1) * dL_da + (Tm(0,0) * Tm(1,1) + Tm(0,1) * Tm(1,0)) * dL_db + 2 * Tm(1,0) * Tm(1,1) * dL_dc;
            dc[2] = 2 * Tm(0,0) * Tm(0,2) * dL_da + (Tm(0,0) * Tm(1,2) + Tm(0,2) * Tm(1,0)) * dL_db + 2 * Tm(1,0) * Tm(1,2) * dL_dc;
            dc[4] = 2 * Tm(0,2) * Tm(0,1) * dL_da + (Tm(0,1) * Tm(1,2) + Tm(0,2) * Tm(1,1)) * dL_db + 2 * Tm(1,1) * Tm(1,2) * dL_dc;
        } else {
            for (int i = 0; i < 6; i++) dc[i] = 0;
        }
#define V(i, j) Vrk.c[i][j]
        float dL_dT00 = 2 * (Tm(0,0) * V(0,0) + Tm(0,1) * V(0,1) + Tm(0,2) * V(0,2)) * dL_da +
                        (Tm(1,0) * V(0,0) + Tm(1,1) * V(0,1) + Tm(1,2) * V(0,2)) * dL_db;
        float dL_dT01 = 2 * (Tm(0,0) * V(1,0) + Tm(0,1) * V(1,1) + Tm(0,2) * V(1,2)) * dL_da +
                        (Tm(1,0) * V(1,0) + Tm(1,1) * V(1,1) + Tm(1,2) * V(1,2)) * dL_db;
        float dL_dT02 = 2 * (Tm(0,0) * V(2,0) + Tm(0,1) * V(2,1) + Tm(0,2) * V(2,2)) * dL_da +
                        (Tm(1,0) * V(2,0) + Tm(1,1) * V(2,1) + Tm(1,2) * V(2,2)) * dL_db;
        float dL_dT10 = 2 * (Tm(1,0) * V(0,0) + Tm(1,1) * V(0,1) + Tm(1,2) * V(0,2)) * dL_dc +
                        (Tm(0,0) * V(0,0) + Tm(0,1) * V(0,1) + Tm(0,2) * V(0,2)) * dL_db;
        float dL_dT11 = 2 * (Tm(1,0) * V(1,0) + Tm(1,1) * V(1,1) + Tm(1,2) * V(1,2)) * dL_dc +
                        (Tm(0,0) * V(1,0) + Tm(0,1) * V(1,1) + Tm(0,2) * V(1,2)) * dL_db;
        float dL_dT12 = 2 * (Tm(1,0) * V(2,0) + Tm(1,1) * V(2,1) + Tm(1,2) * V(2,2)) * dL_dc +
                        (Tm(0,0) * V(2,0) + Tm(0,1) * V(2,1) + Tm(0,2) * V(2,2)) * dL_db;
#undef V
#undef Tm
#define Wg(i, j) Wm.c[i][j]
        float dL_dJ00 = Wg(0,0) * dL_dT00 + Wg(0,1) * dL_dT01 + Wg(0,2) * dL_dT02;
        float dL_dJ02 = Wg(2,0) * dL_dT00 + Wg(2,1) * dL_dT01 + Wg(2,2) * dL_dT02;
        float dL_dJ11 = Wg(1,0) * dL_dT10 + Wg(1,1) * dL_dT11 + Wg(1,2) * dL_dT12;
        float dL_dJ12 = Wg(2,0) * dL_dT10 + Wg(2,1) * dL_dT11 + Wg(2,2) * dL_dT12;
#undef Wg
        float tz = 1.f / t.z;
        float tz2 = tz * tz;
        float tz3 = tz2 * tz;
        float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
        float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
        float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 + (2 * h_y * t.y) * tz3 * dL_dJ12;
        v3 g = { dL_dtx, dL_dty, dL_dtz };
        v3 dL_dmean = transformVec4x3Transpose(g, vm);
        dL_dmeans[3 * idx + 0] += dL_dmean.x;
        dL_dmeans[3 * idx + 1] += dL_dmean.y;
        dL_dmeans[3 * idx + 2] += dL_dmean.z;
    }
    /* kernel 2: preprocessCUDA (bwd) */
    const float *proj = projmatrix;
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        v3 m = { means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2] };
        float m_hom[4];
        transformPoint4x4(m, proj, m_hom);
        float m_w = 1.0f / (m_hom[3] + 0.0000001f);
        float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
        float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
        float mul3 = (proj[2] * m.x + proj[6] * m.y + proj[10] * m.z + proj[14]) * m_w * m_w;
        const float gx_ = dL_dmean2D[3 * idx], gy_ = dL_dmean2D[3 * idx + 1], gz_ = dL_dmean2D[3 * idx + 2];
        float dmx = (proj[0] * m_w - proj[3] * mul1) * gx_ + (proj[1] * m_w - proj[3] * mul2) * gy_ + (proj[2] * m_w - proj[3] * mul3) * gz_;
        float dmy = (proj[4] * m_w - proj[7] * mul1) * gx_ + (proj[5] * m_w - proj[7] * mul2) * gy_ + (proj[6] * m_w - proj[7] * mul3) * gz_;
        float dmz = (proj[8] * m_w - proj[11] * mul1) * gx_ + (proj[9] * m_w - proj[11] * mul2) * gy_ + (proj[10] * m_w - proj[11] * mul3) * gz_;
        /* :414 -- ASSIGNMENT: the covariance-path term added by kernel 1 is overwritten */
        dL_dmeans[3 * idx + 0] = dmx;
        dL_dmeans[3 * idx + 1] = dmy;
        dL_dmeans[3 * idx + 2] = dmz;
        if (shs)
            computeColorFromSH_bwd(idx, D, M, means3D, campos, shs, clamped, dL_dcolor, dL_dmeans, dL_dsh);
        if (scales)
            computeCov3D_bwd(idx, scales + 3 * (size_t)idx, scale_modifier, rotations + 4 * (size_t)idx, dL_dcov, dL_dscale, dL_drot);
    }
}
