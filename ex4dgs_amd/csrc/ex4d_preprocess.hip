// Per-Gaussian preprocess (forward + backward) and markVisible for gfx950.
//
// Replaces (reference, CR/ = submodules/diff_gaussian_rasterization_df/cuda_rasterizer/):
//   preprocessCUDA fwd  CR/forward.cu:165-269  (+ computeCov2D :74-124, computeCov3D :128-162, SH :20-71)
//   computeCov2DCUDA    CR/backward.cu:144-300 and preprocessCUDA bwd :372-423 (+ SH bwd :20-139, cov3D bwd :304-367)
//   checkFrustum        CR/rasterizer_impl.cu:54-68
//
// This translation unit is compiled with -ffp-contract=off: every value that decides an integer
// (cull, radius, tile rect, tiles_touched, depth key) must be bit-identical to the CPU oracle, so no
// FMA fusion, correctly-rounded / and sqrt (hipcc default), and the reference's double promotions
// (CR/auxiliary.h:43, :284; CR/forward.cu:112-116) are kept.  One thread per Gaussian, 256-thread
// blocks (4 wave64); the kernels are HBM-bound (SH read / SH-grad write of 192 B per Gaussian).
#include "ex4d_internal.h"
#include <atomic>
#include <cstdlib>
#include <cstdlib>

namespace {

__constant__ float kSH_C0 = 0.28209479177387814f;
__constant__ float kSH_C1 = 0.4886025119029199f;
__constant__ float kSH_C2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                 -1.0925484305920792f, 0.5462742152960396f };
__constant__ float kSH_C3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                 -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f };

struct Mat3 { float m[3][3]; };   // m[col][row], GLM storage order

// product with GLM's summation order: out[c][r] = a[0][r]*b[c][0] + a[1][r]*b[c][1] + a[2][r]*b[c][2]
__device__ __forceinline__ Mat3 mul(const Mat3 &a, const Mat3 &b)
{
    Mat3 o;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++)
            o.m[c][r] = a.m[0][r] * b.m[c][0] + a.m[1][r] * b.m[c][1] + a.m[2][r] * b.m[c][2];
    return o;
}
__device__ __forceinline__ Mat3 transpose(const Mat3 &a)
{
    Mat3 o;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++)
            o.m[c][r] = a.m[r][c];
    return o;
}
__device__ __forceinline__ Mat3 from_columns(float a0, float a1, float a2, float b0, float b1, float b2, float c0, float c1, float c2)
{
    Mat3 o;
    o.m[0][0] = a0; o.m[0][1] = a1; o.m[0][2] = a2;
    o.m[1][0] = b0; o.m[1][1] = b1; o.m[1][2] = b2;
    o.m[2][0] = c0; o.m[2][1] = c1; o.m[2][2] = c2;
    return o;
}

// float -> int with truncation, saturation and NaN -> 0 (the conversion the reference's target performs)
__device__ __forceinline__ int to_int_sat(float f)
{
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
}

__device__ __forceinline__ float3 xform4x3(float3 p, const float *m)
{
    return make_float3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
                       m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
__device__ __forceinline__ float4 xform4x4(float3 p, const float *m)
{
    return make_float4(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
                       m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14],
                       m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]);
}

// frustum test of CR/auxiliary.h:267-294; returns visibility, p_view and the NDC xy
__device__ __forceinline__ bool frustum_test(float3 p, const float *vm, const float *pm, float min_depth, float max_depth,
                                             float3 &p_view, float &ndc_x, float &ndc_y)
{
    float4 h = xform4x4(p, pm);
    float inv_w = 1.0f / (h.w + 0.0000001f);
    ndc_x = h.x * inv_w;
    ndc_y = h.y * inv_w;
    p_view = xform4x3(p, vm);
    return !((p_view.z <= min_depth) || (p_view.z > max_depth) ||
             ((double)ndc_x < -1.3 || (double)ndc_x > 1.3 || (double)ndc_y < -1.3 || (double)ndc_y > 1.3));
}

__device__ __forceinline__ void tile_rect(float px, float py, int radius, int gx, int gy, int &x0, int &y0, int &x1, int &y1)
{
    // CR/auxiliary.h:46-56 (float arithmetic, truncation toward zero, clamp to the tile grid)
    x0 = min(gx, max(0, to_int_sat((px - (float)radius) / (float)EX4D_TILE)));
    y0 = min(gy, max(0, to_int_sat((py - (float)radius) / (float)EX4D_TILE)));
    x1 = min(gx, max(0, to_int_sat((px + (float)radius + (float)EX4D_TILE - 1.0f) / (float)EX4D_TILE)));
    y1 = min(gy, max(0, to_int_sat((py + (float)radius + (float)EX4D_TILE - 1.0f) / (float)EX4D_TILE)));
}

__device__ __forceinline__ Mat3 rotation_from_quat(float r, float x, float y, float z)
{
    return from_columns(
        1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
        2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
        2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
}

// Screen-space covariance pieces shared by forward and backward (CR/forward.cu:80-106 == CR/backward.cu:171-199)
struct Cov2DCtx { float3 t; float txtz, tytz, limx, limy; Mat3 Wm, T, Vrk, cov; };
__device__ __forceinline__ void cov2d_common(float3 mean, float fx, float fy, float tanx, float tany, const float *cov3D,
                                             const float *vm, Cov2DCtx &c)
{
    c.t = xform4x3(mean, vm);
    c.limx = 1.3f * tanx;
    c.limy = 1.3f * tany;
    c.txtz = c.t.x / c.t.z;
    c.tytz = c.t.y / c.t.z;
    c.t.x = fminf(c.limx, fmaxf(-c.limx, c.txtz)) * c.t.z;
    c.t.y = fminf(c.limy, fmaxf(-c.limy, c.tytz)) * c.t.z;
    Mat3 J = from_columns(fx / c.t.z, 0.0f, -(fx * c.t.x) / (c.t.z * c.t.z),
                          0.0f, fy / c.t.z, -(fy * c.t.y) / (c.t.z * c.t.z),
                          0.0f, 0.0f, 0.0f);
    c.Wm = from_columns(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
    c.T = mul(c.Wm, J);
    c.Vrk = from_columns(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    Mat3 Tt = transpose(c.T);
    Mat3 Vt = transpose(c.Vrk);
    Mat3 TV = mul(Tt, Vt);
    c.cov = mul(TV, c.T);
}

// ---- wave-cooperative stores of narrow per-Gaussian rows -----------------------------------------------
// A [P,3] / [P,6] gradient tensor written one row per lane makes every store instruction touch 64 partial
// lines (measured: 2.3x HBM write amplification in the backward).  The wave's 64 rows are contiguous in
// memory, so they are transposed through LDS and written as W fully coalesced 256-byte stores.
template <int W>
__device__ __forceinline__ void wave_store_rows(float *__restrict__ dst_wave, const float (&vals)[W], float *lds, int nrows, int lane)
{
#pragma unroll
    for (int i = 0; i < W; i++) lds[lane * W + i] = vals[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < W; i++) {
        const int e = i * 64 + lane;
        if (e < nrows * W) dst_wave[e] = lds[e];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Sigma = (S R)^T (S R), CR/forward.cu:128-162; raw (un-normalised) quaternion.  ONE function for the forward kernel and for the backward
// kernel's recomputation: identical operations in identical order (this file is compiled with -ffp-contract=off), identical bits.
__device__ __forceinline__ void cov3d_from_scale_rotation(float scale_modifier, float s0, float s1, float s2, float4 q, float (&cov3D)[6])
{
    Mat3 S = from_columns(1.0f, 0.f, 0.f, 0.f, 1.0f, 0.f, 0.f, 0.f, 1.0f);
    S.m[0][0] = scale_modifier * s0;
    S.m[1][1] = scale_modifier * s1;
    S.m[2][2] = scale_modifier * s2;
    Mat3 R = rotation_from_quat(q.x, q.y, q.z, q.w);
    Mat3 Mx = mul(S, R);
    Mat3 Sigma = mul(transpose(Mx), Mx);
    cov3D[0] = Sigma.m[0][0]; cov3D[1] = Sigma.m[0][1]; cov3D[2] = Sigma.m[0][2];
    cov3D[3] = Sigma.m[1][1]; cov3D[4] = Sigma.m[1][2]; cov3D[5] = Sigma.m[2][2];
}

// ---- wave-cooperative SH staging -------------------------------------------------------------------
// The SH block of the 64 Gaussians of a wave is one contiguous 12 KB span ([P,16,3] floats).  A lane reading
// "its" 48 floats directly issues 12 loads whose 64 lanes are 192 B apart (64 cache lines per instruction);
// instead the wave copies the span with fully coalesced 16-byte accesses through a padded LDS slice
// (row stride 52 floats: conflict-free ds_read/write_b128 for 8-lane groups) and each lane then reads its row.
#define SH_ROW 52

__device__ __forceinline__ void wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The forward kernel does that copy in two steps, so that the global loads are IN FLIGHT while its projection / covariance arithmetic
// runs (a wave used to wait out three dependent memory round trips: means -> scale/rotation -> SH; now one):
// issue = the 12 coalesced 16-byte loads into registers, commit = their transposition into half a padded LDS slice at a time.
// need: bit L set <=> the Gaussian of lane L can be visible (it passed the frustum test); the 16-byte chunks of the other rows are not
// requested.  ~0 = every row.
struct ShPrefetch { float4 v[13]; };
// Bit g (0..63, per lane) of a wave-uniform 64-bit mask, from its two 32-bit halves -- NOT `(need >> g) & 1`: on gfx950 a 64-bit shift
// (v_lshlrev_b64 / v_lshrrev_b64 / v_ashrrev_i64) whose per-lane shift amount the register allocator happens to put into the LAST vector
// register of the wave's allocation shifts by VGPR0 instead, in waves that share their SIMD (the 32-bit amount is range-checked as a
// register pair; tools/dev/micro/topreg_probe.hip).  That -- not its three spilled registers -- made the 128-register build of
// preprocess_bwd_kernel drop the SH rows of a few Gaussians ("Round 4").  ex4dgs_amd/build.py refuses objects that
// hold such an instruction; this form gives the allocator no 64-bit shift by a per-lane amount to place.
__device__ __forceinline__ bool mask_bit(uint64_t need, int g)
{
    const uint32_t half = (g & 32) ? (uint32_t)(need >> 32) : (uint32_t)need;
    return ((half >> (g & 31)) & 1u) != 0u;
}
__device__ __forceinline__ void wave_issue_sh(const float *__restrict__ shs_wave, ShPrefetch &pf, int nrows, int nvec, int lane, uint64_t need)
{
    const float4 *src = reinterpret_cast<const float4 *>(shs_wave);
    // a full wave whose 64 rows are all wanted at degree 3 (every wave but the last of a frame whose Gaussians pass the frustum test):
    // twelve plain loads.  The predicated form below costs ~30 instructions per load (three nested exec regions, divisions by 12, zero fills)
    if (nrows == 64 && nvec == 12 && need == ~0ull) {          // (wave-uniform)
#pragma unroll
        for (int it = 0; it < 12; it++) pf.v[it] = src[it * 64 + lane];
        return;
    }
#pragma unroll
    for (int it = 0; it < 12; it++) {
        const int q = it * 64 + lane;
        const int g = q / 12, v = q - 12 * g;
        pf.v[it] = (g < nrows && v < nvec && mask_bit(need, g)) ? src[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
// split layout: the wave's rows of ONE (dc, rest) tensor pair as two linear 16-byte-aligned spans (rest: 64 x 45 floats = 720 float4,
// dc: 64 x 3 floats = 48 float4); returns false (nothing issued) for the one wave that straddles the static/dynamic boundary or for
// misaligned tensors -- those lanes read their rows straight from memory
__device__ __forceinline__ bool wave_issue_sh_split(const ShSplit &sp, int wave_first, int nrows, ShPrefetch &pf, int lane, uint64_t need)
{
    const bool all_dynamic = wave_first >= sp.n_static, all_static = wave_first + nrows <= sp.n_static;
    if (nrows <= 0 || !(all_dynamic || all_static)) return false;
    const int part = all_dynamic ? 1 : 0;
    const size_t r0 = (size_t)(wave_first - (part ? sp.n_static : 0));
    const float *rest = sp.rest[part] + r0 * 45, *dc = sp.dc[part] + r0 * 3;
    if (((((uintptr_t)rest) | ((uintptr_t)dc)) & 15) != 0 || nrows != 64) return false;
#pragma unroll
    for (int it = 0; it < 12; it++) {
        const int q = it * 64 + lane;
        const int r_lo = (4 * q) / 45, r_hi = (4 * q + 3) / 45;        // the at most two rows a 16-byte chunk of the rest span touches
        const bool want = q < 720 && (mask_bit(need, r_lo & 63) || mask_bit(need, r_hi & 63));      // (never a 64-bit shift by a per-lane amount: mask_bit)
        pf.v[it] = want ? reinterpret_cast<const float4 *>(rest)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    {
        const int r_lo = (4 * lane) / 3, r_hi = (4 * lane + 3) / 3;
        const bool want = lane < 48 && (mask_bit(need, r_lo & 63) || mask_bit(need, r_hi & 63));
        pf.v[12] = want ? reinterpret_cast<const float4 *>(dc)[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return true;
}

// The forward kernel transposes the prefetched block through HALF a slice, rows 32h .. 32h+31 at a time (commit half, the 32
// lanes of that half read their rows, next half): 6.5 KB of LDS per wave instead of 13 -> twice the workgroups per CU.
#define SH_HALF_FLOATS (32 * SH_ROW)              // >= 32 * 45 + 32 * 3 (split layout: rest rows, then dc rows)
#define SH_HALF_DC_OFFSET (32 * 45)
__device__ __forceinline__ void wave_commit_sh_half(float *lds, const ShPrefetch &pf, int lane, int h)
{
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const int it = 6 * h + i;                 // h is a compile-time constant at both call sites
        const int q = it * 64 + lane;
        const int g = q / 12 - 32 * h, v = q % 12;
        *reinterpret_cast<float4 *>(lds + g * SH_ROW + 4 * v) = pf.v[it];
    }
    wave_sync_lds();
}
__device__ __forceinline__ void wave_commit_sh_split_half(float *lds, const ShPrefetch &pf, int lane, int h)
{
#pragma unroll
    for (int it = 0; it < 12; it++) {
        const int q = it * 64 + lane - 360 * h;
        if (q >= 0 && q < 360) reinterpret_cast<float4 *>(lds)[q] = pf.v[it];
    }
    const int d = lane - 24 * h;
    if (d >= 0 && d < 24) reinterpret_cast<float4 *>(lds + SH_HALF_DC_OFFSET)[d] = pf.v[12];
    wave_sync_lds();
}


// half h of the wave's SH block straight into the half slice (no register prefetch): 6 coalesced 16-byte loads per lane
// need: bit L set <=> the Gaussian of lane L is visible; the 16-byte chunks of the other rows (~19 % of them) are not requested
__device__ __forceinline__ void wave_load_sh_half(const float *__restrict__ shs_wave, float *lds, int nrows, int nvec, int lane, int h, uint64_t need)
{
    const float4 *src = reinterpret_cast<const float4 *>(shs_wave);
    float4 t[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const int q = (6 * h + i) * 64 + lane;
        const int g = q / 12, v = q - 12 * g;
        t[i] = (g < nrows && v < nvec && mask_bit(need, g)) ? src[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const int q = (6 * h + i) * 64 + lane;
        const int g = q / 12 - 32 * h, v = q % 12;
        *reinterpret_cast<float4 *>(lds + g * SH_ROW + 4 * v) = t[i];
    }
    wave_sync_lds();
}
// split layout, full aligned wave (see wave_issue_sh_split for the conditions): rows 32h .. 32h+31 of the rest / dc spans
__device__ __forceinline__ bool wave_sh_split_stageable(const ShSplit &sp, int wave_first, int nrows)
{
    const bool all_dynamic = wave_first >= sp.n_static, all_static = wave_first + nrows <= sp.n_static;
    if (nrows != 64 || !(all_dynamic || all_static)) return false;
    const int part = all_dynamic ? 1 : 0;
    const size_t r0 = (size_t)(wave_first - (part ? sp.n_static : 0));
    return ((((uintptr_t)(sp.rest[part] + r0 * 45)) | ((uintptr_t)(sp.dc[part] + r0 * 3))) & 15) == 0;
}
// (the backward reads the coefficients k >= 1 only: the dc rows are not loaded at all; a 16-byte chunk of the rest span is requested
// when one of the at most two rows it touches belongs to a visible Gaussian)
__device__ __forceinline__ void wave_load_sh_split_half(const ShSplit &sp, int wave_first, float *lds, int lane, int h, uint64_t need)
{
    const int part = wave_first >= sp.n_static ? 1 : 0;
    const size_t r0 = (size_t)(wave_first - (part ? sp.n_static : 0)) + 32 * h;
    const float4 *rest = reinterpret_cast<const float4 *>(sp.rest[part] + r0 * 45);
    const uint32_t need_h = (uint32_t)(need >> (32 * h));
    float4 t[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const int q = i * 64 + lane;
        const int r_lo = (4 * q) / 45, r_hi = (4 * q + 3) / 45;       // q < 360: rows 0 .. 31
        const bool want = q < 360 && (((need_h >> r_lo) | (need_h >> (r_hi & 31))) & 1u);
        t[i] = want ? rest[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 6; i++) { const int q = i * 64 + lane; if (q < 360) reinterpret_cast<float4 *>(lds)[q] = t[i]; }
    wave_sync_lds();
}

// split layout, 32 rows of a full, aligned, non-straddling wave (wave_sh_split_stageable): two linear 16-byte copies
__device__ __forceinline__ void wave_store_sh_split_half(const ShSplitGrad &sp, int first, const float *lds, int lane)
{
    wave_sync_lds();
    const int part = first >= sp.n_static ? 1 : 0;
    const size_t r0 = (size_t)(first - (part ? sp.n_static : 0));
    float4 *rest = reinterpret_cast<float4 *>(sp.rest[part] + r0 * 45), *dc = reinterpret_cast<float4 *>(sp.dc[part] + r0 * 3);
#pragma unroll
    for (int i = 0; i < 6; i++) { const int q = i * 64 + lane; if (q < 360) rest[q] = reinterpret_cast<const float4 *>(lds)[q]; }
    if (lane < 24) dc[lane] = reinterpret_cast<const float4 *>(lds + SH_HALF_DC_OFFSET)[lane];
}

// 32 rows (the half slice of the backward kernel)
__device__ __forceinline__ void wave_store_sh_half(float *__restrict__ dst_half, const float *lds, int nrows, int lane)
{
    wave_sync_lds();
    float4 *dst = reinterpret_cast<float4 *>(dst_half);
#pragma unroll
    for (int it = 0; it < 6; it++) {
        const int q = it * 64 + lane;
        const int g = q / 12, v = q - 12 * g;
        if (g < nrows) dst[q] = *reinterpret_cast<const float4 *>(lds + g * SH_ROW + 4 * v);
    }
}

// The same rows taken from the four tensors the model keeps (dc [n,1,3] + rest [n,15,3], static rows first, then dynamic) --
// saves the [P,16,3] concatenation pass (scene/c_gaussian_model.py:351-353) and, in the backward, the split of dL_dsh.
// Inside one tensor the rows of a wave are ONE contiguous span (64 x 45 or 64 x 3 floats), so the copy is linear: no index
// arithmetic, 16-byte accesses when the span is aligned.  The LDS slice keeps that linear layout (rest rows at stride 45,
// dc rows at stride 3 behind them); a lane reads / writes its row with scalar LDS accesses at immediate offsets
// (stride 45 is odd: conflict-free).  Only the single wave that straddles the static/dynamic boundary goes element by element.




// d(colour)/d(direction) sums of the SH backward (CR/backward.cu:57-131): the part that reads the SH VALUES (sh[3 k + ch], k >= 1).
// One inlined function, so that every caller reads through its own address space (LDS slice or global row).
__device__ __forceinline__ void sh_direction_sums(const float *sh, int D, float x, float y, float z,
                                                  float (&dRGBdx)[3], float (&dRGBdy)[3], float (&dRGBdz)[3])
{
#define SHK(k) sh[3 * (k) + ch]
    if (D > 0) {
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            dRGBdx[ch] = -kSH_C1 * SHK(3);
            dRGBdy[ch] = -kSH_C1 * SHK(1);
            dRGBdz[ch] = kSH_C1 * SHK(2);
        }
        if (D > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                dRGBdx[ch] += kSH_C2[0] * y * SHK(4) + kSH_C2[2] * 2.f * -x * SHK(6) + kSH_C2[3] * z * SHK(7) + kSH_C2[4] * 2.f * x * SHK(8);
                dRGBdy[ch] += kSH_C2[0] * x * SHK(4) + kSH_C2[1] * z * SHK(5) + kSH_C2[2] * 2.f * -y * SHK(6) + kSH_C2[4] * 2.f * -y * SHK(8);
                dRGBdz[ch] += kSH_C2[1] * y * SHK(5) + kSH_C2[2] * 2.f * 2.f * z * SHK(6) + kSH_C2[3] * x * SHK(7);
            }
            if (D > 2) {
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    dRGBdx[ch] += (
                        kSH_C3[0] * SHK(9) * 3.f * 2.f * xy +
                        kSH_C3[1] * SHK(10) * yz +
                        kSH_C3[2] * SHK(11) * -2.f * xy +
                        kSH_C3[3] * SHK(12) * -3.f * 2.f * xz +
                        kSH_C3[4] * SHK(13) * (-3.f * xx + 4.f * zz - yy) +
                        kSH_C3[5] * SHK(14) * 2.f * xz +
                        kSH_C3[6] * SHK(15) * 3.f * (xx - yy));
                    dRGBdy[ch] += (
                        kSH_C3[0] * SHK(9) * 3.f * (xx - yy) +
                        kSH_C3[1] * SHK(10) * xz +
                        kSH_C3[2] * SHK(11) * (-3.f * yy + 4.f * zz - xx) +
                        kSH_C3[3] * SHK(12) * -3.f * 2.f * yz +
                        kSH_C3[4] * SHK(13) * -2.f * xy +
                        kSH_C3[5] * SHK(14) * -2.f * yz +
                        kSH_C3[6] * SHK(15) * -3.f * 2.f * xy);
                    dRGBdz[ch] += (
                        kSH_C3[1] * SHK(10) * xy +
                        kSH_C3[2] * SHK(11) * 4.f * 2.f * yz +
                        kSH_C3[3] * SHK(12) * 3.f * (2.f * zz - xx - yy) +
                        kSH_C3[4] * SHK(13) * 4.f * 2.f * xz +
                        kSH_C3[5] * SHK(14) * (xx - yy));
                }
            }
        }
    }
#undef SHK
}

// FAST: the configuration every training / rendering frame of the reference runs (one [P,16,3] SH tensor at degree 3, scale + rotation,
// no precomputed colours or covariances) with its constants known at compile time -- the generic kernel carries runtime tests for
// every other input combination around each of its loads (round 5: 1706 vector + ~500 scalar instructions per wave, a third of them
// such bookkeeping).  Same arithmetic, same order: the results are bit-identical (tests/test_gpu_round5.py).
template <bool FAST>
__global__ __launch_bounds__(256) void preprocess_fwd_kernel(
    int P, int D_, int M_,
    const float *__restrict__ means3D, const float *__restrict__ dir3D, const float *__restrict__ scales, float scale_modifier,
    const float *__restrict__ rotations, const float *__restrict__ opacities, const float *__restrict__ shs,
    const float *__restrict__ cov3D_precomp_, const float *__restrict__ colors_precomp_,
    const float *__restrict__ viewmatrix, const float *__restrict__ projmatrix, const float *__restrict__ campos,
    int W, int H, float tanx, float tany, float fx, float fy, float kernel_size, float min_depth, float max_depth,
    int prefiltered, uint32_t *__restrict__ prefilter_violation,
    int32_t *__restrict__ radii, float4 *__restrict__ records, float *__restrict__ cov3Ds,
    uint8_t *__restrict__ clamped, uint32_t *__restrict__ tiles_touched, uint2 *__restrict__ rects,
    uint32_t *__restrict__ depth_keys, uint32_t *__restrict__ depth_vals, uint32_t depth_key_base, uint32_t depth_key_invisible,
    uint32_t *__restrict__ total_instances, const ShSplit sp_, float *__restrict__ sh_dsums, int sh_predicate, uint32_t *__restrict__ rects4,
    uint32_t *__restrict__ key_range_slots, int global_flags)
{
    const int D = FAST ? 3 : D_, M = FAST ? 16 : M_;
    const float *__restrict__ cov3D_precomp = FAST ? nullptr : cov3D_precomp_;
    const float *__restrict__ colors_precomp = FAST ? nullptr : colors_precomp_;
    ShSplit sp = sp_;
    if (FAST) { sp.dc[0] = sp.dc[1] = sp.rest[0] = sp.rest[1] = nullptr; sp.n_static = 0; __builtin_assume(shs != nullptr); }
    __shared__ __attribute__((aligned(16))) float sh_lds[4 * SH_HALF_FLOATS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Every wave is independent (no workgroup barrier, wave-private LDS slice, per-wave instance count): chunk wc of 64 Gaussians.
    // Round 4, measured and dropped (profiles/archive/r04b_experiments.txt, r04c_experiments.txt): distinct s_setprio levels per wave slot and
    // one-wave workgroups, meant to pull the load / arithmetic / store phases of co-resident waves apart -- no effect (+-2 %).  What the
    // kernel's time is made of (tools/dev/pre_probe.py, 1.0 M Gaussians): 49 us with precomputed colours (no SH path at all), +24 us for
    // the SH staging and evaluation at degree 0 (16 of 192 bytes of SH read per Gaussian), +10 us for the other 176 bytes at degree 3,
    // +10 us for the direction sums left for the backward (which saves 31 us there): it does not follow its bytes.
    // MSD depth sort: the workgroup's key range is collected in LDS (round 6: one pair per workgroup instead of one per wave, so that the
    // depth sort's histogram kernel can reduce the pairs itself and the one-workgroup range kernel in front of it is gone).  The only
    // barrier of the kernel, at its very start; afterwards every wave is on its own.
    __shared__ uint32_t s_kr[4];
    if (key_range_slots) {
        if (threadIdx.x < 4) s_kr[threadIdx.x] = 0u;
        __syncthreads();
    }
    const int wc = (int)blockIdx.x * 4 + wave;
    if (wc >= ((P + 63) >> 6)) return;
    const int idx = wc * 64 + lane;
    const bool in_range = idx < P;
    // the 2x16 camera floats are wave-uniform: they live in SGPRs / the scalar cache
    float vm[16], pm[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { vm[i] = viewmatrix[i]; pm[i] = projmatrix[i]; }

    // ---- every global load of the kernel is issued here, before any arithmetic depends on one of them: the wave's SH block (the
    // bulk: 12 KB, coalesced), the Gaussian's own 12 + 32 + 12 bytes.  Culled Gaussians (~20 %) cost their bytes but no wave waits
    // for a second or third memory round trip any more.
    const int ncoef = (D + 1) * (D + 1);
    const bool split = sp.rest[0] != nullptr || sp.rest[1] != nullptr;        // implies M == 16, shs == nullptr (checked by the API)
    const bool staged = (shs != nullptr || split) && (M == 16);
    const int wave_first = wc * 64;
    const int wave_rows = (P - wave_first) < 64 ? (P - wave_first) : 64;
    ShPrefetch pf;
    bool prefetched = false;
    // sh_predicate (default; option "preprocess_sh_predicate"): the frustum test (CR/auxiliary.h in_frustum: 12 bytes of input) runs
    // BEFORE the SH rows are requested, and the rows of the Gaussians it culls are not requested at all -- one more dependent memory
    // round trip per wave against the SH bytes of every Gaussian the view frustum culls.  (BASELINE's synthetic scenes lose their
    // invisible 19 % AFTER the projection -- empty tile rect -- and 99.98 % pass this test: no traffic saved there, -1.6 us of 92.)
    bool pre_ok = false;
    float3 pre_p = make_float3(0.f, 0.f, 0.f), pre_view = make_float3(0.f, 0.f, 0.f);
    float pre_nx = 0.f, pre_ny = 0.f;
    uint64_t need = ~0ull;
    if (sh_predicate) {
        if (in_range) {
            pre_p = make_float3(means3D[3 * (size_t)idx], means3D[3 * (size_t)idx + 1], means3D[3 * (size_t)idx + 2]);
            pre_ok = frustum_test(pre_p, vm, pm, min_depth, max_depth, pre_view, pre_nx, pre_ny);
        }
        need = __ballot(pre_ok);
    }
    if (staged) {
        if (split) prefetched = wave_issue_sh_split(sp, wave_first, wave_rows, pf, lane, need);
        else { wave_issue_sh(shs + (size_t)wave_first * 48, pf, wave_rows, (ncoef * 3 + 3) / 4, lane, need); prefetched = true; }
    }
    float4 in_q = make_float4(0.f, 0.f, 0.f, 0.f);
    float in_s0 = 0.f, in_s1 = 0.f, in_s2 = 0.f, in_op = 0.f, in_d0 = 0.f, in_d1 = 0.f, in_d2 = 0.f;
    if (in_range) {
        if (!cov3D_precomp) {
            in_q = reinterpret_cast<const float4 *>(rotations)[idx];
            in_s0 = scales[3 * (size_t)idx]; in_s1 = scales[3 * (size_t)idx + 1]; in_s2 = scales[3 * (size_t)idx + 2];
        }
        in_op = opacities[idx];
        if (dir3D) { in_d0 = dir3D[3 * (size_t)idx]; in_d1 = dir3D[3 * (size_t)idx + 1]; in_d2 = dir3D[3 * (size_t)idx + 2]; }
    }

    // rows 0..31 go to the LDS half-slice right away (their 26 staging registers are free during the projection arithmetic; rows
    // 32..63 stay in registers until the first half has been consumed): peak register pressure = arithmetic + half a block
    if (staged && prefetched) {
        float *lds0 = sh_lds + wave * SH_HALF_FLOATS;
        if (split) wave_commit_sh_split_half(lds0, pf, lane, 0);
        else wave_commit_sh_half(lds0, pf, lane, 0);
    }

    int out_radius = 0;
    uint32_t out_tiles = 0;
    uint32_t depth_key = depth_key_invisible;   // invisible Gaussians sort behind every visible one
    uint2 rect = make_uint2(0u, 0u);
    bool visible = false, filtered = false;
    float3 p = make_float3(0.f, 0.f, 0.f), conic = make_float3(0.f, 0.f, 0.f);
    float pix_x = 0.f, pix_y = 0.f, depth = 0.f, coef = 0.f;
    if (in_range) do {
        float3 p_view; float ndc_x, ndc_y;
        bool in_view;
        if (sh_predicate) { p = pre_p; p_view = pre_view; ndc_x = pre_nx; ndc_y = pre_ny; in_view = pre_ok; }
        else {
            p = make_float3(means3D[3 * (size_t)idx], means3D[3 * (size_t)idx + 1], means3D[3 * (size_t)idx + 2]);
            in_view = frustum_test(p, vm, pm, min_depth, max_depth, p_view, ndc_x, ndc_y);
        }
        if (!in_view) {
            if (prefiltered) { filtered = true; if (global_flags) atomicOr(prefilter_violation, 1u); }
            break;
        }
        float cov3D[6];
        if (cov3D_precomp) {
#pragma unroll
            for (int i = 0; i < 6; i++) cov3D[i] = cov3D_precomp[6 * (size_t)idx + i];
        } else {
            cov3d_from_scale_rotation(scale_modifier, in_s0, in_s1, in_s2, in_q, cov3D);
            // the array is only materialised on request (option "geom_debug_arrays"): the backward recomputes it from the same inputs
            if (cov3Ds) {
#pragma unroll
                for (int i = 0; i < 6; i++) cov3Ds[6 * (size_t)idx + i] = cov3D[i];
            }
        }
        Cov2DCtx c;
        cov2d_common(p, fx, fy, tanx, tany, cov3D, vm, c);
        // anti-aliasing coefficient, CR/forward.cu:112-118 (float products, double max / sqrt / compare)
        const float det_0 = (float)fmax(1e-6, (double)(c.cov.m[0][0] * c.cov.m[1][1] - c.cov.m[0][1] * c.cov.m[0][1]));
        const float det_1 = (float)fmax(1e-6, (double)((c.cov.m[0][0] + kernel_size) * (c.cov.m[1][1] + kernel_size) - c.cov.m[0][1] * c.cov.m[0][1]));
        coef = (float)sqrt((double)det_0 / ((double)det_1 + 1e-6) + 1e-6);
        if ((double)det_0 <= 1e-6 || (double)det_1 <= 1e-6) coef = 0.0f;
        const float ca = c.cov.m[0][0] + kernel_size, cb = c.cov.m[0][1], cc = c.cov.m[1][1] + kernel_size;

        const float det = ca * cc - cb * cb;
        if (det == 0.0f) break;
        const float det_inv = 1.f / det;
        conic = make_float3(cc * det_inv, -cb * det_inv, ca * det_inv);
        const float mid = 0.5f * (ca + cc);
        const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        pix_x = (float)((((double)ndc_x + 1.0) * (double)W - 1.0) * 0.5);   // ndc2Pix, CR/auxiliary.h:41-44
        pix_y = (float)((((double)ndc_y + 1.0) * (double)H - 1.0) * 0.5);
        const int gx = (W + EX4D_TILE - 1) / EX4D_TILE, gy = (H + EX4D_TILE - 1) / EX4D_TILE;
        const int ri = to_int_sat(my_radius);
        int x0, y0, x1, y1;
        tile_rect(pix_x, pix_y, ri, gx, gy, x0, y0, x1, y1);
        const uint32_t area = (uint32_t)(x1 - x0) * (uint32_t)(y1 - y0);
        if (area == 0) break;
        rect = make_uint2((uint32_t)x0 | ((uint32_t)y0 << 16), (uint32_t)(x1 - x0) | ((uint32_t)(y1 - y0) << 16));
        visible = true;
        depth = p_view.z;
        out_radius = ri;
        out_tiles = area;
        // depth > min_depth >= 0 => unsigned order of the bit pattern == numeric order: the same bits the reference puts in the low
        // key word (CR/rasterizer_impl.cu:106).  Visible depths lie in (min_depth, max_depth], so their bit patterns lie in a range the
        // host knows: keys are taken relative to its lower end (same order, fewer significant bits -> one radix pass less)
        depth_key = __float_as_uint(p_view.z) - depth_key_base;
    } while (0);

    // MSD depth sort (ex4d_binning.hip): the range [min, max] of the frame's VISIBLE depth keys, so that its top digit is cut from the
    // range the frame occupies and not from [min_depth, max_depth] (a scene inside a narrow depth band would otherwise fall into a
    // handful of buckets).  Every wave leaves (max key, max ~key) of its visible Gaussians, (0, 0) if it has none
    if (key_range_slots) {
        uint32_t kmax = visible ? depth_key : 0u, nkmin = visible ? ~depth_key : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t a = __shfl_xor(kmax, o, 64), b = __shfl_xor(nkmin, o, 64);
            kmax = a > kmax ? a : kmax; nkmin = b > nkmin ? b : nkmin;
        }
        // (one plain 8-byte store per wave; a one-workgroup kernel of the depth sort reduces the P / 64 pairs.  Measured instead, round 5:
        // 2 atomics per wave into 2 x 64 slots +10 us on this kernel wherever they were issued; atomics only where a wave would raise
        // its slot, the slot read first: +55 us -- the slot lines are a hot spot and every wave's first wait included them)
        if (lane == 0) {
            // (LDS atomics of one wave arrive in order, the last wave's ticket behind everybody's maxima)
            atomicMax(&s_kr[0], kmax); atomicMax(&s_kr[1], nkmin);
            const int nchunks = (P + 63) >> 6, mine = nchunks - (int)blockIdx.x * 4;
            const uint32_t t = atomicAdd(&s_kr[2], 1u);
            if ((int)t == (mine < 4 ? mine : 4) - 1)
                reinterpret_cast<uint2 *>(key_range_slots)[blockIdx.x] = make_uint2(__hip_atomic_load(&s_kr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP),
                                                                                    __hip_atomic_load(&s_kr[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        }
    }
    // ---- colour: SH -> RGB (CR/forward.cu:20-71) or precomputed
    float coefv[16][3];
    if (staged) {
        float *lds = sh_lds + wave * SH_HALF_FLOATS;
        float tmp[48];
#pragma unroll
        for (int f = 0; f < 48; f++) tmp[f] = 0.f;
        const int r = lane & 31;
        if (split) {
            if (prefetched) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    if (h == 1) wave_commit_sh_split_half(lds, pf, lane, 1);       // half 0 was committed before the arithmetic
                    if (visible && (lane >> 5) == h) {
#pragma unroll
                        for (int f = 0; f < 3; f++) tmp[f] = lds[SH_HALF_DC_OFFSET + r * 3 + f];
#pragma unroll
                        for (int f = 3; f < 48; f++) if (f < ncoef * 3) tmp[f] = lds[r * 45 + (f - 3)];
                    }
                    wave_sync_lds();
                }
            } else if (visible) {
                // the one wave that straddles the static / dynamic boundary (or misaligned tensors): rows straight from memory
                const int part = idx >= sp.n_static;
                const size_t row = (size_t)(idx - (part ? sp.n_static : 0));
#pragma unroll
                for (int f = 0; f < 3; f++) tmp[f] = sp.dc[part][row * 3 + f];
#pragma unroll
                for (int f = 3; f < 48; f++) if (f < ncoef * 3) tmp[f] = sp.rest[part][row * 45 + (f - 3)];
            }
        } else {
            const int nvec = (ncoef * 3 + 3) / 4;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if (h == 1) wave_commit_sh_half(lds, pf, lane, 1);                 // half 0 was committed before the arithmetic
                if (visible && (lane >> 5) == h) {
                    const float4 *row = reinterpret_cast<const float4 *>(lds + r * SH_ROW);
#pragma unroll
                    for (int v = 0; v < 12; v++) {
                        if (v < nvec) { const float4 t4 = row[v]; tmp[4 * v] = t4.x; tmp[4 * v + 1] = t4.y; tmp[4 * v + 2] = t4.z; tmp[4 * v + 3] = t4.w; }
                    }
                }
                wave_sync_lds();
            }
        }
#pragma unroll
        for (int k = 0; k < 16; k++) { coefv[k][0] = tmp[3 * k]; coefv[k][1] = tmp[3 * k + 1]; coefv[k][2] = tmp[3 * k + 2]; }
    } else if (shs && visible) {
        const float *sh = shs + (size_t)idx * M * 3;
#pragma unroll
        for (int k = 0; k < 16; k++)
#pragma unroll
            for (int ch = 0; ch < 3; ch++) coefv[k][ch] = (k < ncoef && k < M) ? sh[3 * k + ch] : 0.f;
    }
    if (visible) {
        float res[3];
        if (colors_precomp) {
            res[0] = colors_precomp[3 * (size_t)idx]; res[1] = colors_precomp[3 * (size_t)idx + 1]; res[2] = colors_precomp[3 * (size_t)idx + 2];
        } else {
            float dx = p.x - campos[0], dy = p.y - campos[1], dz = p.z - campos[2];
            const float len = sqrtf(dx * dx + dy * dy + dz * dz);
            const float x = dx / len, y = dy / len, z = dz / len;
            uint8_t clamp_bits = 0;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
#define SHK(k) coefv[k][ch]
                float result = kSH_C0 * SHK(0);
                if (D > 0) {
                    result = result - kSH_C1 * y * SHK(1) + kSH_C1 * z * SHK(2) - kSH_C1 * x * SHK(3);
                    if (D > 1) {
                        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                        result = result +
                            kSH_C2[0] * xy * SHK(4) +
                            kSH_C2[1] * yz * SHK(5) +
                            kSH_C2[2] * (2.0f * zz - xx - yy) * SHK(6) +
                            kSH_C2[3] * xz * SHK(7) +
                            kSH_C2[4] * (xx - yy) * SHK(8);
                        if (D > 2) {
                            result = result +
                                kSH_C3[0] * y * (3.0f * xx - yy) * SHK(9) +
                                kSH_C3[1] * xy * z * SHK(10) +
                                kSH_C3[2] * y * (4.0f * zz - xx - yy) * SHK(11) +
                                kSH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SHK(12) +
                                kSH_C3[4] * x * (4.0f * zz - xx - yy) * SHK(13) +
                                kSH_C3[5] * z * (xx - yy) * SHK(14) +
                                kSH_C3[6] * x * (xx - 3.0f * yy) * SHK(15);
                        }
                    }
                }
#undef SHK
                result += 0.5f;
                if (result < 0) clamp_bits |= (uint8_t)(1u << ch);
                res[ch] = fmaxf(result, 0.0f);
            }
            clamped[idx] = clamp_bits;
            if (sh_dsums) {
                // d(colour)/d(direction) sums of the SH backward (CR/backward.cu:57-131): they depend on the SH values and the direction
                // only, both in registers here -- stored (36 B per visible Gaussian, Ex4dParams.prepare_backward) so that the backward
                // does not read the 192-byte SH rows again.  Same function, same operands as the backward's own evaluation: same bits.
                float dRGBdx[3] = { 0, 0, 0 }, dRGBdy[3] = { 0, 0, 0 }, dRGBdz[3] = { 0, 0, 0 };
                sh_direction_sums(&coefv[0][0], D, x, y, z, dRGBdx, dRGBdy, dRGBdz);
                // one 36-byte row per Gaussian (measured against nine planes of P floats: rows 0.100 / 0.076 ms for this kernel / the
                // backward, planes 0.111 / 0.080 -- nine DRAM pages per wave instead of one region)
                float *o = sh_dsums + 9 * (size_t)idx;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) { o[ch] = dRGBdx[ch]; o[3 + ch] = dRGBdy[ch]; o[6 + ch] = dRGBdz[ch]; }
            }
        }
        // per-Gaussian constants of the compositing kernels' quadrant cull (ex4d_composite.hip), evaluated once here
        // instead of once per (Gaussian, tile, quadrant): tau = ln(255 w) + 1 % slack is the largest value of the quadratic
        // form q(d) that still reaches alpha >= 1/255; -inf = never contributes (w < 1/255), +inf = never cull (conic
        // not provably positive definite); k1, k2 = minimisers of q along a vertical / horizontal box edge
        const float w_op = in_op * coef;
        float tau;
        if (w_op < (1.0f / 255.0f)) tau = -__builtin_inff();
        else if (!(conic.x > 0.f && conic.z > 0.f && conic.x * conic.z - conic.y * conic.y > 0.f)) tau = __builtin_inff();
        else tau = logf(255.0f * w_op) + 0.01f;
        float4 *rec = records + 4 * (size_t)idx;
        rec[0] = make_float4(pix_x, pix_y, conic.x, conic.y);
        rec[1] = make_float4(conic.z, tau, -conic.y / conic.z, -conic.y / conic.x);
        rec[2] = make_float4(depth, res[0], res[1], res[2]);
        rec[3] = make_float4(in_d0, in_d1, in_d2, w_op);
    }
    // frame flag for the compositing forward: does any visible Gaussian carry a flow vector?  (plain store of the same value by every
    // wave that sees one: no atomic, no contention; the training loop's dir3D is the all-zero gradient trap and never sets it)
    // Round 6: the two flags also travel in the chunk's count pair (bits 31 / 30 of the segment count, below) -- the synchronous forward reads
    // them there with the counts, and the frame-flag words need no zero-fill launch in front of this kernel (global_flags = 0); the
    // asynchronous forward keeps the words (Ex4dFrameStatus is copied from them).
    const bool chunk_flow = __ballot(visible && (in_d0 != 0.f || in_d1 != 0.f || in_d2 != 0.f)) != 0ull;
    const bool chunk_filtered = prefiltered && __ballot(filtered) != 0ull;
    if (global_flags && chunk_flow) {
        if (lane == 0) prefilter_violation[1] = 1u;
    }
    // frame flag [3]: this frame's direction sums exist (checked by the backward); written either way, so that the word never keeps the
    // mark of an earlier frame on these buffers
    if (idx == 0) prefilter_violation[2] = sh_dsums ? EX4D_DSUMS_MARK : 0u;
    if (in_range) {
        radii[idx] = out_radius;
        if (tiles_touched) tiles_touched[idx] = out_tiles;      // only on request ("geom_debug_arrays"): the rect carries the count
        if (rects) rects[idx] = rect;      // (the 8-byte form: frames whose rects do not fit the packed word, and on request -- "geom_debug_arrays")
        if (rects4) rects4[idx] = (rect.x & 0xFFu) | ((rect.x >> 16) << 8) | ((rect.y & 0xFFu) << 16) | ((rect.y >> 16) << 24);      // (<= 255 x 255 tiles)
        depth_keys[idx] = depth_key;
        if (depth_vals) depth_vals[idx] = (uint32_t)idx;          // (nullptr: the depth sort's first pass takes the index itself)
    }
    // number of tile instances (the reference's num_rendered = last element of the inclusive scan,
    // CR/rasterizer_impl.cu:295-299): it does not depend on the depth order, so it is summed here and read back by the
    // host WHILE the depth sort runs -- the blocking read-back no longer leaves the GPU idle.
    // (one plain store per chunk of 64 Gaussians; the host adds the partial sums -- a single atomic counter would
    // serialise ~12 ns per arrival)
    // (round 6: and the number of tile-row segments -- rect heights -- which sizes the grids of the row-segment tile sort, ex4d_rowsort.hip)
    uint32_t wave_sum = out_tiles, wave_seg = visible ? (rect.y >> 16) : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { wave_sum += __shfl_xor(wave_sum, o, 64); wave_seg += __shfl_xor(wave_seg, o, 64); }
    // (a chunk has at most 64 x 255 segments: bits 30 / 31 of the word are free for the chunk's flags)
    if (lane == 0) reinterpret_cast<uint2 *>(total_instances)[wc] = make_uint2(wave_sum, wave_seg | (chunk_flow ? EX4D_CHUNK_FLOW : 0u) | (chunk_filtered ? EX4D_CHUNK_FILTERED : 0u));        // one pair per 64-Gaussian chunk
}

__global__ __launch_bounds__(256) void mark_visible_kernel(int P, const float *__restrict__ means3D,
    const float *__restrict__ viewmatrix, const float *__restrict__ projmatrix, float min_depth, float max_depth,
    uint8_t *__restrict__ present)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    float vm[16], pm[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { vm[i] = viewmatrix[i]; pm[i] = projmatrix[i]; }
    const float3 p = make_float3(means3D[3 * (size_t)idx], means3D[3 * (size_t)idx + 1], means3D[3 * (size_t)idx + 2]);
    float3 pv; float nx, ny;
    present[idx] = frustum_test(p, vm, pm, min_depth, max_depth, pv, nx, ny) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// Backward of the per-Gaussian stage.  Fuses computeCov2DCUDA + preprocessCUDA(bwd) + the zero-fill /
// unpack of the ten gradient tensors: every output row is written exactly once.
// acc16[idx][0..12] are the sums produced by the compositing backward:
//   layout 0: 0..2 dL_dmean2D.xyz (xy without ln2 W/2, ln2 H/2), 3..5 dL_dconic.(x,y,w) (without -1/2), 6 dL_dopacity,
//             7..9 dL_dcolor, 10..12 dL_ddir
//   layout 1 (matrix-core compositing backward): 0,1 = sum sG dx, sum sG dy; 3..5 = sum sG dx^2, sum sG dx dy, sum sG dy^2 with
//             sG = dL_dG G and d = mean2D - pixel; rest as layout 0.  dG/dmean2D = -G (A dx + B dy, C dy + B dx) (CR/backward.cu:
//             :664-670), dG/dconic = -1/2 G (dx^2, dx dy, dy^2) (:673-675): linear in those sums, conic (A, B, C) from the record.
// ------------------------------------------------------------------------------------------------
// DSUMS = true: the forward kernel left the SH direction sums (GeomState::sh_dsums, Ex4dParams.prepare_backward): the SH
// rows are not read at all (-155 MB of 535 at 1.0 M Gaussians), the staging slice only serves the coalesced stores
template <bool DSUMS>
__global__ __launch_bounds__(256) void preprocess_bwd_kernel(
    int P, int D, int M,
    const float *__restrict__ means3D, const int32_t *__restrict__ radii, const float *__restrict__ shs,
    const uint8_t *__restrict__ clamped, const float *__restrict__ scales, const float *__restrict__ rotations,
    float scale_modifier, const float *__restrict__ cov3Ds, const float *__restrict__ viewmatrix,
    const float *__restrict__ projmatrix, const float *__restrict__ campos,
    int W, int H, float fx, float fy, float tanx, float tany, float kernel_size,
    const float *__restrict__ acc16, const float4 *__restrict__ records,
    float *__restrict__ dL_dmeans2D, float *__restrict__ dL_dcolors, float *__restrict__ dL_dopacity,
    float *__restrict__ dL_dmeans3D, float *__restrict__ dL_dcov3D, float *__restrict__ dL_dsh,
    float *__restrict__ dL_dscales, float *__restrict__ dL_drotations, float *__restrict__ dL_ddir, const ShSplit sp, const ShSplitGrad gsp,
    const float *__restrict__ sh_dsums, const uint32_t *__restrict__ frame_flags)
{
    __shared__ __attribute__((aligned(16))) float sh_lds[4 * SH_HALF_FLOATS];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool in_range = idx < P;
    const bool split = sp.rest[0] != nullptr || sp.rest[1] != nullptr;        // implies M == 16, shs == nullptr (checked by the API)
    const bool staged = (shs != nullptr || split) && (M == 16);
    float *lds_row_base = sh_lds + wave * SH_HALF_FLOATS;
    // the prepare_backward contract (include/ex4d_rasterizer.h): the direction sums exist only when the FORWARD on these buffers ran
    // with prepare_backward = 1 -- it leaves a mark in the frame flags.  A backward that asks for them on other buffers (raw C-ABI
    // callers, replayed snapshots) must not return plausible numbers computed from uninitialised memory: its gradients come out NaN
    const bool dsums_ok = !DSUMS || frame_flags[3] == EX4D_DSUMS_MARK;
    const int wave_first = blockIdx.x * 256 + wave * 64;
    const int wave_rows = (P - wave_first) < 64 ? (P - wave_first) : 64;
    // ---- all global loads are issued before anything waits for one of them (see preprocess_fwd_kernel): the wave's SH block, the
    // Gaussian's accumulator row, mean, covariance, scale / rotation, clamp bits -- also for the ~20 % culled Gaussians, whose rows
    // are then simply not used
    // the SH block is NOT prefetched into registers here (the forward kernel does that; measured: no difference for this kernel, which
    // streams at ~5.2 TB/s either way): the half slices are loaded where they are consumed.  "prefetched" = the wave's rows can be staged
    const bool prefetched = staged && (split ? wave_sh_split_stageable(sp, wave_first, wave_rows) : true);
    int in_radius = 0;
    float4 in_r0 = make_float4(0.f, 0.f, 0.f, 0.f), in_r1 = in_r0, in_r2 = in_r0, in_r3 = in_r0, in_q = in_r0;
    float in_mean[3] = { 0.f, 0.f, 0.f }, in_cov[6] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f }, in_s[3] = { 0.f, 0.f, 0.f };
    uint8_t in_clamped = 0;
    if (in_range) {
        in_radius = radii[idx];
        const float4 *row = reinterpret_cast<const float4 *>(acc16 + 16 * (size_t)idx);
        in_r0 = row[0]; in_r1 = row[1]; in_r2 = row[2]; in_r3 = row[3];
#pragma unroll
        for (int i = 0; i < 3; i++) in_mean[i] = means3D[3 * (size_t)idx + i];
        if (!scales) {          // precomputed covariance: the caller's tensor; otherwise recomputed below from scale / rotation
#pragma unroll
            for (int i = 0; i < 6; i++) in_cov[i] = cov3Ds[6 * (size_t)idx + i];
        }
        if (scales) {
            in_q = reinterpret_cast<const float4 *>(rotations)[idx];
#pragma unroll
            for (int i = 0; i < 3; i++) in_s[i] = scales[3 * (size_t)idx + i];
        }
        if (shs || split) in_clamped = clamped[idx];
    }
    const bool visible = in_range && (in_radius > 0);
    float g_mean2D[3] = { 0, 0, 0 }, g_color[3] = { 0, 0, 0 }, g_dir[3] = { 0, 0, 0 }, g_opacity = 0;
    float g_mean3D[3] = { 0, 0, 0 }, g_cov[6] = { 0, 0, 0, 0, 0, 0 }, g_scale[3] = { 0, 0, 0 }, g_rot[4] = { 0, 0, 0, 0 };
    // ---- SH backward first (CR/backward.cu:20-139): it consumes the prefetched SH block, whose 52 registers are then free for the
    // projection / covariance arithmetic.  Its contribution to dL_dmean3D (sh_j) is added behind the projection terms below.
    float3 sh_dir = make_float3(0.f, 0.f, 0.f), sh_dir_orig = make_float3(0.f, 0.f, 0.f);
    float sh_dRGB[3] = { 0.f, 0.f, 0.f }, sh_j[3] = { 0.f, 0.f, 0.f };
    if (visible && (shs || split)) {
        const float3 dir_orig = make_float3(in_mean[0] - campos[0], in_mean[1] - campos[1], in_mean[2] - campos[2]);
        const float len = sqrtf(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
        sh_dir = make_float3(dir_orig.x / len, dir_orig.y / len, dir_orig.z / len);
        sh_dir_orig = dir_orig;
        const float gc[3] = { in_r1.w, in_r2.x, in_r2.y };       // dL_dcolor (accumulators 7..9)
#pragma unroll
        for (int ch = 0; ch < 3; ch++) sh_dRGB[ch] = gc[ch] * (((in_clamped >> ch) & 1) ? 0.f : 1.f);
    }
    // ---- the part of the SH backward that reads the SH values, CR/backward.cu:57-139.  The prefetched block goes through HALF a slice
    // of LDS, rows 32h .. 32h+31 at a time (6.5 KB per wave instead of 13: twice the workgroups per CU, like the forward kernel);
    // the 32 lanes of the half evaluate the sums while the other 32 idle -- the kernel waits on memory, not on the VALU.
    if (shs || split) {
        float dRGBdx[3] = { 0, 0, 0 }, dRGBdy[3] = { 0, 0, 0 }, dRGBdz[3] = { 0, 0, 0 };
        const int r = lane & 31;
        const uint64_t need_rows = __ballot(visible);
        if (DSUMS) {
            if (visible) {
                const float *o = sh_dsums + 9 * (size_t)idx;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) { dRGBdx[ch] = o[ch]; dRGBdy[ch] = o[3 + ch]; dRGBdz[ch] = o[6 + ch]; }
                if (!dsums_ok) { const float nan = __int_as_float(0x7fc00000); for (int ch = 0; ch < 3; ch++) dRGBdx[ch] = dRGBdy[ch] = dRGBdz[ch] = nan; }
            }
        } else if (staged && prefetched) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if (split) wave_load_sh_split_half(sp, wave_first, lds_row_base, lane, h, need_rows);
                else wave_load_sh_half(shs + (size_t)wave_first * 48, lds_row_base, wave_rows, ((D + 1) * (D + 1) * 3 + 3) / 4, lane, h, need_rows);
                if (visible && (lane >> 5) == h) {
                    // split layout: rest rows at stride 45 (coefficient k >= 1 at 3 (k - 1)); only k >= 1 is read
                    if (split) sh_direction_sums(lds_row_base + r * 45 - 3, D, sh_dir.x, sh_dir.y, sh_dir.z, dRGBdx, dRGBdy, dRGBdz);
                    else sh_direction_sums(lds_row_base + r * SH_ROW, D, sh_dir.x, sh_dir.y, sh_dir.z, dRGBdx, dRGBdy, dRGBdz);
                }
                wave_sync_lds();
            }
        } else if (visible) {
            // rows straight from memory: the one wave that straddles the static / dynamic boundary, misaligned tensors, M != 16
            if (split) {
                const int part = idx >= sp.n_static;
                sh_direction_sums(sp.rest[part] + (size_t)(idx - (part ? sp.n_static : 0)) * 45 - 3, D, sh_dir.x, sh_dir.y, sh_dir.z, dRGBdx, dRGBdy, dRGBdz);
            } else {
                sh_direction_sums(shs + (size_t)idx * M * 3, D, sh_dir.x, sh_dir.y, sh_dir.z, dRGBdx, dRGBdy, dRGBdz);
            }
        }
        if (visible) {
            const float *dRGB = sh_dRGB;
            const float3 dir_orig = sh_dir_orig;
            const float3 dL_ddirv = make_float3(
                dRGBdx[0] * dRGB[0] + dRGBdx[1] * dRGB[1] + dRGBdx[2] * dRGB[2],
                dRGBdy[0] * dRGB[0] + dRGBdy[1] * dRGB[1] + dRGBdy[2] * dRGB[2],
                dRGBdz[0] * dRGB[0] + dRGBdz[1] * dRGB[1] + dRGBdz[2] * dRGB[2]);
            // Jacobian of the direction normalisation, CR/auxiliary.h:235-245
            const float3 v = dir_orig;
            const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
            const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            const float jx = ((+sum2 - v.x * v.x) * dL_ddirv.x - v.y * v.x * dL_ddirv.y - v.z * v.x * dL_ddirv.z) * invsum32;
            const float jy = (-v.x * v.y * dL_ddirv.x + (sum2 - v.y * v.y) * dL_ddirv.y - v.z * v.y * dL_ddirv.z) * invsum32;
            const float jz = (-v.x * v.z * dL_ddirv.x - v.y * v.z * dL_ddirv.y + (sum2 - v.z * v.z) * dL_ddirv.z) * invsum32;
            sh_j[0] = jx; sh_j[1] = jy; sh_j[2] = jz;       // added to dL_dmean3D behind its projection terms (the reference's order)
        }
    }
    if (visible) {
        float vm[16], pm[16];
#pragma unroll
        for (int i = 0; i < 16; i++) { vm[i] = viewmatrix[i]; pm[i] = projmatrix[i]; }
        const float4 r0 = in_r0, r1 = in_r1, r2 = in_r2, r3 = in_r3;
        // factors the compositing backward defers to here (ex4d_composite.hip): ln2 W/2, ln2 H/2 (CR/backward.cu:548-549,
        // :669-670) and -1/2 (:673-675)
        g_mean2D[0] = r0.x * (0.6931471805599453f * (0.5f * W)); g_mean2D[1] = r0.y * (0.6931471805599453f * (0.5f * H));
        g_mean2D[2] = r0.z;
        const float gA = -0.5f * r0.w, gB = -0.5f * r1.x, gC = -0.5f * r1.y;
        g_opacity = r1.z;
        g_color[0] = r1.w; g_color[1] = r2.x; g_color[2] = r2.y;
        g_dir[0] = r2.z; g_dir[1] = r2.w; g_dir[2] = r3.x;

        const float3 mean = make_float3(in_mean[0], in_mean[1], in_mean[2]);
        float cov3D[6];
#pragma unroll
        for (int i = 0; i < 6; i++) cov3D[i] = in_cov[i];
        if (scales) cov3d_from_scale_rotation(scale_modifier, in_s[0], in_s[1], in_s[2], in_q, cov3D);      // == the forward's value, not re-read

        // ---- computeCov2DCUDA, CR/backward.cu:144-300 (the coef-gradient block :201-218 affects no output)
        Cov2DCtx c;
        cov2d_common(mean, fx, fy, tanx, tany, cov3D, vm, c);
        const float a = c.cov.m[0][0] + kernel_size, b = c.cov.m[0][1], cc = c.cov.m[1][1] + kernel_size;
        const float denom = a * cc - b * b;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
#define TT(i, j) c.T.m[i][j]
        if (denom2inv != 0) {
            dL_da = denom2inv * (-cc * cc * gA + 2 * b * cc * gB + (denom - a * cc) * gC);
            dL_dc = denom2inv * (-a * a * gC + 2 * a * b * gB + (denom - a * cc) * gA);
            dL_db = denom2inv * 2 * (b * cc * gA - (denom + 2 * b * b) * gB + a * b * gC);
            g_cov[0] = (TT(0,0) * TT(0,0) * dL_da + TT(0,0) * TT(1,0) * dL_db + TT(1,0) * TT(1,0) * dL_dc);
            g_cov[3] = (TT(0,1) * TT(0,1) * dL_da + TT(0,1) * TT(1,1) * dL_db + TT(1,1) * TT(1,1) * dL_dc);
            g_cov[5] = (TT(0,2) * TT(0,2) * dL_da + TT(0,2) * TT(1,2) * dL_db + TT(1,2) * TT(1,2) * dL_dc);
            g_cov[1] = 2 * TT(0,0) * TT(0,1) * dL_da + (TT(0,0) * TT(1,1) + TT(0,1) * TT(1,0)) * dL_db + 2 * TT(1,0) * TT(1,1) * dL_dc;
            g_cov[2] = 2 * TT(0,0) * TT(0,2) * dL_da + (TT(0,0) * TT(1,2) + TT(0,2) * TT(1,0)) * dL_db + 2 * TT(1,0) * TT(1,2) * dL_dc;
            g_cov[4] = 2 * TT(0,2) * TT(0,1) * dL_da + (TT(0,1) * TT(1,2) + TT(0,2) * TT(1,1)) * dL_db + 2 * TT(1,1) * TT(1,2) * dL_dc;
        }
        // NOTE: the mean gradient this kernel stage produces (CR/backward.cu:261-299) is overwritten by the
        // assignment at CR/backward.cu:414, so it is not computed here at all.
#undef TT

        // ---- preprocessCUDA (bwd), CR/backward.cu:372-423
        const float4 m_hom = xform4x4(mean, pm);
        const float m_w = 1.0f / (m_hom.w + 0.0000001f);
        const float mul1 = (pm[0] * mean.x + pm[4] * mean.y + pm[8] * mean.z + pm[12]) * m_w * m_w;
        const float mul2 = (pm[1] * mean.x + pm[5] * mean.y + pm[9] * mean.z + pm[13]) * m_w * m_w;
        const float mul3 = (pm[2] * mean.x + pm[6] * mean.y + pm[10] * mean.z + pm[14]) * m_w * m_w;
        const float gx = g_mean2D[0], gy = g_mean2D[1], gz = g_mean2D[2];
        g_mean3D[0] = (pm[0] * m_w - pm[3] * mul1) * gx + (pm[1] * m_w - pm[3] * mul2) * gy + (pm[2] * m_w - pm[3] * mul3) * gz;
        g_mean3D[1] = (pm[4] * m_w - pm[7] * mul1) * gx + (pm[5] * m_w - pm[7] * mul2) * gy + (pm[6] * m_w - pm[7] * mul3) * gz;
        g_mean3D[2] = (pm[8] * m_w - pm[11] * mul1) * gx + (pm[9] * m_w - pm[11] * mul2) * gy + (pm[10] * m_w - pm[11] * mul3) * gz;

    }
    if (visible) {
        g_mean3D[0] += sh_j[0]; g_mean3D[1] += sh_j[1]; g_mean3D[2] += sh_j[2];
        if (scales) {
            // cov3D backward, CR/backward.cu:304-367 (gradient w.r.t. the raw quaternion)
            const float4 q = in_q;
            const float r = q.x, x = q.y, y = q.z, z = q.w;
            Mat3 R = rotation_from_quat(r, x, y, z);
            const float s[3] = { scale_modifier * in_s[0], scale_modifier * in_s[1], scale_modifier * in_s[2] };
            Mat3 S = from_columns(1.0f, 0.f, 0.f, 0.f, 1.0f, 0.f, 0.f, 0.f, 1.0f);
            S.m[0][0] = s[0]; S.m[1][1] = s[1]; S.m[2][2] = s[2];
            Mat3 Mx = mul(S, R);
            Mat3 dSigma = from_columns(
                g_cov[0], 0.5f * g_cov[1], 0.5f * g_cov[2],
                0.5f * g_cov[1], g_cov[3], 0.5f * g_cov[4],
                0.5f * g_cov[2], 0.5f * g_cov[4], g_cov[5]);
            Mat3 M2;
#pragma unroll
            for (int cI = 0; cI < 3; cI++)
#pragma unroll
                for (int rI = 0; rI < 3; rI++) M2.m[cI][rI] = Mx.m[cI][rI] * 2.0f;
            Mat3 dM = mul(M2, dSigma);
            Mat3 Rt = transpose(R);
            Mat3 dMt = transpose(dM);
#pragma unroll
            for (int k = 0; k < 3; k++)
                g_scale[k] = Rt.m[k][0] * dMt.m[k][0] + Rt.m[k][1] * dMt.m[k][1] + Rt.m[k][2] * dMt.m[k][2];
#pragma unroll
            for (int k = 0; k < 3; k++)
#pragma unroll
                for (int rI = 0; rI < 3; rI++) dMt.m[k][rI] *= s[k];
#define DM(i, j) dMt.m[i][j]
            g_rot[0] = 2 * z * (DM(0,1) - DM(1,0)) + 2 * y * (DM(2,0) - DM(0,2)) + 2 * x * (DM(1,2) - DM(2,1));
            g_rot[1] = 2 * y * (DM(1,0) + DM(0,1)) + 2 * z * (DM(2,0) + DM(0,2)) + 2 * r * (DM(1,2) - DM(2,1)) - 4 * x * (DM(2,2) + DM(1,1));
            g_rot[2] = 2 * x * (DM(1,0) + DM(0,1)) + 2 * r * (DM(2,0) - DM(0,2)) + 2 * z * (DM(1,2) + DM(2,1)) - 4 * y * (DM(2,2) + DM(0,0));
            g_rot[3] = 2 * r * (DM(0,1) - DM(1,0)) + 2 * x * (DM(2,0) + DM(0,2)) + 2 * y * (DM(1,2) + DM(2,1)) - 4 * z * (DM(1,1) + DM(0,0));
#undef DM
        }
    }
    // every output row written exactly once (replaces the ten torch::zeros of DGR/rasterize_points.cu:178-187)
    // dL_dsh rows, CR/backward.cu:45-131: basis(direction) x dL_dRGB -- formed here, not next to the sums above, so that the 48 values
    // are not live across the covariance backward
    float g_sh[16][3];
#pragma unroll
    for (int k = 0; k < 16; k++) g_sh[k][0] = g_sh[k][1] = g_sh[k][2] = 0.f;
    if (visible && (shs || split)) {
        const float x = sh_dir.x, y = sh_dir.y, z = sh_dir.z;
        const float *dRGB = sh_dRGB;
#define DSH(k, coefexpr) { const float c_ = (coefexpr); g_sh[k][0] = c_ * dRGB[0]; g_sh[k][1] = c_ * dRGB[1]; g_sh[k][2] = c_ * dRGB[2]; }
        DSH(0, kSH_C0)
        if (D > 0) {
            DSH(1, -kSH_C1 * y)
            DSH(2, kSH_C1 * z)
            DSH(3, -kSH_C1 * x)
            if (D > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                DSH(4, kSH_C2[0] * xy)
                DSH(5, kSH_C2[1] * yz)
                DSH(6, kSH_C2[2] * (2.f * zz - xx - yy))
                DSH(7, kSH_C2[3] * xz)
                DSH(8, kSH_C2[4] * (xx - yy))
                if (D > 2) {
                    DSH(9, kSH_C3[0] * y * (3.f * xx - yy))
                    DSH(10, kSH_C3[1] * xy * z)
                    DSH(11, kSH_C3[2] * y * (4.f * zz - xx - yy))
                    DSH(12, kSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy))
                    DSH(13, kSH_C3[4] * x * (4.f * zz - xx - yy))
                    DSH(14, kSH_C3[5] * z * (xx - yy))
                    DSH(15, kSH_C3[6] * x * (xx - 3.f * yy))
                }
            }
        }
#undef DSH
    }
    bool store_staged = prefetched;              // the gradient tensors need the same 16-byte alignment of the wave's spans as the inputs
    if (split && prefetched) {
        const int part = wave_first >= gsp.n_static ? 1 : 0;
        const size_t r0 = (size_t)(wave_first - (part ? gsp.n_static : 0));
        store_staged = ((((uintptr_t)(gsp.rest[part] + r0 * 45)) | ((uintptr_t)(gsp.dc[part] + r0 * 3))) & 15) == 0;
    }
    if (M == 16 && split && !store_staged) {
        // the one wave that straddles the static / dynamic boundary (or misaligned tensors): every lane stores its own row
        if (in_range) {
            const int part = idx >= gsp.n_static;
            const size_t row = (size_t)(idx - (part ? gsp.n_static : 0));
            float *dc = gsp.dc[part] + row * 3, *rest = gsp.rest[part] + row * 45;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) dc[ch] = g_sh[0][ch];
#pragma unroll
            for (int k = 1; k < 16; k++)
#pragma unroll
                for (int ch = 0; ch < 3; ch++) rest[3 * (k - 1) + ch] = g_sh[k][ch];
        }
    } else if (M == 16) {
        // dL_dsh of the wave = one contiguous 12 KB span: stage 32 rows at a time in LDS, store fully coalesced
#pragma unroll
        for (int h = 0; h < 2; h++) {
            wave_sync_lds();
            if ((lane >> 5) == h) {
                const int r = lane & 31;
                if (split) {
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) lds_row_base[SH_HALF_DC_OFFSET + r * 3 + ch] = g_sh[0][ch];
#pragma unroll
                    for (int k = 1; k < 16; k++)
#pragma unroll
                        for (int ch = 0; ch < 3; ch++) lds_row_base[r * 45 + 3 * (k - 1) + ch] = g_sh[k][ch];
                } else {
                    float4 *row = reinterpret_cast<float4 *>(lds_row_base + r * SH_ROW);
#pragma unroll
                    for (int v = 0; v < 12; v++) {
                        float t[4];
#pragma unroll
                        for (int e = 0; e < 4; e++) { const int f = 4 * v + e; t[e] = g_sh[f / 3][f % 3]; }
                        row[v] = make_float4(t[0], t[1], t[2], t[3]);
                    }
                }
            }
            const int first = wave_first + 32 * h;
            const int nrows_sh = (P - first) < 32 ? (P - first) : 32;       // <= 0: nothing
            if (split) wave_store_sh_split_half(gsp, first, lds_row_base, lane);
            else wave_store_sh_half(dL_dsh + (size_t)first * 48, lds_row_base, nrows_sh, lane);
        }
        wave_sync_lds();
    }
    {
        const int nrows = (P - wave_first) < 64 ? (P - wave_first) : 64;    // <= 0 for waves past the end
        wave_store_rows<3>(dL_dmeans2D + 3 * (size_t)wave_first, g_mean2D, lds_row_base, nrows, lane);
        if (dL_dcolors) wave_store_rows<3>(dL_dcolors + 3 * (size_t)wave_first, g_color, lds_row_base, nrows, lane);
        wave_store_rows<3>(dL_dmeans3D + 3 * (size_t)wave_first, g_mean3D, lds_row_base, nrows, lane);
        wave_store_rows<3>(dL_dscales + 3 * (size_t)wave_first, g_scale, lds_row_base, nrows, lane);
        wave_store_rows<3>(dL_ddir + 3 * (size_t)wave_first, g_dir, lds_row_base, nrows, lane);
        if (dL_dcov3D) wave_store_rows<6>(dL_dcov3D + 6 * (size_t)wave_first, g_cov, lds_row_base, nrows, lane);
    }
    if (!in_range) return;
    dL_dopacity[idx] = g_opacity;
    reinterpret_cast<float4 *>(dL_drotations)[idx] = make_float4(g_rot[0], g_rot[1], g_rot[2], g_rot[3]);
    if (M != 16) {
        // constant indices only: a runtime-indexed g_sh[][] would push the whole array into scratch memory
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (k < M) {
#pragma unroll
                for (int ch = 0; ch < 3; ch++) dL_dsh[((size_t)idx * M + k) * 3 + ch] = g_sh[k][ch];
            }
        for (int k = 16; k < M; k++)
            for (int ch = 0; ch < 3; ch++) dL_dsh[((size_t)idx * M + k) * 3 + ch] = 0.f;
    }
}

}  // namespace

// "preprocess_sh_predicate" (ex4d_set_option): the SH rows of frustum-culled Gaussians are not requested (default 1)
static int preprocess_option_default() { const char *e = getenv("EX4D_PREPROCESS_SH_PREDICATE"); return e ? (atoi(e) != 0) : 1; }    // developer override
static std::atomic<int> g_preprocess_fast{1};      // "preprocess_fast_path": the compile-time-specialised forward kernel for the common input combination
void ex4d_set_preprocess_fast(int v) { g_preprocess_fast.store(v); }
int ex4d_get_preprocess_fast() { return g_preprocess_fast.load(); }
static std::atomic<int> g_preprocess_tune{preprocess_option_default()};
void ex4d_set_preprocess_tune(int v) { g_preprocess_tune.store(v != 0); }
int ex4d_get_preprocess_tune() { return g_preprocess_tune.load(); }

hipError_t ex4d_launch_preprocess_fwd(const Ex4dParams &prm, const float *means3D, const float *dir3D, const float *scales,
    const float *rotations, const float *opacities, const float *shs, const float *cov3D_precomp,
    const float *colors_precomp, const float *viewmatrix, const float *projmatrix, const float *campos,
    int32_t *radii, GeomState g, uint32_t *prefilter_violation, ShSplit split,
    uint32_t *depth_keys, uint32_t *depth_vals, uint32_t depth_key_base, uint32_t depth_key_invisible, uint32_t *rects4, hipStream_t stream,
    uint32_t *key_range_slots, bool global_flags)
{
    const float fy = prm.H / (2.0f * prm.tanfovy);   // CR/rasterizer_impl.cu:237-238
    const float fx = prm.W / (2.0f * prm.tanfovx);
    const bool is_split = split.rest[0] != nullptr || split.rest[1] != nullptr;
    const bool fast = g_preprocess_fast.load(std::memory_order_relaxed) != 0 && shs != nullptr && !is_split && prm.D == 3 && prm.M == 16 &&
                      cov3D_precomp == nullptr && colors_precomp == nullptr && scales != nullptr && rotations != nullptr;
#define PF_ARGS prm.P, prm.D, prm.M, means3D, dir3D, scales, prm.scale_modifier, rotations, opacities, shs, cov3D_precomp, colors_precomp, \
        viewmatrix, projmatrix, campos, prm.W, prm.H, prm.tanfovx, prm.tanfovy, fx, fy, prm.kernel_size, \
        prm.min_depth, prm.max_depth, prm.prefiltered, prefilter_violation, \
        radii, g.records, g.cov3D, g.clamped, g.tiles_touched, g.rects, depth_keys, depth_vals, depth_key_base, depth_key_invisible, g.block_totals, split, \
        (prm.prepare_backward && (shs != nullptr || is_split)) ? g.sh_dsums : (float *)nullptr, \
        g_preprocess_tune.load(std::memory_order_relaxed), rects4, key_range_slots, global_flags ? 1 : 0
    if (fast) hipLaunchKernelGGL(preprocess_fwd_kernel<true>, dim3((prm.P + 255) / 256), dim3(256), 0, stream, PF_ARGS);
    else hipLaunchKernelGGL(preprocess_fwd_kernel<false>, dim3((prm.P + 255) / 256), dim3(256), 0, stream, PF_ARGS);
#undef PF_ARGS
    return hipGetLastError();
}

hipError_t ex4d_launch_mark_visible(int P, const float *means3D, const float *viewmatrix, const float *projmatrix,
    float min_depth, float max_depth, uint8_t *present, hipStream_t stream)
{
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, stream,
        P, means3D, viewmatrix, projmatrix, min_depth, max_depth, present);
    return hipGetLastError();
}

hipError_t ex4d_launch_preprocess_bwd(const Ex4dParams &prm, const float *means3D, const int32_t *radii,
    const float *shs, const float *scales, const float *rotations, const float *cov3D_ptr,
    const float *viewmatrix, const float *projmatrix, const float *campos, GeomState g, const float *acc16,
    float *dL_dmeans2D, float *dL_dcolors, float *dL_dopacity, float *dL_dmeans3D, float *dL_dcov3D, float *dL_dsh,
    float *dL_dscales, float *dL_drotations, float *dL_ddir, ShSplit split, ShSplitGrad gsplit, hipStream_t stream)
{
    const float fy = prm.H / (2.0f * prm.tanfovy);   // CR/rasterizer_impl.cu:417-418
    const float fx = prm.W / (2.0f * prm.tanfovx);
#define PB_ARGS prm.P, prm.D, prm.M, means3D, radii, shs, g.clamped, scales, rotations, prm.scale_modifier, cov3D_ptr, \
        viewmatrix, projmatrix, campos, prm.W, prm.H, fx, fy, prm.tanfovx, prm.tanfovy, prm.kernel_size, acc16, g.records, \
        dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, dL_ddir, split, gsplit
    const bool has_sh = shs != nullptr || split.rest[0] != nullptr || split.rest[1] != nullptr;
    if (prm.prepare_backward && has_sh)
        hipLaunchKernelGGL(preprocess_bwd_kernel<true>, dim3((prm.P + 255) / 256), dim3(256), 0, stream, PB_ARGS, (const float *)g.sh_dsums, (const uint32_t *)g.total);
    else
        hipLaunchKernelGGL(preprocess_bwd_kernel<false>, dim3((prm.P + 255) / 256), dim3(256), 0, stream, PB_ARGS, (const float *)nullptr, (const uint32_t *)g.total);
#undef PB_ARGS
    return hipGetLastError();
}
