// Per-tile front-to-back alpha compositing (forward) and its backward for gfx950.
//
// Replaces (CR/ = submodules/diff_gaussian_rasterization_df/cuda_rasterizer/):
//   renderCUDA fwd  CR/forward.cu:274-462     renderCUDA bwd  CR/backward.cu:426-682
//
// MI355X design (not the reference's 256-thread cooperative block):
//   * a 16x16 tile is handled by 4 INDEPENDENT wave64s, one per 8x8 pixel quadrant -- no block barrier
//     anywhere, a quadrant stops as soon as ITS 64 pixels are done;
//   * each wave streams the tile's depth-sorted list 64 entries at a time: one coalesced load of ids, a
//     gather of the 24-byte (mean, conic, opacity) record, an exact conservative ellipse-vs-quadrant test
//     (can the Gaussian reach alpha >= 1/255 anywhere in this quadrant?), wave64 ballot + prefix popcount
//     compaction of the survivors into the wave's private LDS slice (order preserved, original list
//     position kept so n_contrib keeps the reference's meaning);
//   * the 64 lanes then walk the compacted list with wave-uniform LDS broadcasts; colour / depth / flow
//     attributes are staged in LDS too (the reference re-reads them from global per contributing pair);
//   * backward: same traversal in reverse, starting at the quadrant's deepest contributor instead of the
//     list end; the 13 per-Gaussian partials of the 64 pixels are combined with a wave64 reduce-scatter
//     butterfly (17 cross-lane adds instead of 13 x 64 float atomics) and leave the wave as ONE
//     16-lane atomic instruction onto a 64-byte accumulator row.
// Arithmetic follows the reference's expressions; FMA contraction is allowed here (results are compared
// to the oracle within 1e-5, not bit-exactly) and exp() is the hardware v_exp_f32.
#include "ex4d_internal.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

struct alignas(16) StageA { float x, y, depth; uint32_t id; };        // mean2D, depth, Gaussian id
struct alignas(16) StageC { float r, g, b; uint32_t orig; };           // colour, position in the tile list

__device__ __forceinline__ void wave_lds_sync()
{
    // LDS traffic of one wave is processed in order; only the compiler has to be kept from reordering
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { uint32_t t = __shfl_xor(v, o, 64); v = t > v ? t : v; }
    return v;
}

// Can this Gaussian reach alpha >= 1/255 at ANY sample position inside [bx0,bx1]x[by0,by1]?
// alpha = w*exp(-q(d)), q(d) = 0.5 (A dx^2 + C dy^2) + B dx dy  =>  needs  min_box q <= ln(255 w).
// Conservative (never culls a pair the per-pixel test would accept): slack of 1% in alpha plus a
// rounding allowance proportional to the magnitude of the terms; anything not provably convex is kept.
__device__ __forceinline__ bool quadrant_cull(float mx, float my, float4 co, float bx0, float bx1, float by0, float by1)
{
    const float A = co.x, B = co.y, C = co.z, w = co.w;
    if (w < (1.0f / 255.0f)) return true;                  // exp(power) <= 1 => alpha < 1/255 everywhere
    if (!(A > 0.f && C > 0.f && A * C - B * B > 0.f)) return false;
    const float dxc = fminf(fmaxf(mx, bx0), bx1) - mx;     // nearest box point - mean (0 if inside the slab)
    const float dyc = fminf(fmaxf(my, by0), by1) - my;
    if (dxc == 0.f && dyc == 0.f) return false;
    const float tau = __logf(255.0f * w) + 0.01f;
    float qmin = 3.0e38f, mag = 0.f;
    if (dxc != 0.f) {       // facing vertical edge: minimise over dy
        const float dy = fminf(fmaxf(-B * dxc / C, by0 - my), by1 - my);
        const float t1 = 0.5f * A * dxc * dxc, t2 = 0.5f * C * dy * dy, t3 = B * dxc * dy;
        const float q = t1 + t2 + t3;
        if (q < qmin) { qmin = q; mag = fabsf(t1) + fabsf(t2) + fabsf(t3); }
    }
    if (dyc != 0.f) {       // facing horizontal edge: minimise over dx
        const float dx = fminf(fmaxf(-B * dyc / A, bx0 - mx), bx1 - mx);
        const float t1 = 0.5f * A * dx * dx, t2 = 0.5f * C * dyc * dyc, t3 = B * dx * dyc;
        const float q = t1 + t2 + t3;
        if (q < qmin) { qmin = q; mag = fabsf(t1) + fabsf(t2) + fabsf(t3); }
    }
    return qmin > tau + 1e-5f * mag;
}

struct PixelGeom { int px, py; bool inside; float fx, fy; uint32_t pix_id; };

__device__ __forceinline__ PixelGeom pixel_of_lane(int tile, int gx, int W, int H, const float *__restrict__ subpixel_offset)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    PixelGeom p;
    p.px = (tile % gx) * EX4D_TILE + (wave & 1) * 8 + (lane & 7);
    p.py = (tile / gx) * EX4D_TILE + (wave >> 1) * 8 + (lane >> 3);
    p.inside = p.px < W && p.py < H;
    p.pix_id = (uint32_t)(W * p.py + p.px);
    p.fx = (float)p.px; p.fy = (float)p.py;
    if (p.inside && subpixel_offset) {
        const float2 o = reinterpret_cast<const float2 *>(subpixel_offset)[p.pix_id];
        p.fx += o.x; p.fy += o.y;
    }
    return p;
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void composite_fwd_kernel(
    int W, int H, int gx, int num_tiles,
    const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list,
    const float *__restrict__ subpixel_offset, const float2 *__restrict__ means2D,
    const float *__restrict__ features, const float4 *__restrict__ conic_opacity,
    const float *__restrict__ depths, const float *__restrict__ dir3D, const float *__restrict__ bg,
    float max_depth,
    float *__restrict__ final_T, uint32_t *__restrict__ n_contrib,
    float *__restrict__ out_color, float *__restrict__ out_depth, float *__restrict__ out_acc,
    float *__restrict__ out_flow, int32_t *__restrict__ out_idx)
{
    __shared__ StageA s_a[4][64];
    __shared__ float4 s_con[4][64];
    __shared__ StageC s_c[4][64];
    __shared__ float4 s_dir[4][64];

    const int tile = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const PixelGeom p = pixel_of_lane(tile, gx, W, H, subpixel_offset);
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    const float bx0 = wave_min(p.fx), bx1 = wave_max(p.fx), by0 = wave_min(p.fy), by1 = wave_max(p.fy);
    const uint64_t lt = (1ull << lane) - 1ull;

    bool done = !p.inside;
    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dm = 0.f, acc = 0.f, F0 = 0.f, F1 = 0.f, F2 = 0.f, max_vis = 0.f;
    uint32_t last_contributor = 0;
    int32_t best = -1;

    for (int base = 0; base < n; base += 64) {
        if (__ballot(!done) == 0) break;
        const int k = base + lane;
        bool keep = false;
        uint32_t id = 0; float2 xy = make_float2(0.f, 0.f); float4 co = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < n) {
            id = point_list[range.x + k];
            xy = means2D[id];
            co = conic_opacity[id];
            keep = !quadrant_cull(xy.x, xy.y, co, bx0, bx1, by0, by1);
        }
        const uint64_t mask = __ballot(keep);
        const int cnt = __popcll(mask);
        if (keep) {
            const int slot = __popcll(mask & lt);
            StageA a; a.x = xy.x; a.y = xy.y; a.depth = depths[id]; a.id = id;
            s_a[wave][slot] = a;
            s_con[wave][slot] = co;
            StageC c; c.r = features[3 * (size_t)id]; c.g = features[3 * (size_t)id + 1]; c.b = features[3 * (size_t)id + 2]; c.orig = (uint32_t)k;
            s_c[wave][slot] = c;
            float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
            if (dir3D) { d.x = dir3D[3 * (size_t)id]; d.y = dir3D[3 * (size_t)id + 1]; d.z = dir3D[3 * (size_t)id + 2]; }
            s_dir[wave][slot] = d;
        }
        wave_lds_sync();
        for (int j = 0; j < cnt; j++) {
            if (__ballot(!done) == 0) break;
            const StageA a = s_a[wave][j];
            const float4 con_o = s_con[wave][j];
            if (!done) {
                // CR/forward.cu:368-423
                const float dx = a.x - p.fx, dy = a.y - p.fy;
                const float power = -0.5f * (con_o.x * dx * dx + con_o.z * dy * dy) - con_o.y * dx * dy;
                if (power <= 0.0f) {
                    const float alpha = fminf(0.99f, con_o.w * __expf(power));
                    if (!(alpha < 1.0f / 255.0f)) {
                        const float test_T = T * (1 - alpha);
                        if (test_T < 0.0001f) {
                            done = true;
                        } else {
                            const StageC c = s_c[wave][j];
                            const float4 dir = s_dir[wave][j];
                            const float wgt = alpha * T;
                            C0 += c.r * wgt; C1 += c.g * wgt; C2 += c.b * wgt;
                            Dm += a.depth * wgt;
                            acc += wgt;
                            F0 += dir.x * wgt; F1 += dir.y * wgt; F2 += dir.z * wgt;
                            if (wgt > max_vis) { max_vis = wgt; best = (int32_t)a.id; }
                            T = test_T;
                            last_contributor = c.orig + 1;
                        }
                    }
                }
            }
        }
        wave_lds_sync();
    }
    if (p.inside) {
        // CR/forward.cu:426-460
        float Dout, Fx = F0, Fy = F1, Fz = F2;
        if (acc == 0.0f) { Dout = Dm + (1.0f - acc) * max_depth; }
        else { Dout = Dm / acc; Fx = F0 / acc; Fy = F1 / acc; Fz = F2 / acc; }
        const size_t HW = (size_t)H * W;
        final_T[p.pix_id] = T;
        n_contrib[p.pix_id] = last_contributor;
        out_color[p.pix_id] = C0 + T * bg[0];
        out_color[HW + p.pix_id] = C1 + T * bg[1];
        out_color[2 * HW + p.pix_id] = C2 + T * bg[2];
        out_depth[p.pix_id] = Dout;
        out_acc[p.pix_id] = acc;
        out_flow[p.pix_id] = Fx; out_flow[HW + p.pix_id] = Fy; out_flow[2 * HW + p.pix_id] = Fz;
        out_idx[p.pix_id] = best;
    }
}

// ------------------------------------------------------------------------------------------------
// wave64 reduce-scatter of 16 per-lane values: afterwards lane l (all four lanes of its quad) holds the
// wave-wide sum of value number  ((l>>5)&1)*8 + ((l>>4)&1)*4 + ((l>>3)&1)*2 + ((l>>2)&1).
__device__ __forceinline__ float reduce_scatter16(float (&v)[16], int lane)
{
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
    float w8[8], w4[4], w2[2];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const float send = b5 ? v[i] : v[i + 8];
        const float keep = b5 ? v[i + 8] : v[i];
        w8[i] = keep + __shfl_xor(send, 32, 64);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float send = b4 ? w8[i] : w8[i + 4];
        const float keep = b4 ? w8[i + 4] : w8[i];
        w4[i] = keep + __shfl_xor(send, 16, 64);
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const float send = b3 ? w4[i] : w4[i + 2];
        const float keep = b3 ? w4[i + 2] : w4[i];
        w2[i] = keep + __shfl_xor(send, 8, 64);
    }
    const float send = b2 ? w2[0] : w2[1];
    const float keep = b2 ? w2[1] : w2[0];
    float r = keep + __shfl_xor(send, 4, 64);
    r += __shfl_xor(r, 2, 64);
    r += __shfl_xor(r, 1, 64);
    return r;
}

__global__ __launch_bounds__(256) void composite_bwd_kernel(
    int W, int H, int gx,
    const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list,
    const float *__restrict__ subpixel_offset, const float *__restrict__ bg,
    const float2 *__restrict__ means2D, const float4 *__restrict__ conic_opacity,
    const float *__restrict__ colors, const float *__restrict__ depths,
    const float *__restrict__ depth_acc, const float *__restrict__ weight_acc, float min_depth,
    const float *__restrict__ final_Ts, const uint32_t *__restrict__ n_contrib,
    const float *__restrict__ dL_dpixels, const float *__restrict__ dL_ddepths,
    const float *__restrict__ dL_dflows, const float *__restrict__ dL_daccs,
    float *__restrict__ acc16)
{
    __shared__ StageA s_a[4][64];
    __shared__ float4 s_con[4][64];
    __shared__ StageC s_c[4][64];

    const int tile = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const PixelGeom p = pixel_of_lane(tile, gx, W, H, subpixel_offset);
    const uint2 range = ranges[tile];
    const size_t HW = (size_t)H * W;

    // CR/backward.cu:489-549
    const float T_final = p.inside ? final_Ts[p.pix_id] : 0.f;
    float T = T_final;
    const uint32_t last_contributor = p.inside ? n_contrib[p.pix_id] : 0u;
    const float acc = p.inside ? weight_acc[p.pix_id] : 0.f;
    const float final_depth = p.inside ? depth_acc[p.pix_id] : 0.f;
    float gdepth = 0.f, gflow0 = 0.f, gflow1 = 0.f, gflow2 = 0.f, gacc = 0.f, gp0 = 0.f, gp1 = 0.f, gp2 = 0.f;
    if (p.inside) {
        gdepth = dL_ddepths[p.pix_id];
        gp0 = dL_dpixels[p.pix_id]; gp1 = dL_dpixels[HW + p.pix_id]; gp2 = dL_dpixels[2 * HW + p.pix_id];
        if (acc > 0.0f) {
            gdepth /= acc;
            gflow0 = dL_dflows[p.pix_id] / acc; gflow1 = dL_dflows[HW + p.pix_id] / acc; gflow2 = dL_dflows[2 * HW + p.pix_id] / acc;
            gacc = dL_daccs[p.pix_id];
        }
    }
    const float bg_dot_dpixel = bg[0] * gp0 + bg[1] * gp1 + bg[2] * gp2;
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;
    float rec0 = 0.f, rec1 = 0.f, rec2 = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, last_alpha = 0.f;

    const uint32_t deepest = wave_max_u32(last_contributor);     // nothing behind it touches this quadrant
    if (deepest == 0) return;
    const float bx0 = wave_min(p.fx), bx1 = wave_max(p.fx), by0 = wave_min(p.fy), by1 = wave_max(p.fy);
    const uint64_t lt = (1ull << lane) - 1ull;

    for (int base = 0; base < (int)deepest; base += 64) {
        const int k = (int)deepest - 1 - base - lane;            // descending list position
        bool keep = false;
        uint32_t id = 0; float2 xy = make_float2(0.f, 0.f); float4 co = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k >= 0) {
            id = point_list[range.x + k];
            xy = means2D[id];
            co = conic_opacity[id];
            keep = !quadrant_cull(xy.x, xy.y, co, bx0, bx1, by0, by1);
        }
        const uint64_t mask = __ballot(keep);
        const int cnt = __popcll(mask);
        if (keep) {
            const int slot = __popcll(mask & lt);
            StageA a; a.x = xy.x; a.y = xy.y; a.depth = depths[id]; a.id = id;
            s_a[wave][slot] = a;
            s_con[wave][slot] = co;
            StageC c; c.r = colors[3 * (size_t)id]; c.g = colors[3 * (size_t)id + 1]; c.b = colors[3 * (size_t)id + 2]; c.orig = (uint32_t)k;
            s_c[wave][slot] = c;
        }
        wave_lds_sync();
        for (int j = 0; j < cnt; j++) {
            const StageA a = s_a[wave][j];
            const float4 con_o = s_con[wave][j];
            const StageC c = s_c[wave][j];
            // CR/backward.cu:575-590
            bool contributes = p.inside && (c.orig < last_contributor);
            const float dx = a.x - p.fx, dy = a.y - p.fy;
            const float power = -0.5f * (con_o.x * dx * dx + con_o.z * dy * dy) - con_o.y * dx * dy;
            const float G = __expf(power);
            const float alpha = fminf(0.99f, con_o.w * G);
            contributes = contributes && (power <= 0.0f) && !(alpha < 1.0f / 255.0f);
            if (__ballot(contributes) == 0) continue;

            float v[16];
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = 0.f;
            if (contributes) {
                // CR/backward.cu:592-679
                T = T * __builtin_amdgcn_rcpf(1.f - alpha);
                const float dchannel_dcolor = alpha * T;
                float dL_dalpha = 0.0f;
                if ((a.depth > min_depth) & (dchannel_dcolor > 0.0f)) {
                    v[2] = gdepth * dchannel_dcolor;
                    dL_dalpha += (final_depth - a.depth) * gdepth * T;
                }
                rec0 = last_alpha * lc0 + (1.f - last_alpha) * rec0; lc0 = c.r;
                rec1 = last_alpha * lc1 + (1.f - last_alpha) * rec1; lc1 = c.g;
                rec2 = last_alpha * lc2 + (1.f - last_alpha) * rec2; lc2 = c.b;
                dL_dalpha += (c.r - rec0) * gp0;
                dL_dalpha += (c.g - rec1) * gp1;
                dL_dalpha += (c.b - rec2) * gp2;
                v[7] = dchannel_dcolor * gp0; v[8] = dchannel_dcolor * gp1; v[9] = dchannel_dcolor * gp2;
                v[10] = dchannel_dcolor * gflow0; v[11] = dchannel_dcolor * gflow1; v[12] = dchannel_dcolor * gflow2;
                dL_dalpha *= T;
                gacc *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final * __builtin_amdgcn_rcpf(1.f - alpha)) * bg_dot_dpixel;
                const float dL_dG = con_o.w * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * con_o.x - gdy * con_o.y;
                const float dG_ddely = -gdy * con_o.z - gdx * con_o.y;
                v[0] = dL_dG * dG_ddelx * ddelx_dx;
                v[1] = dL_dG * dG_ddely * ddely_dy;
                v[3] = -0.5f * gdx * dx * dL_dG;
                v[4] = -0.5f * gdx * dy * dL_dG;
                v[5] = -0.5f * gdy * dy * dL_dG;
                v[6] = G * dL_dalpha + G * gacc;
            }
            const float r = reduce_scatter16(v, lane);
            if ((lane & 3) == 0) {
                const int which = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
                if (which < 13) unsafeAtomicAdd(&acc16[16 * (size_t)a.id + which], r);
            }
        }
        wave_lds_sync();
    }
}

}  // namespace

hipError_t ex4d_launch_composite_fwd(const Ex4dParams &prm, const uint2 *ranges, const uint32_t *point_list,
    const float *subpixel_offset, const float2 *means2D, const float *features, const float4 *conic_opacity,
    const float *depths, const float *dir3D, const float *bg, float *final_T, uint32_t *n_contrib,
    float *out_color, float *out_depth, float *out_acc, float *out_flow, int32_t *out_idx, hipStream_t stream)
{
    const int gx = (prm.W + EX4D_TILE - 1) / EX4D_TILE, gy = (prm.H + EX4D_TILE - 1) / EX4D_TILE;
    hipLaunchKernelGGL(composite_fwd_kernel, dim3(gx * gy), dim3(256), 0, stream,
        prm.W, prm.H, gx, gx * gy, ranges, point_list, subpixel_offset, means2D, features, conic_opacity, depths, dir3D, bg,
        prm.max_depth, final_T, n_contrib, out_color, out_depth, out_acc, out_flow, out_idx);
    return hipGetLastError();
}

hipError_t ex4d_launch_composite_bwd(const Ex4dParams &prm, const uint2 *ranges, const uint32_t *point_list,
    const float *subpixel_offset, const float *bg, const float2 *means2D, const float4 *conic_opacity,
    const float *colors, const float *depths, const float *out_depth, const float *out_acc,
    const float *final_T, const uint32_t *n_contrib, const float *dL_dpix, const float *dL_ddepth,
    const float *dL_dflow, const float *dL_dacc, float *acc16, hipStream_t stream)
{
    const int gx = (prm.W + EX4D_TILE - 1) / EX4D_TILE, gy = (prm.H + EX4D_TILE - 1) / EX4D_TILE;
    hipLaunchKernelGGL(composite_bwd_kernel, dim3(gx * gy), dim3(256), 0, stream,
        prm.W, prm.H, gx, ranges, point_list, subpixel_offset, bg, means2D, conic_opacity, colors, depths,
        out_depth, out_acc, prm.min_depth, final_T, n_contrib, dL_dpix, dL_ddepth, dL_dflow, dL_dacc, acc16);
    return hipGetLastError();
}
