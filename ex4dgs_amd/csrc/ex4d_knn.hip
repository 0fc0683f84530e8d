// distCUDA2 for gfx950: exact 3-nearest-neighbour mean squared distance, init-time helper of the reference
// (scene/c_gaussian_model.py:395 -> submodules/simple-knn/simple_knn.cu:185-221).
//
// Design: 30-bit Morton codes -> the rasterizer's stable radix sort -> points gathered into Morton order (so a wavefront's
// 64 queries are spatial neighbours) -> two-level bounding boxes (64-point leaves, 1024-point boxes) -> one wavefront per
// 64 consecutive queries walks the boxes; a box / leaf is opened when ANY lane still needs it (ballot), its 64 candidates
// are staged in LDS once and broadcast to all lanes.  Everything stays on the device (no host round trips; the reference
// does two blocking copies for the bounds).  ffp-contract is off: the box lower bound must never exceed a point distance.
#include "ex4d_internal.h"
#include "../../include/ex4d_knn.h"
#include <cfloat>
#include <cstdio>

namespace {

#define KNN_LEAF 64
#define KNN_BOX 1024
#define KNN_LEAVES_PER_BOX (KNN_BOX / KNN_LEAF)

struct Box { float lo[3], hi[3]; };

struct KnnScratch {
    float *bounds_partial;   // [1024][6]
    float *bounds;           // [6]
    uint32_t *keys_a, *vals_a, *keys_b, *vals_b, *hist;
    float4 *sorted;          // [P] xyz + original index bits
    Box *leaves;             // [ceil(P/64)]
    Box *boxes;              // [ceil(P/1024)]
};

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

size_t carve(KnnScratch &s, char *base, int P)
{
    size_t off = 0;
    auto take = [&](size_t bytes) { char *p = base ? base + off : nullptr; off += align256(bytes); return p; };
    const size_t n = (size_t)(P > 0 ? P : 1);
    s.bounds_partial = (float *)take(1024 * 6 * sizeof(float));
    s.bounds = (float *)take(6 * sizeof(float));
    s.keys_a = (uint32_t *)take(n * 4); s.vals_a = (uint32_t *)take(n * 4);
    s.keys_b = (uint32_t *)take(n * 4); s.vals_b = (uint32_t *)take(n * 4);
    s.hist = (uint32_t *)take(ex4d_radix_hist_words((uint32_t)n) * 4);
    s.sorted = (float4 *)take(n * sizeof(float4));
    s.leaves = (Box *)take(((n + KNN_LEAF - 1) / KNN_LEAF) * sizeof(Box));
    s.boxes = (Box *)take(((n + KNN_BOX - 1) / KNN_BOX) * sizeof(Box));
    return off;
}

__device__ __forceinline__ void wave_lds_sync()
{
    // LDS traffic of one wave is processed in order; only the compiler has to be kept from reordering
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float wave_min_f(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64)); return v; }
__device__ __forceinline__ float wave_max_f(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64)); return v; }

__global__ __launch_bounds__(256) void bounds_partial_kernel(int P, const float *__restrict__ pts, float *__restrict__ partial)
{
    __shared__ float s[6][4];
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256)
        for (int a = 0; a < 3; a++) { const float v = pts[3 * (size_t)i + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int a = 0; a < 3; a++) {
        const float l = wave_min_f(lo[a]), h = wave_max_f(hi[a]);
        if (lane == 0) { s[a][wave] = l; s[3 + a][wave] = h; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        float v = s[a][0];
        for (int w = 1; w < 4; w++) v = a < 3 ? fminf(v, s[a][w]) : fmaxf(v, s[a][w]);
        partial[6 * blockIdx.x + a] = v;
    }
}

__global__ __launch_bounds__(64) void bounds_final_kernel(int nblocks, const float *__restrict__ partial, float *__restrict__ bounds)
{
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int b = threadIdx.x; b < nblocks; b += 64)
        for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], partial[6 * b + a]); hi[a] = fmaxf(hi[a], partial[6 * b + 3 + a]); }
    for (int a = 0; a < 3; a++) {
        const float l = wave_min_f(lo[a]), h = wave_max_f(hi[a]);
        if (threadIdx.x == 0) { bounds[a] = l; bounds[3 + a] = h; }
    }
}

__device__ __forceinline__ uint32_t spread10(uint32_t x)      // 10 bits -> every third bit
{
    x &= 0x3ffu;
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

__global__ __launch_bounds__(256) void morton_kernel(int P, const float *__restrict__ pts, const float *__restrict__ bounds,
    uint32_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    uint32_t code = 0;
    for (int a = 0; a < 3; a++) {
        const float lo = bounds[a], ext = bounds[3 + a] - lo;
        float u = ext > 0.f ? (pts[3 * (size_t)i + a] - lo) / ext : 0.f;
        u = fminf(fmaxf(u, 0.f), 1.f);                       // also maps NaN coordinates to cell 0
        code |= spread10((uint32_t)(u * 1023.f)) << a;
    }
    keys[i] = code;
    vals[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void gather_kernel(int P, const float *__restrict__ pts, const uint32_t *__restrict__ order,
    float4 *__restrict__ sorted)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const uint32_t o = order[i];
    sorted[i] = make_float4(pts[3 * (size_t)o], pts[3 * (size_t)o + 1], pts[3 * (size_t)o + 2], __uint_as_float(o));
}

// one workgroup per 1024-point box: 16 waves, one leaf each
__global__ __launch_bounds__(KNN_BOX) void boxes_kernel(int P, const float4 *__restrict__ sorted, Box *__restrict__ leaves, Box *__restrict__ boxes)
{
    __shared__ float s[6][KNN_LEAVES_PER_BOX];
    const int i = blockIdx.x * KNN_BOX + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (i < P) { const float4 q = sorted[i]; lo[0] = hi[0] = q.x; lo[1] = hi[1] = q.y; lo[2] = hi[2] = q.z; }
    for (int a = 0; a < 3; a++) { lo[a] = wave_min_f(lo[a]); hi[a] = wave_max_f(hi[a]); }
    const int leaf = blockIdx.x * KNN_LEAVES_PER_BOX + wave;
    if (lane == 0) {
        if ((size_t)leaf * KNN_LEAF < (size_t)P) { Box b; for (int a = 0; a < 3; a++) { b.lo[a] = lo[a]; b.hi[a] = hi[a]; } leaves[leaf] = b; }
        for (int a = 0; a < 3; a++) { s[a][wave] = lo[a]; s[3 + a][wave] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        Box b;
        for (int a = 0; a < 3; a++) {
            float l = s[a][0], h = s[3 + a][0];
            for (int w = 1; w < KNN_LEAVES_PER_BOX; w++) { l = fminf(l, s[a][w]); h = fmaxf(h, s[3 + a][w]); }
            b.lo[a] = l; b.hi[a] = h;
        }
        boxes[blockIdx.x] = b;
    }
}

// lower bound of the squared distance from p to any point inside the box (simple_knn.cu:117-127 semantics)
__device__ __forceinline__ float box_dist2(const Box &b, float x, float y, float z)
{
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (x < b.lo[0] || x > b.hi[0]) dx = fminf(fabsf(x - b.lo[0]), fabsf(x - b.hi[0]));
    if (y < b.lo[1] || y > b.hi[1]) dy = fminf(fabsf(y - b.lo[1]), fabsf(y - b.hi[1]));
    if (z < b.lo[2] || z > b.hi[2]) dz = fminf(fabsf(z - b.lo[2]), fabsf(z - b.hi[2]));
    return dx * dx + dy * dy + dz * dz;
}

__device__ __forceinline__ void keep3(float d, float &b0, float &b1, float &b2)
{
    const float m0 = fmaxf(b0, d); b0 = fminf(b0, d);
    const float m1 = fmaxf(b1, m0); b1 = fminf(b1, m0);
    b2 = fminf(b2, m1);
}

__device__ __forceinline__ float dist2(float4 q, float cx, float cy, float cz)
{
    const float dx = cx - q.x, dy = cy - q.y, dz = cz - q.z;
    return dx * dx + dy * dy + dz * dz;
}

__global__ __launch_bounds__(256) void knn3_kernel(int P, const float4 *__restrict__ sorted, const Box *__restrict__ leaves,
    const Box *__restrict__ boxes, float *__restrict__ out)
{
    __shared__ float4 s_cand[4][KNN_LEAF];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wbase = (blockIdx.x * 4 + wave) * KNN_LEAF;
    if (wbase >= P) return;
    const int i = wbase + lane;
    const bool valid = i < P;
    const float4 q = sorted[valid ? i : P - 1];
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
    // a first bound from the Morton neighbours (simple_knn.cu:139-146)
    for (int off = -3; off <= 3; off++) {
        const int j = i + off;
        if (off == 0 || j < 0 || j >= P) continue;
        const float4 c = sorted[j];
        keep3(dist2(q, c.x, c.y, c.z), b0, b1, b2);
    }
    const float reject = b2;
    b0 = b1 = b2 = FLT_MAX;
    const int nboxes = (P + KNN_BOX - 1) / KNN_BOX, nleaves = (P + KNN_LEAF - 1) / KNN_LEAF;
    for (int b = 0; b < nboxes; b++) {
        const float d = box_dist2(boxes[b], q.x, q.y, q.z);
        if (__ballot(valid && !(d > reject || d > b2)) == 0) continue;
        const int l1 = min(nleaves, (b + 1) * KNN_LEAVES_PER_BOX);
        for (int leaf = b * KNN_LEAVES_PER_BOX; leaf < l1; leaf++) {
            const float dl = box_dist2(leaves[leaf], q.x, q.y, q.z);
            if (__ballot(valid && !(dl > reject || dl > b2)) == 0) continue;
            const int cbase = leaf * KNN_LEAF, cn = min(KNN_LEAF, P - cbase);
            s_cand[wave][lane] = sorted[min(cbase + lane, P - 1)];
            wave_lds_sync();
            for (int c = 0; c < cn; c++) {
                const float4 cp = s_cand[wave][c];
                float dd = dist2(q, cp.x, cp.y, cp.z);
                dd = (cbase + c == i) ? FLT_MAX : dd;                 // j != i by index (simple_knn.cu:168-169)
                keep3(dd, b0, b1, b2);
            }
            wave_lds_sync();
        }
    }
    if (valid) out[__float_as_uint(q.w)] = (b0 + b1 + b2) / 3.0f;
}

thread_local char g_knn_err[256] = "";

}  // namespace

extern "C" {

const char *ex4d_knn_last_error(void) { return g_knn_err; }

size_t ex4d_dist2_scratch_bytes(int32_t P)
{
    KnnScratch s;
    return carve(s, nullptr, P);
}

int ex4d_dist2(int32_t P, const float *points, float *mean_dist2, void *scratch, void *stream_)
{
    g_knn_err[0] = 0;
    if (P < 0 || (P > 0 && (!points || !mean_dist2 || !scratch))) { snprintf(g_knn_err, sizeof(g_knn_err), "bad argument"); return EX4D_ERR_ARG; }
    if (P == 0) return EX4D_OK;
    hipStream_t stream = (hipStream_t)stream_;
    KnnScratch s;
    carve(s, (char *)scratch, P);
    const int nb = (P + 255) / 256, nred = nb < 1024 ? nb : 1024;
    hipLaunchKernelGGL(bounds_partial_kernel, dim3(nred), dim3(256), 0, stream, P, points, s.bounds_partial);
    hipLaunchKernelGGL(bounds_final_kernel, dim3(1), dim3(64), 0, stream, nred, s.bounds_partial, s.bounds);
    hipLaunchKernelGGL(morton_kernel, dim3(nb), dim3(256), 0, stream, P, points, s.bounds, s.keys_a, s.vals_a);
    bool in_a = true;
    hipError_t e = ex4d_radix_sort_pairs(s.keys_a, s.vals_a, s.keys_b, s.vals_b, (uint32_t)P, 30, s.hist, &in_a, stream);
    if (e == hipSuccess) {
        const uint32_t *order = in_a ? s.vals_a : s.vals_b;
        hipLaunchKernelGGL(gather_kernel, dim3(nb), dim3(256), 0, stream, P, points, order, s.sorted);
        hipLaunchKernelGGL(boxes_kernel, dim3((P + KNN_BOX - 1) / KNN_BOX), dim3(KNN_BOX), 0, stream, P, s.sorted, s.leaves, s.boxes);
        hipLaunchKernelGGL(knn3_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, s.sorted, s.leaves, s.boxes, mean_dist2);
        e = hipGetLastError();
    }
    if (e != hipSuccess) { snprintf(g_knn_err, sizeof(g_knn_err), "launch failed: %s", hipGetErrorString(e)); return EX4D_ERR_HIP; }
    return EX4D_OK;
}

}  // extern "C"
