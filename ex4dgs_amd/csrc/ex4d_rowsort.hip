// Tile sort at ROW-SEGMENT granularity for gfx950 (round 6): the tile lists without ever materialising (tile, id) pairs.
//
// Replaces, together with the depth sort of ex4d_binning.hip (CR/ = submodules/diff_gaussian_rasterization_df/cuda_rasterizer/):
//   duplicateWithKeys                  CR/rasterizer_impl.cu:72-113
//   cub::DeviceRadixSort::SortPairs    CR/rasterizer_impl.cu:321-326
//   identifyTileRanges (+ cudaMemset)  CR/rasterizer_impl.cu:118-140, :328
//
// Rounds 2-5 emitted one (tile, id) pair per instance in depth order (duplicate_kernel: 8 bytes x R written) and sorted the R pairs by
// tile id in two passes (pass A by the high digit, pass B by the low digit): both passes rank EVERY instance with wave ballots, 9.3
// instances per Gaussian at BASELINE config 3, 38 at config 5.  A Gaussian's tile rect is w x h tiles: h ROW SEGMENTS of w consecutive
// tiles.  The same stable order falls out of
//   pass A'  stable partition of the S = sum(h) row segments (id, x0, w, row), generated on the fly from the rects in depth order, by
//            their tile ROW (<= 255 rows): histogram -> row scan -> scatter, ranking 2.7 segments per Gaussian instead of 9.3
//            instances, writing 8 S bytes instead of 8 R + 4 R;
//   pass B'  per tile row, stable counting sort of the segments' instances by tile COLUMN: every workgroup expands its <= 1024
//            segments in LDS, ranks the instances (wave ballots, like the scatter kernels of ex4d_binning.hip) and writes the ids;
//            the (row, column) counts are the tile ranges.
// Stable by row, then stable by column inside a row, both in depth order = the reference's stable order by (tile | depth).
// No instance offsets are needed (no tile scan), no key / value arrays, no packed words.  Histograms are difference arrays:
// a segment [x0, x0 + w) adds +1 at x0 and -1 at x0 + w (two LDS atomics instead of w).
// Applies when the image has at most 255 x 255 tiles (the rects travel packed) and P <= 2^24 (the staged word is column << 24 | id).
#include "ex4d_internal.h"

namespace {

#define ROW_THREADS 256
#define ROWA_GAUSS 512            // Gaussians per workgroup of pass A'
#define ROWA_CAP 4096             // segments a workgroup of pass A' stages in LDS (32 KB); a block with more writes them unstaged
#define ROWB_CAP 8192             // instances a workgroup of pass B' stages in LDS (32 KB)
#define ROW_PAD 260               // counters per wave: 256 rows / columns + the end mark of a segment at 255 + padding to 16 bytes

__device__ __forceinline__ void wsync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct RectU { uint32_t x0, y0, w, h; };
// rect of the k-th Gaussian in depth order: packed (x0 | y0 << 8 | w << 16 | h << 24) or the 8-byte form of the LSD depth-sort path
__device__ __forceinline__ RectU rect_at(const uint32_t *__restrict__ r4, const uint2 *__restrict__ r8, uint32_t k)
{
    if (r4) { const uint32_t p = r4[k]; return { p & 0xFFu, (p >> 8) & 0xFFu, (p >> 16) & 0xFFu, p >> 24 }; }
    const uint2 rc = r8[k];
    return { rc.x & 0xFFFFu, rc.x >> 16, rc.y & 0xFFFFu, rc.y >> 16 };
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
    return x;
}
// exclusive scan over the 256 threads of a workgroup (tmp: 4 words; two barriers, tmp is free afterwards); total = sum over all
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *tmp, uint32_t &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t x = wave_incl_scan(v, lane);
    if (lane == 63) tmp[wave] = x;
    __syncthreads();
    uint32_t excl = x - v;
    total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { const uint32_t t = tmp[w]; if (w < wave) excl += t; total += t; }
    __syncthreads();
    return excl;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_max_u32(uint32_t x)
{
    const uint32_t y = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, ROW_MASK, 0xf, false);
    return y > x ? y : x;
}
__device__ __forceinline__ uint32_t wave_inclusive_max_u32(uint32_t x)
{
    x = dpp_max_u32<0x111, 0xf>(x);       // row_shr:1
    x = dpp_max_u32<0x112, 0xf>(x);       // row_shr:2
    x = dpp_max_u32<0x114, 0xf>(x);       // row_shr:4
    x = dpp_max_u32<0x118, 0xf>(x);       // row_shr:8
    x = dpp_max_u32<0x142, 0xa>(x);       // row_bcast:15 -> rows 1 and 3
    x = dpp_max_u32<0x143, 0xc>(x);       // row_bcast:31 -> rows 2 and 3
    return x;
}

// lanes of the wave holding the same digit as this lane, among the valid ones (see rs_scatter_kernel, ex4d_binning.hip: per bit keep
// the ballot if my bit is set, its complement otherwise; 5 VALU per bit with the width known at compile time)
template <int NBITS>
__device__ __forceinline__ uint64_t same_digit(uint32_t d, int nbits, bool valid)
{
    const uint64_t vmask = __builtin_amdgcn_ballot_w64(valid);
    uint32_t plo = (uint32_t)vmask, phi = (uint32_t)(vmask >> 32);
    if (NBITS > 0) {
#pragma unroll
        for (int b = 0; b < NBITS; b++) {
            const uint64_t bal = __builtin_amdgcn_ballot_w64((d & (1u << b)) != 0u);
            uint32_t flip;
            asm("v_cndmask_b32_e64 %0, -1, 0, %1" : "=v"(flip) : "s"(bal));
            plo &= (uint32_t)bal ^ flip;
            phi &= (uint32_t)(bal >> 32) ^ flip;
        }
    } else {
        for (int b = 0; b < nbits; b++) {
            const uint32_t bit = (d >> b) & 1u;
            const uint64_t bal = __builtin_amdgcn_ballot_w64(bit != 0u);
            const uint32_t flip = bit - 1u;
            plo &= (uint32_t)bal ^ flip;
            phi &= (uint32_t)(bal >> 32) ^ flip;
        }
    }
    return ((uint64_t)phi << 32) | plo;
}
// stable slot of this lane's item: the wave's running counter of its digit + its rank among the wave's lanes with that digit
template <int NBITS>
__device__ __forceinline__ uint32_t ranked_slot(uint32_t *cnt, uint32_t d, int nbits, bool valid, int lane)
{
    const uint64_t peers = same_digit<NBITS>(d, nbits, valid);
    const uint32_t rank = __popcll(peers & ((1ull << lane) - 1ull));
    const uint32_t before = cnt[valid ? d : 0u];                            // the peers of a digit read one word (LDS broadcast)
    if (valid && rank == 0) cnt[d] = before + (uint32_t)__popcll(peers);    // one lane per distinct digit writes
    wsync();
    return before + rank;
}

// a wave's difference array (cnt[d] += 1 at the first digit of an item, -= 1 behind its last) -> counts per digit, in place.
// Lane l owns digits 4 l .. 4 l + 3.
__device__ __forceinline__ void wave_diff_to_counts(uint32_t *cnt, int lane)
{
    uint4 v = *reinterpret_cast<uint4 *>(cnt + 4 * lane);
    v.y += v.x; v.z += v.y; v.w += v.z;
    const uint32_t incl = wave_incl_scan(v.w, lane), excl = incl - v.w;
    *reinterpret_cast<uint4 *>(cnt + 4 * lane) = make_uint4(v.x + excl, v.y + excl, v.z + excl, v.w + excl);
}

// ---------------------------------------------------------------- pass A': histogram
// hist: rows 0 .. NR-1 = segments per (tile row, block), rows NR .. 2 NR - 1 = instances per (tile row, block); [2 NR][nblocks] + totals
__global__ __launch_bounds__(ROW_THREADS) void rows_seg_hist_kernel(uint32_t P, int NR, const uint32_t *__restrict__ r4, const uint2 *__restrict__ r8,
    uint32_t *__restrict__ hist, uint32_t nblocks, uint32_t *__restrict__ frame_total, uint32_t *__restrict__ nb_dev)
{
    __shared__ uint32_t d_seg[ROW_PAD], d_inst[ROW_PAD];
    __shared__ uint32_t tmp[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < ROW_PAD; i += ROW_THREADS) { d_seg[i] = 0u; d_inst[i] = 0u; }
    if (blockIdx.x == 0 && tid == 0) *nb_dev = 0u;           // (the row scan behind this kernel adds the block counts of pass B')
    __syncthreads();
    uint32_t inst = 0;
#pragma unroll
    for (int it = 0; it < ROWA_GAUSS / ROW_THREADS; it++) {
        const uint32_t k = blockIdx.x * ROWA_GAUSS + it * ROW_THREADS + tid;
        if (k < P) {
            const RectU r = rect_at(r4, r8, k);
            if (r.w * r.h != 0u) {
                atomicAdd(&d_seg[r.y0], 1u); atomicAdd(&d_seg[r.y0 + r.h], 0xFFFFFFFFu);
                atomicAdd(&d_inst[r.y0], r.w); atomicAdd(&d_inst[r.y0 + r.h], 0u - r.w);
                inst += r.w * r.h;
            }
        }
    }
    __syncthreads();
    const uint32_t a = d_seg[tid], b = d_inst[tid];
    const uint32_t xa = wave_incl_scan(a, lane), xb = wave_incl_scan(b, lane);
    if (lane == 63) { tmp[wave] = xa; tmp[4 + wave] = xb; }
    __syncthreads();
    uint32_t sa = xa, sb = xb;
    for (int w = 0; w < wave; w++) { sa += tmp[w]; sb += tmp[4 + w]; }
    if (tid < NR) { hist[(size_t)tid * nblocks + blockIdx.x] = sa; hist[(size_t)(NR + tid) * nblocks + blockIdx.x] = sb; }
    if (frame_total) {
        // device-side instance count of the asynchronous forward (one fire-and-forget atomic per workgroup)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) inst += __shfl_xor(inst, o, 64);
        __syncthreads();
        if (lane == 0) tmp[wave] = inst;
        __syncthreads();
        if (tid == 0) atomicAdd(frame_total, tmp[0] + tmp[1] + tmp[2] + tmp[3]);
    }
}

// one workgroup per histogram row: exclusive scan of the first `len` per-block counts (rows are `stride` words apart), row total to
// hist[nrows * stride + row].  chunk != 0: the row's total is a segment count -- add the number of pass-B' blocks it is cut into to
// *nb_dev (rows < nr_seg only).  len_dev != nullptr: scan *len_dev + 1 entries (the blocks pass B' really has, and the one behind them).
__global__ __launch_bounds__(256) void rows_scan_kernel(uint32_t stride, uint32_t len, uint32_t *__restrict__ hist, uint32_t nrows,
    uint32_t chunk, uint32_t nr_seg, uint32_t *__restrict__ nb_dev, const uint32_t *__restrict__ len_dev)
{
    __shared__ uint32_t wave_sums[4];
    __shared__ uint32_t carry_s;
    if (len_dev) { const uint32_t l = *len_dev + 1u; len = l < len ? l : len; }
    uint32_t *row = hist + (size_t)blockIdx.x * stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool vec = ((((uintptr_t)row) & 15) == 0);
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < len; base += 2048) {
        const uint32_t i0 = base + 8 * threadIdx.x;
        uint32_t v[8];
        if (vec && i0 + 8 <= len) {
            const uint4 a = *reinterpret_cast<const uint4 *>(row + i0), b = *reinterpret_cast<const uint4 *>(row + i0 + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = (i0 + k < len) ? row[i0 + k] : 0u;
        }
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { const uint32_t t = v[k]; v[k] = run; run += t; }
        const uint32_t x = wave_incl_scan(run, lane);
        if (lane == 63) wave_sums[wave] = x;
        __syncthreads();
        uint32_t woff = carry_s + x - run;
        for (int w = 0; w < wave; w++) woff += wave_sums[w];
        if (vec && i0 + 8 <= len) {
            *reinterpret_cast<uint4 *>(row + i0) = make_uint4(woff + v[0], woff + v[1], woff + v[2], woff + v[3]);
            *reinterpret_cast<uint4 *>(row + i0 + 4) = make_uint4(woff + v[4], woff + v[5], woff + v[6], woff + v[7]);
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) if (i0 + k < len) row[i0 + k] = woff + v[k];
        }
        __syncthreads();
        if (threadIdx.x == 255) carry_s = woff + run;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        hist[(size_t)nrows * stride + blockIdx.x] = carry_s;
        if (chunk != 0u && blockIdx.x < nr_seg) atomicAdd(nb_dev, (carry_s + chunk - 1u) / chunk);
    }
}

// ---------------------------------------------------------------- pass A': scatter
// Workgroup b owns Gaussians [b ROWA_GAUSS, (b + 1) ROWA_GAUSS) of the depth order, wave w the w-th quarter; the stable order inside
// the block is (wave, Gaussian, row).  Sweep 1 counts the wave's segments per row (difference array), one barrier gives every
// (row, wave) its first block-local slot and every row its global position, sweep 2 generates the segments 64 at a time -- owner of
// output slot t by a max-scan over "first slot" marks, like duplicate_kernel -- ranks them by row and stages them in block-local
// sorted order; the block then streams the staged segments out (runs of one row are contiguous).
// segment = (Gaussian id, x0 | w << 8 | row << 16)
template <int NBITS>
__global__ __launch_bounds__(ROW_THREADS) void rows_seg_scatter_kernel(uint32_t P, int NR, int nbits, const uint32_t *__restrict__ order,
    const uint32_t *__restrict__ r4, const uint2 *__restrict__ r8, const uint32_t *__restrict__ hist, uint32_t nblocks,
    uint2 *__restrict__ segs, uint32_t seg_cap)
{
    __shared__ __attribute__((aligned(16))) uint32_t wave_cnt[4][ROW_PAD];
    __shared__ uint32_t local_start[256], global_base[256];
    __shared__ uint32_t tmp[8];
    __shared__ uint32_t s_total;
    __shared__ uint2 stage[ROWA_CAP];
    __shared__ uint32_t s_mark[4][64];
    __shared__ uint4 s_rec[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int ROUNDS = ROWA_GAUSS / ROW_THREADS;
    const uint32_t first = blockIdx.x * ROWA_GAUSS + wave * (ROWA_GAUSS / 4);
    RectU rc[ROUNDS];
    uint32_t id[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        const uint32_t k = first + r * 64 + lane;
        rc[r] = { 0u, 0u, 0u, 0u }; id[r] = 0u;
        if (k < P) { rc[r] = rect_at(r4, r8, k); id[r] = order[k]; }
        if (rc[r].w * rc[r].h == 0u) rc[r].h = 0u;
    }
    uint32_t *cnt = wave_cnt[wave];
    for (int i = lane; i < ROW_PAD; i += 64) cnt[i] = 0u;
    wsync();
#pragma unroll
    for (int r = 0; r < ROUNDS; r++)
        if (rc[r].h != 0u) { atomicAdd(&cnt[rc[r].y0], 1u); atomicAdd(&cnt[rc[r].y0 + rc[r].h], 0xFFFFFFFFu); }
    wsync();
    wave_diff_to_counts(cnt, lane);
    __syncthreads();
    {
        // thread t = tile row t: exclusive scan over the waves, block-local start of the row, global position of the block's first segment of the row
        uint32_t c[4], tot = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) { c[w] = tid < NR ? wave_cnt[w][tid] : 0u; tot += c[w]; }
        const uint32_t gt = tid < NR ? hist[(size_t)2 * NR * nblocks + tid] : 0u;
        uint32_t total, gtotal;
        const uint32_t ls = block_excl_scan(tot, tmp, total);
        const uint32_t gs = block_excl_scan(gt, tmp + 4, gtotal);
        if (tid < NR) {
            local_start[tid] = ls;
            global_base[tid] = gs + hist[(size_t)tid * nblocks + blockIdx.x];
            wave_cnt[0][tid] = ls; wave_cnt[1][tid] = ls + c[0]; wave_cnt[2][tid] = ls + c[0] + c[1]; wave_cnt[3][tid] = ls + c[0] + c[1] + c[2];
        }
        if (tid == 0) s_total = total;
    }
    __syncthreads();
    const uint32_t total = s_total;
    const bool staged = total <= ROWA_CAP;
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        const uint32_t h = rc[r].h;
        const uint32_t incl = wave_incl_scan(h, lane), off = incl - h;
        const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        s_rec[wave][lane] = make_uint4(off, id[r], rc[r].x0 | (rc[r].w << 8), rc[r].y0);
        uint32_t carry = 0;
        for (uint32_t tb = 0; tb < n; tb += 64) {           // wave-uniform trip count
            s_mark[wave][lane] = 0u;
            wsync();
            if (h != 0u && off - tb < 64u) s_mark[wave][off - tb] = (uint32_t)lane + 1u;
            wsync();
            uint32_t owner = wave_inclusive_max_u32(s_mark[wave][lane]);
            owner = owner > carry ? owner : carry;
            carry = (uint32_t)__builtin_amdgcn_readlane((int)owner, 63);
            const uint32_t t = tb + lane;
            const bool valid = t < n;
            const uint4 rec = s_rec[wave][valid ? owner - 1u : 0u];
            const uint32_t ty = valid ? rec.w + (t - rec.x) : 0u;
            const uint32_t slot = ranked_slot<NBITS>(cnt, ty, nbits, valid, lane);
            if (valid) {
                const uint2 seg = make_uint2(rec.y, rec.z | (ty << 16));
                if (staged) stage[slot] = seg;
                else {
                    const uint32_t dst = global_base[ty] + (slot - local_start[ty]);
                    if (dst < seg_cap) segs[dst] = seg;
                }
            }
        }
        wsync();
    }
    if (!staged) return;            // (uniform)
    __syncthreads();
    for (uint32_t p = tid; p < total; p += ROW_THREADS) {
        const uint2 seg = stage[p];
        const uint32_t ty = seg.y >> 16;
        const uint32_t dst = global_base[ty] + (p - local_start[ty]);
        if (dst < seg_cap) segs[dst] = seg;
    }
}

// ---------------------------------------------------------------- pass B'
// Every tile row (= bucket of pass A') is cut into blocks of <= CHUNK segments, no block straddles two rows.  Each workgroup derives
// its block from the <= 255 row totals itself (two scans + an 8-step search; see ts_locate_block, ex4d_binning.hip).
struct RowBlock { uint32_t start, count, bucket, fb_first, fb_next; };
struct RowLocateLds { uint32_t cnt[257], fb[257], st[257], wave_sums[8]; };
__device__ __forceinline__ RowBlock row_locate_block(const uint32_t *__restrict__ totals, int nbuckets, uint32_t b, uint32_t chunk, RowLocateLds &L)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t cnt = tid < nbuckets ? totals[tid] : 0u;
    const uint32_t nblk = (cnt + chunk - 1) / chunk;
    const uint32_t x = wave_incl_scan(cnt, lane), y = wave_incl_scan(nblk, lane);
    if (lane == 63) { L.wave_sums[wave] = x; L.wave_sums[4 + wave] = y; }
    __syncthreads();
    uint32_t st = x - cnt, fb = y - nblk;
    for (int w = 0; w < wave; w++) { st += L.wave_sums[w]; fb += L.wave_sums[4 + w]; }
    L.cnt[tid] = cnt; L.fb[tid] = fb; L.st[tid] = st;
    if (tid == 255) { L.fb[256] = fb + nblk; L.st[256] = st + cnt; L.cnt[256] = 0; }
    __syncthreads();
    RowBlock t = { 0u, 0u, 0u, 0u, 0u };
    if (b < L.fb[256]) {
        uint32_t h = 0;
#pragma unroll
        for (uint32_t step = 128; step > 0; step >>= 1) if (L.fb[h + step] <= b) h += step;
        const uint32_t within = (b - L.fb[h]) * chunk;
        t.start = L.st[h] + within;
        t.count = L.cnt[h] - within < chunk ? L.cnt[h] - within : chunk;
        t.bucket = h; t.fb_first = L.fb[h]; t.fb_next = L.fb[h + 1];
    }
    return t;
}

// instances per (tile column, block): [NX][nblocks]; workgroups behind the last block write a zero column (the one right behind is read
// as "end of the last row"), workgroups further out leave at once
template <int ITEMS>
__global__ __launch_bounds__(ROW_THREADS) void rows_inst_hist_kernel(const uint2 *__restrict__ segs, uint32_t seg_cap,
    const uint32_t *__restrict__ seg_totals, int NR, int NX, uint32_t *__restrict__ hist, uint32_t nblocks, const uint32_t *__restrict__ nb_dev)
{
    if (blockIdx.x > *nb_dev) return;
    __shared__ uint32_t d[ROW_PAD];
    __shared__ RowLocateLds loc;
    __shared__ uint32_t tmp[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < ROW_PAD; i += ROW_THREADS) d[i] = 0u;
    const RowBlock tb = row_locate_block(seg_totals, NR, blockIdx.x, ROW_THREADS * ITEMS, loc);       // (its barriers publish d = 0)
    if (tb.count != 0u) {
        uint2 s[ITEMS];
#pragma unroll
        for (int it = 0; it < ITEMS; it++) {
            const uint32_t i = it * ROW_THREADS + tid, g = tb.start + i;
            s[it] = (i < tb.count && g < seg_cap) ? segs[g] : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int it = 0; it < ITEMS; it++) {
            const uint32_t x0 = s[it].y & 0xFFu, w = (s[it].y >> 8) & 0xFFu;
            if (w != 0u) { atomicAdd(&d[x0], 1u); atomicAdd(&d[x0 + w], 0xFFFFFFFFu); }
        }
    }
    __syncthreads();
    const uint32_t a = d[tid];
    const uint32_t x = wave_incl_scan(a, lane);
    if (lane == 63) tmp[wave] = x;
    __syncthreads();
    uint32_t s = x;
    for (int w = 0; w < wave; w++) s += tmp[w];
    if (tid < NX) hist[(size_t)tid * nblocks + blockIdx.x] = s;
}

// stage word = column << 24 | Gaussian id
template <int ITEMS, int NBITS>
__global__ __launch_bounds__(ROW_THREADS) void rows_inst_scatter_kernel(const uint2 *__restrict__ segs, uint32_t seg_cap,
    const uint32_t *__restrict__ seg_totals, const uint32_t *__restrict__ inst_totals, int NR, int NX, int nbits,
    const uint32_t *__restrict__ hist, uint32_t nblocks, const uint32_t *__restrict__ nb_dev,
    uint32_t *__restrict__ point_list, uint32_t cap, uint2 *__restrict__ ranges, uint32_t *__restrict__ tile_ids_out)
{
    if (blockIdx.x >= *nb_dev) return;
    __shared__ __attribute__((aligned(16))) uint32_t wave_cnt[4][ROW_PAD];
    __shared__ uint32_t local_start[256], global_base[256];
    __shared__ uint32_t tmp[12];
    __shared__ uint32_t s_total, s_rowstart;
    __shared__ uint32_t stage[ROWB_CAP];
    __shared__ uint32_t s_mark[4][64];
    __shared__ uint4 s_rec[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    RowLocateLds &loc = *reinterpret_cast<RowLocateLds *>(stage);          // the staging area is not in use yet
    const RowBlock tb = row_locate_block(seg_totals, NR, blockIdx.x, ROW_THREADS * ITEMS, loc);
    __syncthreads();
    if (tb.count == 0u) return;             // (uniform)
    const uint32_t row = tb.bucket;
    // the wave's segments: ITEMS rounds of 64 consecutive ones
    uint2 sg[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t i = wave * (64 * ITEMS) + r * 64 + lane, g = tb.start + i;
        sg[r] = (i < tb.count && g < seg_cap) ? segs[g] : make_uint2(0u, 0u);
    }
    uint32_t *cnt = wave_cnt[wave];
    for (int i = lane; i < ROW_PAD; i += 64) cnt[i] = 0u;
    wsync();
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t x0 = sg[r].y & 0xFFu, w = (sg[r].y >> 8) & 0xFFu;
        if (w != 0u) { atomicAdd(&cnt[x0], 1u); atomicAdd(&cnt[x0 + w], 0xFFFFFFFFu); }
    }
    wsync();
    wave_diff_to_counts(cnt, lane);
    __syncthreads();
    {
        // thread t = tile column t
        uint32_t c[4], tot = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) { c[w] = tid < NX ? wave_cnt[w][tid] : 0u; tot += c[w]; }
        // instances of this column inside the row = difference of the column's exclusive prefix at the row's first block and at the
        // next row's first block (the blocks behind the last one hold zero counts: the prefix there is the column total)
        // (fb_next <= the number of blocks <= nblocks - 1 unless the frame overflowed its instance capacity: clamped, the frame is re-run anyway)
        const uint32_t fbn = tb.fb_next < nblocks ? tb.fb_next : nblocks - 1u;
        const uint32_t pf = tid < NX ? hist[(size_t)tid * nblocks + tb.fb_first] : 0u;
        const uint32_t gtot = tid < NX ? hist[(size_t)tid * nblocks + fbn] - pf : 0u;
        const uint32_t rt = tid < NR ? inst_totals[tid] : 0u;
        uint32_t total, gtotal, rtotal;
        const uint32_t ls = block_excl_scan(tot, tmp, total);
        uint32_t gb = block_excl_scan(gtot, tmp + 4, gtotal);
        const uint32_t rs = block_excl_scan(rt, tmp + 8, rtotal);
        if (tid == (int)row) s_rowstart = rs;
        if (tid == 0) s_total = total;
        __syncthreads();
        gb += s_rowstart;
        if (tid < NX) {
            local_start[tid] = ls;
            global_base[tid] = gb + hist[(size_t)tid * nblocks + blockIdx.x] - pf;
            // tile (row, column) occupies [gb, gb + gtot): identifyTileRanges (CR/rasterizer_impl.cu:118-140) without reading a key
            if (blockIdx.x == tb.fb_first && gtot != 0u) {
                const uint32_t e = gb + gtot;
                ranges[(size_t)row * NX + tid] = make_uint2(gb < cap ? gb : cap, e < cap ? e : cap);
            }
            wave_cnt[0][tid] = ls; wave_cnt[1][tid] = ls + c[0]; wave_cnt[2][tid] = ls + c[0] + c[1]; wave_cnt[3][tid] = ls + c[0] + c[1] + c[2];
        }
    }
    __syncthreads();
    const uint32_t total = s_total;
    const bool staged = total <= ROWB_CAP;
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t x0 = sg[r].y & 0xFFu, w = (sg[r].y >> 8) & 0xFFu;
        const uint32_t incl = wave_incl_scan(w, lane), off = incl - w;
        const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        s_rec[wave][lane] = make_uint4(off, sg[r].x, x0, 0u);
        uint32_t carry = 0;
        for (uint32_t t0 = 0; t0 < n; t0 += 64) {           // wave-uniform trip count
            s_mark[wave][lane] = 0u;
            wsync();
            if (w != 0u && off - t0 < 64u) s_mark[wave][off - t0] = (uint32_t)lane + 1u;
            wsync();
            uint32_t owner = wave_inclusive_max_u32(s_mark[wave][lane]);
            owner = owner > carry ? owner : carry;
            carry = (uint32_t)__builtin_amdgcn_readlane((int)owner, 63);
            const uint32_t t = t0 + lane;
            const bool valid = t < n;
            const uint4 rec = s_rec[wave][valid ? owner - 1u : 0u];
            const uint32_t tx = valid ? rec.z + (t - rec.x) : 0u;
            const uint32_t slot = ranked_slot<NBITS>(cnt, tx, nbits, valid, lane);
            if (valid) {
                if (staged) stage[slot] = (tx << 24) | rec.y;
                else {
                    const uint32_t dst = global_base[tx] + (slot - local_start[tx]);
                    if (dst < cap) { point_list[dst] = rec.y; if (tile_ids_out) tile_ids_out[dst] = row * (uint32_t)NX + tx; }
                }
            }
        }
        wsync();
    }
    if (!staged) return;            // (uniform)
    __syncthreads();
#pragma unroll 4
    for (uint32_t p = tid; p < total; p += ROW_THREADS) {
        const uint32_t v = stage[p];
        const uint32_t tx = v >> 24;
        const uint32_t dst = global_base[tx] + (p - local_start[tx]);
        if (dst < cap) { point_list[dst] = v & 0xFFFFFFu; if (tile_ids_out) tile_ids_out[dst] = row * (uint32_t)NX + tx; }
    }
}

static inline int bits_for(int n) { int b = 1; while ((1 << b) < n) b++; return b; }      // digits 0 .. n-1

}  // namespace

bool ex4d_tile_sort_rows_applies(int P, int gx, int gy) { return gx <= 255 && gy <= 255 && (uint32_t)P <= (1u << 24); }
static inline uint32_t rowsA_blocks(uint32_t P) { return (P + ROWA_GAUSS - 1) / ROWA_GAUSS; }
static inline int rowsB_items(uint32_t P, uint32_t R) { return (uint64_t)R > 16ull * P ? 2 : 4; }      // wide rects: smaller blocks, so that their instances fit the LDS stage
static inline uint32_t rowsB_blocks(uint32_t R, int gy, int items) { return R / (uint32_t)(ROW_THREADS * items) + (uint32_t)gy + 2u; }      // S <= R segments in <= gy rows
// pass A' lives in the geometry buffer (sized by P alone: <= 255 rows): [2 gy][nbA] + 2 gy totals + the number of pass-B' blocks
size_t ex4d_tile_sort_rows_geom_words(uint32_t P) { return (size_t)2 * 255 * (rowsA_blocks(P) + 1) + 64; }
// pass B' in the binning buffer (sized by R and the image alone, for the smaller block size): [gx][nbB] + gx totals
size_t ex4d_tile_sort_rows_hist_words(uint32_t R, int gx, int gy) { return (size_t)gx * (rowsB_blocks(R, gy, 2) + 1) + 64; }

// order / (r4 | r8): Gaussian ids and tile rects in depth order; segs: cap entries of scratch; point_list: cap entries; ranges: zeroed by
// the caller (tiles without instances are never written); frame_total (optional): += the frame's instance count
hipError_t ex4d_tile_sort_rows(int P, int gx, int gy, const uint32_t *order, const uint32_t *r4, const uint2 *r8, uint2 *segs,
    uint32_t *point_list, uint32_t *tile_ids_out, uint32_t cap, uint32_t *histA, uint32_t *histB, uint2 *ranges, uint32_t *frame_total, hipStream_t stream)
{
    if (P <= 0 || cap == 0) return hipSuccess;
    const int items = rowsB_items((uint32_t)P, cap);
    const uint32_t nbA = rowsA_blocks((uint32_t)P), nbB = rowsB_blocks(cap, gy, items);
    uint32_t *totA = histA + (size_t)2 * gy * nbA;
    uint32_t *nb_dev = totA + (((size_t)2 * gy + 15) & ~(size_t)15);
    const int by = bits_for(gy), bx = bits_for(gx);
    hipLaunchKernelGGL(rows_seg_hist_kernel, dim3(nbA), dim3(ROW_THREADS), 0, stream, (uint32_t)P, gy, r4, r8, histA, nbA, frame_total, nb_dev);
    hipLaunchKernelGGL(rows_scan_kernel, dim3(2 * gy), dim3(256), 0, stream, nbA, nbA, histA, (uint32_t)(2 * gy), (uint32_t)(ROW_THREADS * items), (uint32_t)gy, nb_dev, (const uint32_t *)nullptr);
#define ROWS_A(NB) hipLaunchKernelGGL((rows_seg_scatter_kernel<NB>), dim3(nbA), dim3(ROW_THREADS), 0, stream, (uint32_t)P, gy, by, order, r4, r8, (const uint32_t *)histA, nbA, segs, cap)
    if (by == 6) ROWS_A(6); else if (by == 7) ROWS_A(7); else if (by == 8) ROWS_A(8); else ROWS_A(0);
#undef ROWS_A
    const uint32_t *seg_tot = totA, *inst_tot = totA + gy;
    if (items == 4) hipLaunchKernelGGL((rows_inst_hist_kernel<4>), dim3(nbB), dim3(ROW_THREADS), 0, stream, (const uint2 *)segs, cap, seg_tot, gy, gx, histB, nbB, (const uint32_t *)nb_dev);
    else hipLaunchKernelGGL((rows_inst_hist_kernel<2>), dim3(nbB), dim3(ROW_THREADS), 0, stream, (const uint2 *)segs, cap, seg_tot, gy, gx, histB, nbB, (const uint32_t *)nb_dev);
    hipLaunchKernelGGL(rows_scan_kernel, dim3(gx), dim3(256), 0, stream, nbB, nbB, histB, (uint32_t)gx, 0u, 0u, (uint32_t *)nullptr, (const uint32_t *)nb_dev);
#define ROWS_B(IT, NB) hipLaunchKernelGGL((rows_inst_scatter_kernel<IT, NB>), dim3(nbB), dim3(ROW_THREADS), 0, stream, (const uint2 *)segs, cap, seg_tot, inst_tot, gy, gx, bx, \
        (const uint32_t *)histB, nbB, (const uint32_t *)nb_dev, point_list, cap, ranges, tile_ids_out)
    if (items == 4) { if (bx == 7) ROWS_B(4, 7); else if (bx == 8) ROWS_B(4, 8); else if (bx == 6) ROWS_B(4, 6); else ROWS_B(4, 0); }
    else { if (bx == 7) ROWS_B(2, 7); else if (bx == 8) ROWS_B(2, 8); else if (bx == 6) ROWS_B(2, 6); else ROWS_B(2, 0); }
#undef ROWS_B
    return hipGetLastError();
}
