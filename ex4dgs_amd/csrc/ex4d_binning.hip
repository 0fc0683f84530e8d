// Binning for gfx950: depth ordering of Gaussians, tile-instance duplication, stable radix sort by tile,
// tile ranges.
//
// Replaces (CR/ = submodules/diff_gaussian_rasterization_df/cuda_rasterizer/):
//   cub::DeviceScan::InclusiveSum      CR/rasterizer_impl.cu:295
//   duplicateWithKeys                  CR/rasterizer_impl.cu:72-113
//   cub::DeviceRadixSort::SortPairs    CR/rasterizer_impl.cu:321-326  (64-bit keys [tile | depth bits], 32-bit values)
//   identifyTileRanges (+ cudaMemset)  CR/rasterizer_impl.cu:118-140, :328
//
// Same result, different factorisation.  The reference sorts R (= instances, ~6x the Gaussian count)
// 96-bit pairs over 45 key bits (6 radix passes over R).  A stable sort by (tile, depth) with ties in
// ascending Gaussian id is identical to: (1) stable-sort the P Gaussians by depth bits (ties: ascending
// id), (2) emit their tile instances in that order, (3) stable-sort the instances by tile id alone.
// Step (1) touches P elements (4 passes), step (3) needs only ceil(log2 T)=13..14 key bits = 2 passes
// over R with 32-bit keys: ~5x less HBM traffic than the 6-pass 64-bit sort, bit-identical point_list.
// All of it is integer work and HBM/latency bound; wave64 ballots give the stable in-wave ranks.
#include "ex4d_internal.h"
#include <atomic>
#include <mutex>

// Round 6: stable ranking by LDS atomics.  The lanes of ONE ds_add_rtn_u32 instruction that hit the same LDS word receive their pre-op
// values in ascending lane order on this part (tools/dev/micro/lds_atomic_order.hip: 4.2e9 lane-trials over every collision pattern, none
// out of order), and a wave's LDS instructions execute in program order: `atomicAdd(&counter[digit], 1)` IS the stable rank of an item
// among the wave's items of its digit -- one LDS instruction per item instead of 5 VALU per digit bit of ballot ranking (35 to 50 per
// item in the scatter passes, which were bound by exactly that).  Not an architectural promise, so the library does not assume it: the
// first forward on a device runs the probe kernel below (a few microseconds, once) and the scatter kernels take the ballot path -- kept,
// and selectable with option "rank_lds_atomics" = 0 -- wherever the probe finds a lane out of order.
__device__ int ex4d_g_rank_lds = 0;

namespace {

__global__ __launch_bounds__(256) void lds_rank_probe_kernel(uint32_t seed, int trials, int ndigits, uint32_t *__restrict__ bad)
{
    __shared__ uint32_t cnt[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = seed ^ (blockIdx.x * 0x9E3779B9u) ^ (threadIdx.x * 0x85EBCA6Bu);
    uint32_t nb = 0;
    for (int t = 0; t < trials; t++) {
        for (int i = lane; i < 256; i += 64) cnt[wave][i] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        x = x * 1664525u + 1013904223u;
        const uint32_t d = (x >> 16) % (uint32_t)ndigits;
        uint32_t expect = 0, same = 0;
        for (int l = 0; l < 64; l++) { const uint32_t dl = __shfl(d, l, 64); if (dl == d) { same++; if (l < lane) expect++; } }
        const uint32_t got = atomicAdd(&cnt[wave][d], 1u);
        const uint32_t got2 = atomicAdd(&cnt[wave][d], 1u);          // a second instruction continues behind the first one's counts
        if (got != expect || got2 != same + expect) nb++;
    }
    if (nb) atomicAdd(bad, nb);
}

__device__ __forceinline__ int to_int_sat(float f)
{
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
}

// ---------------------------------------------------------------- radix sort (8-bit digits, stable)
// pass structure: histogram -> row scan -> scatter.  hist layout: [bin][block] followed by [bin] totals.
// COPIES > 1: lane L counts into copy L % COPIES of the block's histogram (rows padded by one word, so the copies of a bin sit in
// different LDS banks).  The tile sort's first pass sees long runs of equal digits -- the instances of one Gaussian are neighbouring
// tiles, their high digit is the same -- and 64 lanes adding to ONE LDS word serialise (round 3: 14.7 us for a 30 MB read, the LDS
// atomics of 7.5 M instances at one per cycle and CU); with 8 copies a run of 64 equal digits costs 8 serial steps, not 64.
template <int ITEMS, int BINS, int COPIES>
__global__ __launch_bounds__(RS_THREADS) void rs_histogram_kernel(const uint32_t *__restrict__ keys, uint32_t n, int shift,
    uint32_t mask, uint32_t nblocks, uint32_t *__restrict__ hist, const uint32_t *__restrict__ n_dev)
{
    // n_dev (asynchronous forward, Ex4dParams.instance_capacity): the item count lives in device memory, `n` is the capacity the grid
    // was sized for; workgroups behind the last item write an all-zero column
    if (n_dev) { const uint32_t nd = *n_dev; n = nd < n ? nd : n; }
    __shared__ uint32_t h_all[COPIES * (BINS + 1)];
    uint32_t *h = h_all + (COPIES > 1 ? (threadIdx.x % COPIES) * (BINS + 1) : 0);
    for (int b = threadIdx.x; b < COPIES * (BINS + 1); b += RS_THREADS) h_all[b] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (RS_THREADS * ITEMS);
    if (base < n) {              // (uniform)
        // all loads in flight before the first LDS atomic (clamped index instead of a branch per load)
        uint32_t k[ITEMS];
#pragma unroll
        for (int it = 0; it < ITEMS; it++) {
            const uint32_t i = base + it * RS_THREADS + threadIdx.x;
            k[it] = keys[i < n ? i : n - 1];
        }
#pragma unroll
        for (int it = 0; it < ITEMS; it++) {
            const uint32_t i = base + it * RS_THREADS + threadIdx.x;
            if (i < n) atomicAdd(&h[(k[it] >> shift) & mask], 1u);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < BINS; b += RS_THREADS) {
        uint32_t c = 0;
#pragma unroll
        for (int k = 0; k < COPIES; k++) c += h_all[k * (BINS + 1) + b];
        hist[(size_t)b * nblocks + blockIdx.x] = c;
    }
}

// one block per bin: exclusive scan of that bin's per-block counts; bin total to hist[RS_BINS*nblocks + bin].
// 8 consecutive counts per thread (two 16-byte loads when the row is aligned), so rows of up to 2048 blocks take ONE load round trip
// and one block-level scan (the kernel is pure latency: it used to loop over 256 counts at a time)
__global__ __launch_bounds__(256) void rs_scan_rows_kernel(uint32_t nblocks, uint32_t *__restrict__ hist, uint32_t bins)
{
    __shared__ uint32_t wave_sums[4];
    __shared__ uint32_t carry_s;
    uint32_t *row = hist + (size_t)blockIdx.x * nblocks;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool vec = ((((uintptr_t)row) & 15) == 0);
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nblocks; base += 2048) {
        const uint32_t i0 = base + 8 * threadIdx.x;
        uint32_t v[8];
        if (vec && i0 + 8 <= nblocks) {
            const uint4 a = *reinterpret_cast<const uint4 *>(row + i0), b = *reinterpret_cast<const uint4 *>(row + i0 + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = (i0 + k < nblocks) ? row[i0 + k] : 0u;
        }
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { const uint32_t t = v[k]; v[k] = run; run += t; }       // exclusive inside the thread
        uint32_t x = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
        if (lane == 63) wave_sums[wave] = x;
        __syncthreads();
        uint32_t woff = carry_s + x - run;
        for (int w = 0; w < wave; w++) woff += wave_sums[w];
        if (vec && i0 + 8 <= nblocks) {
            *reinterpret_cast<uint4 *>(row + i0) = make_uint4(woff + v[0], woff + v[1], woff + v[2], woff + v[3]);
            *reinterpret_cast<uint4 *>(row + i0 + 4) = make_uint4(woff + v[4], woff + v[5], woff + v[6], woff + v[7]);
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) if (i0 + k < nblocks) row[i0 + k] = woff + v[k];
        }
        __syncthreads();
        if (threadIdx.x == 255) carry_s = woff + run;
        __syncthreads();
    }
    if (threadIdx.x == 0) hist[(size_t)bins * nblocks + blockIdx.x] = carry_s;
}

// ---- top digit of the MSD depth sort (round 5).  The digit is cut from the key range the frame's visible Gaussians OCCUPY
// (dparams = {kmin, shift, invisible key}: derived by dls_range_kernel from the per-wave ranges the per-Gaussian kernel left):
//   digit(key) = (key - kmin) >> shift  in [0, BINS - 2]   for a visible Gaussian,   BINS - 1 for the invisible key
// -- cut from the static [min_depth, max_depth] a scene inside a narrow depth band landed in a handful of buckets.
struct DlsParams { uint32_t kmin, shift, inv_key; };
__device__ __forceinline__ DlsParams dls_params(const uint32_t *__restrict__ dparams) { return { dparams[0], dparams[1], dparams[2] }; }
__device__ __forceinline__ uint32_t dls_digit(uint32_t key, const DlsParams &q, uint32_t bins) { return key == q.inv_key ? bins - 1u : (key - q.kmin) >> q.shift; }

// Round 6: every workgroup of the histogram kernel derives dparams itself from the per-WORKGROUP key ranges the per-Gaussian kernel left
// (P / 256 pairs = 31 KB at 1.0 M, from L2) -- the one-workgroup range kernel and its launch in front of the sort are gone; workgroup 0
// leaves dparams in the frame flags for the partition and the bucket kernel.
__device__ __forceinline__ DlsParams dls_reduce_ranges(const uint2 *__restrict__ ranges, uint32_t nranges, uint32_t inv_key, uint32_t bins, uint32_t *s_red /* 8 words */)
{
    uint32_t kmax = 0u, nkmin = 0u;
    for (uint32_t base = 0; base < nranges; base += 8u * RS_THREADS) {          // 8 loads in flight per thread
        uint2 p[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { const uint32_t i = base + j * RS_THREADS + threadIdx.x; p[j] = i < nranges ? ranges[i] : make_uint2(0u, 0u); }
#pragma unroll
        for (int j = 0; j < 8; j++) { kmax = p[j].x > kmax ? p[j].x : kmax; nkmin = p[j].y > nkmin ? p[j].y : nkmin; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t a = __shfl_xor(kmax, o, 64), b = __shfl_xor(nkmin, o, 64);
        kmax = a > kmax ? a : kmax; nkmin = b > nkmin ? b : nkmin;
    }
    if ((threadIdx.x & 63) == 0) { s_red[threadIdx.x >> 6] = kmax; s_red[4 + (threadIdx.x >> 6)] = nkmin; }
    __syncthreads();
    for (int w = 0; w < RS_THREADS / 64; w++) { kmax = s_red[w] > kmax ? s_red[w] : kmax; nkmin = s_red[4 + w] > nkmin ? s_red[4 + w] : nkmin; }
    uint32_t kmin = ~nkmin, shift = 0;
    if (kmax == 0u) kmin = 0u;           // no visible Gaussian at all
    else {
        const uint32_t range = kmax - kmin;
        while ((range >> shift) > bins - 2u) shift++;
    }
    return { kmin, shift, inv_key };
}

template <int ITEMS, int BINS>
__global__ __launch_bounds__(RS_THREADS) void dls_histogram_kernel(const uint32_t *__restrict__ keys, uint32_t n, uint32_t nblocks,
    uint32_t *__restrict__ hist, uint32_t *__restrict__ dparams, const uint2 *__restrict__ ranges, uint32_t nranges, uint32_t inv_key)
{
    __shared__ uint32_t h[BINS];
    __shared__ uint32_t s_red[8];
    for (int b = threadIdx.x; b < BINS; b += RS_THREADS) h[b] = 0;
    const uint32_t base = blockIdx.x * (RS_THREADS * ITEMS);
    uint32_t k[ITEMS];
    if (base < n) {
#pragma unroll
        for (int it = 0; it < ITEMS; it++) {
            const uint32_t i = base + it * RS_THREADS + threadIdx.x;
            k[it] = keys[i < n ? i : n - 1];
        }
    }
    const DlsParams q = dls_reduce_ranges(ranges, nranges, inv_key, (uint32_t)BINS, s_red);      // (its barrier publishes h = 0)
    if (blockIdx.x == 0 && threadIdx.x == 0) { dparams[0] = q.kmin; dparams[1] = q.shift; dparams[2] = q.inv_key; dparams[3] = 0u; }
    if (base < n) {
#pragma unroll
        for (int it = 0; it < ITEMS; it++) {
            const uint32_t i = base + it * RS_THREADS + threadIdx.x;
            if (i < n) atomicAdd(&h[dls_digit(k[it], q, BINS)], 1u);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < BINS; b += RS_THREADS) hist[(size_t)b * nblocks + blockIdx.x] = h[b];
}

// Scatter pass.  A block owns RS_CHUNK consecutive items; wave w owns the w-th quarter of them and walks it in
// rounds of 64 consecutive items, so the stable order inside the block is (wave, round, lane).
//  phase 1  each wave ranks its items against its own running per-digit counters (wave-private LDS, wave64
//           ballot match, no block barrier);
//  phase 2  one barrier: per digit, exclusive scan over the 4 waves + block-local digit starts + global bases;
//  phase 3  items go to their block-local sorted slot in LDS, one barrier, then the block streams the staged
//           chunk out: neighbouring threads write neighbouring addresses of the same digit run (coalesced).
// MODE 0: (key, value) pairs in, pairs out.
// MODE 1: pairs in, ONE packed word out: (key & low_mask) << (32 - low_bits) | value  (pass A of the tile sort, below).
// MODE 2: packed words in (digit = word >> shift), values out; the workgroup's item range and bucket come from the block table, its
//         global digit starts from the bucket's span of the row-scanned histogram; also writes the tile ranges (pass B).
// MODE 3: (key, value, packed rect) triples in, triples out, and workgroup 0 leaves the first output position of every digit in
//         `bucket_starts` (the MSD partition of the depth sort, below: depth_local_sort_kernel finishes every bucket in LDS).
// Pass B of the tile sort cuts every bucket (= high digit) into blocks of <= 4096 items, so that no block straddles two buckets.
// Every workgroup derives its own block from the <= 256 bucket totals of pass A (two scans + a 8-step search: cheaper than a
// one-workgroup table kernel and its launch in the middle of the sort):
//   fb[h] = first block of bucket h (fb[256] = number of blocks), st[h] = first output position of bucket h
typedef Ex4dTsBlock TsBlock;        // { start, count, bucket, fb_first, fb_next, bucket_start }
// the block's record from the frame's table (two 16-byte-aligned... plain dword loads, L2-resident: 24 bytes)
__device__ __forceinline__ TsBlock ts_block_from_table(const Ex4dTsBlock *__restrict__ table, uint32_t b)
{
    const uint2 *p = reinterpret_cast<const uint2 *>(table + b);
    const uint2 a = p[0], c = p[1], e = p[2];
    return { a.x, a.y, c.x, c.y, e.x, e.y };
}
struct TsLocateLds { uint32_t cnt[257], fb[257], st[257], wave_sums[8]; };
__device__ __forceinline__ TsBlock ts_locate_block(const uint32_t *__restrict__ totals, int nbuckets, uint32_t b, TsLocateLds &L)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;         // 256 threads
    const uint32_t cnt = tid < nbuckets ? totals[tid] : 0u;
    const uint32_t nblk = (cnt + RS_CHUNK - 1) / RS_CHUNK;
    uint32_t x = cnt, y = nblk;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t a = __shfl_up(x, o, 64), c = __shfl_up(y, o, 64); if (lane >= o) { x += a; y += c; } }
    if (lane == 63) { L.wave_sums[wave] = x; L.wave_sums[4 + wave] = y; }
    __syncthreads();
    uint32_t st = x - cnt, fb = y - nblk;
    for (int w = 0; w < wave; w++) { st += L.wave_sums[w]; fb += L.wave_sums[4 + w]; }
    L.cnt[tid] = cnt; L.fb[tid] = fb; L.st[tid] = st;
    if (tid == 255) { L.fb[256] = fb + nblk; L.st[256] = st + cnt; L.cnt[256] = 0; }
    __syncthreads();
    TsBlock t = { 0u, 0u, 0u, 0u, 0u, 0u };
    if (b < L.fb[256]) {
        // the bucket owning block b: the LAST h with fb[h] <= b (empty buckets share their successor's first block and precede it)
        uint32_t h = 0;
#pragma unroll
        for (uint32_t step = 128; step > 0; step >>= 1) if (L.fb[h + step] <= b) h += step;
        const uint32_t within = (b - L.fb[h]) * RS_CHUNK;
        t.start = L.st[h] + within;
        t.count = L.cnt[h] - within < RS_CHUNK ? L.cnt[h] - within : RS_CHUNK;
        t.bucket = h; t.fb_first = L.fb[h]; t.fb_next = L.fb[h + 1]; t.bucket_start = L.st[h];
    }
    return t;
}
// NBITS > 0: digit width known at compile time -- the match loop below unrolls (5 VALU per bit instead of a 12-slot loop body with
// its scalar bookkeeping); 0 = any width at run time.
template <int ITEMS, int BINS, int MODE, int NBITS>
__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(const uint32_t *__restrict__ keys_in,
    const uint32_t *__restrict__ vals_in, uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
    uint32_t n, int shift, int nbits, uint32_t nblocks, const uint32_t *__restrict__ hist,
    int low_bits, const uint32_t *__restrict__ bucket_totals, int nbuckets, uint2 *__restrict__ ranges, const uint32_t *__restrict__ n_dev,
    const uint32_t *__restrict__ rects_in = nullptr, uint32_t *__restrict__ rects_out = nullptr, uint32_t *__restrict__ bucket_starts = nullptr,
    const uint32_t *__restrict__ dparams = nullptr, uint32_t range_stride = 0, uint32_t out_cap = 0xFFFFFFFFu,
    const Ex4dTsBlock *__restrict__ block_table = nullptr)
{
    // MODE 2 only: range_stride = tile ids per bucket (0: 1 << nbits, the binary split of the pair sort; the row-segment sort of round 6
    // has bucket = tile row, digit = tile column: the image's tiles per row); out_cap = capacity of the packed / output arrays (an
    // asynchronous frame whose instance count exceeds it is re-run by the caller: nothing may be touched behind it)
    if (n_dev) { const uint32_t nd = *n_dev; n = nd < n ? nd : n; }      // (see rs_histogram_kernel)
    // MODE 0, last pass of the LSD depth sort in front of the row-segment tile sort (round 6): the zero-fill of the tile ranges rides along
    // (range_stride = number of tiles), and the write-out below gathers every item's packed rect by its value (rects_in -> rects_out)
    if (MODE == 0 && ranges) for (uint32_t i = blockIdx.x * RS_THREADS + threadIdx.x; i < range_stride; i += gridDim.x * RS_THREADS) ranges[i] = make_uint2(0u, 0u);
    if (MODE != 2 && blockIdx.x * (uint32_t)(RS_THREADS * ITEMS) >= n) return;      // behind the last item (uniform: the whole workgroup)
    __shared__ uint32_t wave_cnt[4][BINS];        // per-wave digit counts -> exclusive block-local offsets
    __shared__ uint32_t local_start[BINS];        // first block-local slot of each digit
    __shared__ uint32_t global_base[BINS];        // global position of this block's first item of each digit
    __shared__ uint32_t scan_tmp[8];
    // (key, value) in block-local sorted order; MODE 2 (pass B of the tile sort) sorts packed words and writes ids: ONE word per item -- 16 KB
    // instead of 32, 22.5 KB of LDS per workgroup instead of 38.9: seven workgroups per CU instead of four (round 6)
    constexpr int STAGE_WORDS = (MODE == 2 ? 1 : 2) * RS_THREADS * ITEMS;
    __shared__ __attribute__((aligned(16))) uint32_t stage_w[STAGE_WORDS];
    uint2 *const stage = reinterpret_cast<uint2 *>(stage_w);
    __shared__ uint32_t stage_r[MODE == 3 ? RS_THREADS * ITEMS : 1];      // MODE 3: the third word of the triple
    // (MODE 3 at 16 items per thread -- more than 2 M Gaussians -- is the largest: 16 K of counters + 8 K + 32 K + 16 K = 72 KB, more than
    // the 64 KB of gfx90a-class LDS: this library is gfx950 only, 160 KB per CU)
    static_assert(sizeof(TsLocateLds) <= sizeof(uint32_t) * STAGE_WORDS, "the block table's scratch lives in the staging area");
    static_assert(sizeof(uint32_t) * (4 * BINS + 2 * BINS + 8) + sizeof(uint32_t) * STAGE_WORDS + sizeof(uint32_t) * (MODE == 3 ? RS_THREADS * ITEMS : 1) <= 80 * 1024,
                  "rs_scatter_kernel: static LDS beyond 80 KB (two workgroups per CU)");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (NBITS > 0) nbits = NBITS;
    const uint32_t mask = (1u << nbits) - 1u;
    const int nbins = 1 << nbits;
    // MODE 3: the digit of the MSD depth sort (dls_digit); every other mode: bits [shift, shift + nbits) of the key
    DlsParams dq = { 0u, 0u, 0u };
    if (MODE == 3) dq = dls_params(dparams);
    auto digit_of = [&](uint32_t k) -> uint32_t { return MODE == 3 ? dls_digit(k, dq, (uint32_t)BINS) : ((k >> shift) & mask); };
    for (int i = lane; i < BINS; i += 64) wave_cnt[wave][i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const bool lds_rank = ex4d_g_rank_lds != 0;          // (uniform)
    constexpr uint32_t CHUNK = RS_THREADS * ITEMS;
    uint32_t block_first = blockIdx.x * CHUNK, block_end = n, bucket = 0;
    TsBlock tb = { 0u, 0u, 0u, 0u, 0u, 0u };
    if (MODE == 2) {
        if (block_table) tb = ts_block_from_table(block_table, blockIdx.x);
        else {
            TsLocateLds &loc = *reinterpret_cast<TsLocateLds *>(stage_w);        // the staging area is not in use yet (no extra LDS)
            tb = ts_locate_block(bucket_totals, nbuckets, blockIdx.x, loc);
            __syncthreads();                            // every thread has its copy before the area is reused
        }
        if (tb.count == 0) return;                  // past the last block (uniform: the whole workgroup)
        block_first = tb.start; block_end = tb.start + tb.count; bucket = tb.bucket;
        if (block_first >= out_cap) return;         // (uniform)
        if (block_end > out_cap) block_end = out_cap;
    }
    const uint32_t base = block_first + wave * (CHUNK / 4);
    const uint64_t lt = (1ull << lane) - 1ull;
    // (round 6) the histogram words this thread's digits need behind the ranking -- the block's own prefix, the digit totals / the bucket's
    // span of the row-scanned histogram -- are requested HERE, in front of the items: their round trip (scattered 4-byte loads) hides behind
    // the load + ranking phase instead of standing between two barriers of the workgroup's critical path
    constexpr int BPT_PRE = BINS / RS_THREADS;
    uint32_t pre_pf[BPT_PRE], pre_gt[BPT_PRE], pre_hb[BPT_PRE];
#pragma unroll
    for (int k = 0; k < BPT_PRE; k++) {
        const int d = tid * BPT_PRE + k;
        const bool live = d < nbins;
        if (MODE == 2) {
            pre_pf[k] = live ? hist[(size_t)d * nblocks + tb.fb_first] : 0u;
            pre_gt[k] = live ? hist[(size_t)d * nblocks + tb.fb_next] : 0u;
        } else {
            pre_pf[k] = 0u;
            pre_gt[k] = live ? hist[(size_t)BINS * nblocks + d] : 0u;
        }
        pre_hb[k] = live ? hist[(size_t)d * nblocks + blockIdx.x] : 0u;
    }
    uint32_t key[ITEMS], val[ITEMS], pos[ITEMS], rct[MODE == 3 ? ITEMS : 1];
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        // unconditional loads from a clamped index (no exec-masked branch per item): the waits before the first ranking steps
        // can then count outstanding loads instead of draining all 32
        const uint32_t i = base + it * 64 + lane;
        const uint32_t ic = i < block_end ? i : block_end - 1;
        key[it] = keys_in[ic];
        val[it] = (MODE == 2) ? 0u : (vals_in ? vals_in[ic] : ic);        // vals_in == nullptr: the values are the item indices (first pass of the depth sort)
        if constexpr (MODE == 3) rct[it] = rects_in[ic];
    }
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        const uint32_t i = base + it * 64 + lane;
        const bool valid = i < block_end;
        const uint32_t d = digit_of(key[it]);
        if (lds_rank) {
            // stable rank by the LDS atomic's return value (round 6; see ex4d_g_rank_lds above): index among this wave's items of digit d
            pos[it] = valid ? atomicAdd(&wave_cnt[wave][d], 1u) : 0u;
            continue;
        }
        // lanes holding the same digit: per bit keep the ballot if my bit is set, its complement otherwise -- written as
        // ballot ^ (bit - 1) on 32-bit halves (plain xor/and; a select here compiles to the VOP2 v_cndmask that issues in ~24
        // cycles on gfx950 and made this loop the most expensive part of the pass)
        const uint64_t vmask = __builtin_amdgcn_ballot_w64(valid);
        uint32_t plo = (uint32_t)vmask, phi = (uint32_t)(vmask >> 32);
        if (NBITS > 0) {
#pragma unroll
            for (int b = 0; b < NBITS; b++) {
                const uint64_t bal = __builtin_amdgcn_ballot_w64((d & (1u << b)) != 0u);
                uint32_t flip;                                    // 0 when my bit is set, ~0 otherwise: my bit IS my lane's bit of the ballot
                asm("v_cndmask_b32_e64 %0, -1, 0, %1" : "=v"(flip) : "s"(bal));
                plo &= (uint32_t)bal ^ flip;
                phi &= (uint32_t)(bal >> 32) ^ flip;
            }
        } else {
            for (int b = 0; b < nbits; b++) {
                const uint32_t bit = (d >> b) & 1u;
                const uint64_t bal = __builtin_amdgcn_ballot_w64(bit != 0u);
                const uint32_t flip = bit - 1u;                   // 0 when my bit is set, ~0 otherwise
                plo &= (uint32_t)bal ^ flip;
                phi &= (uint32_t)(bal >> 32) ^ flip;
            }
        }
        const uint64_t peers = ((uint64_t)phi << 32) | plo;
        const uint32_t rank = __popcll(peers & lt);
        // every lane reads its digit's counter (the peers of a digit read ONE word: an LDS broadcast), then the first lane of every
        // digit adds the digit's count -- a wave's LDS accesses execute in order, so the reads see the value before the update
        // (round 5; rounds 1-4: the leader read, wrote and handed `before` to its peers through ds_bpermute: one more LDS round trip)
        const uint32_t before = wave_cnt[wave][valid ? d : 0u];
        if (valid && rank == 0) wave_cnt[wave][d] = before + (uint32_t)__popcll(peers);      // one lane per distinct digit: no two writers share an address
        pos[it] = before + rank;                  // index among this wave's items of digit d
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();
    {
        // thread t owns the BPT consecutive digits t*BPT .. : exclusive scan over waves per digit, then exclusive scan over digits of
        // the block totals (and of the global digit totals, which gives every digit's global start)
        constexpr int BPT = BINS / RS_THREADS;
        uint32_t c[BPT][4], tot[BPT], gtot[BPT], pf[BPT];
        uint32_t tsum = 0, gsum = 0;
#pragma unroll
        for (int k = 0; k < BPT; k++) {
            const int d = tid * BPT + k;
            const bool live = d < nbins;
#pragma unroll
            for (int w = 0; w < 4; w++) c[k][w] = live ? wave_cnt[w][d] : 0u;
            tot[k] = c[k][0] + c[k][1] + c[k][2] + c[k][3];
            static_assert(BPT == BPT_PRE, "the prefetched histogram words follow the digit ownership of this phase");
            if (MODE == 2) {
                // items of this digit inside the bucket = difference of the row's exclusive prefix at the bucket's first block and
                // at the next bucket's first block (blocks past the table hold zero counts, so the prefix there is the row total)
                pf[k] = pre_pf[k];
                gtot[k] = live ? pre_gt[k] - pf[k] : 0u;
            } else {
                pf[k] = 0u;
                gtot[k] = pre_gt[k];
            }
            tsum += tot[k]; gsum += gtot[k];
        }
        uint32_t x = tsum, gx = gsum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o, 64), gy = __shfl_up(gx, o, 64);
            if (lane >= o) { x += y; gx += gy; }
        }
        if (lane == 63) { scan_tmp[wave] = x; scan_tmp[4 + wave] = gx; }
        __syncthreads();
        uint32_t ls = x - tsum, gb = gx - gsum;
        for (int w = 0; w < wave; w++) { ls += scan_tmp[w]; gb += scan_tmp[4 + w]; }
        if (MODE == 2) gb += tb.bucket_start;                   // the bucket's first output position
#pragma unroll
        for (int k = 0; k < BPT; k++) {
            const int d = tid * BPT + k;
            if (d < nbins) {
                local_start[d] = ls;
                global_base[d] = gb + pre_hb[k] - pf[k];
                // tile (bucket, d) occupies [gb, gb + gtot): identifyTileRanges (CR/rasterizer_impl.cu:118-140) without reading the keys
                if (MODE == 2 && blockIdx.x == tb.fb_first && gtot[k] != 0u) {
                    const uint32_t e = gb + gtot[k];
                    ranges[(size_t)bucket * (range_stride ? range_stride : (1u << nbits)) + d] = make_uint2(gb < out_cap ? gb : out_cap, e < out_cap ? e : out_cap);
                }
                if (MODE == 3 && blockIdx.x == 0) bucket_starts[d] = gb;            // first output position of digit d
                wave_cnt[0][d] = ls; wave_cnt[1][d] = ls + c[k][0]; wave_cnt[2][d] = ls + c[k][0] + c[k][1]; wave_cnt[3][d] = ls + c[k][0] + c[k][1] + c[k][2];
            }
            ls += tot[k]; gb += gtot[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        const uint32_t i = base + it * 64 + lane;
        if (i < block_end) {
            const uint32_t d = digit_of(key[it]);
            if constexpr (MODE == 2) stage_w[wave_cnt[wave][d] + pos[it]] = key[it];
            else stage[wave_cnt[wave][d] + pos[it]] = make_uint2(key[it], val[it]);
            if constexpr (MODE == 3) stage_r[wave_cnt[wave][d] + pos[it]] = rct[it];
        }
    }
    __syncthreads();
    const uint32_t count = (block_end - block_first) < CHUNK ? (block_end - block_first) : CHUNK;
#pragma unroll 4
    for (uint32_t p = tid; p < count; p += RS_THREADS) {
        uint2 kv;
        if constexpr (MODE == 2) kv = make_uint2(stage_w[p], 0u); else kv = stage[p];
        const uint32_t d = digit_of(kv.x);
        const uint32_t dst = global_base[d] + (p - local_start[d]);
        if (MODE == 0) { keys_out[dst] = kv.x; vals_out[dst] = kv.y; if (rects_in) rects_out[dst] = rects_in[kv.y]; }
        if constexpr (MODE == 3) { keys_out[dst] = kv.x; vals_out[dst] = kv.y; rects_out[dst] = stage_r[p]; }
        if (MODE == 1) keys_out[dst] = ((kv.x & ((1u << low_bits) - 1u)) << (32 - low_bits)) | kv.y;
        if (MODE == 2 && dst < out_cap) {
            vals_out[dst] = kv.x & (0xFFFFFFFFu >> nbits);
            if (keys_out) keys_out[dst] = bucket * (range_stride ? range_stride : (1u << nbits)) + d;        // the sorted tile ids, on request only
        }
    }
}

// Tile sort, MSD first (the instance count R is ~7.5x the Gaussian count; the sort is HBM-bound):
//   pass A  stable partition by the HIGH digit of the tile id (buckets), writing ONE packed word per instance -- the low digit in the
//           top bits, the Gaussian id below (4 bytes instead of the 8 of a key/value pair);
//   blocks  every bucket is cut into blocks of <= 4096 items, so no block straddles two buckets; each workgroup of pass B derives its
//           block from the bucket totals itself (ts_locate_block);
//   pass B  per bucket, stable counting sort by the LOW digit: block histograms -> row scan -> scatter of the Gaussian ids alone.
//           The (bucket, digit) counts are the tile ranges, so identifyTileRanges never reads 4 R bytes of keys.
// Same result as the reference's single stable sort by (tile | depth): stable by high digit, then stable by low digit inside each
// bucket = stable by the whole tile id.  HBM traffic per instance: 8 written by the duplication + (4 + 8 + 4) + (4 + 4 + 4) = 36 bytes
// instead of 8 + 2 x (4 + 8 + 8) + 4 = 52.
__global__ __launch_bounds__(RS_THREADS) void ts_histogram_kernel(const uint32_t *__restrict__ words, const uint32_t *__restrict__ bucket_totals,
    int nbuckets, int shift, uint32_t nblocks, uint32_t *__restrict__ hist, uint32_t cap = 0xFFFFFFFFu, const Ex4dTsBlock *__restrict__ block_table = nullptr)
{
    __shared__ uint32_t h[256];
    __shared__ TsLocateLds loc;
    h[threadIdx.x] = 0;
    TsBlock tb;
    if (block_table) { tb = ts_block_from_table(block_table, blockIdx.x); __syncthreads(); }      // (the barrier publishes h = 0)
    else tb = ts_locate_block(bucket_totals, nbuckets, blockIdx.x, loc);      // (contains the barriers that publish h = 0)
    if (tb.count != 0) {
        uint32_t k[RS_ITEMS];
#pragma unroll
        for (int it = 0; it < RS_ITEMS; it++) {
            const uint32_t i = it * RS_THREADS + threadIdx.x;
            const uint32_t gi = tb.start + (i < tb.count ? i : tb.count - 1);
            k[it] = words[gi < cap ? gi : cap - 1];
        }
#pragma unroll
        for (int it = 0; it < RS_ITEMS; it++) {
            const uint32_t i = it * RS_THREADS + threadIdx.x;
            if (i < tb.count && tb.start + i < cap) atomicAdd(&h[k[it] >> shift], 1u);
        }
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}


// ---------------------------------------------------------------- depth sort, MSD first: every bucket finished in LDS
// Round 5.  The LSD depth sort was nine launches (3 x histogram / row scan / scatter) of 5-13 us each, bound by launch ramp and
// latency, not by its 8 MB.  Now: ONE global partition of the P (key, id, packed rect) triples by the TOP DLS_MSD_BITS bits of the
// depth key (histogram -> row scan -> scatter, MODE 3 above; stable, so equal keys keep ascending ids) -- a bucket then holds
// ~1.5 k Gaussians at 1.0 M -- and ONE kernel in which a workgroup per bucket sorts its bucket on the remaining `rem` key bits
// entirely in LDS and permutes (id, rect) IN PLACE: the depth order and the rects in depth order (which the tile scan used to
// gather at random: 52 MB of traffic for a 4 MB array) fall out of the same kernel.  4 launches instead of 9 + a streaming scan.
//   * LDS word = (key & remmask) << 13 | arrival index: sorting the words by their key bits with a STABLE LSD radix sort (9-bit digits,
//     wave-private ballot ranking like the scatter kernel) is the stable sort by key; the payload is fetched once at the end through the
//     arrival index (a gather inside the bucket's own 8-32 KB window).
//   * invisible Gaussians carry a key whose top digit is theirs alone (host: one past the digit of max_depth): the partition
//     leaves them in id order behind every visible bucket with their zero rect; nobody touches that bucket again.
//   * a bucket with more than `cap` (<= DLS_CAP = 8192) Gaussians -- a scene squeezed into < 1 % of [min_depth, max_depth] -- is sorted
//     by its workgroup through global memory (ping-pong with the partition's input arrays, whose slice [start, start + n) nobody
//     else uses): correct for any size, slow by design (tests force it with a small `cap`).
#define DLS_ITEMS 16
#define DLS_MAX_CAP 8192
template <int THREADS> struct DlsLds {
    static constexpr int WAVES = THREADS / 64, CAP = THREADS * DLS_ITEMS, IDX_BITS = (THREADS == 512 ? 13 : 12);
    static_assert((1 << IDX_BITS) == CAP, "arrival index bits");
    uint32_t buf[CAP]; uint32_t cnt[WAVES][512]; uint32_t scan_tmp[WAVES];
};
#define DLS_IDX_BITS 13      // (the widest arrival index: what the host checks the key bits against)

__device__ __forceinline__ void dls_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// lanes of the wave that hold the same digit as this lane (among the valid ones); invalid lanes get 0.
// NBITS > 0: digit width known at compile time (unrolled, 5 VALU per bit: see rs_scatter_kernel); 0: `nbits` at run time
template <int NBITS>
__device__ __forceinline__ uint64_t dls_peers(uint32_t d, int nbits, bool valid)
{
    const uint64_t vmask = __builtin_amdgcn_ballot_w64(valid);
    uint32_t plo = (uint32_t)vmask, phi = (uint32_t)(vmask >> 32);
    if (NBITS > 0) {
#pragma unroll
        for (int b = 0; b < NBITS; b++) {
            const uint64_t bal = __builtin_amdgcn_ballot_w64((d & (1u << b)) != 0u);
            uint32_t flip;                                    // 0 when my bit is set, ~0 otherwise: my bit IS my lane's bit of the ballot
            asm("v_cndmask_b32_e64 %0, -1, 0, %1" : "=v"(flip) : "s"(bal));
            plo &= (uint32_t)bal ^ flip;
            phi &= (uint32_t)(bal >> 32) ^ flip;
        }
    } else {
        for (int b = 0; b < nbits; b++) {
            const uint32_t bit = (d >> b) & 1u;
            const uint64_t bal = __builtin_amdgcn_ballot_w64(bit != 0u);
            const uint32_t flip = bit - 1u;                   // 0 when my bit is set, ~0 otherwise
            plo &= (uint32_t)bal ^ flip;
            phi &= (uint32_t)(bal >> 32) ^ flip;
        }
    }
    const uint64_t peers = ((uint64_t)phi << 32) | plo;
    return valid ? peers : 0ull;
}
// stable rank of this lane's item among the wave's items processed so far: returns the index among the wave's items of digit d
// (wave-private counter row `cnt`), and bumps the counter.  All 64 lanes call it.
template <int NBITS>
__device__ __forceinline__ uint32_t dls_rank(uint32_t *cnt, uint32_t d, int nbits, bool valid, int lane)
{
    if (ex4d_g_rank_lds != 0) return valid ? atomicAdd(&cnt[d], 1u) : 0u;      // (uniform; round 6: see ex4d_g_rank_lds)
    const uint64_t peers = dls_peers<NBITS>(d, nbits, valid);
    const uint32_t rank = __popcll(peers & ((1ull << lane) - 1ull));
    const uint32_t before = cnt[valid ? d : 0u];                            // (all peers read one word; see rs_scatter_kernel)
    if (valid && rank == 0) cnt[d] = before + (uint32_t)__popcll(peers);    // one lane per distinct digit
    dls_wave_sync();
    return before + rank;
}
// cnt[w][d] (counts of digit d among wave w's items) -> first destination slot of (d, w): exclusive scan over (digit, wave)
template <int THREADS>
__device__ __forceinline__ void dls_scan_counts(DlsLds<THREADS> &L, int nbits)
{
    constexpr int DLS_WAVES = THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // thread t owns the PER consecutive digits PER * t ...: 512 threads = 512 digits, 256 threads = 2 digits each
    constexpr int PER = 512 / THREADS;
    uint32_t c[PER][DLS_WAVES], tot = 0;
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const bool live = tid * PER + q < (1 << nbits);
#pragma unroll
        for (int w = 0; w < DLS_WAVES; w++) { c[q][w] = live ? L.cnt[w][tid * PER + q] : 0u; tot += c[q][w]; }
    }
    uint32_t x = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
    if (lane == 63) L.scan_tmp[wave] = x;
    __syncthreads();
    uint32_t run = x - tot;
    for (int w = 0; w < wave; w++) run += L.scan_tmp[w];
#pragma unroll
    for (int q = 0; q < PER; q++) {
        if (tid * PER + q < (1 << nbits)) {
#pragma unroll
            for (int w = 0; w < DLS_WAVES; w++) { L.cnt[w][tid * PER + q] = run; run += c[q][w]; }
        }
    }
    __syncthreads();
}
__device__ __forceinline__ int dls_pass_bits(int rem, int lo, int npass, int pass) { return (rem - lo + (npass - pass) - 1) / (npass - pass); }

// n > cap: the workgroup sorts the bucket's (key, id, rect) triples through global memory; X = the partition's output slice (where the
// result has to end), Y = the same slice of the partition's input arrays
template <int THREADS>
__device__ __forceinline__ void dls_sort_through_memory(DlsLds<THREADS> &L, uint32_t *xk, uint32_t *xv, uint32_t *xr, uint32_t *yk, uint32_t *yv, uint32_t *yr,
    uint32_t n, int rem, uint32_t kmin)
{
    constexpr int DLS_WAVES = THREADS / 64, DLS_THREADS = THREADS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int npass = (rem + 8) / 9;
    const uint32_t m = ((n + DLS_WAVES * 64 - 1) / (DLS_WAVES * 64)) * 64;          // items per wave (a multiple of 64)
    const uint32_t wbase = wave * m < n ? wave * m : n, wend = (wbase + m) < n ? (wbase + m) : n;
    uint32_t *sk = xk, *sv = xv, *sr = xr, *dk = yk, *dv = yv, *dr = yr;
    for (int pass = 0, lo = 0; pass < npass; pass++) {
        const int nbits = dls_pass_bits(rem, lo, npass, pass);
        const uint32_t mask = (1u << nbits) - 1u;
        for (int i = lane; i < 512; i += 64) L.cnt[wave][i] = 0;
        dls_wave_sync();
        for (uint32_t i = wbase + lane; i < wend; i += 64) atomicAdd(&L.cnt[wave][((sk[i] - kmin) >> lo) & mask], 1u);
        __syncthreads();
        dls_scan_counts(L, nbits);
        for (uint32_t base = wbase; base < wend; base += 64) {        // wave-uniform trip count, items in order
            const uint32_t i = base + lane;
            const bool valid = i < wend;
            const uint32_t k = valid ? sk[i] : 0u, v = valid ? sv[i] : 0u, r = valid ? sr[i] : 0u;
            const uint32_t d = ((k - kmin) >> lo) & mask;
            const uint32_t dst = dls_rank<0>(L.cnt[wave], d, nbits, valid, lane);
            if (valid) { dk[dst] = k; dv[dst] = v; dr[dst] = r; }
        }
        __threadfence();
        __syncthreads();
        uint32_t *t;
        t = sk; sk = dk; dk = t; t = sv; sv = dv; dv = t; t = sr; sr = dr; dr = t;
        lo += nbits;
    }
    if (sv != xv) {                 // odd number of passes: the result sits in Y
        for (uint32_t i = tid; i < n; i += DLS_THREADS) { xv[i] = sv[i]; xr[i] = sr[i]; }
    }
}

// Tile counts of the bucket's Gaussians in depth order -> bucket-local inclusive scan (local_incl[s + p]) and the bucket's instance
// count: the tile scan of CR/rasterizer_impl.cu:295 without a kernel of its own -- duplicate_kernel adds the exclusive prefix of the
// <= 1024 bucket sums itself.  Generic form (any n): chunks of DLS_CAP rects read back from memory.
__device__ __forceinline__ uint32_t rect4_count(uint32_t r) { return ((r >> 16) & 0xFFu) * (r >> 24); }
// counts of one chunk sit in L.buf[0 .. cn): inclusive scan in place (+ carry); returns the chunk total.  All threads call it.
template <int THREADS>
__device__ __forceinline__ uint32_t dls_scan_chunk(DlsLds<THREADS> &L, uint32_t cn, uint32_t carry)
{
    constexpr int DLS_WAVES = THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t v[DLS_ITEMS], run = 0;
#pragma unroll
    for (int q = 0; q < DLS_ITEMS; q++) { const uint32_t i = tid * DLS_ITEMS + q; run += (i < cn) ? L.buf[i] : 0u; v[q] = run; }
    uint32_t x = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
    if (lane == 63) L.scan_tmp[wave] = x;
    __syncthreads();
    uint32_t excl = carry + x - run, total = 0;
    for (int w = 0; w < DLS_WAVES; w++) { const uint32_t t = L.scan_tmp[w]; if (w < wave) excl += t; total += t; }
#pragma unroll
    for (int q = 0; q < DLS_ITEMS; q++) { const uint32_t i = tid * DLS_ITEMS + q; if (i < cn) L.buf[i] = excl + v[q]; }
    __syncthreads();
    return total;
}
template <int THREADS>
__device__ __forceinline__ uint32_t dls_scan_from_memory(DlsLds<THREADS> &L, const uint32_t *rects, uint32_t *local_incl, uint32_t n)
{
    constexpr int DLS_THREADS = THREADS, DLS_CAP = THREADS * DLS_ITEMS;
    const int tid = threadIdx.x;
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < n; c0 += DLS_CAP) {
        const uint32_t cn = (n - c0) < DLS_CAP ? (n - c0) : DLS_CAP;
        uint32_t r[DLS_ITEMS];
#pragma unroll
        for (int j = 0; j < DLS_ITEMS; j++) { const uint32_t p = tid + j * DLS_THREADS; r[j] = p < cn ? rects[c0 + p] : 0u; }
#pragma unroll
        for (int j = 0; j < DLS_ITEMS; j++) { const uint32_t p = tid + j * DLS_THREADS; if (p < cn) L.buf[p] = rect4_count(r[j]); }
        __syncthreads();
        carry += dls_scan_chunk(L, cn, carry);
#pragma unroll
        for (int j = 0; j < DLS_ITEMS; j++) { const uint32_t p = tid + j * DLS_THREADS; if (p < cn) local_incl[c0 + p] = L.buf[p]; }
        __syncthreads();
    }
    return carry;
}

// one stable LSD pass over the n words in L.buf on the digit (word >> sh) & (2^nbits - 1): wave w owns the w-th run of m words
template <int THREADS, int NBITS>
__device__ __forceinline__ void dls_lds_pass(DlsLds<THREADS> &L, uint32_t n, uint32_t m, int sh, int nbits)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (NBITS > 0) nbits = NBITS;
    const uint32_t mask = (1u << nbits) - 1u;
    const uint32_t wbase = wave * m;
    for (int i = lane; i < (1 << nbits); i += 64) L.cnt[wave][i] = 0;
    dls_wave_sync();
    uint32_t word[DLS_ITEMS], pos[DLS_ITEMS];
#pragma unroll
    for (int it = 0; it < DLS_ITEMS; it++) {
        word[it] = 0u; pos[it] = 0u;
        if ((uint32_t)(it * 64) < m) {                    // (wave-uniform)
            const uint32_t i = wbase + it * 64 + lane;
            const bool valid = i < n;
            word[it] = valid ? L.buf[i] : 0u;
            pos[it] = dls_rank<NBITS>(L.cnt[wave], (word[it] >> sh) & mask, nbits, valid, lane);
        }
    }
    __syncthreads();                   // every word is in a register, every count final
    dls_scan_counts(L, nbits);
#pragma unroll
    for (int it = 0; it < DLS_ITEMS; it++) {
        if ((uint32_t)(it * 64) < m) {
            const uint32_t i = wbase + it * 64 + lane;
            if (i < n) L.buf[L.cnt[wave][(word[it] >> sh) & mask] + pos[it]] = word[it];
        }
    }
    __syncthreads();
}

// THREADS = 256: buckets of <= 4096 in LDS, 24 KB of LDS (the choice up to ~1.2 M Gaussians: ~1.5 k per bucket, every workgroup of the
// grid resident at once); 512: <= 8192, 48 KB (more Gaussians per bucket: 2 M and up)
template <int THREADS>
__global__ __launch_bounds__(THREADS, THREADS == 512 ? 4 : 5) void depth_local_sort_kernel(uint32_t *kb, uint32_t *vb, uint32_t *rb, uint32_t *ka, uint32_t *va, uint32_t *ra,
    const uint32_t *__restrict__ starts, const uint32_t *__restrict__ totals, const uint32_t *__restrict__ dparams, uint32_t cap,
    uint32_t *__restrict__ local_incl, uint32_t *__restrict__ bucket_sums, int T, uint2 *__restrict__ ranges, uint32_t *__restrict__ watch)
{
    constexpr int DLS_THREADS = THREADS, DLS_WAVES = THREADS / 64, DLS_CAP = THREADS * DLS_ITEMS, IDX_BITS = DlsLds<THREADS>::IDX_BITS;
    __shared__ DlsLds<THREADS> L;
    const uint32_t b = blockIdx.x;
    const int tid = threadIdx.x;
    // the zero-fill of the tile ranges (cudaMemset at CR/rasterizer_impl.cu:328) rides along here (it used to ride in the tile scan)
    if (ranges) for (int i = b * DLS_THREADS + tid; i < T; i += gridDim.x * DLS_THREADS) ranges[i] = make_uint2(0u, 0u);
    const uint32_t n = totals[b], s = starts[b];
    const DlsParams dq = dls_params(dparams);
    const int rem = (int)dq.shift;                        // key bits below the top digit: what is left to sort inside a bucket
    const uint32_t inv_digit = gridDim.x - 1u;            // (one workgroup per digit)
    if (b == inv_digit || n == 0u) {                     // invisible Gaussians: already in id order, rect 0; nobody reads their offsets
        if (bucket_sums && tid == 0) bucket_sums[b] = 0u;
        return;
    }
    const bool sorted_already = n < 2u || rem == 0;      // (rem == 0: all keys of a bucket are equal, the partition was stable)
    if (sorted_already || n > cap) {
        // a bucket beyond the LDS capacity: one workgroup sorts it through global memory (or, all its keys being equal, only scans it from
        // there) -- correct, and slow (hundreds of microseconds for a wall of Gaussians at one depth).  `watch` (pinned host word, optional)
        // tells the host it happened: in its "auto" mode the library then orders the following frames with the LSD sort, which does not
        // care (ex4d_api.hip: depth_sort_auto_msd)
        if (n > cap && watch && tid == 0) __hip_atomic_store(watch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (!sorted_already) {
            dls_sort_through_memory(L, kb + s, vb + s, rb + s, ka + s, va + s, ra + s, n, rem, dq.kmin); __threadfence(); __syncthreads();
        }
        if (local_incl) {
            const uint32_t sum = dls_scan_from_memory(L, rb + s, local_incl + s, n);
            if (tid == 0) bucket_sums[b] = sum;
        }
        return;
    }
    const uint32_t remmask = (1u << rem) - 1u;
    {
        uint32_t k[DLS_ITEMS];                           // every load in flight before the first LDS store
#pragma unroll
        for (int j = 0; j < DLS_ITEMS; j++) { const uint32_t p = tid + j * DLS_THREADS; k[j] = p < n ? kb[s + p] : 0u; }
#pragma unroll
        for (int j = 0; j < DLS_ITEMS; j++) { const uint32_t p = tid + j * DLS_THREADS; if (p < n) L.buf[p] = (((k[j] - dq.kmin) & remmask) << IDX_BITS) | p; }
    }
    __syncthreads();
    const int npass = (rem + 8) / 9;
    const uint32_t m = ((n + DLS_WAVES * 64 - 1) / (DLS_WAVES * 64)) * 64;          // items per wave: a multiple of 64, <= 1024
    // digit widths of the common key ranges at compile time (rem = 16: depths in (4, 300]; 17: (0.01, 300]), any other at run time
    if (rem == 16) { dls_lds_pass<THREADS, 8>(L, n, m, IDX_BITS, 8); dls_lds_pass<THREADS, 8>(L, n, m, IDX_BITS + 8, 8); }
    else if (rem == 17) { dls_lds_pass<THREADS, 9>(L, n, m, IDX_BITS, 9); dls_lds_pass<THREADS, 8>(L, n, m, IDX_BITS + 9, 8); }
    else {
        for (int pass = 0, lo = 0; pass < npass; pass++) {
            const int nbits = dls_pass_bits(rem, lo, npass, pass);
            dls_lds_pass<THREADS, 0>(L, n, m, IDX_BITS + lo, nbits);
            lo += nbits;
        }
    }
    // payload: position p takes the (id, rect) that arrived at index idx(p).  In place: every load lands before any store goes out
    uint32_t v[DLS_ITEMS], r[DLS_ITEMS];
#pragma unroll
    for (int j = 0; j < DLS_ITEMS; j++) {
        const uint32_t p = tid + j * DLS_THREADS;
        v[j] = 0u; r[j] = 0u;
        if (p < n) { const uint32_t idx = L.buf[p] & (DLS_CAP - 1u); v[j] = vb[s + idx]; r[j] = rb[s + idx]; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int j = 0; j < DLS_ITEMS; j++) {
        const uint32_t p = tid + j * DLS_THREADS;
        if (p < n) { vb[s + p] = v[j]; rb[s + p] = r[j]; if (local_incl) L.buf[p] = rect4_count(r[j]); }
    }
    if (local_incl) {
        __syncthreads();
        const uint32_t sum = dls_scan_chunk(L, n, 0u);
#pragma unroll
        for (int j = 0; j < DLS_ITEMS; j++) { const uint32_t p = tid + j * DLS_THREADS; if (p < n) local_incl[s + p] = L.buf[p]; }
        if (tid == 0) bucket_sums[b] = sum;
    }
}

// ---------------------------------------------------------------- scan of tiles_touched in depth order
// rects4 (optional): the same rects packed into 32 bits (x0 | y0 << 8 | w << 16 | h << 24; images of at most 255 x 255 tiles).  The
// gather in depth order is random: from the 8-byte array every read pulls a whole cache line of an 8 MB array that no L2 holds (round 3:
// 87 MB of traffic for 8 MB of rects); the packed array is 4 MB at 1.0 M Gaussians and stays resident in each XCD's 4 MB L2.
__device__ __forceinline__ uint2 unpack_rect(uint32_t p) { return make_uint2((p & 0xFFu) | (((p >> 8) & 0xFFu) << 16), ((p >> 16) & 0xFFu) | ((p >> 24) << 16)); }
__global__ __launch_bounds__(256) void scan_tiles_local_kernel(int P, const uint2 *__restrict__ rects, const uint32_t *__restrict__ rects4,
    const uint32_t *__restrict__ order, uint2 *__restrict__ sorted_rects, uint32_t *__restrict__ out, uint32_t *__restrict__ block_sums,
    int T, uint2 *__restrict__ ranges, uint32_t *__restrict__ frame_total)
{
    // every kernel launch costs ~5 us of ramp and tail on this part: the zero-fill of the tile ranges (cudaMemset at
    // CR/rasterizer_impl.cu:328) rides along here instead of being its own launch
    for (int i = blockIdx.x * 256 + threadIdx.x; i < T; i += gridDim.x * 256) ranges[i] = make_uint2(0u, 0u);
    // coalesced (striped) global accesses, blocked scan: counts go through LDS; each thread scans 8 consecutive items,
    // then wave + block scan of the thread totals
    __shared__ uint32_t cnt[SCAN_CHUNK];
    __shared__ uint32_t wave_sums[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int block_first = blockIdx.x * SCAN_CHUNK;
#pragma unroll
    for (int it = 0; it < SCAN_CHUNK / 256; it++) {
        const int i = block_first + it * 256 + threadIdx.x;
        // the only random gather of the binning stage: the rect of the i-th Gaussian in depth order (8 bytes), written
        // back in that order so that the duplication kernel streams it
        uint2 rc = make_uint2(0u, 0u);
        // (order == nullptr: rects4 already IS in depth order -- the MSD depth sort carried it along -- and is streamed; nothing is written back)
        if (i < P) { rc = rects4 ? unpack_rect(rects4[order ? order[i] : (uint32_t)i]) : rects[order[i]]; if (sorted_rects) sorted_rects[i] = rc; }
        cnt[it * 256 + threadIdx.x] = (rc.y & 0xFFFFu) * (rc.y >> 16);
    }
    __syncthreads();
    uint32_t v[8];
    uint32_t run = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { run += cnt[threadIdx.x * 8 + k]; v[k] = run; }
    uint32_t x = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
    if (lane == 63) wave_sums[wave] = x;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; w++) woff += wave_sums[w];
    const uint32_t excl = woff + x - run;
#pragma unroll
    for (int k = 0; k < 8; k++) cnt[threadIdx.x * 8 + k] = excl + v[k];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < SCAN_CHUNK / 256; it++) {
        const int i = block_first + it * 256 + threadIdx.x;
        if (i < P) out[i] = cnt[it * 256 + threadIdx.x];
    }
    if (threadIdx.x == 255) {
        block_sums[blockIdx.x] = woff + x;
        // the frame's instance count in device memory (cleared with the frame flags): what the kernels behind this one read when the
        // forward runs asynchronously (a few hundred fire-and-forget atomics spread over the kernel's duration)
        atomicAdd(frame_total, woff + x);
    }
}

// ---------------------------------------------------------------- duplication
// The 64 Gaussians of a wave are consecutive in depth order, so their instances form ONE contiguous output
// range.  The wave walks that range 64 outputs at a time and issues fully coalesced stores -- no serial per-Gaussian
// loops, no tail behind large rects.  Which Gaussian owns output slot t?
// Round 2 answered with a 6-step binary search over the lanes' offsets: 6 dependent ds_bpermute + 5 more to fetch the
// owner's fields per 64 outputs -- the kernel was bound by that LDS round-trip chain (rocprofv3: 7.3e6 VALU instructions
// but 28 us; SQ_WAIT_INST_LDS 1.8e7).  Round 3: owners are monotone in t, so every Gaussian that STARTS inside the
// current 64 outputs drops its lane index at its first slot (one LDS write), an inclusive max-scan over the lanes (DPP,
// no LDS) spreads it to the slots behind, and the owner's fields come out of one 20-byte LDS record: 2 LDS writes + 3
// reads per 64 outputs, a dependent chain of 3 round trips instead of 7.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_max_u32(uint32_t x)
{
    // lanes without a source (or outside row_mask) keep x: max(x, x) = x
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
struct DupRec { uint32_t off, gid, xy, w, magic; };
__global__ __launch_bounds__(256) void duplicate_kernel(int P, int gx, int gy, const uint32_t *__restrict__ order,
    const uint32_t *__restrict__ sorted_offsets, const uint32_t *__restrict__ block_sums,
    const uint2 *__restrict__ sorted_rects, const uint32_t *__restrict__ sorted_rects4,
    uint32_t *__restrict__ tile_keys, uint32_t *__restrict__ vals, uint32_t cap,
    const uint32_t *__restrict__ bucket_keys, const uint32_t *__restrict__ dparams, const uint32_t *__restrict__ bucket_sums, int nbuckets, uint32_t *__restrict__ frame_total)
{
    // cap: capacity of the output arrays -- the instance count itself (synchronous forward) or Ex4dParams.instance_capacity (an
    // instance count above it truncates the stream: the caller sees that in the frame status and re-runs the frame)
    // bucket_keys != nullptr (MSD depth sort with the scan fused into its bucket kernel, round 5): sorted_offsets holds the inclusive
    // scan INSIDE each depth bucket, bucket_sums the instance count of every bucket; the bucket of position k is the top digit of
    // bucket_keys[k] (dls_digit; the partition's key output: all keys of a bucket's slice share it).  Every workgroup scans the <= 1024 bucket sums
    // itself (4 KB from L2) -- there is no tile-scan kernel on this path; workgroup 0 leaves the frame's instance count in frame_total.
    __shared__ DupRec s_rec[4][64];
    __shared__ uint32_t s_mark[4][64];
    __shared__ uint32_t s_base[1024];
    __shared__ uint32_t s_wsum[4];
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (bucket_keys) {
        const int per = nbuckets >> 8;               // 2 or 4 consecutive buckets per thread (nbuckets = 512 or 1024)
        uint32_t c[4], run = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) { c[q] = q < per ? bucket_sums[threadIdx.x * per + q] : 0u; }
#pragma unroll
        for (int q = 0; q < 4; q++) { const uint32_t t = c[q]; c[q] = run; run += t; }
        uint32_t x = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
        if (lane == 63) s_wsum[wave] = x;
        __syncthreads();
        uint32_t excl = x - run;
        for (int wv = 0; wv < wave; wv++) excl += s_wsum[wv];
#pragma unroll
        for (int q = 0; q < 4; q++) if (q < per) s_base[threadIdx.x * per + q] = excl + c[q];
        if (blockIdx.x == 0 && threadIdx.x == 255) *frame_total = excl + run;
        __syncthreads();
    }
    uint32_t gid = 0, off = 0, count = 0;
    int x0 = 0, y0 = 0, w = 1;
    if (k < P) {
        gid = order[k];
        const uint2 rc = sorted_rects4 ? unpack_rect(sorted_rects4[k]) : sorted_rects[k];      // getRect (CR/auxiliary.h:46-56) was evaluated once, by the preprocess kernel
        x0 = (int)(rc.x & 0xFFFFu); y0 = (int)(rc.x >> 16);
        w = (int)(rc.y & 0xFFFFu);
        count = (uint32_t)w * (rc.y >> 16);
        if (w <= 0) w = 1;
        // exclusive offset = inclusive scan value of the previous element (+ its scan chunk's base, below); bucket form: the bucket's
        // base + the Gaussian's own inclusive value - its count (no neighbour, no bucket-boundary case)
        if (!bucket_keys) off = (k == 0) ? 0u : sorted_offsets[k - 1];
        else if (count != 0u) off = s_base[dls_digit(bucket_keys[k], dls_params(dparams), (uint32_t)nbuckets)] + sorted_offsets[k] - count;
    }
    // base of a scan chunk = sum of the chunk totals before it.  The (k-1) of a wave lie in at most two chunks; the wave sums
    // the few hundred totals itself instead of a one-workgroup scan kernel in between (one launch less)
    if (!bucket_keys) {
        const int kf = blockIdx.x * 256 + (threadIdx.x & ~63) - 1;            // k - 1 of the wave's first lane (may be -1)
        const int c0 = kf < 0 ? 0 : kf / SCAN_CHUNK;
        uint32_t part = 0;
        for (int j = lane; j < c0; j += 64) part += block_sums[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        if (k > 0 && k < P) {
            const int c = (k - 1) / SCAN_CHUNK;
            off += part + (c > c0 ? block_sums[c0] : 0u);
        }
    }
    // row = local / w without a per-output division: local < 2^16 (a rect has at most gx*gy tiles; images with more than 65535 tiles
    // take the plain division below) and w < 2^16, so floor(local / w) == umulhi(local, floor((2^32 - 1) / w) + 1) exactly; one division per Gaussian
    const uint32_t magic = 0xFFFFFFFFu / (uint32_t)w + 1u;
    // the wave's output range [start, end): from the first slot of its first Gaussian with instances to the last slot of its last one
    // (Gaussians without instances -- invisible ones, lanes behind P -- take no part: their offsets may be anything)
    uint32_t start = count != 0u ? off : 0xFFFFFFFFu, end = count != 0u ? off + count : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t t = __shfl_xor(end, o, 64), u = __shfl_xor(start, o, 64);
        end = t > end ? t : end; start = u < start ? u : start;
    }
    s_rec[wave][lane] = { off, gid, (uint32_t)x0 | ((uint32_t)y0 << 16), (uint32_t)w, magic };
    uint32_t carry = 0;                      // owner (lane index + 1) of the last output of the previous round
    for (uint32_t tb = start; tb < end; tb += 64) {       // wave-uniform trip count
        const uint32_t t = tb + lane;
        s_mark[wave][lane] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // Gaussians with instances whose first slot lies in this round (distinct slots: their ranges are disjoint)
        if (count != 0u && off - tb < 64u) s_mark[wave][off - tb] = (uint32_t)lane + 1u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t owner = wave_inclusive_max_u32(s_mark[wave][lane]);
        owner = owner > carry ? owner : carry;
        carry = (uint32_t)__builtin_amdgcn_readlane((int)owner, 63);
        if (t < end && t < cap) {            // owner >= 1 here: slot `start` belongs to the first Gaussian with instances
            const DupRec r = s_rec[wave][owner - 1u];
            const uint32_t local = t - r.off;
            uint32_t row = (r.w == 1u) ? local : __umulhi(local, r.magic);           // magic wraps to 0 for w == 1
            if (gx * gy > 65535) row = local / r.w;                                   // > 4080 x 4080 pixels: plain division
            const uint32_t ty = (r.xy >> 16) + row, tx = (r.xy & 0xFFFFu) + (local - row * r.w);
            tile_keys[t] = ty * (uint32_t)gx + tx;
            vals[t] = r.gid;
        }
    }
}

// ---------------------------------------------------------------- zero fill
// The library clears its frame flags and the backward's accumulator rows with a kernel of its own instead of hipMemsetAsync: a memset
// NODE of a captured graph was observed (ROCm 7.2, round 4: tools/dev/dbg_graph.py) to clear its target on the first replay only --
// from the second replay on the frame-flag words held stale pointers-like values and the instance count came out as garbage + R.
__global__ __launch_bounds__(256) void zero_fill_kernel(uint4 *__restrict__ p, size_t n16, uint32_t tail_words)
{
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = z;
    if (blockIdx.x == 0 && threadIdx.x < tail_words) reinterpret_cast<uint32_t *>(p + n16)[threadIdx.x] = 0u;       // < 4 words behind the last 16 bytes
}

// ---------------------------------------------------------------- tile ranges
__global__ __launch_bounds__(256) void tile_ranges_kernel(uint32_t R, const uint32_t *__restrict__ tile_ids, uint2 *__restrict__ ranges,
    const uint32_t *__restrict__ n_dev)
{
    if (n_dev) { const uint32_t nd = *n_dev; R = nd < R ? nd : R; }
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= R) return;
    const uint32_t cur = tile_ids[i];
    if (i == 0) ranges[cur].x = 0;
    else {
        const uint32_t prev = tile_ids[i - 1];
        if (cur != prev) { ranges[prev].y = i; ranges[cur].x = i; }
    }
    if (i == R - 1) ranges[cur].y = R;
}

}  // namespace

// ---- "rank_lds_atomics": -1 = probe once per device (default), 0 = ballot ranking, 1 = ranking by LDS atomics without asking
static std::atomic<int> g_rank_lds_option{-1};
static std::mutex g_rank_lds_mu;
static int g_rank_lds_device = -1, g_rank_lds_value = 0;
void ex4d_set_rank_lds(int v) { std::lock_guard<std::mutex> lock(g_rank_lds_mu); g_rank_lds_option.store(v); g_rank_lds_device = -1; }
int ex4d_get_rank_lds() { return g_rank_lds_option.load(); }
int ex4d_rank_lds_in_use() { return g_rank_lds_value; }
// makes the device flag of the current device match the option; capturing streams keep whatever the device already has
hipError_t ex4d_prepare_rank_lds(hipStream_t stream)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return hipGetLastError();
    std::lock_guard<std::mutex> lock(g_rank_lds_mu);
    if (dev == g_rank_lds_device) return hipSuccess;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone) return hipSuccess;
    (void)hipGetLastError();
    int want = g_rank_lds_option.load();
    if (want < 0) {
        uint32_t *bad = nullptr, h = 1;
        hipError_t e = hipMalloc((void **)&bad, sizeof(uint32_t));
        if (e != hipSuccess) return e;
        (void)hipMemset(bad, 0, sizeof(uint32_t));
        for (int nd : { 1, 3, 17, 64, 256 })
            hipLaunchKernelGGL(lds_rank_probe_kernel, dim3(256), dim3(256), 0, 0, 0x5EEDu + nd, 200, nd, bad);
        e = hipMemcpy(&h, bad, sizeof(uint32_t), hipMemcpyDeviceToHost);
        (void)hipFree(bad);
        if (e != hipSuccess) return e;
        want = h == 0u ? 1 : 0;
    }
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(ex4d_g_rank_lds), &want, sizeof(int));
    if (e != hipSuccess) return e;
    g_rank_lds_device = dev; g_rank_lds_value = want;
    return hipSuccess;
}

// small inputs (the per-Gaussian depth sort) use 1024-item chunks so that every CU gets several workgroups, and 9-bit digits
// (512 bins): its passes are launch / latency bound, so one pass less is worth more than the two extra ballots per item;
// large ones (the per-instance tile sort) 4096-item chunks and <= 8-bit digits
#define RS_SMALL_ITEMS 8
static inline int rs_items_for(uint32_t n) { return n <= (2u << 20) ? RS_SMALL_ITEMS : RS_ITEMS; }
static inline uint32_t rs_blocks_for(uint32_t n) { const uint32_t c = RS_THREADS * rs_items_for(n); return (n + c - 1) / c; }
static inline int rs_max_bits_for(uint32_t n) { return rs_items_for(n) == RS_SMALL_ITEMS ? 9 : 8; }
size_t ex4d_radix_hist_words(uint32_t n) { return (size_t)1024 * rs_blocks_for(n) + 1024; }      // (1024: the MSD depth sort's bins)
int ex4d_radix_passes(uint32_t n, int end_bit) { const int mb = rs_max_bits_for(n); return (end_bit + mb - 1) / mb; }

hipError_t ex4d_radix_sort_pairs(uint32_t *keys_a, uint32_t *vals_a, uint32_t *keys_b, uint32_t *vals_b,
    uint32_t n, int end_bit, uint32_t *hist, bool *result_in_a, hipStream_t stream, const uint32_t *n_dev, bool iota_values,
    const uint32_t *gather_in, uint32_t *gather_out, uint2 *zero_ranges, int num_ranges)
{
    // gather_in / gather_out (optional): the LAST pass also writes gather_out[sorted position] = gather_in[value] (the packed tile rects in
    // depth order for the row-segment tile sort: no gathering scan kernel behind the sort); zero_ranges: cleared by that pass
    *result_in_a = true;
    if (n == 0) return hipSuccess;
    const uint32_t nb = rs_blocks_for(n);
    const bool small = rs_items_for(n) == RS_SMALL_ITEMS;
    uint32_t *kin = keys_a, *vin = iota_values ? nullptr : vals_a, *kout = keys_b, *vout = vals_b;
    uint32_t *vnext = vals_a;          // (ping-pong partner of vals_b after the first pass)
    // balanced digits (13 bits -> 7 + 6, not 8 + 5): a pass with fewer bins writes longer runs per digit and workgroup
    const int npass = ex4d_radix_passes(n, end_bit);
    for (int pass = 0, shift = 0; pass < npass; pass++) {
        const int nbits = (end_bit - shift + (npass - pass) - 1) / (npass - pass);
        const uint32_t mask = (1u << nbits) - 1u;
        if (small) {
            hipLaunchKernelGGL((rs_histogram_kernel<RS_SMALL_ITEMS, 512, 1>), dim3(nb), dim3(RS_THREADS), 0, stream, kin, n, shift, mask, nb, hist, n_dev);
            hipLaunchKernelGGL(rs_scan_rows_kernel, dim3(1u << nbits), dim3(256), 0, stream, nb, hist, 512u);
            const bool last = pass == npass - 1;
#define RS_SMALL_SCATTER(NB) hipLaunchKernelGGL((rs_scatter_kernel<RS_SMALL_ITEMS, 512, 0, NB>), dim3(nb), dim3(RS_THREADS), 0, stream, kin, vin, kout, vout, n, shift, nbits, nb, hist, 0, nullptr, 0, \
                last ? zero_ranges : (uint2 *)nullptr, n_dev, last ? gather_in : (const uint32_t *)nullptr, last ? gather_out : (uint32_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr, (uint32_t)num_ranges)
            if (nbits == 9) RS_SMALL_SCATTER(9); else if (nbits == 8) RS_SMALL_SCATTER(8); else RS_SMALL_SCATTER(0);
#undef RS_SMALL_SCATTER
        } else {
            hipLaunchKernelGGL((rs_histogram_kernel<RS_ITEMS, 256, 1>), dim3(nb), dim3(RS_THREADS), 0, stream, kin, n, shift, mask, nb, hist, n_dev);
            hipLaunchKernelGGL(rs_scan_rows_kernel, dim3(1u << nbits), dim3(256), 0, stream, nb, hist, 256u);
            const bool last = pass == npass - 1;
            hipLaunchKernelGGL((rs_scatter_kernel<RS_ITEMS, 256, 0, 0>), dim3(nb), dim3(RS_THREADS), 0, stream, kin, vin, kout, vout, n, shift, nbits, nb, hist, 0, nullptr, 0,
                last ? zero_ranges : (uint2 *)nullptr, n_dev, last ? gather_in : (const uint32_t *)nullptr, last ? gather_out : (uint32_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr, (uint32_t)num_ranges);
        }
        uint32_t *t = kin; kin = kout; kout = t;
        t = (pass == 0) ? vnext : vin; vin = vout; vout = t;
        *result_in_a = !*result_in_a;
        shift += nbits;
    }
    return hipGetLastError();
}

// ---- MSD depth sort (depth_local_sort_kernel above): histogram / row scan / partition on the top digit of (key - kmin), buckets in LDS.
// (ka, va, ra): keys / ids / packed rects as the preprocess kernel wrote them (the ids are NOT read: they are 0 .. n-1, the partition
// generates them; all three arrays are scratch of the bucket kernel afterwards); (kb, vb, rb): the result --
// vb = Gaussian ids in depth order, rb = their packed rects in that order.  inv_key: the key of invisible Gaussians (above every
// visible key); flags: the frame-flag words (the digit parameters land there); wave_ranges: the per-wave key ranges of the per-Gaussian kernel; hist: ex4d_radix_hist_words(n) words; starts: 1 << DLS_MSD_BITS words.
// key_bits (host bound on the visible keys): the bits below the digit must fit the LDS word next to the arrival index
bool ex4d_depth_sort_msd_applies(uint32_t n, int key_bits) { return n <= (1u << 26) && key_bits - (EX4D_DLS_MSD_BITS - 1) + DLS_IDX_BITS <= 32; }
// Width of the top digit (round 6): 9 bits -- 511 visible buckets of ~1.6 k Gaussians at 1.0 M, each finished by a 512-thread workgroup
// (<= 8192 in LDS) -- up to 1.3 M Gaussians where the key bits under it fit the LDS word; 10 bits beyond (buckets twice as small: the
// margin below the LDS capacity that 2 M Gaussians need).  Same-box A/B at 1.0 M (stage depth_sort, ms): 10 bits / 256 threads 0.0708,
// 10 / 512 0.0768, 9 / 512 0.0632, 9 / 256 0.0744 (oversize buckets); config 2 (100 k) 0.0420 -> 0.0402: the partition's histogram is half as
// large (its ~0.5 M scattered 4-byte stores are what that kernel costs), the bucket kernel runs 511 workgroups instead of 1023.
int ex4d_depth_sort_msd_bits(uint32_t n, int key_bits)
{
    const int narrow = EX4D_DLS_MSD_BITS - 1;
    return (n <= 1300000u && key_bits - (narrow - 1) + DLS_IDX_BITS <= 32) ? narrow : EX4D_DLS_MSD_BITS;
}
template <int MB>
static hipError_t depth_sort_msd_launch(uint32_t *ka, uint32_t *va, uint32_t *ra, uint32_t *kb, uint32_t *vb, uint32_t *rb, uint32_t n, uint32_t inv_key,
    uint32_t *flags, const uint2 *wave_ranges, uint32_t *hist, uint32_t *starts, uint32_t local_cap, hipStream_t stream,
    uint32_t *local_incl, uint32_t *bucket_sums, int T, uint2 *ranges, int local_threads, uint32_t *watch)
{
    if (n == 0) return hipSuccess;
    constexpr int BINS = 1 << MB;
    const uint32_t nb = rs_blocks_for(n);
    const bool small = rs_items_for(n) == RS_SMALL_ITEMS;
    uint32_t *dparams = flags + EX4D_FLAG_DPARAMS;
    const uint32_t nranges = (n + 255u) / 256u;          // one (max key, max ~key) pair per workgroup of the per-Gaussian kernel
    if (local_threads != 256 && local_threads != 512) local_threads = (MB < EX4D_DLS_MSD_BITS || n > 1200000u) ? 512 : 256;      // (the 9-bit digit's buckets are twice as large)
    const uint32_t kcap = (uint32_t)local_threads * DLS_ITEMS;
    if (local_cap == 0 || local_cap > kcap) local_cap = kcap;
    if (small) {
        hipLaunchKernelGGL((dls_histogram_kernel<RS_SMALL_ITEMS, BINS>), dim3(nb), dim3(RS_THREADS), 0, stream, ka, n, nb, hist, dparams, wave_ranges, nranges, inv_key);
        hipLaunchKernelGGL(rs_scan_rows_kernel, dim3(BINS), dim3(256), 0, stream, nb, hist, (uint32_t)BINS);
        hipLaunchKernelGGL((rs_scatter_kernel<RS_SMALL_ITEMS, BINS, 3, MB>), dim3(nb), dim3(RS_THREADS), 0, stream, ka, (const uint32_t *)nullptr, kb, vb, n, 0, MB, nb, hist,
            0, (const uint32_t *)nullptr, 0, (uint2 *)nullptr, (const uint32_t *)nullptr, ra, rb, starts, (const uint32_t *)dparams);
    } else {
        hipLaunchKernelGGL((dls_histogram_kernel<RS_ITEMS, BINS>), dim3(nb), dim3(RS_THREADS), 0, stream, ka, n, nb, hist, dparams, wave_ranges, nranges, inv_key);
        hipLaunchKernelGGL(rs_scan_rows_kernel, dim3(BINS), dim3(256), 0, stream, nb, hist, (uint32_t)BINS);
        hipLaunchKernelGGL((rs_scatter_kernel<RS_ITEMS, BINS, 3, MB>), dim3(nb), dim3(RS_THREADS), 0, stream, ka, (const uint32_t *)nullptr, kb, vb, n, 0, MB, nb, hist,
            0, (const uint32_t *)nullptr, 0, (uint2 *)nullptr, (const uint32_t *)nullptr, ra, rb, starts, (const uint32_t *)dparams);
    }
    if (local_threads == 256)
        hipLaunchKernelGGL(depth_local_sort_kernel<256>, dim3(BINS), dim3(256), 0, stream, kb, vb, rb, ka, va, ra, starts, hist + (size_t)BINS * nb, (const uint32_t *)dparams, local_cap,
            local_incl, bucket_sums, T, ranges, watch);
    else
        hipLaunchKernelGGL(depth_local_sort_kernel<512>, dim3(BINS), dim3(512), 0, stream, kb, vb, rb, ka, va, ra, starts, hist + (size_t)BINS * nb, (const uint32_t *)dparams, local_cap,
            local_incl, bucket_sums, T, ranges, watch);
    return hipGetLastError();
}

hipError_t ex4d_depth_sort_msd(uint32_t *ka, uint32_t *va, uint32_t *ra, uint32_t *kb, uint32_t *vb, uint32_t *rb, uint32_t n, uint32_t inv_key,
    uint32_t *flags, const uint2 *wave_ranges, uint32_t *hist, uint32_t *starts, uint32_t local_cap, hipStream_t stream,
    uint32_t *local_incl, uint32_t *bucket_sums, int T, uint2 *ranges, int local_threads, uint32_t *watch, int msd_bits)
{
    if (msd_bits == EX4D_DLS_MSD_BITS - 1)
        return depth_sort_msd_launch<EX4D_DLS_MSD_BITS - 1>(ka, va, ra, kb, vb, rb, n, inv_key, flags, wave_ranges, hist, starts, local_cap, stream, local_incl, bucket_sums, T, ranges, local_threads, watch);
    return depth_sort_msd_launch<EX4D_DLS_MSD_BITS>(ka, va, ra, kb, vb, rb, n, inv_key, flags, wave_ranges, hist, starts, local_cap, stream, local_incl, bucket_sums, T, ranges, local_threads, watch);
}

// ---- MSD tile sort (see ts_block_table_kernel).  Applies when the tile id needs 9..16 bits and the Gaussian ids fit under the low digit.
bool ex4d_tile_sort_msd_applies(int P, int tile_bits)
{
    if (tile_bits < 9 || tile_bits > 16) return false;
    const int low_bits = (tile_bits + 1) / 2;
    return (uint64_t)P <= (1ull << (32 - low_bits));
}
static inline int ts_clamp_bits(int tile_bits) { return tile_bits > 16 ? 16 : (tile_bits < 2 ? 2 : tile_bits); }
static inline uint32_t ts_max_blocks(uint32_t R, int tile_bits) { return rs_num_blocks(R) + (1u << (tile_bits - (tile_bits + 1) / 2)) + 1u; }
// two histograms: pass A's [256][nbA] + totals stays readable (bucket totals) while pass B fills its own [256][nbB] + totals
size_t ex4d_tile_sort_hist_words(uint32_t R, int tile_bits) { return (size_t)256 * (rs_num_blocks(R) + 1) + (size_t)256 * (ts_max_blocks(R, ts_clamp_bits(tile_bits)) + 1); }

// keys / vals: the instances in depth order (from the duplication); packed: R words of scratch; point_list: the result; tile_ids_out:
// optional (nullptr = not materialised); ranges must be zero (tiles without instances are never written)
hipError_t ex4d_tile_sort_msd(const uint32_t *keys, const uint32_t *vals, uint32_t *packed, uint32_t *point_list, uint32_t *tile_ids_out,
    uint32_t R, int tile_bits, uint32_t *hist, uint2 *ranges, hipStream_t stream, const uint32_t *n_dev)
{
    if (R == 0) return hipSuccess;
    const int low_bits = (tile_bits + 1) / 2, high_bits = tile_bits - low_bits;
    const uint32_t nbA = rs_num_blocks(R), nbB = ts_max_blocks(R, tile_bits);
    uint32_t *histB = hist + (size_t)256 * (nbA + 1);
    const uint32_t *totals = hist + (size_t)256 * nbA;
    hipLaunchKernelGGL((rs_histogram_kernel<RS_ITEMS, 256, 8>), dim3(nbA), dim3(RS_THREADS), 0, stream, keys, R, low_bits, (1u << high_bits) - 1u, nbA, hist, n_dev);
    hipLaunchKernelGGL(rs_scan_rows_kernel, dim3(1u << high_bits), dim3(256), 0, stream, nbA, hist, 256u);
#define TS_SCATTER_A(NB) hipLaunchKernelGGL((rs_scatter_kernel<RS_ITEMS, 256, 1, NB>), dim3(nbA), dim3(RS_THREADS), 0, stream, keys, vals, packed, (uint32_t *)nullptr, \
        R, low_bits, high_bits, nbA, hist, low_bits, (const uint32_t *)nullptr, 0, (uint2 *)nullptr, n_dev)
    if (high_bits == 6) TS_SCATTER_A(6); else if (high_bits == 7) TS_SCATTER_A(7); else if (high_bits == 8) TS_SCATTER_A(8); else TS_SCATTER_A(0);
#undef TS_SCATTER_A
    hipLaunchKernelGGL(ts_histogram_kernel, dim3(nbB), dim3(RS_THREADS), 0, stream, packed, totals, 1 << high_bits, 32 - low_bits, nbB, histB);
    hipLaunchKernelGGL(rs_scan_rows_kernel, dim3(1u << low_bits), dim3(256), 0, stream, nbB, histB, 256u);
#define TS_SCATTER_B(NB) hipLaunchKernelGGL((rs_scatter_kernel<RS_ITEMS, 256, 2, NB>), dim3(nbB), dim3(RS_THREADS), 0, stream, packed, (const uint32_t *)nullptr, tile_ids_out, point_list, \
        R, 32 - low_bits, low_bits, nbB, histB, low_bits, totals, 1 << high_bits, ranges, (const uint32_t *)nullptr)
    if (low_bits == 7) TS_SCATTER_B(7); else if (low_bits == 8) TS_SCATTER_B(8); else TS_SCATTER_B(0);
#undef TS_SCATTER_B
    return hipGetLastError();
}

// Pass B of the MSD tile sort on its own (round 6: behind the row-segment partition of ex4d_rowsort.hip): `packed` holds, bucket after
// bucket (bucket = tile row, `totals[nbuckets]` instances each, in depth order inside a bucket), one word per instance,
// column << (32 - low_bits) | Gaussian id.  Writes point_list, the tile ranges (tile = bucket * stride + column) and, on request, the tile ids.
// [256][blocks] + totals, then the block table of the row-segment sort (6 words per block, 8-byte aligned: the word count in front of it is even)
size_t ex4d_tile_sort_pass_b_hist_words(uint32_t R) { return (size_t)256 * (rs_num_blocks(R) + 258) + (size_t)6 * (rs_num_blocks(R) + 258); }
size_t ex4d_tile_sort_pass_b_table_offset(uint32_t R) { return (size_t)256 * (rs_num_blocks(R) + 258); }
hipError_t ex4d_tile_sort_pass_b(const uint32_t *packed, const uint32_t *totals, int nbuckets, int low_bits, uint32_t stride, uint32_t R, uint32_t cap,
    uint32_t *histB, uint32_t *point_list, uint32_t *tile_ids_out, uint2 *ranges, hipStream_t stream, const Ex4dTsBlock *block_table)
{
    if (R == 0) return hipSuccess;
    const uint32_t nbB = ex4d_tile_sort_pass_b_blocks(R, nbuckets);
    hipLaunchKernelGGL(ts_histogram_kernel, dim3(nbB), dim3(RS_THREADS), 0, stream, packed, totals, nbuckets, 32 - low_bits, nbB, histB, cap, block_table);
    hipLaunchKernelGGL(rs_scan_rows_kernel, dim3(1u << low_bits), dim3(256), 0, stream, nbB, histB, 256u);
#define TS_SCATTER_B(NB) hipLaunchKernelGGL((rs_scatter_kernel<RS_ITEMS, 256, 2, NB>), dim3(nbB), dim3(RS_THREADS), 0, stream, packed, (const uint32_t *)nullptr, tile_ids_out, point_list, \
        R, 32 - low_bits, low_bits, nbB, histB, low_bits, totals, nbuckets, ranges, (const uint32_t *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, \
        (const uint32_t *)nullptr, stride, cap, block_table)
    if (low_bits == 7) TS_SCATTER_B(7); else if (low_bits == 8) TS_SCATTER_B(8); else TS_SCATTER_B(0);
#undef TS_SCATTER_B
    return hipGetLastError();
}

hipError_t ex4d_launch_scan_tiles(int P, const uint2 *rects, const uint32_t *rects4, const uint32_t *order, uint2 *sorted_rects, uint32_t *sorted_offsets,
    uint32_t *block_sums, int T, uint2 *ranges, uint32_t *frame_total, hipStream_t stream)
{
    const int nb = (P + SCAN_CHUNK - 1) / SCAN_CHUNK;
    hipLaunchKernelGGL(scan_tiles_local_kernel, dim3(nb), dim3(256), 0, stream, P, rects, rects4, order, sorted_rects, sorted_offsets, block_sums, T, ranges, frame_total);
    return hipGetLastError();
}

hipError_t ex4d_launch_duplicate(int P, int W, int H, const uint32_t *order, const uint32_t *sorted_offsets,
    const uint32_t *block_sums, const uint2 *sorted_rects, const uint32_t *sorted_rects4, uint32_t *tile_keys, uint32_t *vals, uint32_t cap, hipStream_t stream,
    const uint32_t *bucket_keys, const uint32_t *dparams, const uint32_t *bucket_sums, uint32_t *frame_total, int msd_bits)
{
    const int gx = (W + EX4D_TILE - 1) / EX4D_TILE, gy = (H + EX4D_TILE - 1) / EX4D_TILE;
    hipLaunchKernelGGL(duplicate_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, gx, gy, order, sorted_offsets, block_sums,
        sorted_rects, sorted_rects4, tile_keys, vals, cap, bucket_keys, dparams, bucket_sums, 1 << msd_bits, frame_total);
    return hipGetLastError();
}

hipError_t ex4d_launch_zero(void *ptr, size_t bytes, hipStream_t stream)
{
    // ptr: 16-byte aligned; bytes: a multiple of 4
    const size_t n16 = bytes / 16;
    const uint32_t tail_words = (uint32_t)((bytes % 16) / 4);
    if (bytes == 0) return hipSuccess;
    size_t blocks = (n16 + 1023) / 1024;                 // 4 stores of 16 bytes per thread
    if (blocks > 8192) blocks = 8192;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (uint4 *)ptr, n16, tail_words);
    return hipGetLastError();
}

hipError_t ex4d_launch_tile_ranges(uint32_t R, int T, const uint32_t *tile_ids, uint2 *ranges, hipStream_t stream, const uint32_t *n_dev)
{
    (void)T;      // the ranges were zeroed by the scan kernel
    if (R > 0)
        hipLaunchKernelGGL(tile_ranges_kernel, dim3((R + 255) / 256), dim3(256), 0, stream, R, tile_ids, ranges, n_dev);
    return hipGetLastError();
}
