// Tile sort at ROW-SEGMENT granularity for gfx950 (round 6): the first half of the tile sort without (tile, id) pairs.
//
// Replaces, together with the depth sort and pass B of ex4d_binning.hip (CR/ = submodules/diff_gaussian_rasterization_df/cuda_rasterizer/):
//   duplicateWithKeys                  CR/rasterizer_impl.cu:72-113
//   cub::DeviceRadixSort::SortPairs    CR/rasterizer_impl.cu:321-326
//   identifyTileRanges (+ cudaMemset)  CR/rasterizer_impl.cu:118-140, :328
//
// Rounds 2-5 emitted one (tile, id) pair per instance in depth order (duplicate_kernel: 8 bytes x R written) and sorted the R pairs by
// tile id in two passes (pass A by the high digit, pass B by the low digit): both passes rank EVERY instance with wave ballots, 9.3
// instances per Gaussian at BASELINE config 3, 38 at config 5.  A Gaussian's tile rect is w x h tiles: h ROW SEGMENTS of w consecutive
// tiles.  The same stable order falls out of
//   pass A'  stable partition of the S = sum(h) row segments, generated on the fly from the rects in depth order, by their tile ROW
//            (<= 255 rows): histogram -> row scan -> scatter.  It ranks 2.7 segments per Gaussian instead of 9.3 instances, with
//            coverage masks instead of ranking ballots (below), and the write-out of a workgroup expands every staged segment into its
//            w instance words  column << (32 - column bits) | id  -- the input format of
//   pass B   (ex4d_binning.hip: ex4d_tile_sort_pass_b, the second pass of rounds 2-5 with bucket = tile row, digit = tile column):
//            per tile row, stable counting sort of the instance words by column; the (row, column) counts are the tile ranges.
// Stable by row, then stable by column inside a row, both in depth order = the reference's stable order by (tile | depth).
// No instance offsets (no tile scan), no duplication kernel, no key / value arrays.  Histograms are difference arrays: a Gaussian
// covering rows [y0, y0 + h) adds +1 at y0 and -1 at y0 + h (two LDS atomics instead of h).
// Measured and dropped on the way (round 6): pass B' on segments (every workgroup expanding 1024 segments in LDS: 52 us
// against pass B's 27 -- the expansion, by owner search or by coverage masks, costs more than ranking ready-made words).
// Applies when the image has at most 255 x 255 tiles (the rects travel packed) and P <= 2^24.
#include "ex4d_internal.h"

namespace {

#define ROW_THREADS 256
#define ROWA_GAUSS 256            // Gaussians per workgroup of pass A'
#define ROWA_HINTS 1024           // entries of the word -> segment hint table of its write-out
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

// the same for two values at once (tmp: 8 words)
__device__ __forceinline__ void block_excl_scan2(uint32_t a, uint32_t b, uint32_t *tmp, uint32_t &ea, uint32_t &eb, uint32_t &ta, uint32_t &tb)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t xa = wave_incl_scan(a, lane), xb = wave_incl_scan(b, lane);
    if (lane == 63) { tmp[wave] = xa; tmp[4 + wave] = xb; }
    __syncthreads();
    ea = xa - a; eb = xb - b; ta = 0; tb = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { const uint32_t t = tmp[w], u = tmp[4 + w]; if (w < wave) { ea += t; eb += u; } ta += t; tb += u; }
    __syncthreads();
}

// a wave's difference array (cnt[d] += 1 at the first digit of an item, -= 1 behind its last) -> counts per digit, in place.
// Lane l owns the PER consecutive digits PER l ... (NDIG = 64 PER digits)
template <int NDIG>
__device__ __forceinline__ void wave_diff_to_counts(uint32_t *cnt, int lane)
{
    constexpr int PER = NDIG / 64;
    uint32_t v[PER], run = 0;
#pragma unroll
    for (int q = 0; q < PER; q++) { run += cnt[PER * lane + q]; v[q] = run; }
    const uint32_t excl = wave_incl_scan(run, lane) - run;
#pragma unroll
    for (int q = 0; q < PER; q++) cnt[PER * lane + q] = v[q] + excl;
}

// ---------------------------------------------------------------- pass A': histogram
// hist: rows 0 .. NR-1 = segments per (tile row, block), rows NR .. 2 NR - 1 = instances per (tile row, block); [2 NR][nblocks] + totals
// (every wave counts one block of ROWA_GAUSS Gaussians on its own -- wave-private difference arrays, no workgroup barrier: the kernel is
// a few LDS atomics per Gaussian behind one memory round trip, what it costs is workgroups in flight and launches)
__global__ __launch_bounds__(ROW_THREADS) void rows_seg_hist_kernel(uint32_t P, int NR, const uint32_t *__restrict__ r4, const uint2 *__restrict__ r8,
    uint32_t *__restrict__ hist, uint32_t nblocks, uint32_t *__restrict__ frame_total)
{
    __shared__ __attribute__((aligned(16))) uint32_t d_seg[4][ROW_PAD], d_inst[4][ROW_PAD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t blk = blockIdx.x * 4 + wave;
    if (blk >= nblocks) return;              // (a whole wave)
    uint32_t *ds = d_seg[wave], *di = d_inst[wave];
    RectU r[ROWA_GAUSS / 64];
#pragma unroll
    for (int it = 0; it < ROWA_GAUSS / 64; it++) {
        const uint32_t k = blk * ROWA_GAUSS + it * 64 + lane;
        r[it] = { 0u, 0u, 0u, 0u };
        if (k < P) r[it] = rect_at(r4, r8, k);
    }
    for (int i = lane; i < ROW_PAD; i += 64) { ds[i] = 0u; di[i] = 0u; }
    wsync();
    uint32_t inst = 0;
#pragma unroll
    for (int it = 0; it < ROWA_GAUSS / 64; it++) {
        if (r[it].w * r[it].h != 0u) {
            atomicAdd(&ds[r[it].y0], 1u); atomicAdd(&ds[r[it].y0 + r[it].h], 0xFFFFFFFFu);
            atomicAdd(&di[r[it].y0], r[it].w); atomicAdd(&di[r[it].y0 + r[it].h], 0u - r[it].w);
            inst += r[it].w * r[it].h;
        }
    }
    wsync();
    wave_diff_to_counts<256>(ds, lane);
    wave_diff_to_counts<256>(di, lane);
    wsync();
    for (int row = lane; row < NR; row += 64) {
        hist[(size_t)row * nblocks + blk] = ds[row];
        hist[(size_t)(NR + row) * nblocks + blk] = di[row];
    }
    (void)inst; (void)frame_total;      // (the device-side instance count of the asynchronous forward is summed by the row scan below: one atomic per
                                        // tile row instead of one per block -- 3907 atomics on ONE word cost this kernel 25 us at 1.0 M Gaussians)
}

// one workgroup per histogram row: exclusive scan of its `len` per-block counts, row total to hist[nrows * len + row]
// frame_total (optional): += the totals of the rows first_total_row .. nrows - 1 (the instance rows: the frame's instance count)
__global__ __launch_bounds__(256) void rows_scan_kernel(uint32_t len, uint32_t *__restrict__ hist, uint32_t nrows, uint32_t *__restrict__ frame_total, uint32_t first_total_row)
{
    __shared__ uint32_t wave_sums[4];
    __shared__ uint32_t carry_s;
    uint32_t *row = hist + (size_t)blockIdx.x * len;
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
        hist[(size_t)nrows * len + blockIdx.x] = carry_s;
        if (frame_total && blockIdx.x >= first_total_row && carry_s != 0u) atomicAdd(frame_total, carry_s);
    }
}

// ---------------------------------------------------------------- placement by coverage masks
// Lanes = the 64 items of a round (Gaussians in pass A', segments in pass B'); item l covers the digits [d0, d0 + len).  Every lane ORs
// its bit into cov[d] of the digits it covers (loop over j < len: no lane waits for another, the LDS atomics return nothing), then
// the j-th element of item l goes to slot  cnt[d] + popcount(cov[d] & lanes below l)  -- the elements of one digit in item order,
// behind everything earlier rounds placed there: a stable counting sort without ranking ballots (4 VALU per element instead of 5 per
// digit bit) and without looking up which item owns an output slot.  Afterwards the counters take popcount(cov[d]) and the masks are
// cleared.  Both loops run max(len) times: rounds with one long item among short ones lose lanes (rects are mostly 2-4 tiles wide / high).
struct CovLds { uint2 cov[4][ROW_PAD]; };
struct LaneBits { uint32_t bit_lo, bit_hi, lt_lo, lt_hi; };
__device__ __forceinline__ LaneBits lane_bits(int lane)
{
    // (32-bit shifts only: a 64-bit shift by a per-lane amount is the gfx950 last-register hazard build.py looks for)
    LaneBits b;
    const uint32_t one = 1u << (lane & 31);
    b.bit_lo = lane < 32 ? one : 0u; b.bit_hi = lane < 32 ? 0u : one;
    b.lt_lo = lane < 32 ? one - 1u : 0xFFFFFFFFu; b.lt_hi = lane < 32 ? 0u : one - 1u;
    return b;
}
__device__ __forceinline__ void cov_publish(uint2 *cov, uint32_t d0, uint32_t len, const LaneBits &lb)
{
    for (uint32_t j = 0; __builtin_amdgcn_ballot_w64(j < len) != 0ull; j++)
        if (j < len) { if (lb.bit_lo) atomicOr(&cov[d0 + j].x, lb.bit_lo); else atomicOr(&cov[d0 + j].y, lb.bit_hi); }
    wsync();
}
// counters += items per digit, masks cleared (digits below ndig, a multiple of 64)
__device__ __forceinline__ void cov_retire(uint2 *cov, uint32_t *cnt, int ndig, int lane)
{
    wsync();
    for (int c = lane; c < ndig; c += 64) {
        const uint2 m = cov[c];
        const uint32_t k = __popc(m.x) + __popc(m.y);
        if (k != 0u) { cnt[c] += k; cov[c] = make_uint2(0u, 0u); }
    }
    wsync();
}

// ---------------------------------------------------------------- pass A': scatter
// Workgroup b owns Gaussians [b ROWA_GAUSS, (b + 1) ROWA_GAUSS) of the depth order, wave w the w-th 64 of them; the stable order inside
// the block is (wave, lane, row) = (Gaussian, row).  Sweep 1 counts the wave's segments per row (difference array), one barrier gives
// every (row, wave) its first block-local slot, sweep 2 places the wave's segments (coverage masks, above) in block-local sorted
// order: stage[p] = (id, x0 | w << 8 | row << 16).  An exclusive scan of the widths over the staged segments (winc) gives every
// segment its first instance inside the block; the block's instance words then leave in block order, neighbouring threads writing
// neighbouring words of a row's run: thread q finds the segment of word q by a binary search in winc (10 LDS reads; expanding the
// segments into an LDS buffer first cost 15 KB of LDS per workgroup -- 3 workgroups per CU -- and was slower: every phase of this
// kernel is a short chain of dependent LDS round trips, what pays is the number of workgroups in flight).  A row's run starts at the
// row's first instance + the instances earlier blocks put into the row (both from the histogram's second half).
// Measured on the way (round 6, 1.0 M Gaussians): every thread writing the w words of its segments straight to memory 40 us; expansion
// through LDS with a per-word row tag 46 us; persistent workgroups prefetching their next block 54 us; this version: see DESIGN.md.
// A block with more than CAP segments takes a sequential path: thread = tile row, the Gaussians one after the other.
__device__ unsigned long long g_rows_prof[8];      // developer profile (option "rows_probe"): cycles per phase summed over blocks, [7] = blocks
template <int NRP, int CAP>
__global__ __launch_bounds__(ROW_THREADS) void rows_seg_scatter_kernel(uint32_t P, int NR, const uint32_t *__restrict__ order,
    const uint32_t *__restrict__ r4, const uint2 *__restrict__ r8, const uint32_t *__restrict__ hist, uint32_t nblocks,
    uint32_t *__restrict__ words, uint32_t cap, int shift, int probe, Ex4dTsBlock *__restrict__ block_table, uint32_t table_blocks)
{
    constexpr int PAD = NRP + 4;
    __shared__ __attribute__((aligned(16))) uint32_t wave_cnt[4][PAD];
    __shared__ __attribute__((aligned(16))) uint2 wave_cov[4][PAD];
    __shared__ uint32_t local_start[NRP], inst_base[NRP], row_word[NRP];
    __shared__ uint32_t tmp[8];
    __shared__ uint32_t s_total, s_words;
    __shared__ uint2 stage[CAP];
    __shared__ uint32_t winc[CAP + 256];
    __shared__ uint16_t hint[ROWA_HINTS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long ts[8];
#define ROWS_TS(i) do { if (probe) ts[i] = __builtin_amdgcn_s_memtime(); } while (0)
    ROWS_TS(0);
    const uint32_t k = blockIdx.x * ROWA_GAUSS + tid;
    RectU rc = { 0u, 0u, 0u, 0u };
    uint32_t id = 0u;
    if (k < P) { rc = rect_at(r4, r8, k); id = order[k]; }
    if (rc.w * rc.h == 0u) rc.h = 0u;
    // (the global numbers of the barrier phase are requested before the counting sweep)
    const uint32_t it = tid < NR ? hist[(size_t)2 * NR * nblocks + NR + tid] : 0u;          // instances of tile row t in the frame
    const uint32_t ipre = tid < NR ? hist[(size_t)(NR + tid) * nblocks + blockIdx.x] : 0u;  // ... of them in earlier blocks
    uint32_t *cnt = wave_cnt[wave];
    uint2 *cov = wave_cov[wave];
    for (int i = lane; i < PAD; i += 64) { cnt[i] = 0u; cov[i] = make_uint2(0u, 0u); }
    wsync();
    if (rc.h != 0u) { atomicAdd(&cnt[rc.y0], 1u); atomicAdd(&cnt[rc.y0 + rc.h], 0xFFFFFFFFu); }
    wsync();
    wave_diff_to_counts<NRP>(cnt, lane);
    ROWS_TS(1);
    __syncthreads();
    {
        // thread t = tile row t: exclusive scan over the waves, block-local start of the row's segments, first instance of the row's run
        uint32_t c[4], tot = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) { c[w] = tid < NR ? wave_cnt[w][tid] : 0u; tot += c[w]; }
        uint32_t total, itotal, ls, is;
        block_excl_scan2(tot, it, tmp, ls, is, total, itotal);
        if (block_table && blockIdx.x == 0) {              // (workgroup-uniform)
            // Pass B's block table (Ex4dTsBlock, ex4d_internal.h): thread t = tile row t knows the row's instances (`it`) and its first output
            // position (`is`); an exclusive scan of the rows' block counts gives the row's first block.  Records behind the last block: empty.
            const uint32_t nblk = tid < NR ? (it + RS_CHUNK - 1u) / RS_CHUNK : 0u;
            uint32_t fb_total;
            const uint32_t fb = block_excl_scan(nblk, tmp, fb_total);
            for (uint32_t j = 0; j < nblk; j++) {
                const uint32_t within = j * RS_CHUNK;
                if (fb + j < table_blocks)            // (an asynchronous frame beyond its capacity: invalid anyway, but nothing is written out of bounds)
                    block_table[fb + j] = { is + within, (it - within) < RS_CHUNK ? (it - within) : (uint32_t)RS_CHUNK, (uint32_t)tid, fb, fb + nblk, is };
            }
            for (uint32_t e = fb_total + tid; e < table_blocks; e += ROW_THREADS) block_table[e] = { 0u, 0u, 0u, 0u, 0u, 0u };
        }
        if (tid < NRP) { local_start[tid] = ls; inst_base[tid] = is + ipre; }      // (rows >= NR: the block's total -- the end of the last row's run)
        if (tid < NR) { wave_cnt[0][tid] = ls; wave_cnt[1][tid] = ls + c[0]; wave_cnt[2][tid] = ls + c[0] + c[1]; wave_cnt[3][tid] = ls + c[0] + c[1] + c[2]; }
        if (tid == 0) s_total = total;
    }
    __syncthreads();
    const uint32_t total = s_total;
    ROWS_TS(2);
    if (total == 0u) return;                 // (uniform)
    if (total > CAP) {
        // sequential path (uniform): thread t = tile row t keeps the row's write position, the block's Gaussians (rect + id, through LDS)
        // pass by in depth order
        stage[tid] = make_uint2(rc.x0 | (rc.y0 << 8) | (rc.w << 16) | (rc.h << 24), id);      // (h = 0: no instances)
        __syncthreads();
        if (tid < NR) {
            uint32_t pos = inst_base[tid];
            for (int kk = 0; kk < ROWA_GAUSS; kk++) {
                const uint2 e = stage[kk];
                const uint32_t x0 = e.x & 0xFFu, y0 = (e.x >> 8) & 0xFFu, w = (e.x >> 16) & 0xFFu, h = e.x >> 24;
                if ((uint32_t)tid - y0 < h) {
                    for (uint32_t j = 0; j < w; j++) if (pos + j < cap) words[pos + j] = ((x0 + j) << shift) | e.y;
                    pos += w;
                }
            }
        }
        return;
    }
    {
        const LaneBits lb = lane_bits(lane);
        const uint32_t h = rc.h, y0 = rc.y0;
        if (__builtin_amdgcn_ballot_w64(h != 0u) != 0ull) {      // (uniform per wave)
            cov_publish(cov, y0, h, lb);
            const uint32_t xw = rc.x0 | (rc.w << 8);
            for (uint32_t j0 = 0; __builtin_amdgcn_ballot_w64(j0 < h) != 0ull; j0 += 2) {
                uint2 m[2]; uint32_t c[2];
#pragma unroll
                for (int u = 0; u < 2; u++) { const uint32_t i = j0 + u < h ? y0 + j0 + u : 0u; m[u] = cov[i]; c[u] = cnt[i]; }
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    if (j0 + u < h) {
                        const uint32_t ty = y0 + j0 + u;
                        const uint32_t slot = c[u] + __popc(m[u].x & lb.lt_lo) + __popc(m[u].y & lb.lt_hi);
                        stage[slot] = make_uint2(id, xw | (ty << 16));
                    }
                }
            }
        }
    }
    ROWS_TS(3);
    __syncthreads();                        // every segment staged
    {
        // exclusive scan of the widths over the staged segments: thread t owns the KP consecutive segments t KP ...
        constexpr int KP = CAP / ROW_THREADS;
        uint32_t wv[KP], sum = 0;
#pragma unroll
        for (int q = 0; q < KP; q++) { const uint32_t p = tid * KP + q; wv[q] = p < total ? (stage[p].y >> 8) & 0xFFu : 0u; sum += wv[q]; }
        uint32_t all;
        uint32_t run = block_excl_scan(sum, tmp, all);
        // hint[b] = the segment that holds word b << sh (sh: the block's words fit ROWA_HINTS buckets; 3 unless the rects are wide), so that the
        // search below starts sh steps from its end.  Every bucket boundary lies in exactly one segment: one writer per entry.
        int sh = 3;
        while (((all - 1u) >> sh) >= ROWA_HINTS) sh++;
#pragma unroll
        for (int q = 0; q < KP; q++) {
            const uint32_t p = tid * KP + q;
            winc[p] = p < total ? run : 0xFFFFFFFFu;      // (behind the last segment: never <= a word index)
            if (wv[q] != 0u) for (uint32_t bkt = (run + (1u << sh) - 1u) >> sh; bkt <= (run + wv[q] - 1u) >> sh; bkt++) hint[bkt] = (uint16_t)p;
            run += wv[q];
        }
        winc[CAP + tid] = 0xFFFFFFFFu;
        if (tid == 0) s_words = all;
    }
    __syncthreads();
    const uint32_t nwords = s_words;
    ROWS_TS(4);
    // first word of every row's run inside the block (rows without segments: the next row's; behind the last row: the block's total)
    if (tid < NRP) { const uint32_t ls = local_start[tid]; row_word[tid] = ls < total ? winc[ls] : nwords; }
    __syncthreads();
    ROWS_TS(5);
    int sh = 3;
    while (((nwords - 1u) >> sh) >= ROWA_HINTS) sh++;
    // word q of the block belongs to the last segment p with winc[p] <= q: at most 2^sh - 1 segments behind hint[q >> sh]
    // (four words per thread and trip: four independent chains of dependent LDS reads in flight)
    for (uint32_t q0 = tid; q0 < nwords; q0 += 4 * ROW_THREADS) {
        uint32_t q[4], p[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { q[u] = q0 + u * ROW_THREADS; p[u] = hint[(q[u] < nwords ? q[u] : q0) >> sh]; }
        for (uint32_t step = 1u << (sh - 1); step > 0; step >>= 1) {
            uint32_t wv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) wv[u] = winc[p[u] + step];
#pragma unroll
            for (int u = 0; u < 4; u++) if (wv[u] <= q[u]) p[u] += step;
        }
        uint2 seg[4]; uint32_t w0[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { seg[u] = stage[p[u]]; w0[u] = winc[p[u]]; }
        uint32_t ib[4], rw[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t ty = seg[u].y >> 16; ib[u] = inst_base[ty]; rw[u] = row_word[ty]; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t dst = ib[u] + (q[u] - rw[u]);
            if (q[u] < nwords && dst < cap) words[dst] = (((seg[u].y & 0xFFu) + (q[u] - w0[u])) << shift) | seg[u].x;
        }
    }
    ROWS_TS(6);
    if (probe && tid == 0) { for (int i = 0; i < 6; i++) atomicAdd(&g_rows_prof[i], ts[i + 1] - ts[i]); atomicAdd(&g_rows_prof[7], 1ull); }
#undef ROWS_TS
}

static inline int bits_for(int n) { int b = 1; while ((1 << b) < n) b++; return b; }      // digits 0 .. n-1

}  // namespace

hipError_t ex4d_tile_sort_pass_b(const uint32_t *packed, const uint32_t *totals, int nbuckets, int low_bits, uint32_t stride, uint32_t R, uint32_t cap,
    uint32_t *histB, uint32_t *point_list, uint32_t *tile_ids_out, uint2 *ranges, hipStream_t stream, const Ex4dTsBlock *block_table);
size_t ex4d_tile_sort_pass_b_hist_words(uint32_t R);
size_t ex4d_tile_sort_pass_b_table_offset(uint32_t R);

static int g_rows_probe = 0;
void ex4d_set_rows_probe(int v) { g_rows_probe = v; }
hipError_t ex4d_rows_prof(unsigned long long *out8, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_rows_prof), 8 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) { unsigned long long z[8] = { 0 }; e = hipMemcpyToSymbol(HIP_SYMBOL(g_rows_prof), z, sizeof(z)); }
    return e;
}
bool ex4d_tile_sort_rows_applies(int P, int gx, int gy) { return gx <= 255 && gy <= 255 && (uint32_t)P <= (1u << 24); }
// pass A' lives in the geometry buffer (sized by P alone: <= 255 rows, blocks of >= 256 Gaussians): [2 gy][nbA] + 2 gy totals
size_t ex4d_tile_sort_rows_geom_words(uint32_t P) { return (size_t)2 * 255 * ((P + 255) / 256 + 1) + 64; }
size_t ex4d_tile_sort_rows_hist_words(uint32_t R) { return ex4d_tile_sort_pass_b_hist_words(R); }

// order / (r4 | r8): Gaussian ids and tile rects in depth order; words: cap words of scratch; point_list: cap entries; R: the frame's
// instance count (synchronous forward) or the capacity; S: its number of row segments if known (0: not); ranges: zeroed by the
// caller (tiles without instances are never written); frame_total (optional): += the frame's instance count
hipError_t ex4d_tile_sort_rows(int P, int gx, int gy, const uint32_t *order, const uint32_t *r4, const uint2 *r8, uint32_t *words,
    uint32_t *point_list, uint32_t *tile_ids_out, uint32_t cap, uint32_t S, uint32_t *histA, uint32_t *histB, uint2 *ranges, uint32_t *frame_total, hipStream_t stream)
{
    if (P <= 0 || cap == 0) return hipSuccess;
    const uint32_t nbA = ((uint32_t)P + ROWA_GAUSS - 1) / ROWA_GAUSS;
    uint32_t *totA = histA + (size_t)2 * gy * nbA;
    const int bx = bits_for(gx);
    hipLaunchKernelGGL(rows_seg_hist_kernel, dim3((nbA + 3) / 4), dim3(ROW_THREADS), 0, stream, (uint32_t)P, gy, r4, r8, histA, nbA, frame_total);
    hipLaunchKernelGGL(rows_scan_kernel, dim3(2 * gy), dim3(256), 0, stream, nbA, histA, (uint32_t)(2 * gy), frame_total, (uint32_t)gy);
    // rects more than 3.5 rows high on average (config 5: 6): the larger segment stage
    const bool big = S == 0 ? (uint64_t)cap > 20ull * (uint32_t)P : 2ull * S > 7ull * (uint32_t)P;
    // pass B's block table behind its histogram (same buffer): written by workgroup 0 of the scatter kernel below
    Ex4dTsBlock *block_table = reinterpret_cast<Ex4dTsBlock *>(histB + ex4d_tile_sort_pass_b_table_offset(cap));
    const uint32_t table_blocks = ex4d_tile_sort_pass_b_blocks(cap, gy);
#define ROWS_A(NRP, CAP) hipLaunchKernelGGL((rows_seg_scatter_kernel<NRP, CAP>), dim3(nbA), dim3(ROW_THREADS), 0, stream, (uint32_t)P, gy, order, r4, r8, (const uint32_t *)histA, nbA, \
        words, cap, 32 - bx, g_rows_probe, block_table, table_blocks)
    if (gy <= 64) { if (big) ROWS_A(64, 2048); else ROWS_A(64, 1024); }
    else if (gy <= 128) { if (big) ROWS_A(128, 2048); else ROWS_A(128, 1024); }
    else { if (big) ROWS_A(256, 2048); else ROWS_A(256, 1024); }
#undef ROWS_A
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return ex4d_tile_sort_pass_b(words, totA + gy, gy, bx, (uint32_t)gx, cap, cap, histB, point_list, tile_ids_out, ranges, stream, block_table);
}
