// Internal declarations shared by the HIP translation units of libex4d_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/ex4d_rasterizer.h"
#include "../../include/ex4d_attributes.h"

#define EX4D_ALIGN 256

static inline size_t ex4d_align_up(size_t x) { return (x + (EX4D_ALIGN - 1)) & ~(size_t)(EX4D_ALIGN - 1); }

// Typed views over the three opaque scratch buffers (layouts reported by ex4d_*_layout()).
// One 64-byte record per Gaussian, written by preprocess_fwd for visible Gaussians, gathered by the
// compositing kernels (one cache line per gather):
//   float4 #0: mean2D.x, mean2D.y, conic.x (A), conic.y (B)
//   float4 #1: conic.z (C), tau, k1 = -B/C, k2 = -B/A   (cull constants: tau = ln(255 w) + 0.01, +-inf = never / always cull)
//   float4 #2: depth (p_view.z), r, g, b          (SH colour or colors_precomp)
//   float4 #3: dir3D.x, dir3D.y, dir3D.z, w       (per-Gaussian "flow" channel, zeros if absent; w = opacity*coef)
#define EX4D_RECORD_FLOATS 16
#define EX4D_DLS_MSD_BITS 10          // the widest top digit (array sizes); the sort runs with 9 or 10 bits: ex4d_depth_sort_msd_bits()
#define EX4D_CHUNK_FLOW 0x80000000u       // per-chunk count pair, second word: some visible Gaussian of the chunk carries a non-zero dir3D
#define EX4D_CHUNK_FILTERED 0x40000000u   // ... a Gaussian of the chunk failed the frustum test although `prefiltered` was set
#define EX4D_DSUMS_MARK 0x44535553u     // frame flag [3]: the forward stored the SH direction sums (Ex4dParams.prepare_backward)

struct GeomState {
    float4 *records;          // [P][4]
    float *cov3D;
    uint8_t *clamped;
    uint32_t *tiles_touched;
    uint2 *rects;             // [P] tile rect of every Gaussian: .x = x0 | y0 << 16, .y = w | h << 16 (w*h == tiles_touched)
    uint2 *sorted_rects;      // [P] the same in depth order (gathered once by the scan kernel, streamed by duplicate)
    uint32_t *rects4;         // [P] the rects packed into 32 bits (x0 | y0 << 8 | w << 16 | h << 24): the array the scan kernel gathers from when the image has at most 255 x 255 tiles
    uint32_t *depth_order;
    uint32_t *sorted_offsets;
    // scratch used only inside forward (not needed by backward)
    uint32_t *sort_keys_a, *sort_keys_b, *sort_vals_b;   // depth-sort ping-pong
    uint32_t *bucket_sums;                               // MSD depth sort with the fused tile scan: instance count of every depth bucket
    uint2 *key_ranges;                                   // MSD depth sort: (max key, max ~key) of every 64-Gaussian chunk's visible Gaussians
    uint32_t *sort_vals_a, *rects4_b, *bucket_starts;    // MSD depth sort: the ids as preprocess wrote them, the packed rects in depth order, first position of every bucket
    uint32_t *scan_block_sums;                           // per-block totals of the tiles_touched scan
    uint32_t *sort_hist;                                 // radix histogram table for the depth sort
    uint32_t *total;                                     // frame flags (Ex4dFrameStatus): [0] instance count (summed by the tile scan), [1] prefilter violation, [2] some visible Gaussian has dir3D != 0, [3] EX4D_DSUMS_MARK when sh_dsums was written
    uint32_t *block_totals;                              // instance counts of preprocess_fwd per chunk of 64 Gaussians (summed on the host)
    uint32_t *row_hist;                                  // row-segment tile sort (ex4d_rowsort.hip): histogram of its pass A' (sized by P)
    float *sh_dsums;                                     // [P][9] d(colour)/d(direction) sums of the SH backward, left by the forward per-Gaussian kernel on request (Ex4dParams.prepare_backward)
};
struct BinState {
    uint32_t *point_list;     // final sorted Gaussian ids
    uint32_t *tile_ids;       // final sorted tile ids
    uint32_t *vals_tmp, *keys_tmp;   // ping-pong (keys_tmp: the instance words between the two passes of the tile sort)
    uint32_t *sort_hist;
    // The forward compositing kernel leaves, per (tile, quadrant), the COMPACTED list of the entries that survived its quadrant cull and
    // were composited: (Gaussian id, position in the tile list), in list order.  The backward streams these lists back to front
    // instead of walking the tile list and culling again (round 2 kept one survivor bit mask per 64-entry chunk and quadrant: the
    // backward still paid a load + compaction round per chunk for ~4 survivors).
    // Quadrant q of a tile whose range is [r0, r1) owns qlist[4 r0 + q (r1 - r0) ...) (capacity = the list length: every entry could
    // survive); qcount[4 tile + q] entries of it are valid.  Only the valid prefix is ever touched (~6 % of 4 R entries).
    uint32_t *qlist;          // (round 6: list positions only -- the id is point_list[range.x + position])
    uint32_t *qcount;
};
struct ImgState {
    float *final_T;
    uint32_t *n_contrib;
    uint2 *ranges;
};

// radix sort geometry (shared by size computations and kernels)
#define RS_THREADS 256
#define RS_ITEMS 16                       // items per thread per block
#define RS_CHUNK (RS_THREADS * RS_ITEMS)  // 4096 items per block
#define RS_BINS 256
#define SCAN_CHUNK 2048                   // items per block in the tiles_touched scan

static inline uint32_t rs_num_blocks(uint32_t n) { return (n + RS_CHUNK - 1) / RS_CHUNK; }
// Pass B of the tile sort cuts every bucket (tile row) into blocks of <= RS_CHUNK words, no block straddling two buckets.  One record per block;
// round 6: written ONCE per frame by workgroup 0 of the row partition's scatter kernel (ex4d_rowsort.hip) instead of re-derived by every
// workgroup of both pass-B kernels from the bucket totals (two scans, three barriers and a search each).
struct Ex4dTsBlock { uint32_t start, count, bucket, fb_first, fb_next, bucket_start; };
static inline uint32_t ex4d_tile_sort_pass_b_blocks(uint32_t R, int nbuckets) { return rs_num_blocks(R) + (uint32_t)nbuckets + 1u; }      // upper bound = the pass-B grids

// SH coefficients as the model stores them (include/ex4d_rasterizer.h: Ex4dSplitSH); all pointers null = one [P,M,3] tensor
struct ShSplit { const float *dc[2]; const float *rest[2]; int n_static; };
struct ShSplitGrad { float *dc[2]; float *rest[2]; int n_static; };

// ---- launchers (each launches on `stream`, returns hipGetLastError()) -------------------------
hipError_t ex4d_launch_preprocess_fwd(const Ex4dParams &prm, const float *means3D, const float *dir3D, const float *scales,
    const float *rotations, const float *opacities, const float *shs, const float *cov3D_precomp,
    const float *colors_precomp, const float *viewmatrix, const float *projmatrix, const float *campos,
    int32_t *radii, GeomState g, uint32_t *prefilter_violation, ShSplit split,
    uint32_t *depth_keys, uint32_t *depth_vals, uint32_t depth_key_base, uint32_t depth_key_invisible, uint32_t *rects4, hipStream_t stream,
    uint32_t *key_range_slots = nullptr, bool global_flags = true);      // global_flags: also raise the frame-flag words [1] / [2] (else the flags travel in the per-chunk count pairs only); key_range_slots: uint2[(P + 63) / 64] <- (max, max(~)) of every wave's visible depth keys (MSD depth sort)

hipError_t ex4d_launch_mark_visible(int P, const float *means3D, const float *viewmatrix, const float *projmatrix,
    float min_depth, float max_depth, uint8_t *present, hipStream_t stream);

hipError_t ex4d_launch_preprocess_bwd(const Ex4dParams &prm, const float *means3D, const int32_t *radii,
    const float *shs, const float *scales, const float *rotations, const float *cov3D_ptr,
    const float *viewmatrix, const float *projmatrix, const float *campos, GeomState g, const float *acc16,
    float *dL_dmeans2D, float *dL_dcolors, float *dL_dopacity, float *dL_dmeans3D, float *dL_dcov3D, float *dL_dsh,
    float *dL_dscales, float *dL_drotations, float *dL_ddir, ShSplit split, ShSplitGrad gsplit, hipStream_t stream);

// stable LSD radix sort of (key,value) uint32 pairs over key bits [0, end_bit); result lands in
// (keys_out, vals_out) which must be one of the two ping-pong pairs; returns which through *result_in_a.
// n_dev (optional): the item count in device memory, n then is the capacity the grids are sized for (asynchronous forward)
hipError_t ex4d_radix_sort_pairs(uint32_t *keys_a, uint32_t *vals_a, uint32_t *keys_b, uint32_t *vals_b,
    uint32_t n, int end_bit, uint32_t *hist, bool *result_in_a, hipStream_t stream, const uint32_t *n_dev = nullptr, bool iota_values = false,
    const uint32_t *gather_in = nullptr, uint32_t *gather_out = nullptr, uint2 *zero_ranges = nullptr, int num_ranges = 0);      // last pass: gather_out[position] = gather_in[value]; zero_ranges[0 .. num_ranges) cleared      // iota_values: the input values are 0 .. n-1 (vals_a is not read by the first pass)
size_t ex4d_radix_hist_words(uint32_t n);
int ex4d_radix_passes(uint32_t n, int end_bit);      // number of passes ex4d_radix_sort_pairs will run (decides where the result lands)

// MSD-first tile sort on packed words (ex4d_binning.hip); writes point_list, the tile ranges and, on request, the sorted tile ids
bool ex4d_tile_sort_msd_applies(int P, int tile_bits);
size_t ex4d_tile_sort_hist_words(uint32_t R, int tile_bits);
hipError_t ex4d_tile_sort_msd(const uint32_t *keys, const uint32_t *vals, uint32_t *packed, uint32_t *point_list, uint32_t *tile_ids_out,
    uint32_t R, int tile_bits, uint32_t *hist, uint2 *ranges, hipStream_t stream, const uint32_t *n_dev = nullptr);

// Tile sort at row-segment granularity (ex4d_rowsort.hip, round 6): point_list + tile ranges straight from the rects in depth order --
// no instance offsets, no (tile, id) pairs.  Images of at most 255 x 255 tiles, P <= 2^24.
bool ex4d_tile_sort_rows_applies(int P, int gx, int gy);
size_t ex4d_tile_sort_rows_geom_words(uint32_t P);                      // histogram of pass A' (geometry buffer)
size_t ex4d_tile_sort_rows_hist_words(uint32_t R);                      // histogram of pass B (binning buffer)
void ex4d_set_rows_probe(int v);      // developer profile of pass A' (cycles per phase)
hipError_t ex4d_rows_prof(unsigned long long *out8, int reset);
hipError_t ex4d_tile_sort_rows(int P, int gx, int gy, const uint32_t *order, const uint32_t *r4, const uint2 *r8, uint32_t *words,
    uint32_t *point_list, uint32_t *tile_ids_out, uint32_t cap, uint32_t S, uint32_t *histA, uint32_t *histB, uint2 *ranges, uint32_t *frame_total, hipStream_t stream);

hipError_t ex4d_launch_scan_tiles(int P, const uint2 *rects, const uint32_t *rects4, const uint32_t *order, uint2 *sorted_rects, uint32_t *sorted_offsets,
    uint32_t *block_sums, int T, uint2 *ranges, uint32_t *frame_total, hipStream_t stream);
hipError_t ex4d_launch_duplicate(int P, int W, int H, const uint32_t *order, const uint32_t *sorted_offsets,
    const uint32_t *block_sums, const uint2 *sorted_rects, const uint32_t *sorted_rects4, uint32_t *tile_keys, uint32_t *vals, uint32_t cap, hipStream_t stream,
    const uint32_t *bucket_keys = nullptr, const uint32_t *dparams = nullptr, const uint32_t *bucket_sums = nullptr, uint32_t *frame_total = nullptr,
    int msd_bits = EX4D_DLS_MSD_BITS);      // msd_bits: the digit width the MSD depth sort ran with (fused tile scan: bucket bases)

// MSD-first depth sort (ex4d_binning.hip): one global partition on the top digit of (key - smallest visible key), every bucket finished in LDS.
// Frame-flag words used by it (GeomState::total): [EX4D_FLAG_DPARAMS ..+3] = {kmin, shift, invisible key, 0} written by its range kernel
#define EX4D_FLAG_DPARAMS 8
#define EX4D_FLAG_WORDS 64
bool ex4d_depth_sort_msd_applies(uint32_t n, int key_bits);
hipError_t ex4d_depth_sort_msd(uint32_t *ka, uint32_t *va, uint32_t *ra, uint32_t *kb, uint32_t *vb, uint32_t *rb, uint32_t n, uint32_t inv_key,
    uint32_t *flags, const uint2 *wave_ranges, uint32_t *hist, uint32_t *starts, uint32_t local_cap, hipStream_t stream,
    uint32_t *local_incl, uint32_t *bucket_sums, int T, uint2 *ranges, int local_threads,       // local_incl / bucket_sums: the tile scan fused into the bucket kernel (nullptr = not)
    uint32_t *watch = nullptr,                                                                   // watch: pinned host word set to 1 when a bucket exceeded the LDS capacity
    int msd_bits = EX4D_DLS_MSD_BITS);                                                           // EX4D_DLS_MSD_BITS or one less: ex4d_depth_sort_msd_bits()
int ex4d_depth_sort_msd_bits(uint32_t n, int key_bits);      // digit width by Gaussian count (9 bits up to 1.3 M, 10 beyond)
// stable ranking by LDS atomics in the scatter kernels (ex4d_binning.hip): probed once per device, option "rank_lds_atomics"
void ex4d_set_rank_lds(int v);
int ex4d_get_rank_lds();
int ex4d_rank_lds_in_use();
hipError_t ex4d_prepare_rank_lds(hipStream_t stream);
hipError_t ex4d_launch_zero(void *ptr, size_t bytes, hipStream_t stream);      // ptr 16-byte aligned, bytes a multiple of 4 (a kernel, not hipMemsetAsync: ex4d_binning.hip)
hipError_t ex4d_launch_tile_ranges(uint32_t R, int T, const uint32_t *tile_ids, uint2 *ranges, hipStream_t stream, const uint32_t *n_dev = nullptr);

void ex4d_set_preprocess_tune(int v);   // ex4d_preprocess.hip: 1 (default) = SH rows of frustum-culled Gaussians are not requested
int ex4d_get_preprocess_tune();
void ex4d_set_preprocess_fast(int v);   // 1 (default) = frames with one [P,16,3] SH tensor at degree 3 and scale + rotation take the specialised kernel
int ex4d_get_preprocess_fast();
void ex4d_set_fwd_asm(int on);           // compositing forward: hand-scheduled entry walk (default) or the compiler's loop
int ex4d_get_fwd_asm();
void ex4d_set_clamp_always(int on);     // compositing: 1 = evaluate min(0.99, w G) everywhere (rounds 1-5), 0 (default) = only where w > 0.99 can reach it
int ex4d_get_clamp_always();
void ex4d_set_bwd_pairs(int on);        // compositing backward: two pixels per lane on packed math (default 1)
int ex4d_get_bwd_pairs();
hipError_t ex4d_launch_composite_fwd(const Ex4dParams &prm, const uint2 *ranges, const uint32_t *point_list,
    const float *subpixel_offset, const float4 *records, const float *bg, float *final_T, uint32_t *n_contrib,
    float *out_color, float *out_depth, float *out_acc, float *out_flow, int32_t *out_idx, uint32_t *qlist, uint32_t *qcount,
    bool has_flow, hipStream_t stream);

hipError_t ex4d_launch_composite_bwd(const Ex4dParams &prm, const uint2 *ranges, const uint32_t *point_list,
    const float *subpixel_offset, const float *bg, const float4 *records, const float *out_depth, const float *out_acc,
    const float *final_T, const uint32_t *n_contrib, const float *dL_dpix, const float *dL_ddepth,
    const float *dL_dflow, const float *dL_dacc, float *acc16, const uint32_t *qlist, const uint32_t *qcount,
    int variant, hipStream_t stream);

// developer statistics of the scan compositing backward (variant 8)
hipError_t ex4d_bwd_stats(unsigned long long *out, int count, int reset);
