/*
 * ex4d_rasterizer.h -- C ABI of the MI355X-native (gfx950) differentiable 4D-Gaussian rasterizer.
 *
 * Drop-in boundary for the reference's native layer
 *   CudaRasterizer::Rasterizer::{forward, backward, markVisible}
 *     (submodules/diff_gaussian_rasterization_df/cuda_rasterizer/rasterizer.h:20-106)
 * as it is driven by the torch binding
 *   RasterizeGaussiansCUDA / RasterizeGaussiansBackwardCUDA / markVisible
 *     (submodules/diff_gaussian_rasterization_df/rasterize_points.cu:35-133, :135-234, :236-259).
 *
 * Plain pointers and sizes only: every pointer is a DEVICE pointer of the current HIP device unless
 * stated otherwise; `stream` is a hipStream_t passed as void* (0 = null stream).  Optional inputs
 * are passed as NULL (the reference uses "empty tensor => null data pointer",
 * cuda_rasterizer/forward.cu:218,254).  All matrices use the reference's memory convention
 * mem[c*4+r] = M[r][c] (cuda_rasterizer/auxiliary.h:68-87).
 *
 * The three scratch buffers (geometry / binning / image state) are opaque to the caller, exactly as
 * in the reference (rasterizer_impl.h:29-65); the caller owns their storage and provides it through
 * the allocation callbacks (reference: std::function<char*(size_t)>, rasterize_points.cu:27-33).
 * Their INTERNAL layout is this library's own (ex4d_*_layout below reports it for tests).
 */
#ifndef EX4D_RASTERIZER_H_INCLUDED
#define EX4D_RASTERIZER_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EX4D_TILE 16          /* cuda_rasterizer/config.h:16-17 (BLOCK_X, BLOCK_Y): part of the parity contract */
#define EX4D_CHANNELS 3       /* cuda_rasterizer/config.h:15 */

/* Scalar arguments of Rasterizer::forward / ::backward (rasterizer.h:33-67, :69-104). */
typedef struct Ex4dParams {
    int32_t P;                /* number of Gaussians (means3D.size(0)) */
    int32_t D;                /* active SH degree */
    int32_t M;                /* SH coefficients stored per Gaussian per channel (sh.size(1)), 0 if no SH */
    int32_t W, H;             /* image width / height */
    float tanfovx, tanfovy;
    float kernel_size;        /* 2D mip filter added to the screen-space covariance */
    float scale_modifier;
    float min_depth, max_depth;
    int32_t prefiltered;      /* reference traps on a culled Gaussian when set; here: error EX4D_ERR_PREFILTERED */
    int32_t debug;            /* synchronise + check after every kernel (auxiliary.h:296-303) */
    int32_t prepare_backward; /* forward: a backward will follow -- the per-Gaussian kernel also stores the d(colour)/d(direction) sums of the
                                 SH backward (36 B per visible Gaussian, inside the geometry buffer) while it has the SH rows in registers;
                                 pass the SAME value to the backward call that consumes this forward's buffers: it then does not read
                                 the SH tensors at all (-155 MB of 535 at 1.0 M Gaussians).  0 = the backward reads them itself.
                                 The forward marks the geometry buffer when it stored the sums; a backward that asks for them on a
                                 buffer whose forward ran with 0 returns NaN gradients (never numbers computed from uninitialised memory). */
    int32_t instance_capacity;/* 0 (default): the reference's behaviour -- the forward reads the instance count back (one blocking 4-byte D2H,
                                 rasterizer_impl.cu:298-299) and sizes the binning buffer exactly.
                                 > 0: ASYNCHRONOUS forward.  The binning buffer is sized for this many (Gaussian, tile) instances, every kernel
                                 behind the tile scan reads the actual count from device memory, NOTHING blocks the host and every launch
                                 has a host-constant grid (the call sequence can be captured into a hipGraph).  `num_rendered` must then
                                 point to PINNED HOST memory (or device memory) of Ex4dFrameStatus size, filled by an asynchronous copy on
                                 `stream`; the caller looks at it after synchronising with the stream (or an event recorded behind the call).  An instance count
                                 above the capacity truncates the tile lists: status.num_rendered > capacity tells, the frame's outputs
                                 are then invalid and the caller re-runs it with a larger capacity.  The matching backward takes
                                 num_rendered = this capacity (it sizes the same buffer layout; the kernels read the actual ranges). */
    int32_t assume_no_flow;   /* asynchronous forward only: 1 = the caller asserts that every dir3D of the frame is zero (the training loop's
                                 gradient trap, gaussian_renderer/__init__.py:66-70) and the flow-free compositing kernel is launched;
                                 status.has_flow != 0 reports a violated assertion (flow output invalid).  0 = the kernel with flow.
                                 (The synchronous forward picks the kernel from the frame flag it reads back.) */
    int32_t reserved;
} Ex4dParams;

/* What an ASYNCHRONOUS forward (Ex4dParams.instance_capacity > 0) leaves in the caller's pinned host memory */
typedef struct Ex4dFrameStatus {
    uint32_t num_rendered;        /* (Gaussian, tile) instances of the frame; > capacity: lists truncated, frame invalid */
    uint32_t prefilter_violation; /* != 0: prefiltered was set and a Gaussian was culled (the synchronous call returns EX4D_ERR_PREFILTERED) */
    uint32_t has_flow;            /* != 0: some visible Gaussian carries a non-zero dir3D */
    uint32_t reserved[5];
} Ex4dFrameStatus;

/* Scratch allocation callback: must return a device pointer to at least `bytes` bytes, 256-byte aligned,
 * that stays valid until the matching backward has run (the reference resizes a torch byte tensor). */
typedef void *(*ex4d_alloc_fn)(void *user, size_t bytes);

enum {
    EX4D_OK = 0,
    EX4D_ERR_ARG = 1,          /* bad argument (shape/NULL/colour source), std::runtime_error in the reference */
    EX4D_ERR_HIP = 2,          /* a HIP runtime call or kernel failed */
    EX4D_ERR_ALLOC = 3,        /* allocation callback returned NULL */
    EX4D_ERR_PREFILTERED = 4   /* prefiltered=1 but a Gaussian was culled (auxiliary.h:286-290 traps) */
};

/* Thread-local message of the last failing call on this thread ("" if none). */
const char *ex4d_last_error(void);

/* Library/ABI version and the gfx target the kernels were compiled for (e.g. "gfx950"). */
int ex4d_abi_version(void);
const char *ex4d_target_arch(void);

/*
 * Forward: replaces CudaRasterizer::Rasterizer::forward (rasterizer_impl.cu:204-363).
 * Outputs follow rasterize_points.cu:73-78: out_color[3,H,W], radii[P] (int32), out_depth[1,H,W],
 * out_acc[1,H,W], out_flow[3,H,W], out_idx[1,H,W] (int32); all are fully written by the call
 * (no pre-initialisation needed; out_idx = -1 where nothing contributed).
 * *num_rendered receives the number of (Gaussian, tile) instances (host int, one blocking 4-byte D2H
 * exactly like rasterizer_impl.cu:298-299).  P == 0 is handled by the caller (rasterize_points.cu:90).
 * With Ex4dParams.instance_capacity > 0 the call is asynchronous and `num_rendered` is an Ex4dFrameStatus in pinned host
 * memory (see Ex4dParams); such a call enqueues only kernels, memsets and one device-to-host copy, so it may run under
 * hipStreamBeginCapture (with the per-stage profiler off and allocation callbacks that are legal during capture).
 */
int ex4d_forward(
    const Ex4dParams *prm,
    const float *background,      /* [3] */
    const float *means3D,         /* [P,3] */
    const float *dir3D,           /* [P,3] per-Gaussian "flow" channel composited into out_flow; NULL = zeros */
    const float *shs,             /* [P,M,3] or NULL */
    const float *colors_precomp,  /* [P,3] or NULL (exactly one of shs / colors_precomp) */
    const float *opacities,       /* [P] */
    const float *scales,          /* [P,3] or NULL */
    const float *rotations,       /* [P,4] (r,x,y,z), NOT normalised (forward.cu:137), or NULL */
    const float *cov3D_precomp,   /* [P,6] or NULL (exactly one of scales+rotations / cov3D_precomp) */
    const float *viewmatrix,      /* [16] */
    const float *projmatrix,      /* [16] */
    const float *campos,          /* [3] */
    const float *subpixel_offset, /* [H,W,2] or NULL = zeros */
    ex4d_alloc_fn geom_alloc, void *geom_user,
    ex4d_alloc_fn binning_alloc, void *binning_user,
    ex4d_alloc_fn img_alloc, void *img_user,
    float *out_color, int32_t *radii, float *out_depth, float *out_acc, float *out_flow, int32_t *out_idx,
    void *stream,
    int32_t *num_rendered);

/*
 * Backward: replaces CudaRasterizer::Rasterizer::backward (rasterizer_impl.cu:367-486) including the
 * zero-fill of the ten gradient tensors (rasterize_points.cu:178-187): every output below is fully
 * written for all P Gaussians (zeros for invisible ones), so the caller may pass uninitialised memory.
 * dL_dcolors and dL_dcov3D may be NULL (a caller that rendered from SH / from scale + rotation has no tensor to receive them): they
 * are then not written -- 36 bytes per Gaussian less HBM traffic; every other output is required.
 * `bwd_scratch` must hold ex4d_backward_scratch_bytes(P) bytes (internal per-Gaussian accumulators,
 * the reference's dL_dconic[P,2,2] among them).
 * Reference-specific semantics reproduced exactly: dL_dopacity is w.r.t. opacity*coef,
 * dL_dmeans3D = projection path + SH path only (backward.cu:414 overwrites the covariance path), etc.
 */
int ex4d_backward(
    const Ex4dParams *prm, int32_t num_rendered,
    const float *background, const float *means3D, const int32_t *radii,
    const float *shs, const float *colors_precomp, const float *scales, const float *rotations,
    const float *cov3D_precomp, const float *viewmatrix, const float *projmatrix, const float *campos,
    const float *subpixel_offset,
    const float *out_depth, const float *out_acc,            /* forward outputs [1,H,W] */
    const void *geom_buffer, const void *binning_buffer, const void *img_buffer,
    const float *dL_dout_color /*[3,H,W]*/, const float *dL_dout_depth /*[1,H,W]*/,
    const float *dL_dout_flow /*[3,H,W]*/, const float *dL_dout_acc /*[1,H,W]*/,   /* each may be NULL = zeros */
    float *dL_dmeans2D /*[P,3]*/, float *dL_dcolors /*[P,3]*/, float *dL_dopacity /*[P,1]*/,
    float *dL_dmeans3D /*[P,3]*/, float *dL_dcov3D /*[P,6]*/, float *dL_dsh /*[P,M,3]*/,
    float *dL_dscales /*[P,3]*/, float *dL_drotations /*[P,4]*/, float *dL_ddir /*[P,3]*/,
    void *bwd_scratch,
    void *stream);

size_t ex4d_backward_scratch_bytes(int32_t P);

/*
 * The same two calls with the SH coefficients given AS THE MODEL STORES THEM instead of one concatenated [P,16,3] tensor:
 * rows [0, n_static) are (dc[0] [n,1,3], rest[0] [n,15,3]), rows [n_static, P) are (dc[1], rest[1]) -- the four tensors
 * CGaussianModel.get_features concatenates every frame (scene/c_gaussian_model.py:337-353: two torch.cat of 192 B/Gaussian,
 * and their split in the backward).  M must be 16; colors_precomp is not used.  Everything else as ex4d_forward/backward;
 * the backward writes dL/dsh straight into the four gradient tensors (fully written).
 */
typedef struct Ex4dSplitSH { const float *dc[2]; const float *rest[2]; int32_t n_static; } Ex4dSplitSH;
typedef struct Ex4dSplitSHGrad { float *dc[2]; float *rest[2]; int32_t n_static; } Ex4dSplitSHGrad;

int ex4d_forward_split_sh(
    const Ex4dParams *prm, const float *background, const float *means3D, const float *dir3D, const Ex4dSplitSH *shs,
    const float *opacities, const float *scales, const float *rotations, const float *cov3D_precomp,
    const float *viewmatrix, const float *projmatrix, const float *campos, const float *subpixel_offset,
    ex4d_alloc_fn geom_alloc, void *geom_user, ex4d_alloc_fn binning_alloc, void *binning_user,
    ex4d_alloc_fn img_alloc, void *img_user,
    float *out_color, int32_t *radii, float *out_depth, float *out_acc, float *out_flow, int32_t *out_idx,
    void *stream, int32_t *num_rendered);

int ex4d_backward_split_sh(
    const Ex4dParams *prm, int32_t num_rendered,
    const float *background, const float *means3D, const int32_t *radii,
    const Ex4dSplitSH *shs, const float *scales, const float *rotations,
    const float *cov3D_precomp, const float *viewmatrix, const float *projmatrix, const float *campos,
    const float *subpixel_offset, const float *out_depth, const float *out_acc,
    const void *geom_buffer, const void *binning_buffer, const void *img_buffer,
    const float *dL_dout_color, const float *dL_dout_depth, const float *dL_dout_flow, const float *dL_dout_acc,
    float *dL_dmeans2D, float *dL_dcolors, float *dL_dopacity, float *dL_dmeans3D, float *dL_dcov3D, const Ex4dSplitSHGrad *dL_dsh,
    float *dL_dscales, float *dL_drotations, float *dL_ddir, void *bwd_scratch, void *stream);

/* markVisible: replaces Rasterizer::markVisible (rasterizer_impl.cu:143-159); present[P] bytes (0/1). */
int ex4d_mark_visible(int32_t P, const float *means3D, const float *viewmatrix, const float *projmatrix,
                      float min_depth, float max_depth, uint8_t *present, void *stream);

/* Sizes the allocation callbacks will be asked for (rasterizer_impl.h required<T>(N) equivalent).
 * Footprints: geometry ~190 B per Gaussian; binning 24 B per (Gaussian, tile) instance + histograms (rounds 3-5: 48) -- point_list, the
 * sorted tile ids, and 16 B of CAPACITY for the per-quadrant compacted lists the forward leaves for the backward (uint32[4 * num_rendered]:
 * every entry of a tile list could survive the cull of each of its four quadrants; ~6 % is touched); the tile sort's two scratch arrays
 * live inside that region (the sort is over before the first list entry is written).  1.0 M Gaussians, 1352x1014, 7.5 M instances:
 * 190 MB; the deep-overlap configuration (2048x1088, 31.5 M instances): 0.78 GB; with Ex4dParams.instance_capacity the capacity takes
 * the place of num_rendered. */
size_t ex4d_geom_bytes(int32_t P);
size_t ex4d_binning_bytes(int32_t num_rendered, int32_t W, int32_t H);
size_t ex4d_img_bytes(int32_t W, int32_t H);

/* Byte offsets of the internal arrays inside the opaque buffers -- for parity tests only.
 * (reference counterparts: GeometryState / BinningState / ImageState, rasterizer_impl.h:29-65) */
typedef struct Ex4dGeomLayout {
    size_t records;         /* float[P][16]        one 64-byte record per Gaussian:
                                                   [0..1] mean2D, [2..4] conic.xyz, [5..7] cull constants, [8] depth (p_view.z),
                                                   [9..11] rgb (SH colour or colors_precomp), [12..14] dir3D, [15] opacity*coef */
    size_t cov3D;           /* float[6P]           with option "geom_debug_arrays" = 1 only */
    size_t clamped;         /* uint8[P]            bit c set <=> channel c clamped at 0 (forward.cu:67-69) */
    size_t tiles_touched;   /* uint32[P]           with option "geom_debug_arrays" = 1 only */
    size_t depth_order;     /* uint32[P]           Gaussian ids, stable-sorted by depth key (visible first) */
    size_t sorted_offsets;  /* uint32[P]           INTERNAL SCRATCH, not an interface: written only by the pair-sort path ("tile_sort_rows" = 0; its meaning
                                                   depends on the depth sort that ran); the default row-segment tile sort of round 6 needs no instance offsets */
    size_t rects;           /* uint2[P]            tile rect (getRect, auxiliary.h:46-56): .x = x0 | y0 << 16, .y = w | h << 16;
                                                   w * h == tiles_touched; defined for visible Gaussians.  Written for images of more than
                                                   255 x 255 tiles and with option "geom_debug_arrays" = 1; smaller images carry their rects in
                                                   an internal packed 4-byte form only (round 6) */
    size_t total;
} Ex4dGeomLayout;
typedef struct Ex4dBinningLayout {
    size_t point_list;      /* uint32[R]           Gaussian ids sorted by (tile, depth, id): == reference point_list */
    size_t tile_ids;        /* uint32[R]           tile id of every sorted instance (high word of the reference key); with option
                                                   "binning_tile_ids" = 1 only, see ex4d_set_option */
    size_t qlist;           /* uint32[4 R]         per (tile, 8x8 quadrant) the compacted list the forward compositing kernel leaves for the
                                                   backward: the positions in the tile list of the entries that survived its quadrant
                                                   cull, ascending (the id is point_list[r0 + position]); quadrant q of a tile with range [r0, r1) owns
                                                   [4 r0 + q (r1 - r0), ...), qcount[4 tile + q] entries are valid */
    size_t qcount;          /* uint32[4 T] */
    size_t total;
} Ex4dBinningLayout;
typedef struct Ex4dImgLayout {
    size_t final_T;         /* float[H*W]          reference accum_alpha */
    size_t n_contrib;       /* uint32[H*W] */
    size_t ranges;          /* uint2[T] */
    size_t total;
} Ex4dImgLayout;
void ex4d_geom_layout(int32_t P, Ex4dGeomLayout *out);
void ex4d_binning_layout(int32_t num_rendered, int32_t W, int32_t H, Ex4dBinningLayout *out);
void ex4d_img_layout(int32_t W, int32_t H, Ex4dImgLayout *out);
/* bwd_scratch holds the packed per-Gaussian accumulator rows float[P][16] at offset 0 (read by the parity tests).
 * Row: 0..2 dL_dmean2D.xyz (xy without the factors ln2 W/2, ln2 H/2), 3..5 dL_dconic.(x,y,w) (without -1/2), 6 dL_dopacity,
 *      7..9 dL_dcolor, 10..12 dL_ddir, 13..15 unused */

/* Tuning knobs (process-wide; results are the same within float rounding whatever the setting):
 *   "composite_bwd_variant"  4 = (Gaussian, pixel-slot) lanes with register accumulation (default), 8 = 4 + developer statistics
 *                            (ex4d_debug_bwd_stats).  Round 1's per-pixel kernel (0) and the matrix-core formulation of the sums (2)
 *                            were measured slower and are no longer part of the library.
 *   "composite_fwd_asm"      1 (default) = the flow-free compositing forward walks its staged entries with the hand-scheduled
 *                            inline-asm loop, 0 = with the compiler's loop (what frames with a non-zero dir3D always run); bit-identical.
 *   "binning_tile_ids"       1 = also write the sorted tile ids (Ex4dBinningLayout.tile_ids); 0 (default) = that region is scratch of
 *                            the tile sort -- nothing downstream reads the ids, the tile ranges carry the same information.  (Images with
 *                            <= 256 or > 65536 tiles, or more than 2^(32 - ceil(tile bits / 2)) Gaussians, take the key/value sort,
 *                            which always writes them.)
 *   "preprocess_sh_predicate" 1 (default) = the per-Gaussian forward kernel runs its frustum test before it requests the SH rows and
 *                            does not request the rows of frustum-culled Gaussians (pays in views whose frustum culls; the synthetic
 *                            BASELINE scenes lose their invisible Gaussians after the projection); 0 = all loads up front.  Same results.
 *   "geom_debug_arrays"      1 = also write Ex4dGeomLayout.cov3D and .tiles_touched; 0 (default) = those regions stay untouched: the
 *                            backward recomputes the covariance from scale / rotation (same function, same bits), the tile rect carries
 *                            the count.
 *   "preprocess_fast_path"   1 (default) = frames with one [P,16,3] SH tensor at degree 3 and scale + rotation (no precomputed colours or
 *                            covariances) take the per-Gaussian forward kernel specialised for exactly that; 0 = always the generic one.
 *                            Bit-identical results.
 *   "depth_sort_msd"         0 = the Gaussians are ordered by depth with a 3-pass LSD radix sort, the tile scan gathers their rects;
 *                            2 = MSD-first depth sort: one partition on the top digit of the key range the frame occupies, every bucket finished
 *                            in LDS (4 launches instead of 10; ~13 us per frame faster at 1.0 M Gaussians spread in depth; behind the pair sort
 *                            -- "tile_sort_rows" = 0 -- the tile scan is fused into the bucket kernel); 1 = the same with the tile scan as a
 *                            kernel of its own.  Identical results.  A bucket of more
 *                            than 4096 (8192 beyond 1.2 M Gaussians) is sorted by one workgroup through global memory -- a fronto-parallel
 *                            wall holding a third of the Gaussians costs 1.4 ms there ("Round 5").  Hence
 *                            3 (default) = auto: the MSD sort, until its bucket kernel reports an oversize bucket (a word in pinned host
 *                            memory); the following 64 frames then take the LSD sort, and every further report doubles that hold (so that
 *                            of F frames at most log2 F pay the slow path).  Calls being recorded into a graph take the LSD sort.  Setting
 *                            the option resets the hold; "depth_sort_hold" / "depth_sort_trips" (read-only) show frames left on the LSD
 *                            sort / reports seen.  The MSD sort needs 0 <= min_depth < max_depth with at most 28 significant key bits and
 *                            an image of at most 255 x 255 tiles; other frames take the LSD sort whatever the option says.
 *   "tile_sort_rows"         1 (default, round 6) = the tile lists come from the row-segment sort (ex4d_rowsort.hip): a stable partition of the
 *                            rects' row segments by tile row whose write-out expands them into instance words, then a counting sort by
 *                            tile column per row -- no duplication kernel, no instance offsets, no (tile, id) pairs; 0 = duplication + the
 *                            MSD-first pair sort of rounds 2-5.  Identical point_list / ranges.  Needs an image of at most 255 x 255 tiles
 *                            and P <= 2^24; other frames take the pair sort whatever the option says.
 *   "rank_lds_atomics"       -1 (default) = the scatter kernels rank their items by the return value of an LDS atomic IF a probe kernel,
 *                            run once per device by the first forward, finds that the lanes of one ds_add_rtn instruction receive their
 *                            pre-op values in ascending lane order (gfx950: yes; not an architectural promise), else by wave ballots;
 *                            0 = always ballots; 1 = LDS atomics without asking.  Identical results where the probe holds.
 *                            "rank_lds_atomics_in_use" (read-only): what the last forward's device uses.
 *   "composite_clamp_always" 0 (default) = the flow-free compositing forward evaluates alpha = min(0.99, w G) only in chunks that staged an
 *                            entry with w > 0.99 (the clamp cannot bind elsewhere: same bits); 1 = everywhere (rounds 1-5; A/B runs).
 *   "composite_bwd_pairs"    1 (default, round 6) = quadrants with integer pixel positions and no upstream dL_dacc run the compositing backward
 *                            with two pixels per lane on packed math (v_pk_*_f32); 0 = one pixel per lane and step everywhere (rounds 2-5).
 *                            Same decisions, sums reassociated (even / odd columns accumulate apart): equal within the gradient bars.
 *   "readback_side_stream"   1 (default, round 6) = the synchronous forward's one read-back (instance count, frame flags) is copied on a
 *                            stream of the library's own behind an event recorded after the per-Gaussian kernel, so that the depth sort
 *                            does not queue behind the copy and its system-scope release (-14 us per frame at 1.0 M Gaussians);
 *                            0 = the copy sits on the caller's stream (rounds 1-5).  Same results; the call still returns only after the
 *                            count has arrived.
 *   "rows_probe"             developer: 1 = the row partition's scatter kernel records shader-clock cycles per phase (ex4d_debug_rows_prof).
 *   "depth_sort_msd_bits"    0 (default) = the MSD depth sort cuts its top digit 9 bits wide up to 1.3 M Gaussians (511 visible buckets of
 *                            ~1.6 k at 1.0 M, each finished by a 512-thread workgroup: <= 8192 in LDS) and 10 bits beyond; 9 / 10 = forced
 *                            (tests, A/B runs; a forced 9 only where the key bits under the digit fit the bucket kernel's LDS word).
 *                            Same order either way.
 *   "depth_sort_local_cap"   tests: largest bucket (0 = the kernel's capacity) the MSD depth sort finishes in LDS.
 *   "depth_sort_local_threads" 0 (default: 512 under the 9-bit digit and beyond 1.2 M Gaussians, else 256) / 256 / 512 = workgroup size of the MSD depth sort's bucket kernel.
 * Returns EX4D_OK / the value, or an error / -1. */
int ex4d_set_option(const char *name, int value);
int ex4d_get_option(const char *name);
/* developer counters of "composite_bwd_variant" 8 since the last reset: [0] batches, [1] valid Gaussians, [2] steps run,
 * [3] steps skipped, [4] contributing (pixel, Gaussian) pairs, [5] Gaussians with a contributing pair, [6..7] spare */
int ex4d_debug_bwd_stats(unsigned long long *out8, int reset);
/* ... and the extended set: [6] pairs alive by list position, [7] / [8] / [9] Gaussians contributing in the top four rows of their
 * quadrant only / the bottom four only / both, [10] / [11] steps run in the top / bottom half, [12] / [13] batches whose top / bottom
 * half has no contributing pair, [14..15] spare */
int ex4d_debug_bwd_stats16(unsigned long long *out16, int reset);
/* developer profile of the row-segment partition (option "rows_probe" = 1): shader-clock cycles per phase of its scatter kernel summed over
 * workgroups ([0] loads + count, [1] barrier scan, [2] placement, [3] width scan, [4] expansion, [5] write-out), [7] = workgroups */
int ex4d_debug_rows_prof(unsigned long long *out8, int reset);

/* Optional per-stage timing (hipEvents on the caller's stream, single host thread; used by bench.py).
 * ex4d_profile_read(which = 0 forward / 1 backward) waits for the last recorded call of that kind and
 * returns the number of stages written to ms[] / names[]. */
void ex4d_profile_enable(int on);
int ex4d_profile_read(int which, float *ms, const char **names, int max_stages);

#ifdef __cplusplus
}
#endif
#endif
