// C-ABI entry points of libex4d_hip.so: scratch-buffer carving, kernel sequencing, error reporting.
//
// Replaces CudaRasterizer::Rasterizer::{forward, backward, markVisible}
// (submodules/diff_gaussian_rasterization_df/cuda_rasterizer/rasterizer_impl.cu:204-363, :367-486, :143-159)
// and the GeometryState / BinningState / ImageState::fromChunk carving (:161-200).
#include "ex4d_internal.h"
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace {

std::atomic<bool> g_prof_on{false};
// tuning knobs (ex4d_set_option): which compositing-backward kernel runs (ex4d_composite.hip: ex4d_launch_composite_bwd)
std::atomic<int> g_bwd_variant{4};
std::atomic<int> g_tile_ids{0};
// "tile_sort_rows": 1 (default) = the tile lists come from the row-segment sort of round 6 (ex4d_rowsort.hip: no duplication kernel, no
// instance offsets); 0 = duplication + the MSD-first pair sort of rounds 2-5.  Same point_list / ranges bit for bit.
std::atomic<int> g_tile_rows{1};
// "depth_sort_msd": 0 = the 3-pass LSD depth sort + gathering tile scan; 2 = the MSD-first depth sort of round 5 (one partition on the
// top digit of the occupied key range, every bucket finished in LDS, tile scan fused in: 5 launches instead of 10, -22 us of kernel time at
// 1.0 M Gaussians with well-spread depths); 1 = the same with the streaming tile-scan kernel; 3 (default) = "auto": the MSD sort until
// one of its buckets does not fit the LDS, then the LSD sort for a while.  The MSD sort's bucket kernel sorts a bucket of more than
// 4096 / 8192 Gaussians with ONE workgroup through global memory -- a large surface at one depth (a fronto-parallel wall: tens of
// thousands of Gaussians inside 0.3 % of the depth range) costs hundreds of microseconds there, the LSD sort does not care (DESIGN.md
// section 4, "Round 5").  Both sorts give the same order bit for bit, so which one ran shows in the frame time only.
std::atomic<int> g_depth_msd{3};
// auto mode: the bucket kernel sets `word` (pinned host memory) when it met an oversize bucket; the next forward that sees it orders
// `backoff` frames with the LSD sort and doubles `backoff` (never reset: over a run of F frames at most log2(F) frames pay the slow
// path, whatever the scene does).  Process-wide: a conservative signal shared by every stream and thread.
struct DepthSortWatch {
    std::mutex mu;
    uint32_t *word = nullptr;
    bool tried = false;
    std::atomic<int> hold{0}, backoff{64}, trips{0};
    uint32_t *get()
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!tried) {
            tried = true;
            if (hipHostMalloc((void **)&word, 64, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); word = nullptr; }
            else memset(word, 0, 64);
        }
        return word;
    }
    void reset()
    {
        std::lock_guard<std::mutex> lock(mu);
        hold.store(0); backoff.store(64); trips.store(0);
        if (word) __atomic_store_n(word, 0u, __ATOMIC_RELAXED);
    }
} g_depth_watch;
// does this frame use the MSD sort in auto mode?  (*watch: the word its bucket kernel reports to)
bool depth_sort_auto_msd(hipStream_t stream, bool async, uint32_t **watch)
{
    *watch = nullptr;
    if (async) {    // a call being recorded into a graph replays without the host: no way to switch afterwards
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (st != hipStreamCaptureStatusNone) return false;
    }
    uint32_t *w = g_depth_watch.get();
    if (!w) return false;
    // (test and clear in ONE exchange: a separate store could erase the report of a bucket kernel of a frame still in flight -- ADVICE r05)
    if (__atomic_load_n(w, __ATOMIC_RELAXED) != 0u && __atomic_exchange_n(w, 0u, __ATOMIC_RELAXED) != 0u) {
        const int b = g_depth_watch.backoff.load();
        g_depth_watch.hold.store(b);
        g_depth_watch.backoff.store(b < (1 << 20) ? 2 * b : b);
        g_depth_watch.trips.fetch_add(1);
    }
    for (int h = g_depth_watch.hold.load(); h > 0; )       // (threads share the hold: take one frame of it, never below zero)
        if (g_depth_watch.hold.compare_exchange_weak(h, h - 1)) return false;
    *watch = w;
    return true;
}
std::atomic<int> g_depth_msd_bits{0};       // "depth_sort_msd_bits": 0 (default) = by Gaussian count (9 bits up to 1.3 M, 10 beyond); 9 / 10 = forced (tests, A/B runs)
std::atomic<int> g_readback_side{1};        // "readback_side_stream": 1 (default, round 6) = the synchronous forward's read-back copy runs on a side stream; 0 = on the caller's stream (rounds 1-5)
std::atomic<int> g_depth_local_cap{0};      // "depth_sort_local_cap": largest bucket the MSD depth sort finishes in LDS (0 = the kernel's capacity; tests force the through-memory path with a small value)
std::atomic<int> g_depth_local_threads{0};  // "depth_sort_local_threads": 256 / 512 = workgroup size of the depth sort's bucket kernel, 0 = by Gaussian count
std::atomic<int> g_geom_debug{0};    // "geom_debug_arrays": also write cov3D[P,6] and tiles_touched[P] into the geometry buffer (tests)      // "binning_tile_ids": also write the sorted tile ids (tests, debugging)
thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(EX4D_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// auxiliary.h:296-303 (CHECK_CUDA): in debug mode synchronise after every stage and surface the error
#define STAGE(expr, prm, stream)                                                                   \
    do {                                                                                           \
        HIP_TRY(expr);                                                                             \
        if ((prm)->debug) HIP_TRY(hipStreamSynchronize(stream));                                   \
    } while (0)
#define MARK(which, name) g_prof.mark(which, name, stream)

// Optional per-stage timing with hipEvents on the caller's stream (used by bench.py for the roofline line;
// off by default).  HIP events belong to the device that was current when they were created: the events are
// (re)created whenever the calling thread's current device differs from the one they were made on.  One profiler per process
// (forward and backward of one frame run on different host threads under torch autograd), serialised by a mutex.
struct StageProfiler {
    static const int kMax = 16;
    hipEvent_t ev[2][kMax + 1];
    const char *names[2][kMax];
    int n[2] = { 0, 0 };
    bool created = false;
    int device = -1;
    std::mutex mu;
    void begin(int which, hipStream_t s)
    {
        if (!g_prof_on.load(std::memory_order_relaxed)) return;
        std::lock_guard<std::mutex> lock(mu);
        int dev = -1;
        (void)hipGetDevice(&dev);
        if (created && dev != device) {
            for (int w = 0; w < 2; w++) for (int i = 0; i <= kMax; i++) (void)hipEventDestroy(ev[w][i]);
            created = false; n[0] = n[1] = 0;
        }
        if (!created) { for (int w = 0; w < 2; w++) for (int i = 0; i <= kMax; i++) (void)hipEventCreate(&ev[w][i]); created = true; device = dev; }
        n[which] = 0;
        (void)hipEventRecord(ev[which][0], s);
    }
    void mark(int which, const char *name, hipStream_t s)
    {
        if (!g_prof_on.load(std::memory_order_relaxed)) return;
        std::lock_guard<std::mutex> lock(mu);
        if (!created || n[which] >= kMax) return;
        names[which][n[which]] = name;
        n[which]++;
        (void)hipEventRecord(ev[which][n[which]], s);
    }
};
StageProfiler g_prof;

// pinned host word + event for the instance-count read-back (one per host thread, created on first use; the event is
// re-created when the thread's current device changes -- events are device-bound)
struct Readback {
    uint32_t *host = nullptr;
    size_t words = 0;
    hipEvent_t ev;
    // "readback_side_stream": the copy runs on a stream of its own behind `ev_pre` (recorded on the caller's stream after the per-Gaussian
    // kernel), so that the depth sort's kernels do not queue behind the copy and its system-scope release
    hipEvent_t ev_pre;
    hipStream_t side = nullptr;
    bool have_ev = false;
    int device = -1;
    bool init(size_t need_words)
    {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        if (have_ev && dev != device) { (void)hipEventDestroy(ev); (void)hipEventDestroy(ev_pre); (void)hipStreamDestroy(side); side = nullptr; have_ev = false; }
        if (!have_ev) {
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return false;
            if (hipEventCreateWithFlags(&ev_pre, hipEventDisableTiming) != hipSuccess) { (void)hipEventDestroy(ev); return false; }
            if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess) { (void)hipEventDestroy(ev); (void)hipEventDestroy(ev_pre); side = nullptr; return false; }
            have_ev = true; device = dev;
        }
        if (need_words > words) {
            if (host) (void)hipHostFree(host);
            host = nullptr; words = 0;
            if (hipHostMalloc((void **)&host, need_words * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) return false;
            words = need_words;
        }
        return true;
    }
};
thread_local Readback g_readback;

struct Carver {
    char *base; size_t off;
    explicit Carver(void *b) : base((char *)b), off(0) {}
    template <typename T> T *take(size_t count)
    {
        T *p = base ? (T *)(base + off) : nullptr;
        off = ex4d_align_up(off + count * sizeof(T));
        return p;
    }
};

GeomState carve_geom(void *buf, int P, Ex4dGeomLayout *lay, size_t *total)
{
    Carver c(buf);
    GeomState g;
    Ex4dGeomLayout l;
    l.records = c.off;        g.records = c.take<float4>(4 * (size_t)P);
    l.cov3D = c.off;          g.cov3D = c.take<float>(6 * (size_t)P);
    l.clamped = c.off;        g.clamped = c.take<uint8_t>(P);
    l.tiles_touched = c.off;  g.tiles_touched = c.take<uint32_t>(P);
    l.rects = c.off;          g.rects = c.take<uint2>(P);
    g.sorted_rects = c.take<uint2>(P);
    g.rects4 = c.take<uint32_t>(P);
    l.depth_order = c.off;    g.depth_order = c.take<uint32_t>(P);
    l.sorted_offsets = c.off; g.sorted_offsets = c.take<uint32_t>(P);
    g.sort_keys_a = c.take<uint32_t>(P);
    g.sort_keys_b = c.take<uint32_t>(P);
    g.sort_vals_b = c.take<uint32_t>(P);
    g.sort_vals_a = c.take<uint32_t>(P);
    g.rects4_b = c.take<uint32_t>(P);
    g.bucket_starts = c.take<uint32_t>((size_t)1 << EX4D_DLS_MSD_BITS);
    g.key_ranges = c.take<uint2>(((size_t)P + 63) / 64);
    g.bucket_sums = c.take<uint32_t>((size_t)1 << EX4D_DLS_MSD_BITS);
    g.scan_block_sums = c.take<uint32_t>((P + SCAN_CHUNK - 1) / SCAN_CHUNK + 1);
    g.sort_hist = c.take<uint32_t>(ex4d_radix_hist_words((uint32_t)P));
    g.total = c.take<uint32_t>(EX4D_FLAG_WORDS);       // [1] prefilter violation flag, followed by the per-chunk instance counts
    g.block_totals = c.take<uint32_t>(2 * (((size_t)P + 63) / 64));        // (instance count, tile-row segment count) of every 64-Gaussian chunk
    g.row_hist = c.take<uint32_t>(ex4d_tile_sort_rows_geom_words((uint32_t)P));
    g.sh_dsums = c.take<float>(9 * (size_t)P);
    l.total = c.off;
    if (lay) *lay = l;
    if (total) *total = c.off;
    return g;
}

// number of key bits needed for tile ids 0..T-1 (the reference's getHigherMsb over-estimates by design,
// rasterizer_impl.cu:35-50; any bit count covering every tile id yields the identical stable order)
int tile_bits(int T)
{
    int b = 1;
    while ((1 << b) < T) b++;
    return b;
}

BinState carve_binning(void *buf, uint32_t R, int W, int H, Ex4dBinningLayout *lay, size_t *total)
{
    const int T = ((W + EX4D_TILE - 1) / EX4D_TILE) * ((H + EX4D_TILE - 1) / EX4D_TILE);
    Carver c(buf);
    BinState b;
    Ex4dBinningLayout l;
    const size_t n = R ? R : 1;
    l.point_list = c.off; b.point_list = c.take<uint32_t>(n);
    l.tile_ids = c.off;   b.tile_ids = c.take<uint32_t>(n);
    const int tb = tile_bits(T);
    const int cgx = (W + EX4D_TILE - 1) / EX4D_TILE, cgy = (H + EX4D_TILE - 1) / EX4D_TILE;
    const size_t hw = ex4d_radix_hist_words(R), hw2 = ex4d_tile_sort_hist_words(R, tb);
    const size_t hw3 = (cgx <= 255 && cgy <= 255) ? ex4d_tile_sort_rows_hist_words(R) : 0;
    const size_t hwm = hw > hw2 ? hw : hw2;
    b.sort_hist = c.take<uint32_t>(hwm > hw3 ? hwm : hw3);
    l.qlist = c.off;  b.qlist = c.take<uint32_t>(4 * n);
    // (round 6) the two scratch arrays of the tile sort live INSIDE the compacted-list region: the sort is over before the compositing
    // forward writes its first list entry (8 of the 16 bytes per instance the lists reserve)
    b.vals_tmp = b.qlist;
    b.keys_tmp = b.qlist ? b.vals_tmp + ex4d_align_up(n * sizeof(uint32_t)) / sizeof(uint32_t) : nullptr;
    l.qcount = c.off; b.qcount = c.take<uint32_t>(4 * (size_t)T);
    l.total = c.off;
    if (lay) *lay = l;
    if (total) *total = c.off;
    return b;
}

ImgState carve_img(void *buf, int W, int H, Ex4dImgLayout *lay, size_t *total)
{
    Carver c(buf);
    ImgState s;
    Ex4dImgLayout l;
    const size_t HW = (size_t)W * H;
    const int T = ((W + EX4D_TILE - 1) / EX4D_TILE) * ((H + EX4D_TILE - 1) / EX4D_TILE);
    l.final_T = c.off;   s.final_T = c.take<float>(HW);
    l.n_contrib = c.off; s.n_contrib = c.take<uint32_t>(HW);
    l.ranges = c.off;    s.ranges = c.take<uint2>(T);
    l.total = c.off;
    if (lay) *lay = l;
    if (total) *total = c.off;
    return s;
}

}  // namespace

extern "C" {

const char *ex4d_last_error(void) { return g_err; }
int ex4d_abi_version(void) { return 5; }      // 5 (round 6): binning buffer 24 B per instance, 4-byte compacted-list entries, per-chunk (instance, segment) counts
const char *ex4d_target_arch(void) { return "gfx950"; }

size_t ex4d_geom_bytes(int32_t P) { size_t t; carve_geom(nullptr, P, nullptr, &t); return t; }
size_t ex4d_binning_bytes(int32_t R, int32_t W, int32_t H) { size_t t; carve_binning(nullptr, (uint32_t)R, W, H, nullptr, &t); return t; }
size_t ex4d_img_bytes(int32_t W, int32_t H) { size_t t; carve_img(nullptr, W, H, nullptr, &t); return t; }
void ex4d_geom_layout(int32_t P, Ex4dGeomLayout *out) { carve_geom(nullptr, P, out, nullptr); }
void ex4d_binning_layout(int32_t R, int32_t W, int32_t H, Ex4dBinningLayout *out) { carve_binning(nullptr, (uint32_t)R, W, H, out, nullptr); }
void ex4d_img_layout(int32_t W, int32_t H, Ex4dImgLayout *out) { carve_img(nullptr, W, H, out, nullptr); }
size_t ex4d_backward_scratch_bytes(int32_t P) { return ex4d_align_up((size_t)P * 16 * sizeof(float)); }

static int forward_impl(
    const Ex4dParams *prm, ShSplit split,
    const float *background, const float *means3D, const float *dir3D, const float *shs, const float *colors_precomp,
    const float *opacities, const float *scales, const float *rotations, const float *cov3D_precomp,
    const float *viewmatrix, const float *projmatrix, const float *campos, const float *subpixel_offset,
    ex4d_alloc_fn geom_alloc, void *geom_user, ex4d_alloc_fn binning_alloc, void *binning_user,
    ex4d_alloc_fn img_alloc, void *img_user,
    float *out_color, int32_t *radii, float *out_depth, float *out_acc, float *out_flow, int32_t *out_idx,
    void *stream_, int32_t *num_rendered)
{
    g_err[0] = 0;
    hipStream_t stream = (hipStream_t)stream_;
    if (!prm || !num_rendered) return fail(EX4D_ERR_ARG, "null params");
    const int P = prm->P, W = prm->W, H = prm->H;
    if (P <= 0 || W <= 0 || H <= 0) return fail(EX4D_ERR_ARG, "P, W, H must be positive (P == 0 is handled by the caller)");
    if (!means3D || !opacities || !background || !viewmatrix || !projmatrix || !campos)
        return fail(EX4D_ERR_ARG, "means3D, opacities, background, viewmatrix, projmatrix, campos are required");
    const bool is_split = split.rest[0] != nullptr || split.rest[1] != nullptr;
    if (((shs != nullptr || is_split) ? 1 : 0) + (colors_precomp != nullptr ? 1 : 0) != 1 || (shs != nullptr && is_split))
        return fail(EX4D_ERR_ARG, "Please provide excatly one of either SHs or precomputed colors!");
    if (is_split && prm->M != 16) return fail(EX4D_ERR_ARG, "split SH needs M == 16 (dc [n,1,3] + rest [n,15,3])");
    if (((scales == nullptr || rotations == nullptr) && cov3D_precomp == nullptr) ||
        ((scales != nullptr || rotations != nullptr) && cov3D_precomp != nullptr))
        return fail(EX4D_ERR_ARG, "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
    if (shs && prm->M < (prm->D + 1) * (prm->D + 1)) return fail(EX4D_ERR_ARG, "sh has fewer coefficients than the active degree needs");
    if (!out_color || !radii || !out_depth || !out_acc || !out_flow || !out_idx) return fail(EX4D_ERR_ARG, "null output");
    const int gx = (W + EX4D_TILE - 1) / EX4D_TILE, gy = (H + EX4D_TILE - 1) / EX4D_TILE;
    const int T = gx * gy;
    // asynchronous forward (Ex4dParams.instance_capacity): no read-back, no host wait, host-constant grids; `num_rendered` is an
    // Ex4dFrameStatus in pinned host memory that an asynchronous copy fills
    const bool async = prm->instance_capacity > 0;
    if (prm->instance_capacity < 0) return fail(EX4D_ERR_ARG, "instance_capacity must be >= 0");
    if (async && prm->debug) return fail(EX4D_ERR_ARG, "debug (synchronise after every stage) and instance_capacity > 0 (asynchronous forward) exclude each other");

    void *geom_buf = geom_alloc(geom_user, ex4d_geom_bytes(P));
    if (!geom_buf) return fail(EX4D_ERR_ALLOC, "geometry buffer allocation failed");
    GeomState g = carve_geom(geom_buf, P, nullptr, nullptr);
    void *img_buf = img_alloc(img_user, ex4d_img_bytes(W, H));
    if (!img_buf) return fail(EX4D_ERR_ALLOC, "image buffer allocation failed");
    ImgState im = carve_img(img_buf, W, H, nullptr, nullptr);

    // depth-sort keys: the bit patterns of the visible depths lie in (bits(min_depth), bits(max_depth)] when 0 <= min_depth < max_depth;
    // taken relative to bits(min_depth) they need 26 bits for (4, 300], 27 for (0.01, 300] instead of 32 -> 3 radix passes, not 4
    uint32_t key_base = 0, key_invisible = 0xFFFFFFFFu;
    int key_bits = 32;
    if (prm->min_depth >= 0.0f && prm->max_depth > prm->min_depth && prm->max_depth < 3.0e38f) {
        uint32_t lo, hi;
        memcpy(&lo, &prm->min_depth, 4); memcpy(&hi, &prm->max_depth, 4);
        key_base = lo;
        key_invisible = hi - lo + 1u;
        key_bits = 1;
        while (key_bits < 32 && (key_invisible >> key_bits) != 0u) key_bits++;
    }
    const bool packed_rects = gx <= 255 && gy <= 255;       // the tile scan gathers 32-bit packed rects (L2-resident) instead of the 8-byte ones
    // MSD-first depth sort (round 5; ex4d_binning.hip: depth_local_sort_kernel): applies when the rects travel packed and the key bits
    // under the top digit fit its LDS word.  Its top digit is cut from the key range the frame's visible Gaussians occupy (the
    // per-Gaussian kernel leaves that range in the frame flags), the invisible key gets the last digit to itself
    int msd_mode = g_depth_msd.load(std::memory_order_relaxed);
    uint32_t *msd_watch = nullptr;
    if (!(packed_rects && ex4d_depth_sort_msd_applies((uint32_t)P, key_bits))) msd_mode = 0;
    else if (msd_mode == 3) msd_mode = depth_sort_auto_msd(stream, async, &msd_watch) ? 2 : 0;
    const bool msd_depth = msd_mode != 0;
    int msd_bits = ex4d_depth_sort_msd_bits((uint32_t)P, key_bits);      // width of its top digit: 9 bits up to 1.3 M Gaussians, 10 beyond (round 6)
    {   // (option "depth_sort_msd_bits": a forced 9 only where the key bits under a 9-bit digit fit the bucket kernel's LDS word)
        const int forced = g_depth_msd_bits.load(std::memory_order_relaxed);
        if (forced == EX4D_DLS_MSD_BITS || (forced == EX4D_DLS_MSD_BITS - 1 && ex4d_depth_sort_msd_bits(1u, key_bits) == forced)) msd_bits = forced;
    }
    // the LSD sort ping-pongs between the (a) and (b) pairs and has to end in (a) = depth_order: start in (b) for an odd pass count
    const bool start_in_b = (ex4d_radix_passes((uint32_t)P, key_bits) & 1) != 0;
    uint32_t *keys0 = start_in_b ? g.sort_keys_b : g.sort_keys_a, *vals0 = start_in_b ? g.sort_vals_b : g.depth_order;
    uint32_t *keys1 = start_in_b ? g.sort_keys_a : g.sort_keys_b, *vals1 = start_in_b ? g.depth_order : g.sort_vals_b;
    if (msd_depth) { keys0 = g.sort_keys_a; vals0 = g.sort_vals_a; }      // partition: (keys_a, vals_a, rects4) -> (keys_b, depth_order, rects4_b)

    HIP_TRY(ex4d_prepare_rank_lds(stream));          // (first forward on a device: the probe of the LDS-atomic ranking, once)
    g_prof.begin(0, stream);
    // the frame flags / Ex4dFrameStatus words: zeroed for the asynchronous forward (its status copy and the device-side instance count live
    // there); the synchronous forward reads its flags from the per-chunk count pairs (round 6: one launch less in front of the per-Gaussian kernel)
    if (async) HIP_TRY(ex4d_launch_zero(g.total, 8 * sizeof(uint32_t), stream));
    // 1. per-Gaussian preprocess
    GeomState gw = g;          // cov3D[P,6] / tiles_touched[P] are written on request only: nothing downstream reads them
    if (!g_geom_debug.load(std::memory_order_relaxed)) {
        gw.cov3D = nullptr; gw.tiles_touched = nullptr;
        if (packed_rects) gw.rects = nullptr;      // (round 6) every consumer takes the packed 4-byte rects when the image has <= 255 x 255 tiles: 8 MB less written at 1.0 M Gaussians
    }
    STAGE(ex4d_launch_preprocess_fwd(*prm, means3D, dir3D, scales, rotations, opacities, shs, cov3D_precomp, colors_precomp,
                                     viewmatrix, projmatrix, campos, radii, gw, g.total + 1, split,
                                     keys0, (uint32_t *)nullptr, key_base, key_invisible, packed_rects ? g.rects4 : nullptr, stream,
                                     msd_depth ? reinterpret_cast<uint32_t *>(g.key_ranges) : nullptr, async), prm, stream);
    MARK(0, "preprocess_fwd");
    // the one read-back the reference also has (rasterizer_impl.cu:298-299), started here: the instance count was summed by
    // the preprocess kernel and travels to a pinned host word while the depth sort below keeps the GPU busy
    // (g.total[0..63] and the per-chunk counts are adjacent in the geometry buffer: one copy)
    const size_t nblk = (size_t)(P + 63) / 64;
    const size_t rb_words = (size_t)(g.block_totals - g.total) + 2 * nblk;
    if (!async) {
        if (!g_readback.init(rb_words)) return fail(EX4D_ERR_HIP, "pinned read-back buffer allocation failed");
        if (g_readback_side.load(std::memory_order_relaxed)) {
            HIP_TRY(hipEventRecord(g_readback.ev_pre, stream));
            HIP_TRY(hipStreamWaitEvent(g_readback.side, g_readback.ev_pre, 0));
            HIP_TRY(hipMemcpyAsync(g_readback.host, g.total, rb_words * sizeof(uint32_t), hipMemcpyDeviceToHost, g_readback.side));
            HIP_TRY(hipEventRecord(g_readback.ev, g_readback.side));
        } else {
            HIP_TRY(hipMemcpyAsync(g_readback.host, g.total, rb_words * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipEventRecord(g_readback.ev, stream));
        }
    }
    // 2. order Gaussians by depth (stable; invisible ones last); keys/ids were emitted by the preprocess kernel
    // depth_sort_msd == 2 (and auto): the tile scan is fused into the bucket kernel of the depth sort (bucket-local inclusive scans +
    // bucket sums; duplicate_kernel adds the bucket bases): no scan kernel at all.  1: the streaming scan kernel.
    // tile lists by the row-segment sort (round 6): it reads the rects in depth order and nothing else -- no instance offsets
    const bool rows_sort = g_tile_rows.load(std::memory_order_relaxed) != 0 && ex4d_tile_sort_rows_applies(P, gx, gy);
    const bool fused_scan = msd_mode == 2 && !rows_sort;
    const bool lsd_gather = !msd_depth && rows_sort && packed_rects;      // LSD depth sort in front of the row-segment sort: its last pass hands over the rects
    if (msd_depth) {
        STAGE(ex4d_depth_sort_msd(g.sort_keys_a, g.sort_vals_a, g.rects4, g.sort_keys_b, g.depth_order, g.rects4_b, (uint32_t)P, key_invisible,
                                  g.total, g.key_ranges, g.sort_hist, g.bucket_starts, (uint32_t)g_depth_local_cap.load(std::memory_order_relaxed), stream,
                                  fused_scan ? g.sorted_offsets : nullptr, fused_scan ? g.bucket_sums : nullptr, T, (fused_scan || rows_sort) ? im.ranges : nullptr,
                                  g_depth_local_threads.load(std::memory_order_relaxed), msd_watch, msd_bits), prm, stream);
        MARK(0, "depth_sort");
        // 3. instance offsets in depth order + total: the rects arrive in depth order (rects4_b), nothing to gather
        if (!fused_scan && !rows_sort) {
            STAGE(ex4d_launch_scan_tiles(P, nullptr, g.rects4_b, nullptr, nullptr, g.sorted_offsets, g.scan_block_sums, T, im.ranges, g.total, stream), prm, stream);
            MARK(0, "scan_tiles");
        }
    } else {
        bool in_first = true;
        // (the ids the sort starts from are 0 .. P-1: its first pass generates them instead of reading an array the per-Gaussian kernel would have to write)
        // round 6, in front of the row-segment tile sort: the sort's last pass gathers the packed rects into depth order and clears the tile
        // ranges itself -- no gathering scan kernel (21 us at 1.0 M Gaussians) behind the LSD sort: scenes with a depth wall, calls in a graph
        STAGE(ex4d_radix_sort_pairs(keys0, vals0, keys1, vals1, (uint32_t)P, key_bits, g.sort_hist, &in_first, stream, nullptr, true,
                                    lsd_gather ? g.rects4 : nullptr, lsd_gather ? g.rects4_b : nullptr, lsd_gather ? im.ranges : nullptr, T), prm, stream);
        if (in_first == start_in_b) return fail(EX4D_ERR_HIP, "internal: depth sort ended in the wrong buffer");
        MARK(0, "depth_sort");
        if (!lsd_gather) {
            // 3. instance offsets in depth order + total (the total also lands in g.total[0]: device-side instance count)
            STAGE(ex4d_launch_scan_tiles(P, g.rects, packed_rects ? g.rects4 : nullptr, g.depth_order, g.sorted_rects, g.sorted_offsets, g.scan_block_sums, T, im.ranges, g.total, stream), prm, stream);
            MARK(0, "scan_tiles");
        }
    }
    const uint2 *dup_rects = (msd_depth || lsd_gather) ? nullptr : g.sorted_rects;
    const uint32_t *dup_rects4 = (msd_depth || lsd_gather) ? g.rects4_b : nullptr;
    uint32_t R = 0;                      // instance count (synchronous) or capacity (asynchronous): sizes the binning buffer and the grids
    uint32_t segment_sum = 0;            // tile-row segments of the frame (synchronous forward; asynchronous: bounded by the capacity)
    const uint32_t *n_dev = nullptr;     // asynchronous: the kernels read the actual count here
    bool has_flow;
    if (async) {
        static_assert(sizeof(Ex4dFrameStatus) == 8 * sizeof(uint32_t), "Ex4dFrameStatus mirrors the first eight frame-flag words");
        // (hipMemcpyDefault: the status may live in pinned host memory or -- e.g. for calls recorded into a graph -- in device memory)
        // (fused scan: the instance count is written by the duplication kernel -- the copy follows it, below)
        // (row-segment sort behind the MSD depth sort: its first kernel sums the instance count -- the copy follows it, below)
        const bool count_later = fused_scan || (rows_sort && msd_depth) || lsd_gather;
        if (!count_later) HIP_TRY(hipMemcpyAsync(num_rendered, g.total, sizeof(Ex4dFrameStatus), hipMemcpyDefault, stream));
        R = (uint32_t)prm->instance_capacity;
        segment_sum = 0;          // (not known on the host: the row partition picks its segment stage from the capacity -- round 6: passing the capacity
                                  // here chose the large stage for every frame, 32 us instead of 25 at config 3)
        n_dev = g.total;
        has_flow = prm->assume_no_flow == 0;
    } else {
        // 4. wait for the read-back only (not for the sort / scan kernels queued behind it)
        {   // (busy-wait on the event instead of a blocking synchronise: the copy lands while the depth sort runs, and every microsecond
            // between its landing and the launches below is a microsecond the GPU may run dry)
            static const bool spin = []{ const char *e = getenv("EX4D_READBACK_SPIN"); return e ? atoi(e) != 0 : true; }();
            if (spin) {
                // (bounded: ~100 us of polling cover the preprocess kernel + the copy; behind that -- an oversubscribed host, several ranks
                // per core, a GPU that hangs -- the thread blocks instead of burning its core: ADVICE r05)
                hipError_t q = hipErrorNotReady;
                for (int i = 0; i < 20000 && (q = hipEventQuery(g_readback.ev)) == hipErrorNotReady; i++) { }
                if (q == hipErrorNotReady) q = hipEventSynchronize(g_readback.ev);
                HIP_TRY(q);
            } else HIP_TRY(hipEventSynchronize(g_readback.ev));
        }
        uint32_t instance_sum = 0;      // uint32 wrap-around like the reference's scan
        uint32_t chunk_flags = 0;       // EX4D_CHUNK_FLOW / EX4D_CHUNK_FILTERED of every chunk, in the high bits of its segment count
        for (size_t i = 0; i < nblk; i++) {
            const uint32_t seg = g_readback.host[(size_t)(g.block_totals - g.total) + 2 * i + 1];
            instance_sum += g_readback.host[(size_t)(g.block_totals - g.total) + 2 * i];
            segment_sum += seg & 0x3FFFFFFFu;
            chunk_flags |= seg;
        }
        has_flow = (chunk_flags & EX4D_CHUNK_FLOW) != 0u;      // flag of the preprocess kernel: some visible Gaussian carries a non-zero dir3D
        if (prm->prefiltered && (chunk_flags & EX4D_CHUNK_FILTERED))
            return fail(EX4D_ERR_PREFILTERED, "Point is filtered although prefiltered is set. This shouldn't happen!");
        R = instance_sum;
        if (R > 0x7FFFFFFFu) return fail(EX4D_ERR_ARG, "more than 2^31-1 tile instances");
        *num_rendered = (int32_t)R;
    }

    void *bin_buf = binning_alloc(binning_user, ex4d_binning_bytes((int32_t)R, W, H));
    if (!bin_buf) return fail(EX4D_ERR_ALLOC, "binning buffer allocation failed");
    BinState b = carve_binning(bin_buf, R, W, H, nullptr, nullptr);

    // 5. emit (tile, id) instances in depth order, 6. stable sort by tile, 7. ranges
    const int passes = ex4d_radix_passes(R, tile_bits(T));
    uint32_t *k0 = (passes % 2 == 0) ? b.tile_ids : b.keys_tmp;   // start so that the result lands in (tile_ids, point_list)
    uint32_t *v0 = (passes % 2 == 0) ? b.point_list : b.vals_tmp;
    uint32_t *k1 = (passes % 2 == 0) ? b.keys_tmp : b.tile_ids;
    uint32_t *v1 = (passes % 2 == 0) ? b.vals_tmp : b.point_list;
    if (rows_sort) {
        if (R > 0) {
            STAGE(ex4d_tile_sort_rows(P, gx, gy, g.depth_order, dup_rects4, dup_rects, b.keys_tmp, b.point_list,
                                      g_tile_ids.load(std::memory_order_relaxed) ? b.tile_ids : nullptr, R, segment_sum, g.row_hist, b.sort_hist, im.ranges,
                                      (async && (msd_depth || lsd_gather)) ? g.total : nullptr, stream), prm, stream);
        }
        if (async && (msd_depth || lsd_gather)) HIP_TRY(hipMemcpyAsync(num_rendered, g.total, sizeof(Ex4dFrameStatus), hipMemcpyDefault, stream));
        MARK(0, "tile_sort");
    } else if (R > 0 && ex4d_tile_sort_msd_applies(P, tile_bits(T))) {
        // MSD-first sort on packed words; the tile ranges fall out of its second pass (ex4d_binning.hip: ex4d_tile_sort_msd).
        // The sorted tile ids are materialised on request only (option "binning_tile_ids"): nothing downstream reads them
        STAGE(ex4d_launch_duplicate(P, W, H, g.depth_order, g.sorted_offsets, g.scan_block_sums, dup_rects, dup_rects4, b.tile_ids, b.vals_tmp, R, stream,
                                    fused_scan ? g.sort_keys_b : nullptr, g.total + EX4D_FLAG_DPARAMS, g.bucket_sums, g.total, msd_bits), prm, stream);
        if (async && fused_scan) HIP_TRY(hipMemcpyAsync(num_rendered, g.total, sizeof(Ex4dFrameStatus), hipMemcpyDefault, stream));
        MARK(0, "duplicate");
        STAGE(ex4d_tile_sort_msd(b.tile_ids, b.vals_tmp, b.keys_tmp, b.point_list, g_tile_ids.load(std::memory_order_relaxed) ? b.tile_ids : nullptr,
                                 R, tile_bits(T), b.sort_hist, im.ranges, stream, n_dev), prm, stream);
        MARK(0, "tile_sort");
        MARK(0, "tile_ranges");
    } else {
        if (R > 0) {
            STAGE(ex4d_launch_duplicate(P, W, H, g.depth_order, g.sorted_offsets, g.scan_block_sums, dup_rects, dup_rects4, k0, v0, R, stream,
                                        fused_scan ? g.sort_keys_b : nullptr, g.total + EX4D_FLAG_DPARAMS, g.bucket_sums, g.total, msd_bits), prm, stream);
            if (async && fused_scan) HIP_TRY(hipMemcpyAsync(num_rendered, g.total, sizeof(Ex4dFrameStatus), hipMemcpyDefault, stream));
            MARK(0, "duplicate");
            bool res_a = true;
            STAGE(ex4d_radix_sort_pairs(k0, v0, k1, v1, R, tile_bits(T), b.sort_hist, &res_a, stream, n_dev), prm, stream);
        }
        MARK(0, "tile_sort");
        STAGE(ex4d_launch_tile_ranges(R, T, b.tile_ids, im.ranges, stream, n_dev), prm, stream);
        MARK(0, "tile_ranges");
    }
    // 8. compositing
    STAGE(ex4d_launch_composite_fwd(*prm, im.ranges, b.point_list, subpixel_offset, g.records, background, im.final_T, im.n_contrib,
                                    out_color, out_depth, out_acc, out_flow, out_idx, b.qlist, b.qcount, has_flow, stream), prm, stream);
    MARK(0, "composite_fwd");
    return EX4D_OK;
}

static int backward_impl(
    const Ex4dParams *prm, ShSplit split, ShSplitGrad gsplit, int32_t num_rendered,
    const float *background, const float *means3D, const int32_t *radii,
    const float *shs, const float *colors_precomp, const float *scales, const float *rotations,
    const float *cov3D_precomp, const float *viewmatrix, const float *projmatrix, const float *campos,
    const float *subpixel_offset, const float *out_depth, const float *out_acc,
    const void *geom_buffer, const void *binning_buffer, const void *img_buffer,
    const float *dL_dout_color, const float *dL_dout_depth, const float *dL_dout_flow, const float *dL_dout_acc,
    float *dL_dmeans2D, float *dL_dcolors, float *dL_dopacity, float *dL_dmeans3D, float *dL_dcov3D, float *dL_dsh,
    float *dL_dscales, float *dL_drotations, float *dL_ddir, void *bwd_scratch, void *stream_)
{
    g_err[0] = 0;
    hipStream_t stream = (hipStream_t)stream_;
    if (!prm) return fail(EX4D_ERR_ARG, "null params");
    const int P = prm->P, W = prm->W, H = prm->H;
    if (P <= 0 || W <= 0 || H <= 0) return fail(EX4D_ERR_ARG, "P, W, H must be positive (P == 0 is handled by the caller)");
    if (!geom_buffer || !binning_buffer || !img_buffer || !bwd_scratch) return fail(EX4D_ERR_ARG, "null state buffer");
    if ((size_t)P * 64 > 0xFFFFFFFFull) return fail(EX4D_ERR_ARG, "more than 2^26 Gaussians: the accumulator rows are addressed with 32-bit byte offsets");
    // any of the four upstream gradients may be NULL (= zeros: that output is not part of the loss)
    if (!dL_dmeans2D || !dL_dopacity || !dL_dmeans3D || !dL_dscales || !dL_drotations || !dL_ddir || (prm->M > 0 && !dL_dsh && !(gsplit.rest[0] || gsplit.rest[1])))
        return fail(EX4D_ERR_ARG, "null gradient output");
    GeomState g = carve_geom((void *)geom_buffer, P, nullptr, nullptr);
    BinState b = carve_binning((void *)binning_buffer, (uint32_t)num_rendered, W, H, nullptr, nullptr);
    ImgState im = carve_img((void *)img_buffer, W, H, nullptr, nullptr);
    float *acc16 = (float *)bwd_scratch;

    const int variant = g_bwd_variant.load(std::memory_order_relaxed);
    g_prof.begin(1, stream);
    HIP_TRY(ex4d_launch_zero(acc16, (size_t)P * 16 * sizeof(float), stream));
    MARK(1, "zero_accumulators");
    // colours (SH or precomputed, rasterizer_impl.cu:426) already sit in the records
    if (num_rendered > 0)
        STAGE(ex4d_launch_composite_bwd(*prm, im.ranges, b.point_list, subpixel_offset, background, g.records,
                                        out_depth, out_acc, im.final_T, im.n_contrib,
                                        dL_dout_color, dL_dout_depth, dL_dout_flow, dL_dout_acc, acc16, b.qlist, b.qcount, variant, stream), prm, stream);
    MARK(1, "composite_bwd");
    // rasterizer_impl.cu:460 takes the forward's stored covariance when none was passed in; here the kernel recomputes it from
    // scale / rotation with the forward's own function (identical bits), so only a caller-provided covariance is read
    const float *cov3D_ptr = cov3D_precomp;
    STAGE(ex4d_launch_preprocess_bwd(*prm, means3D, radii, shs, scales, rotations, cov3D_ptr, viewmatrix, projmatrix, campos, g, acc16,
                                     dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, dL_ddir,
                                     split, gsplit, stream), prm, stream);
    MARK(1, "preprocess_bwd");
    return EX4D_OK;
}

static const ShSplit kNoSplit = { { nullptr, nullptr }, { nullptr, nullptr }, 0 };
static const ShSplitGrad kNoSplitGrad = { { nullptr, nullptr }, { nullptr, nullptr }, 0 };

int ex4d_forward(
    const Ex4dParams *prm,
    const float *background, const float *means3D, const float *dir3D, const float *shs, const float *colors_precomp,
    const float *opacities, const float *scales, const float *rotations, const float *cov3D_precomp,
    const float *viewmatrix, const float *projmatrix, const float *campos, const float *subpixel_offset,
    ex4d_alloc_fn geom_alloc, void *geom_user, ex4d_alloc_fn binning_alloc, void *binning_user,
    ex4d_alloc_fn img_alloc, void *img_user,
    float *out_color, int32_t *radii, float *out_depth, float *out_acc, float *out_flow, int32_t *out_idx,
    void *stream_, int32_t *num_rendered)
{
    return forward_impl(prm, kNoSplit, background, means3D, dir3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                        viewmatrix, projmatrix, campos, subpixel_offset, geom_alloc, geom_user, binning_alloc, binning_user,
                        img_alloc, img_user, out_color, radii, out_depth, out_acc, out_flow, out_idx, stream_, num_rendered);
}

static bool split_ok(const Ex4dParams *prm, const float *const dc[2], const float *const rest[2], int n_static)
{
    if (!prm || n_static < 0 || n_static > prm->P) return false;
    if (n_static > 0 && (!dc[0] || !rest[0])) return false;
    if (n_static < prm->P && (!dc[1] || !rest[1])) return false;
    return true;
}

int ex4d_forward_split_sh(
    const Ex4dParams *prm, const float *background, const float *means3D, const float *dir3D, const Ex4dSplitSH *shs,
    const float *opacities, const float *scales, const float *rotations, const float *cov3D_precomp,
    const float *viewmatrix, const float *projmatrix, const float *campos, const float *subpixel_offset,
    ex4d_alloc_fn geom_alloc, void *geom_user, ex4d_alloc_fn binning_alloc, void *binning_user,
    ex4d_alloc_fn img_alloc, void *img_user,
    float *out_color, int32_t *radii, float *out_depth, float *out_acc, float *out_flow, int32_t *out_idx,
    void *stream_, int32_t *num_rendered)
{
    g_err[0] = 0;
    if (!shs || !split_ok(prm, shs->dc, shs->rest, shs->n_static)) return fail(EX4D_ERR_ARG, "split SH: null tensor or n_static outside [0, P]");
    const ShSplit sp = { { shs->dc[0], shs->dc[1] }, { shs->rest[0], shs->rest[1] }, shs->n_static };
    return forward_impl(prm, sp, background, means3D, dir3D, nullptr, nullptr, opacities, scales, rotations, cov3D_precomp,
                        viewmatrix, projmatrix, campos, subpixel_offset, geom_alloc, geom_user, binning_alloc, binning_user,
                        img_alloc, img_user, out_color, radii, out_depth, out_acc, out_flow, out_idx, stream_, num_rendered);
}

int ex4d_backward(
    const Ex4dParams *prm, int32_t num_rendered,
    const float *background, const float *means3D, const int32_t *radii,
    const float *shs, const float *colors_precomp, const float *scales, const float *rotations,
    const float *cov3D_precomp, const float *viewmatrix, const float *projmatrix, const float *campos,
    const float *subpixel_offset, const float *out_depth, const float *out_acc,
    const void *geom_buffer, const void *binning_buffer, const void *img_buffer,
    const float *dL_dout_color, const float *dL_dout_depth, const float *dL_dout_flow, const float *dL_dout_acc,
    float *dL_dmeans2D, float *dL_dcolors, float *dL_dopacity, float *dL_dmeans3D, float *dL_dcov3D, float *dL_dsh,
    float *dL_dscales, float *dL_drotations, float *dL_ddir, void *bwd_scratch, void *stream_)
{
    return backward_impl(prm, kNoSplit, kNoSplitGrad, num_rendered, background, means3D, radii, shs, colors_precomp, scales, rotations,
                         cov3D_precomp, viewmatrix, projmatrix, campos, subpixel_offset, out_depth, out_acc, geom_buffer, binning_buffer,
                         img_buffer, dL_dout_color, dL_dout_depth, dL_dout_flow, dL_dout_acc, dL_dmeans2D, dL_dcolors, dL_dopacity,
                         dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, dL_ddir, bwd_scratch, stream_);
}

int ex4d_backward_split_sh(
    const Ex4dParams *prm, int32_t num_rendered,
    const float *background, const float *means3D, const int32_t *radii,
    const Ex4dSplitSH *shs, const float *scales, const float *rotations,
    const float *cov3D_precomp, const float *viewmatrix, const float *projmatrix, const float *campos,
    const float *subpixel_offset, const float *out_depth, const float *out_acc,
    const void *geom_buffer, const void *binning_buffer, const void *img_buffer,
    const float *dL_dout_color, const float *dL_dout_depth, const float *dL_dout_flow, const float *dL_dout_acc,
    float *dL_dmeans2D, float *dL_dcolors, float *dL_dopacity, float *dL_dmeans3D, float *dL_dcov3D, const Ex4dSplitSHGrad *dL_dsh,
    float *dL_dscales, float *dL_drotations, float *dL_ddir, void *bwd_scratch, void *stream_)
{
    g_err[0] = 0;
    if (!shs || !dL_dsh || !split_ok(prm, shs->dc, shs->rest, shs->n_static) || dL_dsh->n_static != shs->n_static ||
        !split_ok(prm, (const float *const *)dL_dsh->dc, (const float *const *)dL_dsh->rest, dL_dsh->n_static))
        return fail(EX4D_ERR_ARG, "split SH: null tensor, n_static outside [0, P] or mismatching gradient split");
    if (prm->M != 16) return fail(EX4D_ERR_ARG, "split SH needs M == 16 (dc [n,1,3] + rest [n,15,3])");
    const ShSplit sp = { { shs->dc[0], shs->dc[1] }, { shs->rest[0], shs->rest[1] }, shs->n_static };
    const ShSplitGrad gsp = { { dL_dsh->dc[0], dL_dsh->dc[1] }, { dL_dsh->rest[0], dL_dsh->rest[1] }, dL_dsh->n_static };
    return backward_impl(prm, sp, gsp, num_rendered, background, means3D, radii, nullptr, nullptr, scales, rotations,
                         cov3D_precomp, viewmatrix, projmatrix, campos, subpixel_offset, out_depth, out_acc, geom_buffer, binning_buffer,
                         img_buffer, dL_dout_color, dL_dout_depth, dL_dout_flow, dL_dout_acc, dL_dmeans2D, dL_dcolors, dL_dopacity,
                         dL_dmeans3D, dL_dcov3D, nullptr, dL_dscales, dL_drotations, dL_ddir, bwd_scratch, stream_);
}

int ex4d_set_option(const char *name, int value)
{
    if (name && !strcmp(name, "composite_bwd_variant") && (value == 4 || value == 8)) { g_bwd_variant.store(value); return EX4D_OK; }
    if (name && !strcmp(name, "composite_fwd_asm") && (value == 0 || value == 1)) { ex4d_set_fwd_asm(value); return EX4D_OK; }
    if (name && !strcmp(name, "binning_tile_ids") && (value == 0 || value == 1)) { g_tile_ids.store(value); return EX4D_OK; }
    if (name && !strcmp(name, "composite_clamp_always") && (value == 0 || value == 1)) { ex4d_set_clamp_always(value); return EX4D_OK; }
    if (name && !strcmp(name, "composite_bwd_pairs") && (value == 0 || value == 1)) { ex4d_set_bwd_pairs(value); return EX4D_OK; }
    if (name && !strcmp(name, "tile_sort_rows") && (value == 0 || value == 1)) { g_tile_rows.store(value); return EX4D_OK; }
    if (name && !strcmp(name, "rank_lds_atomics") && value >= -1 && value <= 1) { ex4d_set_rank_lds(value); return EX4D_OK; }
    if (name && !strcmp(name, "rows_probe")) { ex4d_set_rows_probe(value); return EX4D_OK; }
    if (name && !strcmp(name, "preprocess_sh_predicate") && (value == 0 || value == 1)) { ex4d_set_preprocess_tune(value); return EX4D_OK; }
    if (name && !strcmp(name, "geom_debug_arrays") && (value == 0 || value == 1)) { g_geom_debug.store(value); return EX4D_OK; }
    if (name && !strcmp(name, "preprocess_fast_path") && (value == 0 || value == 1)) { ex4d_set_preprocess_fast(value); return EX4D_OK; }
    if (name && !strcmp(name, "depth_sort_msd") && value >= 0 && value <= 3) { g_depth_msd.store(value); g_depth_watch.reset(); return EX4D_OK; }
    if (name && !strcmp(name, "depth_sort_local_cap") && value >= 0 && value <= 8192) { g_depth_local_cap.store(value); return EX4D_OK; }
    if (name && !strcmp(name, "readback_side_stream") && (value == 0 || value == 1)) { g_readback_side.store(value); return EX4D_OK; }
    if (name && !strcmp(name, "depth_sort_msd_bits") && (value == 0 || value == EX4D_DLS_MSD_BITS - 1 || value == EX4D_DLS_MSD_BITS)) { g_depth_msd_bits.store(value); return EX4D_OK; }
    if (name && !strcmp(name, "depth_sort_local_threads") && (value == 0 || value == 256 || value == 512)) { g_depth_local_threads.store(value); return EX4D_OK; }
    return fail(EX4D_ERR_ARG, "unknown option or value out of range");
}

int ex4d_debug_bwd_stats(unsigned long long *out8, int reset) { return ex4d_bwd_stats(out8, 8, reset) == hipSuccess ? EX4D_OK : EX4D_ERR_HIP; }
int ex4d_debug_rows_prof(unsigned long long *out8, int reset) { return ex4d_rows_prof(out8, reset) == hipSuccess ? EX4D_OK : EX4D_ERR_HIP; }
int ex4d_debug_bwd_stats16(unsigned long long *out16, int reset) { return ex4d_bwd_stats(out16, 16, reset) == hipSuccess ? EX4D_OK : EX4D_ERR_HIP; }

int ex4d_get_option(const char *name)
{
    if (name && !strcmp(name, "composite_bwd_variant")) return g_bwd_variant.load();
    if (name && !strcmp(name, "composite_fwd_asm")) return ex4d_get_fwd_asm();
    if (name && !strcmp(name, "binning_tile_ids")) return g_tile_ids.load();
    if (name && !strcmp(name, "composite_clamp_always")) return ex4d_get_clamp_always();
    if (name && !strcmp(name, "composite_bwd_pairs")) return ex4d_get_bwd_pairs();
    if (name && !strcmp(name, "tile_sort_rows")) return g_tile_rows.load();
    if (name && !strcmp(name, "rank_lds_atomics")) return ex4d_get_rank_lds();
    if (name && !strcmp(name, "rank_lds_atomics_in_use")) return ex4d_rank_lds_in_use();      // (read-only) what the probe decided for the current device
    if (name && !strcmp(name, "preprocess_sh_predicate")) return ex4d_get_preprocess_tune();
    if (name && !strcmp(name, "geom_debug_arrays")) return g_geom_debug.load();
    if (name && !strcmp(name, "preprocess_fast_path")) return ex4d_get_preprocess_fast();
    if (name && !strcmp(name, "depth_sort_msd")) return g_depth_msd.load();
    if (name && !strcmp(name, "depth_sort_hold")) return g_depth_watch.hold.load();        // (read-only) auto mode: frames left on the LSD sort
    if (name && !strcmp(name, "depth_sort_trips")) return g_depth_watch.trips.load();      // (read-only) auto mode: oversize buckets seen since the option was set
    if (name && !strcmp(name, "depth_sort_local_cap")) return g_depth_local_cap.load();
    if (name && !strcmp(name, "readback_side_stream")) return g_readback_side.load();
    if (name && !strcmp(name, "depth_sort_msd_bits")) return g_depth_msd_bits.load();
    if (name && !strcmp(name, "depth_sort_local_threads")) return g_depth_local_threads.load();
    return -1;
}

void ex4d_profile_enable(int on) { g_prof_on.store(on != 0); }

int ex4d_profile_read(int which, float *ms, const char **names, int max_stages)
{
    std::lock_guard<std::mutex> lock(g_prof.mu);
    if (which < 0 || which > 1 || !g_prof.created) return 0;
    const int n = g_prof.n[which] < max_stages ? g_prof.n[which] : max_stages;
    if (n == 0) return 0;
    if (hipEventSynchronize(g_prof.ev[which][g_prof.n[which]]) != hipSuccess) return 0;
    for (int i = 0; i < n; i++) {
        float t = 0.f;
        (void)hipEventElapsedTime(&t, g_prof.ev[which][i], g_prof.ev[which][i + 1]);
        ms[i] = t;
        if (names) names[i] = g_prof.names[which][i];
    }
    return n;
}

int ex4d_mark_visible(int32_t P, const float *means3D, const float *viewmatrix, const float *projmatrix,
                      float min_depth, float max_depth, uint8_t *present, void *stream_)
{
    g_err[0] = 0;
    if (P <= 0) return EX4D_OK;
    if (!means3D || !viewmatrix || !projmatrix || !present) return fail(EX4D_ERR_ARG, "null argument");
    HIP_TRY(ex4d_launch_mark_visible(P, means3D, viewmatrix, projmatrix, min_depth, max_depth, present, (hipStream_t)stream_));
    return EX4D_OK;
}

}  // extern "C"
