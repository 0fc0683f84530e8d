// Per-tile front-to-back alpha compositing (forward) and its backward for gfx950.
//
// Replaces (CR/ = submodules/diff_gaussian_rasterization_df/cuda_rasterizer/):
//   renderCUDA fwd  CR/forward.cu:274-462     renderCUDA bwd  CR/backward.cu:426-682
//
// MI355X design (not the reference's 256-thread cooperative block):
//   * a 16x16 tile is handled by 4 INDEPENDENT wave64s, one per 8x8 pixel quadrant -- no block barrier
//     anywhere, a quadrant stops as soon as ITS 64 pixels are done;
//   * every visible Gaussian has ONE 64-byte record (ex4d_internal.h: SplatRecord) written by the
//     preprocess kernel, so a gather touches exactly one cache line instead of five arrays;
//   * each wave streams the tile's depth-sorted list 64 entries at a time: one coalesced load of ids, a
//     gather of the first 24 record bytes, an exact conservative ellipse-vs-quadrant test (can this
//     Gaussian reach alpha >= 1/255 anywhere in the quadrant?), wave64 ballot + prefix-popcount compaction
//     of the survivors into the wave's private LDS slice (order preserved; the original list position is
//     kept so n_contrib keeps the reference's meaning);
//   * the 64 lanes then walk the compacted list with wave-uniform LDS broadcasts, one flat predicate per
//     pair (no nested divergent branches: SALU mask traffic costs as much issue bandwidth as VALU here);
//   * backward: lanes = (Gaussian, pixel slot) -- batches of 16 list entries x 4 pixels per step, the per-pixel recurrences as
//     16-lane DPP prefix scans with carries, the 13 per-Gaussian partial sums accumulated over TIME in registers, and one
//     64-byte-row atomic instruction per 4 Gaussians (see "Compositing backward" below); it streams the forward's per-quadrant
//     compacted lists (BinState::qlist) back to front, starting at the quadrant's deepest contributor.
// Tiles are assigned to workgroups XCD-aware: workgroup b runs on XCD b % 8, so each XCD gets a
// contiguous band of the screen and neighbouring tiles (which share Gaussians) hit the same L2.
// Arithmetic follows the reference's expressions; FMA contraction is allowed here (results are compared
// to the oracle within 1e-5, not bit-exactly) and exp() is the hardware v_exp_f32.
#include "ex4d_internal.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <atomic>

namespace {

__device__ __forceinline__ void wave_lds_sync()
{
    // LDS traffic of one wave is processed in order; only the compiler has to be kept from reordering
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// gfx950 issues the VOP2 encoding of v_cndmask_b32 (implicit VCC) in ~24 cycles but the VOP3 encoding in ~5 (measured,
// tools/dev/micro/pk_rate.hip), and the compiler shrinks every select whose condition sits in VCC to VOP2.  Selects and
// wave votes inside the per-Gaussian loops are therefore spelled out: condition as a 64-bit lane mask in SGPRs.
// Conditions are kept as 64-bit lane masks in SGPRs: a ballot of ONE compare is the v_cmp itself, masks combine on the
// scalar unit, "no lane" is a scalar test (a ballot of a combined bool costs a v_cndmask + v_cmp round trip instead).
typedef unsigned long long lanemask;
#define LANES(cmp) __builtin_amdgcn_ballot_w64(cmp)
// gfx940 / gfx950 hazards the compiler pads for its OWN instructions but not for a consumer inside an asm string (it treats the
// statement as one opaque instruction and only pads behind it): a transcendental result (v_exp / v_rcp / v_log ...) read by the next
// VALU instruction needs 1 wait state, an SGPR pair written by a VALU compare and read as lane mask needs 2.  Round 3: once the
// scheduler had sunk v_exp_f32 behind the skip branch of the backward step, `v_exp_f32 v2, ...` sat directly in front of
// `v_cndmask_b32_e64 v21, v5, v2, s[8:9]` and every gradient came out wrong (deterministically; the forward was unaffected) -- while
// the same source with one more instruction in between was exact.  The selects therefore carry their own two wait states.
__device__ __forceinline__ float select_f(lanemask m, float if_set, float if_clear)
{
    float r;
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
    return r;
}
__device__ __forceinline__ int select_i(lanemask m, int if_set, int if_clear)
{
    int r;
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
    return r;
}

// Round 3: the kernels evaluate q2 = -power2 >= 0 (the same contraction with every operand negated: bit for bit -power2, and
// +0 -- never -0 -- at dx = dy = 0 whatever the sign of B), so that the reference's two skips
//     power > 0.0f  (CR/forward.cu:372)   and   alpha < 1/255  (:379),  alpha = min(0.99, w exp(power)),
// become ONE unsigned compare of bit patterns:  bits(q2) <= bits(tauq),  tauq = log2(255 w) >= 0:  a negative q2 (power > 0) has its
// sign bit set and fails; q2 > tauq <=> w 2^-q2 < 1/255 up to the rounding of log2 / exp2 (1e-7 relative in alpha: inside the band
// of 1e-4 the parity contract calls fragile).  Forward and backward take the decision from the same q2 and tauq bits, so the
// backward still re-derives exactly the contributor set the forward composited.  G = exp2(-q2) costs nothing (source modifier).
__device__ __forceinline__ float q2_rows(float dx, float ap, float bdy, float cdydy) { return __builtin_fmaf(dx, __builtin_fmaf(dx, -ap, -bdy), -cdydy); }
__device__ __forceinline__ float q2_of(float dx, float dy, float ap, float bp, float cp) { return q2_rows(dx, ap, bp * dy, (cp * dy) * dy); }
__device__ __forceinline__ uint32_t tauq_bits_of(float w) { return __float_as_uint(__builtin_amdgcn_logf(255.0f * w)); }     // v_log_f32 = log2

// a wave-uniform float into an SGPR (the builtin takes an int: go through the bit pattern, not through a value conversion)
__device__ __forceinline__ float uniform_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

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

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { uint32_t t = __shfl_xor(v, o, 64); v = t < v ? t : v; }
    return v;
}

// Can this Gaussian reach alpha >= 1/255 at ANY sample position inside [bx0,bx1]x[by0,by1]?
// alpha = w*exp(-q(d)), q(d) = 0.5 (A dx^2 + C dy^2) + B dx dy  =>  needs  min_box q <= ln(255 w) =: tau.
// Conservative (never culls a pair the per-pixel test would accept): tau carries a slack of 1% in alpha, plus a
// rounding allowance proportional to the magnitude of the terms.  tau, k1 = -B/C and k2 = -B/A come precomputed
// from the preprocess kernel (tau = -inf: w < 1/255, cull always; tau = +inf: conic not provably convex, keep).
__device__ __forceinline__ bool quadrant_cull(float mx, float my, float A, float B, float C, float tau, float k1, float k2,
                                              float bx0, float bx1, float by0, float by1)
{
    const float dxc = fminf(fmaxf(mx, bx0), bx1) - mx;     // nearest box point - mean (0 if inside the slab)
    const float dyc = fminf(fmaxf(my, by0), by1) - my;
    if (dxc == 0.f && dyc == 0.f) return tau < -3.0e38f;   // mean inside the box: only a never-visible Gaussian is culled
    float qmin = 3.0e38f, mag = 0.f;
    if (dxc != 0.f) {       // facing vertical edge: minimise over dy
        const float dy = fminf(fmaxf(k1 * dxc, by0 - my), by1 - my);
        const float t1 = 0.5f * A * dxc * dxc, t2 = 0.5f * C * dy * dy, t3 = B * dxc * dy;
        const float q = t1 + t2 + t3;
        if (q < qmin) { qmin = q; mag = fabsf(t1) + fabsf(t2) + fabsf(t3); }
    }
    if (dyc != 0.f) {       // facing horizontal edge: minimise over dx
        const float dx = fminf(fmaxf(k2 * dyc, bx0 - mx), bx1 - mx);
        const float t1 = 0.5f * A * dx * dx, t2 = 0.5f * C * dyc * dyc, t3 = B * dx * dyc;
        const float q = t1 + t2 + t3;
        if (q < qmin) { qmin = q; mag = fabsf(t1) + fabsf(t2) + fabsf(t3); }
    }
    return qmin > tau + 1e-5f * mag;
}

__constant__ const float kHalfLog2e = -0.5f * 1.4426950408889634f;   // -1/2 log2(e)
__constant__ const float kNegLog2e = -1.4426950408889634f;

struct PixelGeom { int px, py; bool inside; float fx, fy; uint32_t pix_id; };

__device__ __forceinline__ PixelGeom pixel_of_lane(int tile, int quad, int gx, int W, int H, const float *__restrict__ subpixel_offset)
{
    const int lane = threadIdx.x & 63;
    PixelGeom p;
    p.px = (tile % gx) * EX4D_TILE + (quad & 1) * 8 + (lane & 7);
    p.py = (tile / gx) * EX4D_TILE + (quad >> 1) * 8 + (lane >> 3);
    p.inside = p.px < W && p.py < H;
    p.pix_id = (uint32_t)(W * p.py + p.px);
    p.fx = (float)p.px; p.fy = (float)p.py;
    if (p.inside && subpixel_offset) {
        const float2 o = reinterpret_cast<const float2 *>(subpixel_offset)[p.pix_id];
        p.fx += o.x; p.fy += o.y;
    }
    return p;
}

// workgroup -> (tile, quadrant), XCD-aware: consecutive workgroup ids round-robin over the 8 XCDs (each with its own L2), so XCD x
// gets the contiguous tile band [x*chunk, (x+1)*chunk).  One workgroup = one wave = one quadrant of a tile: the four quadrant waves of
// a tile are independent, and as separate workgroups each gives its wave slot and LDS back as soon as IT is done (round 3: as one
// 256-thread workgroup the forward held four slots until its slowest quadrant finished, 0.150 -> 0.144 ms).
// Measured and dropped in round 3 (profiles/archive/r03_compositing_experiments.txt): stripes of 1 / 2 / 4 / 8 tile rows dealt to the XCDs
// instead of bands, and "heaviest tiles of a band first" (a counting sort of the band's tiles by list length in front of the
// forward) -- neither moved either compositing kernel by more than the box-to-box noise: they do not end on a few late heavy tiles.
__device__ __forceinline__ void tile_of_block(int num_tiles, int &tile, int &quad)
{
    const int chunk = (num_tiles + 7) >> 3;
    const int i = (int)(blockIdx.x >> 3);
    tile = (int)(blockIdx.x & 7) * chunk + (i >> 2);
    quad = i & 3;
}

// ------------------------------------------------------------------------------------------------
// Compositing forward: four independent quadrant waves per tile (no workgroup barrier), each streaming the tile list in chunks of 64.
// FLOW = false: no Gaussian of the frame carries a non-zero dir3D (the training loop passes the all-zero gradient-trap tensor,
// gaussian_renderer/__init__.py:66-70): the flow image is zero, its three accumulations per pair and the staging of dir3D are skipped.
//
// Round 3, measured on MI355X (1.0 M Gaussians, rocprofv3 counters in profiles/archive/r03_fwd_experiment.txt): the kernel is VALU-issue bound
// to the instruction -- 9.10e7 wave instructions x 4 cycles / 1024 SIMDs = 0.148 ms = its duration; a quadrant composites ~147 list
// entries (3.2 M (entry, quadrant) pairs reach the pixel loop, 87 % of them with a contributing pixel), half of the entries of the
// chunks it looks at survive the quadrant cull, and it stops at a fifth of the list.  Tried and rejected with numbers:
//   * b'dy and (c'dy)dy per (Gaussian, quadrant row) from an LDS table: -11 % instructions, but the per-lane (non-broadcast) LDS read
//     in the pair loop and the larger footprint made it SLOWER (0.165 ms with 64 staging slots, 0.153 with 32);
//   * tile-cooperative culling (one gather and four culls per entry and tile, per-quadrant queues, two barriers per 256 entries):
//     0.192 ms -- a fifth of the list is all a tile ever reads, and the barriers serialise its four latency chains into one.
// What stayed: T (1 - alpha) as one fma and T -= weight instead of a select; the two skips of the reference as one unsigned compare
// (q2_rows above); the pair loop unrolled by two by hand (immediate LDS offsets).  Per staged Gaussian and pixel: 13 VALU before the
// decision, 12 more when a pixel contributes.
// The dominant index (CR/forward.cu:411-415: strict `>` on the full-precision weight alpha T, i.e. the FIRST entry attaining the
// maximum) is exact (round 4; rounds 1-3 compared weights quantised to 26 bits).  A contributing weight lies in [1e-4 / 255, 0.99], so
// its sign and the three high exponent bits are constant: `bits << 4` keeps all 28 significant bits and leaves 4 bits for the position
// inside a GROUP of 16 staged entries (15 - j, larger for earlier entries: equal weights keep the first one).  Inside a group the
// running best is one v_lshl_or_b32 + v_max_u32 per contributing pair; at the end of every group the group's best is folded into the
// pixel's running (key, location) with the strict compare `key > (best_key | 15)` -- 4 VALU per 16 entries.
// Every chunk's survivors are appended to the quadrant's compacted list (BinState::qlist) -- all the backward reads of the tile list.
std::atomic<int> g_fwd_asm{1};
std::atomic<int> g_alpha_clamp_always{0};
std::atomic<int> g_bwd_pairs{1};                 // "composite_bwd_pairs": the compositing backward handles two pixels per lane on packed math where it applies (round 6)      // "composite_clamp_always" = 1: every chunk / batch evaluates min(0.99, w G) (rounds 1-5; A/B runs)      // compositing forward: hand-scheduled walk (1, default) or the compiler's (0)

struct FwdLds {                  // per wave: the survivors of one 64-entry chunk (4.5 KB; the kernel's VGPRs, not LDS, bound its occupancy)
    float4 q0[64];               // mean.x, mean.y, a', b'
    float4 q1[64];               // c', w, tauq (bits), blue
    float4 c[64];                // red, green, depth, 1.0   (two register pairs: (C0, C1) += (r, g) wgt and (Dm, acc) += (depth, 1) wgt are one v_pk_fma_f32 each)
    float4 f[64];                // dir3D (frames with flow only)
    uint2 it[64];                // Gaussian id, position in the tile list
};
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) f32x4 *lds_float4_ptr;

// One staged entry of the hand-scheduled walk (see composite_fwd_body): CUR0 / CUR1 = the register quads holding this entry's test
// operands (mean.x mean.y a' b' | c' w tauq blue), NXT0 / NXT1 = the quads the NEXT entry's operands are requested into; this entry's
// (red, green, depth, 1) is requested into CUR0 as soon as the mean and the conic have been consumed; v60-v63 temporaries.  Branch targets carry the statement's unique id (%=) and the set's letter.
#define EX4D_FWD_CLAMP "v_min_f32 v60, 0x3f7d70a4, v60\n\t"              /* alpha = min(0.99, w G) */
#define EX4D_FWD_NOCLAMP ""                                            /* w <= 0.99 and G <= 1 in range: w G <= 0.99 already */
#define EX4D_FWD_ENTRY(S, CLAMP, CUR0, CX, CY, CA, CB, CC, CW, CT, CBLUE, NXT0, NXT1, G2RG, G2DA, ONXT0, ONXT1, OCOL, LAST, ADVANCE) \
    "ds_read_b128 " NXT0 ", %[va] offset:" ONXT0 "\n\t" \
    "ds_read_b128 " NXT1 ", %[va] offset:" ONXT1 "\n\t" \
    "s_waitcnt lgkmcnt(2)\n\t"                                   /* this entry's operands (requested one entry earlier) have arrived */ \
    "v_sub_f32 v60, " CY ", %[fy]\n\t" \
    "v_sub_f32 v61, " CX ", %[fx]\n\t" \
    "v_mul_f32 v62, v60, " CC "\n\t" \
    "v_mul_f32_e64 v63, " CB ", -v60\n\t" \
    "v_fma_f32 v63, v61, -" CA ", v63\n\t" \
    "ds_read_b128 " CUR0 ", %[va] offset:" OCOL "\n\t"           /* (red, green, depth, 1) into the quad whose mean / conic are consumed */ \
    "v_mul_f32_e64 v62, v62, -v60\n\t" \
    "v_fmac_f32 v62, v61, v63\n\t"                               /* q2 = -power log2(e) */ \
    "v_exp_f32_e64 v60, -v62\n\t" \
    "v_cmp_ge_u32_e64 %[ok], " CT ", v62\n\t"                    /* bits(q2) <= bits(tauq): CR/forward.cu:372 and :379 in one compare */ \
    "s_and_b64 %[ok], %[ok], %[live]\n\t" \
    "s_cbranch_scc0 Lskip" S "_%=\n\t"                           /* no live lane in range: nothing to do for this entry */ \
    "v_mul_f32 v60, " CW ", v60\n\t" \
    CLAMP \
    "v_fma_f32 v61, -%[T], v60, %[T]\n\t"                        /* test_T = T (1 - alpha), CR/forward.cu:383 */ \
    "v_cmp_gt_f32_e64 %[stop], %[thr], v61\n\t" \
    "s_and_b64 %[stop], %[stop], %[ok]\n\t" \
    "s_cbranch_scc1 Lrare" S "_%=\n"                              /* some lane saturates here (CR/forward.cu:384-388): off the common path */ \
    "Ladd" S "_%=:\n\t" \
    "s_mov_b64 exec, %[ok]\n\t"                                  /* CR/forward.cu:389-422 for the contributing lanes only */ \
    "v_mul_f32 v62, %[T], v60\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t"                                   /* this entry's colours */ \
    "v_pk_fma_f32 %[crg], " G2RG ", v[62:63], %[crg] op_sel_hi:[1,0,1]\n\t" \
    "v_pk_fma_f32 %[dacc], " G2DA ", v[62:63], %[dacc] op_sel_hi:[1,0,1]\n\t" \
    "v_fmac_f32 %[c2], " CBLUE ", v62\n\t" \
    "v_lshl_or_b32 v61, v62, 4, %[jkey]\n\t"                     /* dominant index key: (weight bits << 4) | (15 - j mod 16), exact */ \
    "v_sub_f32 %[T], %[T], v62\n\t" \
    "v_max_u32 %[best], %[best], v61\n\t" \
    LAST                                                         /* the lane's last contributing entry, as its LDS address */ \
    "s_mov_b64 exec, -1\n" \
    "Lskip" S "_%=:\n\t" \
    "s_add_i32 %[jkey], %[jkey], -1\n\t" \
    ADVANCE \
    "s_cmp_lg_u32 %[jkey], %[jend]\n\t"
#define EX4D_FWD_RARE(S, V) \
    "Lrare" S "_%=:\n\t" \
    "s_andn2_b64 %[live], %[live], %[stop]\n\t" \
    "s_andn2_b64 %[ok], %[ok], %[stop]\n\t"                      /* the lanes that still add */ \
    "s_cbranch_scc1 Ladd" S "_%=\n\t" \
    "s_cmp_eq_u64 %[live], 0\n\t" \
    "s_cbranch_scc1 Ldead" S "_%=\n\t" \
    "s_branch Lskip" S "_%=\n"

// the walk over the staged entries of a chunk (labels carry the variant tag V: both variants live in ONE asm statement -- two statements in
// the arms of an if made the compiler carry the lane masks in vector registers)
#define EX4D_FWD_WALK_BODY(V, CLAMP) \
                    "Lgroup" V "_%=:\n\t" \
                    "s_sub_i32 %[jend], 15, %[rem]\n\t" \
                    "s_max_i32 %[jend], %[jend], -1\n\t" \
                    "s_mov_b32 %[jkey], 15\n" \
                    "Lloop" V "_%=:\n\t" \
                    EX4D_FWD_ENTRY("a" V, CLAMP, "v[40:43]", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v[48:51]", "v[52:55]", "v[40:41]", "v[42:43]", \
                                   "16", "1040", "2048", "v_mov_b32 %[last], %[va]\n\t", "") \
                    "s_cbranch_scc0 Lgodd" V "_%=\n\t" \
                    EX4D_FWD_ENTRY("b" V, CLAMP, "v[48:51]", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v[40:43]", "v[44:47]", "v[48:49]", "v[50:51]", \
                                   "32", "1056", "2064", "v_add_u32 %[last], 16, %[va]\n\t", "v_add_u32 %[va], 32, %[va]\n\t") \
                    "s_cbranch_scc1 Lloop" V "_%=\n" \
                    "Lgdone" V "_%=:\n\t" \
                    "v_or_b32 v60, 15, %[dkey]\n\t" \
                    "v_cmp_gt_u32_e64 %[ok], %[best], v60\n\t" \
                    "s_add_i32 %[rem], %[rem], -16\n\t" \
                    "s_nop 1\n\t" \
                    "v_cndmask_b32_e64 %[dkey], %[dkey], %[best], %[ok]\n\t" \
                    "v_cndmask_b32_e64 %[dloc], %[dloc], %[va], %[ok]\n\t" \
                    "v_mov_b32 %[best], 0\n\t" \
                    "s_cmp_gt_i32 %[rem], 0\n\t" \
                    "s_cbranch_scc1 Lgroup" V "_%=\n\t" \
                    "s_branch Ldone_%=\n" \
                    EX4D_FWD_RARE("a" V, V) \
                    EX4D_FWD_RARE("b" V, V) \
                    "Lgodd" V "_%=:\n\t"                         /* the group ended on set a (the chunk's last, odd count): va behind that entry */ \
                    "v_add_u32 %[va], 16, %[va]\n\t" \
                    "s_branch Lgdone" V "_%=\n" \
                    "Ldeadb" V "_%=:\n\t"                        /* every lane is done: va behind the current entry, nothing left of the chunk */ \
                    "v_add_u32 %[va], 16, %[va]\n" \
                    "Ldeada" V "_%=:\n\t" \
                    "v_add_u32 %[va], 16, %[va]\n\t" \
                    "s_mov_b32 %[rem], 0\n\t" \
                    "s_branch Lgdone" V "_%=\n"

template <bool FLOW, bool ASMLOOP>
__device__ __forceinline__ void composite_fwd_body(
    int W, int H, int gx, int tile, int wave, const PixelGeom p,
    const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list,
    const float4 *__restrict__ records,
    const float *__restrict__ bg, float max_depth,
    float *__restrict__ final_T, uint32_t *__restrict__ n_contrib,
    float *__restrict__ out_color, float *__restrict__ out_depth, float *__restrict__ out_acc,
    float *__restrict__ out_flow, int32_t *__restrict__ out_idx, uint32_t *__restrict__ qlist, uint32_t *__restrict__ qcount,
    FwdLds &L, int clamp_always)
{
    const int lane = threadIdx.x & 63;
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    const float bx0 = wave_min(p.fx), bx1 = wave_max(p.fx), by0 = wave_min(p.fy), by1 = wave_max(p.fy);
    const uint64_t lt = (1ull << lane) - 1ull;
    // this quadrant's compacted list for the backward pass (ex4d_internal.h: BinState::qlist)
    uint32_t *const my_list = qlist + 4 * (size_t)range.x + (size_t)wave * (size_t)n;
    int consumed = 0;

    lanemask live = LANES(p.inside);          // lanes still compositing (CR/forward.cu: !done)
    float T = 1.0f, C2 = 0.f, F0 = 0.f, F1 = 0.f, F2 = 0.f;
    f32x2 Crg = { 0.f, 0.f }, Dacc = { 0.f, 0.f };           // (C0, C1), (sum of depth weight, sum of weights)
    uint32_t last_contributor = 0;
    int32_t best = -1;
    // dominant index (CR/forward.cu:411-415), exact: the hand-scheduled walk keeps the pixel's largest weight as the integer key
    // (bits << 4) | (15 - position in its group of 16), the C++ walk as the float itself; both orders are the reference's strict `>`
    uint32_t dom_key = 0;
    float max_vis = 0.f;
    // LDS byte address of this wave's staging area (the low half of the flat address of a __shared__ object IS its LDS address)
    const uint32_t lds_base = (uint32_t)(uintptr_t)&L.q0[0];

    for (int base = 0; base < n; base += 64) {
        if (live == 0) break;
        const int k = base + lane;
        bool keep = false;
        uint32_t id = 0;
        float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 q1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < n) {
            id = point_list[range.x + k];
            const float4 *r = records + 4 * (size_t)id;
            q0 = r[0];
            q1 = r[1];
            keep = !quadrant_cull(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, bx0, bx1, by0, by1);
        }
        const uint64_t mask = __ballot(keep);
        const int cnt = __popcll(mask);
        bool sat = false;
        if (keep) {
            const int slot = __popcll(mask & lt);
            const float4 *r = records + 4 * (size_t)id;
            const float4 q2 = r[2];
            const float4 q3 = r[3];
            sat = clamp_always || !(q3.w <= 0.99f);
            // alpha = w exp(power) = w exp2(dx (a' dx + b' dy) + c' dy^2): fold -1/2 and log2(e) once per Gaussian
            L.q0[slot] = make_float4(q0.x, q0.y, q0.z * kHalfLog2e, q0.w * kNegLog2e);
            L.q1[slot] = make_float4(q1.x * kHalfLog2e, q3.w, __uint_as_float(tauq_bits_of(q3.w)), q2.w);
            L.c[slot] = make_float4(q2.y, q2.z, q2.x, 1.0f);
            if (FLOW) L.f[slot] = q3;
            const uint2 item = make_uint2(id, (uint32_t)k);
            L.it[slot] = item;
            my_list[consumed + slot] = (uint32_t)k;   // the survivor's position in the tile list: 4 bytes, coalesced (round 6; rounds 3-5: 8 with the id)
        }
        consumed += cnt;
        const lanemask any_sat = LANES(sat);      // some staged entry can reach the clamp of alpha (rare: opacity x coefficient > 0.99)
        wave_lds_sync();
        // The walk over the staged entries keeps the entry's LDS address in a VGPR that the compiler cannot prove uniform (a uniform
        // index lives in an SGPR and costs a v_mov per ds_read): one v_add per two entries, every read at an immediate offset, and the
        // last contributing entry is remembered as that address (even / odd entries apart) instead of as a materialised index.
        uint32_t va;
        asm volatile("v_mov_b32 %0, %1" : "=v"(va) : "s"(lds_base));
        int last_even = -1, last_odd = -1;     // LDS address of the pair whose even / odd entry contributed last
        int dom_j = -1;                        // staged entry that became the pixel's dominant contributor in this chunk (none: -1)
        if (ASMLOOP && !FLOW) {
            // Hand-scheduled walk (round 3): the compiler's version of this loop spends as many instructions on lane-mask and loop
            // bookkeeping as on arithmetic and waits twice per entry for LDS.  Here: the next entry's operands are requested while the
            // current one is evaluated (two register sets trade places), the contributing lanes are selected through EXEC (no selects,
            // no wait-state padding), an entry no live lane reaches leaves after 9 VALU, and a saturating lane (rare) is handled off
            // the common path.  Same operations on the same operands in the same order as the C++ walk below: same bits.
            // Round 4: the entries run in groups of 16 (jkey = 15 .. 0 is both the loop counter and the low bits of the dominant-index
            // key); at the end of a group its best key is folded into (dom_key, dom_loc) with the reference's strict compare, the
            // operand prefetch runs on across the group boundary (a full group ends on set b, the next one starts on set a).
            if (cnt > 0) {
                uint32_t jkey, jend, rem = (uint32_t)cnt;
                lanemask ok_m, stop_m;
                int last_addr = -1;
                uint32_t best_key = 0, dom_loc = 0;
                // Round 6: alpha = min(0.99, w G) needs its v_min only for entries with w > 0.99 (G <= 1 for every lane in range, and the
                // product of w <= 0.99 with a factor <= 1 rounds to at most w): chunks without such an entry -- all but a few per cent --
                // walk without it (one of 22 VALU instructions per entry and lane; same bits).
                asm volatile(
                    "s_waitcnt lgkmcnt(0)\n\t"                      // nothing of the compiler's in flight: the counts below are this block's own
                    "ds_read_b128 v[40:43], %[va]\n\t"
                    "ds_read_b128 v[44:47], %[va] offset:1024\n\t"
                    "s_cmp_lg_u64 %[sat], 0\n\t"
                    "s_cbranch_scc1 Lgroup1_%=\n"
                    EX4D_FWD_WALK_BODY("0", EX4D_FWD_NOCLAMP)
                    EX4D_FWD_WALK_BODY("1", EX4D_FWD_CLAMP)
                    "Ldone_%=:\n\t"
                    "s_mov_b64 exec, -1\n\t"
                    "s_waitcnt lgkmcnt(0)"
                    : [T] "+v"(T), [crg] "+v"(Crg), [dacc] "+v"(Dacc), [c2] "+v"(C2), [best] "+v"(best_key), [last] "+v"(last_addr),
                      [dkey] "+v"(dom_key), [dloc] "+v"(dom_loc), [va] "+v"(va), [live] "+s"(live), [rem] "+s"(rem),
                      [jkey] "=&s"(jkey), [jend] "=&s"(jend), [ok] "=&s"(ok_m), [stop] "=&s"(stop_m)
                    : [fx] "v"(p.fx), [fy] "v"(p.fy), [thr] "s"(0.0001f), [sat] "s"(any_sat)
                    : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55",
                      "v60", "v61", "v62", "v63", "vcc", "scc", "memory");
                last_even = last_addr;
                // dom_loc = the address behind the last entry of the group that holds the new dominant contributor (0: none in this chunk)
                if (dom_loc != 0u) dom_j = (int)((((dom_loc - 16u - lds_base) >> 4) & ~15u) + 15u - (dom_key & 15u));
            }
        } else {
            // one (pixel, staged Gaussian) pair per lane
            int dom_addr = -1;                 // LDS address of the entry that became the dominant contributor in this chunk
            auto pair = [&](const int off, int &last_addr) {
                const f32x4 g0 = *(lds_float4_ptr)(uintptr_t)(va + off);
                const f32x4 g1 = *(lds_float4_ptr)(uintptr_t)(va + off + 1024);
                // CR/forward.cu:368-387 as one flat predicate; q2 = -power log2(e)
                const float q2 = q2_of(g0.x - p.fx, g0.y - p.fy, g0.z, g0.w, g1.x);
                const float alpha = fminf(0.99f, g1.y * __builtin_amdgcn_exp2f(-q2));
                const float test_T = __builtin_fmaf(-T, alpha, T);   // T (1 - alpha), CR/forward.cu:383, as the asm walk's one fused multiply-add
                const float wgt_all = alpha * T;
                const lanemask ok = live & LANES(__float_as_uint(q2) <= __float_as_uint(g1.z));
                const lanemask stop = ok & LANES(test_T < 0.0001f);
                live &= ~stop;
                const lanemask add = ok & ~stop;
                if (add == 0) return;
                // CR/forward.cu:389-422 for all lanes: lanes outside `add` accumulate a zero weight
                const f32x4 g2 = *(lds_float4_ptr)(uintptr_t)(va + off + 2048);
                const float wgt = select_f(add, wgt_all, 0.f);
                const f32x2 ww = { wgt, wgt };
                Crg = __builtin_elementwise_fma((f32x2){ g2.x, g2.y }, ww, Crg);
                Dacc = __builtin_elementwise_fma((f32x2){ g2.z, g2.w }, ww, Dacc);
                C2 = __builtin_fmaf(g1.w, wgt, C2);
                if (FLOW) { const f32x4 g3 = *(lds_float4_ptr)(uintptr_t)(va + off + 3072); F0 += g3.x * wgt; F1 += g3.y * wgt; F2 += g3.z * wgt; }
                // dominant index (CR/forward.cu:411-415): strict > on the weight (lanes outside `add` carry a zero weight: never larger)
                const lanemask better = LANES(wgt > max_vis);
                max_vis = fmaxf(max_vis, wgt);
                dom_addr = select_i(better, (int)va + off, dom_addr);
                T -= wgt;                                          // the contributing lanes' new transmittance
                last_addr = select_i(add, (int)va, last_addr);
            };
            {
                int j = 0;
                for (; j + 1 < cnt; j += 2, va += 32) {
                    if (live == 0) break;
                    pair(0, last_even);
                    if (live == 0) break;
                    pair(16, last_odd);
                }
                if (j < cnt && live != 0) pair(0, last_even);          // odd count: the last entry (a break above leaves live == 0)
            }
            if (dom_addr >= 0) dom_j = (int)(((uint32_t)dom_addr - lds_base) >> 4);
        }
        const int je = last_even < 0 ? -1 : (int)(((uint32_t)last_even - lds_base) >> 4);
        const int jo = last_odd < 0 ? -1 : (int)(((uint32_t)last_odd - lds_base) >> 4) + 1;
        const int last_j = je > jo ? je : jo;
        if (last_j >= 0) last_contributor = L.it[last_j].y + 1;
        if (dom_j >= 0) best = (int32_t)L.it[dom_j].x;
        wave_lds_sync();
    }
    if (lane == 0) qcount[4 * tile + wave] = (uint32_t)consumed;
    if (p.inside) {
        // CR/forward.cu:426-460
        const float C0 = Crg.x, C1 = Crg.y, Dm = Dacc.x, acc = Dacc.y;
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

// FLOW is a launch-time choice (the frame flag travels with the instance count's read-back): the flow-free kernel needs 43 VGPRs, the
// one with both paths behind a run-time branch needed 66
template <bool FLOW, bool ASMLOOP>
__global__ __launch_bounds__(64) void composite_fwd_kernel(
    int W, int H, int gx, int num_tiles,
    const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list,
    const float *__restrict__ subpixel_offset, const float4 *__restrict__ records,
    const float *__restrict__ bg, float max_depth,
    float *__restrict__ final_T, uint32_t *__restrict__ n_contrib,
    float *__restrict__ out_color, float *__restrict__ out_depth, float *__restrict__ out_acc,
    float *__restrict__ out_flow, int32_t *__restrict__ out_idx, uint32_t *__restrict__ qlist, uint32_t *__restrict__ qcount, int clamp_always)
{
    __shared__ FwdLds lds;
    int tile, quad;
    tile_of_block(num_tiles, tile, quad);
    if (tile >= num_tiles) return;
    const PixelGeom p = pixel_of_lane(tile, quad, gx, W, H, subpixel_offset);
    composite_fwd_body<FLOW, ASMLOOP>(W, H, gx, tile, quad, p, ranges, point_list, records, bg, max_depth, final_T, n_contrib, out_color, out_depth, out_acc,
                             out_flow, out_idx, qlist, qcount, lds, clamp_always);
}

// ------------------------------------------------------------------------------------------------
// Compositing backward, scan version (default) -- lanes = (Gaussian, pixel slot).
//
// The per-pixel version above spends a third of its VALU instructions forming the 13 per-Gaussian partials and summing them over
// the 64 pixels of the wave (13 LDS stores + 4 x 16-byte reads + 17 adds + the atomic, per Gaussian).  Here the lanes are
// re-assigned so that those sums accumulate over TIME in registers instead of across lanes: lane (n, g) = (entry n of a batch of
// 16 list entries, pixel slot g); step s of a batch handles pixels 4s+g (g = 0..3) of the 8x8 quadrant for all 16 Gaussians at
// once, so after 16 steps every lane holds the partial sums of ITS Gaussian over ITS 16 pixels; the four pixel-slot lanes of a
// Gaussian are added once per batch.  The per-pixel recurrences of CR/backward.cu:571-680 run ACROSS the 16 Gaussian lanes of a
// DPP row as prefix scans with a per-pixel carry (in LDS) from batch to batch (list order descending = back to front):
//     T_i    = T_carry prod_{j<=i} 1/(1 - alpha_j)                                   (T = T / (1 - alpha))
//     E_i    = E_carry + sum_{j<i} alpha_j T_j (c_j . dL_dpixel)                     (accum_rec in closed form:
//              (c_i - accum_rec_i) . dL_dpixel T_i = (c_i . dL_dpixel) T_i - E_i / (1 - alpha_i))
//     gacc_i = gacc_carry prod_{j<=i, contributing} T_j                              (dL_dacc *= T)
// List compaction comes for free: the forward kernel leaves, per (tile, quadrant), the compacted list of the entries that survived its
// quadrant cull (BinState::qlist / qcount: Gaussian id + list position, in list order); this kernel streams it back to front and
// gathers the records of the survivors only.
//
// Measured on MI355X, 1.0 M Gaussians / 7.5 M instances (profiles/archive/r02_*): per-pixel kernel 0.458 ms, 2.32e8 VALU instructions;
// this kernel 0.35 ms, 1.79e8 (86 % VALU-busy; 55 % of the lanes of a step hold a contributing pair, 96 % of the staged
// Gaussians contribute somewhere in their quadrant).
//
// A formulation of the 13 sums as contractions on the matrix cores (three v_mfma_f32_16x16x4_f32 per step: lane (n, g) IS the B-operand
// layout) was built and measured in round 2: correct, but slower than 15 VALU FMAs per step (0.40 vs 0.36 ms) -- the f32-input MFMA runs
// at the f32 VECTOR rate and does not overlap the VALU stream on this part (tools/dev/micro/mfma_dpp_probe.hip).  Removed in round 3
// together with round 1's per-pixel kernel (DESIGN.md section 4 keeps the measurements).
// Accumulator row (per-Gaussian backward kernel, ex4d_preprocess.hip):  0,1 dL_dmean2D.xy without the factors ln2 W/2, ln2 H/2
//     2 dL_dmean2D.z   3..5 dL_dconic.(x,y,w) without -1/2   6 dL_dopacity   7..9 dL_dcolor   10..12 dL_ddir
// ------------------------------------------------------------------------------------------------
#define DPP_ROW_SHR(n) (0x110 + (n))

// lanes whose DPP source lies outside their 16-lane row keep `old`
template <int CTRL> __device__ __forceinline__ float dpp_or(float old, float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// Inclusive prefix product / sum over the 16 lanes of a DPP row (Hillis-Steele, shifts 1 2 4 8).  Product: one instruction
// per step -- v_mul_f32_dpp leaves lanes without a source untouched (the compiler would emit v_mov_dpp + v_mul); a DPP read of a
// VGPR written by the previous VALU instruction needs two wait states, which nobody inserts inside an asm statement but us.
__device__ __forceinline__ float row_scan_mul(float x)
{
    asm("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf" : "+v"(x));
    return x;
}
// list entries staged per wave (<= 15 left over + 64 new): 96 slots (indices wrap by a conditional subtraction) keep the wave's LDS at
// 9.75 KB = 16 waves per CU, what the registers allow as well
#define BWD_RING 96
template <int R> __device__ __forceinline__ int ring_wrap(int i) { return i >= R ? i - R : i; }       // 0 <= i < 2 R
#define DUMP_FLOATS 320          // per-wave scratch: junk target of the non-carry lanes (64 lanes + 15 steps x 16 floats), epilogue staging

// developer statistics of the scan kernel (variant 8 only): [0] batches, [1] valid Gaussians, [2] steps run, [3] steps skipped,
// [4] contributing (pixel, Gaussian) pairs, [5] Gaussians with at least one contributing pair in their batch,
// [6] (pixel, Gaussian) pairs whose Gaussian lies in front of the pixel's last contributor (alive by list position, in range or not),
// [7] Gaussians contributing in the TOP four rows of the quadrant only, [8] in the BOTTOM four rows only, [9] in both,
// [10] steps run in the top half, [11] in the bottom half, [12] batches with no contributing pair in the top half, [13] ... bottom half
__device__ unsigned long long g_bwd_stats[16];

// Round 6, pixel PAIRS: the per-pixel constants of the two pixels a lane handles in one row-step -- (row r, column g) and (row r, column 4 + g),
// pair index 4 r + g -- interleaved so that every quantity is a register pair (a, b) straight from a 16-byte read: the step then runs on
// v_pk_*_f32 (bwd_batch_pairs below).  Same LDS as the per-pixel arrays (a quadrant uses one layout or the other).
struct BwdPairConsts {
    float4 A[32];                // gp0.a gp0.b gp1.a gp1.b
    float4 A2[32];               // gp2.a gp2.b gdepth.a gdepth.b
    float4 B[32];                // final_depth.a .b  last_contributor.a .b
    float4 C[32];                // carries: T.a T.b (bgT - E).a .b
    float4 F[32];                // gflow0.a .b gflow1.a .b
    float2 F2[32];               // gflow2.a .b
};
template <int RING> struct BwdLdsT {
    float4 ring[3][RING];        // [0] x y a' b'   [1] c' w depth id   [2] r g b list-position     (also: transposition scratch at setup)
    union {
        struct {
            float4 pa[64];       // per pixel: gp0 gp1 gp2 gdepth
            float4 pb[64];       //            final_depth  last_contributor  T carry  (bgT - E) carry      (the two carries: one 8-byte store)
            float4 pc[64];       //            gflow0 gflow1 gflow2  gacc carry      (read only with depth / flow / acc gradients)
            float4 pd[64];       //            fx  fy  -  -                          (read only with sub-pixel offsets)
        };
        BwdPairConsts pp;
    };
    float dump[DUMP_FLOATS];
};

// a reduced over lane ^ 32 in lanes 0-31, b in lanes 32-63 (v_permlane32_swap: lanes 32-63 of the first operand trade places with
// lanes 0-31 of the second); the same over lane ^ 16 with rows: a in rows 0 and 2, b in rows 1 and 3
__device__ __forceinline__ float halves_sum(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float rows_sum(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ float row_scan_add_asm(float x)
{
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf" : "+v"(x));
    return x;
}

// The same two scans with the wait states of their DPP steps (a DPP read needs two wait states behind the VALU write of its source:
// `s_nop 1`, an instruction slot that does nothing) filled by independent work of the step: the colour dot product c . dL_dpixel beside
// the product scan, the dcc-weighted sums (and, with depth / flow gradients, the depth term) beside the sum scan.  Same operations on
// the same operands as the plain versions -- only their place in the instruction stream differs.
__device__ __forceinline__ float row_scan_mul_with_dot(float x, float &dot, float ax, float ay, float az, float bx, float by, float bz)
{
    asm("v_mul_f32 %[d], %[ax], %[bx]\n\t"
        "v_fmac_f32 %[d], %[ay], %[by]\n\t"
        "v_mul_f32_dpp %[x], %[x], %[x] row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32 %[d], %[az], %[bz]\n\t"
        "s_nop 0\n\t"
        "v_mul_f32_dpp %[x], %[x], %[x] row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %[x], %[x], %[x] row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %[x], %[x], %[x] row_shr:8 row_mask:0xf bank_mask:0xf"
        : [x] "+v"(x), [d] "=&v"(dot) : [ax] "v"(ax), [ay] "v"(ay), [az] "v"(az), [bx] "v"(bx), [by] "v"(by), [bz] "v"(bz));
    return x;
}
// sum scan of e; beside it: (c78.x, c78.y) += (p.x, p.y) d,  c9 += p.z d      (d = the low half of dd)
__device__ __forceinline__ float row_scan_add_with_sums(float e, f32x2 &c78, float &c9, f32x2 pxy, float pz, f32x2 dd, float d)
{
    asm("v_pk_fma_f32 %[c78], %[pxy], %[dd], %[c78] op_sel_hi:[1,0,1]\n\t"
        "v_fmac_f32 %[c9], %[pz], %[d]\n\t"
        "v_add_f32_dpp %[e], %[e], %[e] row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %[e], %[e], %[e] row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %[e], %[e], %[e] row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %[e], %[e], %[e] row_shr:8 row_mask:0xf bank_mask:0xf"
        : [e] "+v"(e), [c78] "+v"(c78), [c9] "+v"(c9) : [pxy] "v"(pxy), [pz] "v"(pz), [dd] "v"(dd), [d] "v"(d));
    return e;
}
// ... and with depth / flow gradients also (c1011.x, c1011.y) += (f.x, f.y) d,  c12 += f.z d,  gdT = gd T,  v2 += alpha gdT
__device__ __forceinline__ float row_scan_add_with_sums_extra(float e, f32x2 &c78, float &c9, f32x2 pxy, float pz, f32x2 dd, float d,
                                                              f32x2 &c1011, float &c12, f32x2 fxy, float fz, float &gdT, float gd, float T, float &v2, float alpha)
{
    asm("v_pk_fma_f32 %[c78], %[pxy], %[dd], %[c78] op_sel_hi:[1,0,1]\n\t"
        "v_fmac_f32 %[c9], %[pz], %[d]\n\t"
        "v_add_f32_dpp %[e], %[e], %[e] row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_pk_fma_f32 %[c1011], %[fxy], %[dd], %[c1011] op_sel_hi:[1,0,1]\n\t"
        "v_fmac_f32 %[c12], %[fz], %[d]\n\t"
        "v_add_f32_dpp %[e], %[e], %[e] row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32 %[gdT], %[gd], %[T]\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %[e], %[e], %[e] row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32 %[v2], %[alpha], %[gdT]\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %[e], %[e], %[e] row_shr:8 row_mask:0xf bank_mask:0xf"
        : [e] "+v"(e), [c78] "+v"(c78), [c9] "+v"(c9), [c1011] "+v"(c1011), [c12] "+v"(c12), [gdT] "=&v"(gdT), [v2] "+v"(v2)
        : [pxy] "v"(pxy), [pz] "v"(pz), [dd] "v"(dd), [d] "v"(d), [fxy] "v"(fxy), [fz] "v"(fz), [gd] "v"(gd), [T] "v"(T), [alpha] "v"(alpha));
    return e;
}

// The 13 sums of a batch leave the wave: every lane holds partial sums over ITS pixels -- add the four pixel-slot lanes of a Gaussian
// (lanes n, n+16, n+32, n+48), transpose through 1 KB of LDS, one atomic instruction per 4 Gaussians covering whole 64-byte rows.
template <bool NOLAST, typename LDS>
__device__ __forceinline__ void bwd_batch_finish(LDS &L, int nvalid, float *__restrict__ acc16, const float (&v)[13], uint32_t id, int n, int g)
{
    float *out = L.dump + 17 * n;                  // [16 Gaussians][16 slots], row stride 17: the 16 writers of a slot hit 16 banks
    {
        // every lane holds partial sums over ITS 16 pixels: add the four pixel-slot lanes of a Gaussian (lanes n, n+16, n+32, n+48).
        // Half / row exchanges reduce two values per instruction: v_permlane32_swap(a, b) leaves a.lo | b.lo and a.hi | b.hi, whose
        // sum holds a reduced over the halves in lanes 0-31 and b in lanes 32-63; v_permlane16_swap does the same for odd / even
        // rows.  Four sums (a, b, c, d) end up fully reduced in ONE register: row 0 a, row 1 c, row 2 b, row 3 d.
        // 12 exchanges + 12 additions instead of 26 ds_bpermute + 26 additions.
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float quad4[4];
#pragma unroll
            for (int q = 0; q < 4; q++) quad4[q] = (4 * j + q < 13) ? v[4 * j + q] : 0.f;
            float x, y;
            if (j < 3) {
                x = halves_sum(quad4[0], quad4[1]);
                y = halves_sum(quad4[2], quad4[3]);
            } else {
                x = halves_sum(quad4[0], quad4[0]);            // only v[12]: reduced in both halves ...
                y = x;
            }
            const float r = rows_sum(x, y);                     // ... and in all four rows
            const int q = 4 * j + ((g & 1) << 1) + (g >> 1);    // rows 0 1 2 3 hold sums a c b d
            if (q < 13) out[q] = r;
        }
    }
    // byte offset of this lane's Gaussian's accumulator row (all four pixel-slot lanes write the same value); slots 304..319 of the
    // dump area lie behind everything the steps and the transposition touch
    reinterpret_cast<uint32_t *>(L.dump)[304 + n] = 64u * id;
    wave_lds_sync();
    // round r: lane (n, g) adds slot n of Gaussian 4r + g -- one atomic instruction covers whole 64-byte rows of four Gaussians.  The
    // row offsets and the sums sit at immediate offsets of two per-lane LDS addresses; the global address is the (scalar) base of
    // the accumulator array + a 32-bit lane offset: 1 VALU instruction per round besides the two LDS reads and the atomic
    // (round 2 recomputed the ring slot, the LDS addresses and a 64-bit address per round: 12 VALU, one of them the ~24-cycle
    // VOP2 v_cndmask of the ring wrap)
    {
        const float *pv = L.dump + 17 * g + n;
        const uint32_t *po = reinterpret_cast<const uint32_t *>(L.dump) + 304 + g;
        const uint32_t n4 = 4u * (uint32_t)n;
        char *base = reinterpret_cast<char *>(acc16);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float val = pv[68 * r];
            const uint32_t off = po[4 * r] | n4;               // 64 id + 4 n: the row offset has its low six bits clear
            if (n < 13 && (NOLAST || 4 * r + g < nvalid)) unsafeAtomicAdd(reinterpret_cast<float *>(base + off), val);
        }
    }
    wave_lds_sync();
}

// Every lane accumulates the 13 Gaussian-centred partial sums of its 16 pixels in registers (15 VALU per step), the four pixel-slot
// lanes of a Gaussian are added in the epilogue.  STATS = true additionally counts into g_bwd_stats (developer variant 8).
// EXTRA = false: no pixel of the quadrant has an upstream depth or flow gradient (training on the image alone) -- the depth term of
// dL_dalpha and four of the 13 sums drop out (compile-time, so the common all-gradients path carries no extra branches)
// SEP = true: every pixel of the quadrant sits at its integer coordinates (no sub-pixel offsets), so dx / dy are not read per pixel
// NOLAST = true: the batch is full and all of its entries lie in front of every pixel's last contributor (wave-uniform, decided
// by the caller from the first = deepest entry): the per-pair test `list position < last contributor` is dropped
// GACC = true: some pixel of the quadrant has an upstream dL_dacc (a third scan per step and its carry)
// (Round 6, measured and dropped: batches without an entry of w > 0.99 running a variant without the min of alpha = min(0.99, w G) --
// one VALU instruction per step less, the same bits; same-box A/B 0.2433 vs 0.2420 ms: nothing.  The forward keeps its version of it.)
template <bool STATS, bool EXTRA, bool SEP, bool NOLAST, bool GACC>
__device__ __forceinline__ void bwd_batch(BwdLdsT<BWD_RING> &L, int head, int nvalid, float ox, float oy, float min_depth,
                                          float *__restrict__ acc16)
{
    constexpr bool use_gacc = GACC;
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    constexpr int RING = BWD_RING;
    const int slot = ring_wrap<RING>(head + n);
    const float4 g0 = L.ring[0][slot], g1 = L.ring[1][slot], g2 = L.ring[2][slot];
    const bool valid = n < nvalid;
    const float ap = g0.z, bp = g0.w, cp = g1.x, w = g1.y, dep = g1.z;
    // invalid lanes never pass `orig < last_contributor` (select spelled out: otherwise the compiler turns the compare below into
    // `valid && ...` and pays a v_cndmask + v_cmp per step to re-materialise the lane mask)
    const uint32_t orig = (uint32_t)select_i(LANES(valid), (int)__float_as_uint(g2.w), -1);
    const uint32_t tauq = tauq_bits_of(w);                  // the forward's value: same instruction, same bits
    const float flagf = dep > min_depth ? 1.f : 0.f;       // CR/backward.cu:603
    const float depflag = dep * flagf;
    // The 13 sums live in explicit register PAIRS fed by v_pk_fma_f32 (pair += pair * broadcast scalar).  Left to itself the compiler
    // packs scalar accumulators into pairs on its own and shuffles them in and out of the pairs with ~10 v_mov per step (round 3:
    // 53 VALU per step, 10 of them moves).  The factor w of sG = w G dL_dalpha is applied once per batch instead of once per step.
    f32x2 SYe = { 0.f, 0.f }, SYo = { 0.f, 0.f };        // MOMENTS: (sum s6, sum s6 dy) over the even / odd steps
    f32x2 M01 = { 0.f, 0.f }, M34 = { 0.f, 0.f };        // otherwise: (sum s6 dx, sum s6 dy), (sum s6 dx dx, sum s6 dx dy)
    f32x2 M56 = { 0.f, 0.f };                            // (sum s6 dy dy, sum s6 [+ G gacc])
    f32x2 C78 = { 0.f, 0.f }, C1011 = { 0.f, 0.f };      // (v7, v8), (v10, v11)
    float c9 = 0.f, c12 = 0.f, v2 = 0.f;
    f32x2 one_dy = { 1.f, 0.f }, dy2_one = { 0.f, 1.f }; // (1, dy), (dy dy, 1): the multiplicands of the moment pairs
    f32x2 dxy = { 0.f, 0.f };
    // carries of pixel 4s+g are written by lane n == 15 of row g; the other lanes write into the dump area (no exec masking)
    float2 *wTQ = (n == 15) ? reinterpret_cast<float2 *>(&L.pb[g].z) : reinterpret_cast<float2 *>(&L.dump[2 * (lane & 31)]);
    float *wG = (n == 15) ? (&L.pc[g].w) : (&L.dump[lane]);
    constexpr bool rd_pc = EXTRA || use_gacc;
    // the per-pixel constants / carries of step s + 1 are requested before step s runs (its LDS latency hides behind the step;
    // the carries of pixels 4(s+1)+g are last written one batch earlier, so the early read sees the right values)
    float4 pa_n = L.pa[g], pb_n = L.pb[g], pc_n = make_float4(0.f, 0.f, 0.f, 0.f), pd_n = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rd_pc) pc_n = L.pc[g];
    if (!SEP) pd_n = L.pd[g];
    unsigned long long st_run = 0, st_skip = 0, st_pairs = 0, st_any = 0, st_alive = 0, st_top = 0, st_bot = 0, st_run_top = 0;
    const float dxe = g0.x - (ox + (float)g), dxo = g0.x - (ox + (float)(4 + g));
    float dyr = 0.f, bdy = 0.f, cdydy = 0.f;
    constexpr bool MOMENTS = SEP;     // see the accumulation below
#pragma unroll
    for (int s = 0; s < 16; s++) {
        const float4 pa = pa_n, pb = pb_n, pc = pc_n, pd = pd_n;
        if (s < 15) {
            pa_n = L.pa[4 * s + 4 + g]; pb_n = L.pb[4 * s + 4 + g];
            if (rd_pc) pc_n = L.pc[4 * s + 4 + g];
            if (!SEP) pd_n = L.pd[4 * s + 4 + g];
        }
        // the forward kernel's arithmetic: identical alpha, identical decisions.  Pixel 4s+g sits in column 4(s&1)+g, row s>>1 of
        // the quadrant: without sub-pixel offsets (SEP) dx takes two values per batch, dy / b'dy / (c'dy)dy change every other step
        float dx, dy;
        if (SEP) {
            dx = (s & 1) ? dxo : dxe;
            if ((s & 1) == 0) { dyr = g0.y - (oy + (float)(s >> 1)); bdy = bp * dyr; cdydy = (cp * dyr) * dyr; one_dy.y = dyr; dy2_one.x = dyr * dyr; }
            dy = dyr;
        } else {
            dxy = (f32x2){ g0.x, g0.y } - (f32x2){ pd.x, pd.y };
            dx = dxy.x; dy = dxy.y;
            bdy = bp * dy; cdydy = (cp * dy) * dy;
        }
        const float q2 = q2_rows(dx, ap, bdy, cdydy);
        const lanemask ok = NOLAST ? LANES(__float_as_uint(q2) <= tauq)
                                   : (LANES(orig < __float_as_uint(pb.y)) & LANES(__float_as_uint(q2) <= tauq));
        if (STATS) {
            if (ok == 0) st_skip++; else { st_run++; if (s < 8) st_run_top++; }
            st_pairs += __popcll(ok); st_any |= ok;
            if (s < 8) st_top |= ok; else st_bot |= ok;
            st_alive += __popcll(LANES(orig < __float_as_uint(pb.y)));
        }
        if (ok == 0) continue;                 // nothing changes: T, E, gacc carries stay, the sums get zeros
        // G and alpha of the contributing lanes, exact zeros elsewhere (one select: w is finite, so w * 0 = 0)
        const float G_m = select_f(ok, __builtin_amdgcn_exp2f(-q2), 0.f);
        const float alpha_m = fminf(0.99f, w * G_m);
        const float inv = __builtin_amdgcn_rcpf(1.f - alpha_m);
        // T_i = T_carry * prod_{j <= i} inv_j   (the carry is row-uniform: every lane of the row read it from LDS)
        // (the colour dot product c . dL_dpixel is evaluated in the wait states of the product scan)
        float cgp;
        const float T = pb.z * row_scan_mul_with_dot(inv, cgp, g2.x, g2.y, g2.z, pa.x, pa.y, pa.z);
        const float dcc = alpha_m * T;                                  // dchannel_dcolor
        const float e = dcc * cgp;
        f32x2 dd;
        dd.x = dcc;                                                     // (only the low half is read: op_sel_hi broadcasts it)
        // the carry slot holds Q = bgT - E (E = sum of e over everything behind this batch): one subtraction gives bgT - E_inclusive;
        // the dcc-weighted sums (v7..v9, and v10..v12, gdT = dL_ddepth T, v2 with depth / flow gradients) fill the scan's wait states
        float gdT = 0.f, Ssum;
        if (EXTRA) Ssum = row_scan_add_with_sums_extra(e, C78, c9, (f32x2){ pa.x, pa.y }, pa.z, dd, dcc, C1011, c12, (f32x2){ pc.x, pc.y }, pc.z, gdT, pa.w, T, v2, alpha_m);
        else Ssum = row_scan_add_with_sums(e, C78, c9, (f32x2){ pa.x, pa.y }, pa.z, dd, dcc);
        const float Q = pb.w - Ssum;
        // dL_dalpha, CR/backward.cu:592-662:  ((final_depth - dep) gdepth T + (c - accum_rec) . dL_dpixel) T + bgT / (1 - alpha).
        // With e inv = (c . dL_dpixel) T (inv - 1) the colour and background terms collapse to inv ((c . dL_dpixel) T + bgT - E)
        // (E inclusive); the depth flag (0 or 1) is folded into per-Gaussian constants: (final_depth - dep) flag = final_depth flag - dep flag
        float dLa = (cgp * T + Q) * inv;
        if (EXTRA) dLa += ((pb.x * flagf - depflag) * gdT) * T;         // gdT: dL_ddepth T; the flag of the dL_dmean2D.z sum is applied per Gaussian
        const float s6 = G_m * dLa;                                     // dL_dG G / w
        const f32x2 s66 = { s6, s6 };
        if (use_gacc) {
            // dL_dacc *= T for every contributor (CR/backward.cu:650), then dL_dopacity += G (dL_dalpha + dL_dacc)
            const float ga = pc.w * row_scan_mul(select_f(ok, T, 1.f));
            M56.y = __builtin_fmaf(G_m, ga, M56.y);
            wG[16 * s] = ga;
        }
        if (MOMENTS) {
            // dx takes one value on the even steps and one on the odd steps of a batch: the sums over s6 dx, s6 dx^2, s6 dx dy
            // follow from S = sum s6 and Y = sum s6 dy kept separately for the two step parities
            if (s & 1) SYo = __builtin_elementwise_fma(one_dy, s66, SYo); else SYe = __builtin_elementwise_fma(one_dy, s66, SYe);
        } else {
            const float tdx = s6 * dx;
            dy2_one.x = dy * dy;
            M01 = __builtin_elementwise_fma(dxy, s66, M01);
            M34 = __builtin_elementwise_fma(dxy, (f32x2){ tdx, tdx }, M34);
        }
        M56 = __builtin_elementwise_fma(dy2_one, s66, M56);
        wTQ[8 * s] = make_float2(T, Q);
    }
    // The sums leave the wave like in the per-pixel kernel -- one atomic instruction covers whole 64-byte accumulator rows
    // (13 neighbouring floats per Gaussian, 4 Gaussians per instruction) -- after a 1 KB transposition through LDS; 13 separate
    // 16-lane atomics per batch would send 13x the requests to L2 (measured: 4.4x the kernel time).
    if (STATS && lane == 0) {
        const unsigned long long any16 = (st_any | (st_any >> 16) | (st_any >> 32) | (st_any >> 48)) & 0xffffull;
        atomicAdd(&g_bwd_stats[0], 1ull); atomicAdd(&g_bwd_stats[1], (unsigned long long)nvalid); atomicAdd(&g_bwd_stats[2], st_run);
        atomicAdd(&g_bwd_stats[3], st_skip); atomicAdd(&g_bwd_stats[4], st_pairs); atomicAdd(&g_bwd_stats[5], (unsigned long long)__popcll(any16));
        const unsigned long long top16 = (st_top | (st_top >> 16) | (st_top >> 32) | (st_top >> 48)) & 0xffffull;
        const unsigned long long bot16 = (st_bot | (st_bot >> 16) | (st_bot >> 32) | (st_bot >> 48)) & 0xffffull;
        atomicAdd(&g_bwd_stats[6], st_alive);
        atomicAdd(&g_bwd_stats[7], (unsigned long long)__popcll(top16 & ~bot16)); atomicAdd(&g_bwd_stats[8], (unsigned long long)__popcll(bot16 & ~top16));
        atomicAdd(&g_bwd_stats[9], (unsigned long long)__popcll(top16 & bot16));
        atomicAdd(&g_bwd_stats[10], st_run_top); atomicAdd(&g_bwd_stats[11], st_run - st_run_top);
        atomicAdd(&g_bwd_stats[12], top16 == 0 ? 1ull : 0ull); atomicAdd(&g_bwd_stats[13], bot16 == 0 ? 1ull : 0ull);
    }
    wave_lds_sync();
    {
        float v[13];
        if (MOMENTS) {
            const float Se = SYe.x, Ye = SYe.y, So = SYo.x, Yo = SYo.y;
            v[0] = dxe * Se + dxo * So;
            v[1] = Ye + Yo;
            v[3] = (dxe * dxe) * Se + (dxo * dxo) * So;
            v[4] = dxe * Ye + dxo * Yo;
        } else {
            v[0] = M01.x; v[1] = M01.y; v[3] = M34.x; v[4] = M34.y;
        }
        v[0] *= w; v[1] *= w; v[3] *= w; v[4] *= w;           // sG = w s6
        v[5] = w * M56.x;
        v[2] = v2; v[6] = M56.y; v[7] = C78.x; v[8] = C78.y; v[9] = c9; v[10] = C1011.x; v[11] = C1011.y; v[12] = c12;
        // accumulator layout 0 (what the per-pixel kernel writes): dL_dmean2D.xy as sums of sG (2 a' dx + b' dy), sG (2 c' dy + b' dx)
        // with the pre-scaled conic -- linear in the moments, so the conic is applied once per lane here
        {
            const float m0 = v[0], m1 = v[1];
            v[0] = (2.f * ap) * m0 + bp * m1;
            v[1] = (2.f * cp) * m1 + bp * m0;
            v[2] *= flagf;
        }
        bwd_batch_finish<NOLAST>(L, nvalid, acc16, v, __float_as_uint(g1.w), n, g);
    }
}

// Two inclusive scans over the 16 lanes of every DPP row, interleaved (round 6: one per pixel of a pair): between the dependent steps of one
// chain sit the other chain's instruction and one s_nop -- the two wait states a DPP read needs behind the VALU write of its source.
__device__ __forceinline__ void row_scan_mul2(float &a, float &b)
{
    asm("s_nop 1\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void row_scan_add2(float &a, float &b)
{
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0" : "+v"(a), "+v"(b));
}

// the two sum scans of a pixel pair into FRESH registers, ending in  q = c - inclusive scan(e):  the first step of each chain reads e in
// place (bound_ctrl: lanes without a DPP source add zero), so neither e's register pair nor the result needs a copy
__device__ __forceinline__ void row_scan_add2_from(float &qa, float &qb, float ea, float eb, float ca, float cb)
{
    asm("s_nop 1\n\t"                  /* (e may come straight out of the preceding VALU instruction: two wait states in front of a DPP read) */
        "v_add_f32_dpp %0, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_sub_f32 %0, %4, %0\n\t"
        "v_sub_f32 %1, %5, %1"
        : "=&v"(qa), "=&v"(qb) : "v"(ea), "v"(eb), "v"(ca), "v"(cb));
}

// ------------------------------------------------------------------------------------------------
// Round 6: the batch with TWO PIXELS PER LANE.  Lane (n, g) handles, in row-step r = 0..7, the pixels (row r, column g) and (row r, column
// 4 + g) of the quadrant -- what bwd_batch does in steps 2 r and 2 r + 1 -- with every per-pixel quantity as a register pair (a, b): the
// exponent, alpha, 1 - alpha, the colour dot product, T, the weights, the sums all run on v_pk_{fma,mul,add}_f32 (28 packed + 28 plain
// VALU instructions per row-step against 83 for the two steps; a packed instruction costs 1.75 plain ones on this part, the transcendentals,
// compares, selects and the 16 DPP scan steps do not pack).  The row terms of the exponent (dy, b'dy, c'dy^2) are shared by the pair, the
// two pixels' scans interleave (each fills the other's DPP wait states), one skip test covers both pixels.  Quadrants without sub-pixel
// offsets and without an upstream dL_dacc (the training loop's); same decisions as the forward (the same q2 / tauq bits), sums
// reassociated (column a and column b accumulate apart: they are the even / odd step parities of bwd_batch).
template <bool EXTRA, bool NOLAST>
__device__ __forceinline__ void bwd_batch_pairs(BwdLdsT<BWD_RING> &L, int head, int nvalid, float ox, float oy, float min_depth, float *__restrict__ acc16)
{
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    constexpr int RING = BWD_RING;
    const int slot = ring_wrap<RING>(head + n);
    const float4 g0 = L.ring[0][slot], g1 = L.ring[1][slot], g2 = L.ring[2][slot];
    const bool valid = n < nvalid;
    const float ap = g0.z, bp = g0.w, cp = g1.x, w = g1.y, dep = g1.z;
    const uint32_t orig = (uint32_t)select_i(LANES(valid), (int)__float_as_uint(g2.w), -1);
    const uint32_t tauq = tauq_bits_of(w);
    const float flagf = dep > min_depth ? 1.f : 0.f;
    const float depflag = dep * flagf;
    const float dxe = g0.x - (ox + (float)g), dxo = g0.x - (ox + (float)(4 + g));
    const f32x2 dx2 = { dxe, dxo };
    f32x2 S2 = { 0.f, 0.f }, Y2 = { 0.f, 0.f }, M5 = { 0.f, 0.f };                 // (sum s6, sum s6 dy, sum s6 dy dy) per column
    f32x2 c7 = { 0.f, 0.f }, c8 = { 0.f, 0.f }, c9 = { 0.f, 0.f }, c10 = { 0.f, 0.f }, c11 = { 0.f, 0.f }, c12 = { 0.f, 0.f }, v2 = { 0.f, 0.f };
    // carries of pixel pair 4 r + g are written by lane n == 15 of row g; the other lanes write into the dump area (no exec masking)
    float4 *wC = (n == 15) ? &L.pp.C[g] : reinterpret_cast<float4 *>(&L.dump[4 * n]);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        // (the pair's constants are requested at the top of the step and first needed behind the exponent: their LDS latency hides there;
        // requesting them a step ahead like bwd_batch does costs 22 registers the 128-register budget does not have)
        const float4 A = L.pp.A[4 * r + g], A2 = L.pp.A2[4 * r + g], B = L.pp.B[4 * r + g], C = L.pp.C[4 * r + g];
        float4 F = make_float4(0.f, 0.f, 0.f, 0.f);
        float2 F2 = make_float2(0.f, 0.f);
        if (EXTRA) { F = L.pp.F[4 * r + g]; F2 = L.pp.F2[4 * r + g]; }
        const float dyr = g0.y - (oy + (float)r);
        const float bdy = bp * dyr, cdydy = (cp * dyr) * dyr, dy2 = dyr * dyr;
        // the forward kernel's arithmetic per pixel: q2_rows(dx, ap, bdy, cdydy) = fma(dx, fma(dx, -ap, -bdy), -cdydy), identical bits
        const f32x2 inner = __builtin_elementwise_fma(dx2, (f32x2){ -ap, -ap }, (f32x2){ -bdy, -bdy });
        const f32x2 q2 = __builtin_elementwise_fma(dx2, inner, (f32x2){ -cdydy, -cdydy });
        const lanemask ok_a = NOLAST ? LANES(__float_as_uint(q2.x) <= tauq) : (LANES(orig < __float_as_uint(B.z)) & LANES(__float_as_uint(q2.x) <= tauq));
        const lanemask ok_b = NOLAST ? LANES(__float_as_uint(q2.y) <= tauq) : (LANES(orig < __float_as_uint(B.w)) & LANES(__float_as_uint(q2.y) <= tauq));
        if ((ok_a | ok_b) == 0) continue;          // nothing changes: the carries stay, the sums get zeros
        f32x2 G;
        G.x = select_f(ok_a, __builtin_amdgcn_exp2f(-q2.x), 0.f);
        G.y = select_f(ok_b, __builtin_amdgcn_exp2f(-q2.y), 0.f);
        f32x2 alpha = (f32x2){ w, w } * G;
        alpha.x = fminf(0.99f, alpha.x); alpha.y = fminf(0.99f, alpha.y);
        const f32x2 om = (f32x2){ 1.f, 1.f } - alpha;
        f32x2 inv;
        inv.x = __builtin_amdgcn_rcpf(om.x); inv.y = __builtin_amdgcn_rcpf(om.y);
        // c . dL_dpixel of both pixels
        f32x2 cgp = (f32x2){ g2.x, g2.x } * (f32x2){ A.x, A.y };
        cgp = __builtin_elementwise_fma((f32x2){ g2.y, g2.y }, (f32x2){ A.z, A.w }, cgp);
        cgp = __builtin_elementwise_fma((f32x2){ g2.z, g2.z }, (f32x2){ A2.x, A2.y }, cgp);
        // T = T_carry * prod_{j <= i} inv_j, both pixels
        float sa = inv.x, sb = inv.y;
        row_scan_mul2(sa, sb);
        const f32x2 T = (f32x2){ C.x, C.y } * (f32x2){ sa, sb };
        const f32x2 dcc = alpha * T;
        const f32x2 e = dcc * cgp;
        // (the scans write fresh registers -- their first step reads the pair's halves in place, lanes without a DPP source add zero --
        // and end in the subtraction from the carry: no copy into or out of a register pair around the sixteen DPP steps)
        float qa, qb;
        row_scan_add2_from(qa, qb, e.x, e.y, C.z, C.w);
        c7 = __builtin_elementwise_fma((f32x2){ A.x, A.y }, dcc, c7);
        c8 = __builtin_elementwise_fma((f32x2){ A.z, A.w }, dcc, c8);
        c9 = __builtin_elementwise_fma((f32x2){ A2.x, A2.y }, dcc, c9);
        const f32x2 Q = { qa, qb };                                     // (bgT - E) carry - inclusive scan of e
        f32x2 dLa = __builtin_elementwise_fma(cgp, T, Q) * inv;
        if (EXTRA) {
            c10 = __builtin_elementwise_fma((f32x2){ F.x, F.y }, dcc, c10);
            c11 = __builtin_elementwise_fma((f32x2){ F.z, F.w }, dcc, c11);
            c12 = __builtin_elementwise_fma((f32x2){ F2.x, F2.y }, dcc, c12);
            const f32x2 gdT = (f32x2){ A2.z, A2.w } * T;
            v2 = __builtin_elementwise_fma(alpha, gdT, v2);
            const f32x2 t = __builtin_elementwise_fma((f32x2){ B.x, B.y }, (f32x2){ flagf, flagf }, (f32x2){ -depflag, -depflag }) * gdT;
            dLa = __builtin_elementwise_fma(t, T, dLa);
        }
        const f32x2 s6 = G * dLa;
        S2 = S2 + s6;
        Y2 = __builtin_elementwise_fma((f32x2){ dyr, dyr }, s6, Y2);
        M5 = __builtin_elementwise_fma((f32x2){ dy2, dy2 }, s6, M5);
        wC[4 * r] = make_float4(T.x, T.y, Q.x, Q.y);
    }
    wave_lds_sync();
    {
        float v[13];
        const float Se = S2.x, So = S2.y, Ye = Y2.x, Yo = Y2.y;
        v[0] = dxe * Se + dxo * So;
        v[1] = Ye + Yo;
        v[3] = (dxe * dxe) * Se + (dxo * dxo) * So;
        v[4] = dxe * Ye + dxo * Yo;
        v[0] *= w; v[1] *= w; v[3] *= w; v[4] *= w;           // sG = w s6
        v[5] = w * (M5.x + M5.y);
        v[2] = v2.x + v2.y; v[6] = Se + So; v[7] = c7.x + c7.y; v[8] = c8.x + c8.y; v[9] = c9.x + c9.y;
        v[10] = c10.x + c10.y; v[11] = c11.x + c11.y; v[12] = c12.x + c12.y;
        {
            const float m0 = v[0], m1 = v[1];
            v[0] = (2.f * ap) * m0 + bp * m1;
            v[1] = (2.f * cp) * m1 + bp * m0;
            v[2] *= flagf;
        }
        bwd_batch_finish<NOLAST>(L, nvalid, acc16, v, __float_as_uint(g1.w), n, g);
    }
}

template <int WPB, bool STATS>
__global__ __launch_bounds__(64 * WPB, 4) void composite_bwd_scan_kernel(
    int W, int H, int gx, int num_tiles,
    const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list,
    const float *__restrict__ subpixel_offset, const float *__restrict__ bg,
    const float4 *__restrict__ records,
    const float *__restrict__ depth_acc, const float *__restrict__ weight_acc, float min_depth,
    const float *__restrict__ final_Ts, const uint32_t *__restrict__ n_contrib,
    const float *__restrict__ dL_dpixels, const float *__restrict__ dL_ddepths,
    const float *__restrict__ dL_dflows, const float *__restrict__ dL_daccs,
    float *__restrict__ acc16, const uint32_t *__restrict__ qlist, const uint32_t *__restrict__ qcount, int pairs_on)
{
    constexpr int RING = BWD_RING;
    __shared__ BwdLdsT<RING> lds[WPB];
    int tile, quad;
    tile_of_block(num_tiles, tile, quad);
    if (tile >= num_tiles) return;
    const int wave = (WPB == 4) ? (threadIdx.x >> 6) : 0, lane = threadIdx.x & 63;
    BwdLdsT<RING> &L = lds[wave];
    const PixelGeom p = pixel_of_lane(tile, quad, gx, W, H, subpixel_offset);      // setup: lane = pixel (x = lane & 7, y = lane >> 3)
    const uint2 range = ranges[tile];
    const size_t HW = (size_t)H * W;
    const uint32_t last_contributor = p.inside ? n_contrib[p.pix_id] : 0u;
    const uint32_t deepest = wave_max_u32(last_contributor);     // nothing behind it touches this quadrant
    if (deepest == 0) return;
    const uint32_t min_last = wave_min_u32(last_contributor);    // entries in front of it are in front of every pixel's last contributor

    // ---- per-pixel constants, CR/backward.cu:489-549
    const float T_final = p.inside ? final_Ts[p.pix_id] : 0.f;
    const float acc = p.inside ? weight_acc[p.pix_id] : 0.f;
    const float final_depth = p.inside ? depth_acc[p.pix_id] : 0.f;
    float gdepth = 0.f, gflow0 = 0.f, gflow1 = 0.f, gflow2 = 0.f, gacc = 0.f, gp0 = 0.f, gp1 = 0.f, gp2 = 0.f;
    if (p.inside) {
        if (dL_ddepths) gdepth = dL_ddepths[p.pix_id];
        if (dL_dpixels) { gp0 = dL_dpixels[p.pix_id]; gp1 = dL_dpixels[HW + p.pix_id]; gp2 = dL_dpixels[2 * HW + p.pix_id]; }
        if (acc > 0.0f) {
            gdepth /= acc;
            if (dL_dflows) { gflow0 = dL_dflows[p.pix_id] / acc; gflow1 = dL_dflows[HW + p.pix_id] / acc; gflow2 = dL_dflows[2 * HW + p.pix_id] / acc; }
            if (dL_daccs) gacc = dL_daccs[p.pix_id];
        }
    }
    const float bgT = -T_final * (bg[0] * gp0 + bg[1] * gp1 + bg[2] * gp2);
    const float ox = (float)((tile % gx) * EX4D_TILE + (quad & 1) * 8), oy = (float)((tile / gx) * EX4D_TILE + (quad >> 1) * 8);
    // wave-uniform: which optional upstream gradients take part in this quadrant at all (training on the image alone has none of them)
    const bool use_gacc = LANES(gacc != 0.0f) != 0;
    const bool sep = LANES(p.fx != ox + (float)(lane & 7) || p.fy != oy + (float)(lane >> 3)) == 0;    // no sub-pixel offsets in this quadrant
    const bool use_extra = LANES(gdepth != 0.0f || gflow0 != 0.0f || gflow1 != 0.0f || gflow2 != 0.0f) != 0;
    // round 6: two pixels per lane (bwd_batch_pairs) where the quadrant has integer pixel positions and no upstream dL_dacc
    const bool pairs = !STATS && pairs_on && sep && !use_gacc;
    if (pairs) {
        // lane = pixel (x = lane & 7, y = lane >> 3): pair 4 y + (x & 3), half x >> 2
        const int pi = 4 * (lane >> 3) + (lane & 3), hf = (lane >> 2) & 1;
        reinterpret_cast<float *>(L.pp.A)[4 * pi + hf] = gp0;  reinterpret_cast<float *>(L.pp.A)[4 * pi + 2 + hf] = gp1;
        reinterpret_cast<float *>(L.pp.A2)[4 * pi + hf] = gp2; reinterpret_cast<float *>(L.pp.A2)[4 * pi + 2 + hf] = gdepth;
        reinterpret_cast<float *>(L.pp.B)[4 * pi + hf] = final_depth; reinterpret_cast<float *>(L.pp.B)[4 * pi + 2 + hf] = __uint_as_float(last_contributor);
        reinterpret_cast<float *>(L.pp.C)[4 * pi + hf] = T_final; reinterpret_cast<float *>(L.pp.C)[4 * pi + 2 + hf] = bgT;     // (bgT - E, E = 0 behind the deepest contributor)
        reinterpret_cast<float *>(L.pp.F)[4 * pi + hf] = gflow0; reinterpret_cast<float *>(L.pp.F)[4 * pi + 2 + hf] = gflow1;
        reinterpret_cast<float *>(L.pp.F2)[2 * pi + hf] = gflow2;
    } else {
        L.pa[lane] = make_float4(gp0, gp1, gp2, gdepth);
        L.pb[lane] = make_float4(final_depth, __uint_as_float(last_contributor), T_final, bgT);     // w: bgT - E, E = 0 behind the deepest contributor
        L.pc[lane] = make_float4(gflow0, gflow1, gflow2, gacc);
        L.pd[lane] = make_float4(p.fx, p.fy, 0.f, 0.f);
    }

    // the ring holds finite values from the start (stale entries are read by invalid lanes)
    for (int i = lane; i < RING; i += 64) {
        L.ring[0][i] = make_float4(0.f, 0.f, 0.f, 0.f); L.ring[1][i] = make_float4(0.f, 0.f, 0.f, 0.f); L.ring[2][i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    wave_lds_sync();

    int head = 0, tail = 0, count = 0;                            // wave-uniform ring indices in [0, RING) and the number of staged entries
    {
        // The forward kernel left this quadrant's compacted list -- (Gaussian id, list position) of every entry it composited, in list
        // order.  Chunks of 64 from the back; inside a chunk lane l takes entry 64 c + 63 - l, so the ring fills in descending list
        // order.  Entries at or behind the quadrant's deepest contributor touch no pixel and are dropped (they form the tail of the list).
        const uint32_t qn = (uint32_t)__builtin_amdgcn_readfirstlane((int)qcount[4 * tile + quad]);
        const uint32_t list_len = range.y - range.x;
        const uint32_t *ql = qlist + 4 * (size_t)range.x + (size_t)quad * (size_t)list_len;
        const uint32_t *pl = point_list + range.x;
        const int nchunks = (int)((qn + 63u) >> 6);
        const int b = 63 - lane;
        const uint64_t lt = (1ull << lane) - 1ull;
        // Round 6: the list holds positions only (16 instead of 32 bytes of capacity per instance); the Gaussian id is point_list[position].
        // Two stages run ahead of the chunk being processed -- the positions of chunk c - 2 and the ids of chunk c - 1 are requested while
        // chunk c is processed -- so a chunk still waits for ONE memory round trip (its records).
        uint32_t pos_n = 0xFFFFFFFFu, pos_n2 = 0xFFFFFFFFu, id_n = 0u;       // (entries past the end carry position 0xFFFFFFFF)
        if (nchunks > 0) { const uint32_t e = 64u * (uint32_t)(nchunks - 1) + (uint32_t)b; if (e < qn) pos_n = ql[e]; }
        if (nchunks > 1) pos_n2 = ql[64u * (uint32_t)(nchunks - 2) + (uint32_t)b];          // chunk c - 1 lies wholly inside the list
        if (pos_n != 0xFFFFFFFFu) id_n = pl[pos_n];
        for (int c = nchunks - 1; c >= 0; c--) {
            const uint2 it = make_uint2(id_n, pos_n);
            pos_n = pos_n2;
            if (c > 0) id_n = pl[pos_n];
            if (c > 1) pos_n2 = ql[64u * (uint32_t)(c - 2) + (uint32_t)b];
            const bool valid = it.y < deepest;                    // (entries past the end carry position 0xFFFFFFFF)
            const uint64_t mask = __ballot(valid);
            if (valid) {
                const float4 *r = records + 4 * (size_t)it.x;
                const float4 q0 = r[0], q2 = r[2];
                const int slot = ring_wrap<RING>(tail + __popcll(mask & lt));
                L.ring[0][slot] = make_float4(q0.x, q0.y, q0.z * kHalfLog2e, q0.w * kNegLog2e);
                L.ring[1][slot] = make_float4(r[1].x * kHalfLog2e, r[3].w, q2.x, __uint_as_float(it.x));
                L.ring[2][slot] = make_float4(q2.y, q2.z, q2.w, __uint_as_float(it.y));
            }
            tail = ring_wrap<RING>(tail + __popcll(mask));
            count += __popcll(mask);
            wave_lds_sync();
            while (count >= 16 || (c == 0 && count > 0)) {
                const int nb = count < 16 ? count : 16;
                // first entry of the batch = its deepest: wave-uniform read of its list position
                const uint32_t kfirst = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(L.ring[2][head].w));
                const bool nolast = nb == 16 && kfirst < min_last;
#define BATCH(E, S, N, G) bwd_batch<STATS, E, S, N, G>(L, head, nb, ox, oy, min_depth, acc16)
                // (quadrants with sub-pixel offsets take the general variant: all scans, all sums)
                if (pairs) {
                    if (use_extra) { if (nolast) bwd_batch_pairs<true, true>(L, head, nb, ox, oy, min_depth, acc16); else bwd_batch_pairs<true, false>(L, head, nb, ox, oy, min_depth, acc16); }
                    else { if (nolast) bwd_batch_pairs<false, true>(L, head, nb, ox, oy, min_depth, acc16); else bwd_batch_pairs<false, false>(L, head, nb, ox, oy, min_depth, acc16); }
                }
                else if (!sep) BATCH(true, false, false, true);
                else if (use_extra) {
                    if (use_gacc) { if (nolast) BATCH(true, true, true, true); else BATCH(true, true, false, true); }
                    else { if (nolast) BATCH(true, true, true, false); else BATCH(true, true, false, false); }
                } else if (use_gacc) { if (nolast) BATCH(false, true, true, true); else BATCH(false, true, false, true); }
                else { if (nolast) BATCH(false, true, true, false); else BATCH(false, true, false, false); }
#undef BATCH
                head = ring_wrap<RING>(head + nb);
                count -= nb;
                wave_lds_sync();
            }
        }
    }
}

}  // namespace

void ex4d_set_fwd_asm(int on) { g_fwd_asm.store(on); }
void ex4d_set_clamp_always(int on) { g_alpha_clamp_always.store(on); }
void ex4d_set_bwd_pairs(int on) { g_bwd_pairs.store(on); }
int ex4d_get_bwd_pairs() { return g_bwd_pairs.load(); }
int ex4d_get_clamp_always() { return g_alpha_clamp_always.load(); }
int ex4d_get_fwd_asm() { return g_fwd_asm.load(); }

hipError_t ex4d_bwd_stats(unsigned long long *out, int count, int reset)
{
    if (count > 16) count = 16;
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bwd_stats), count * sizeof(unsigned long long));
    if (e == hipSuccess && reset) { unsigned long long z[16] = { 0 }; e = hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_stats), z, sizeof(z)); }
    return e;
}

hipError_t ex4d_launch_composite_fwd(const Ex4dParams &prm, const uint2 *ranges, const uint32_t *point_list,
    const float *subpixel_offset, const float4 *records, const float *bg, float *final_T, uint32_t *n_contrib,
    float *out_color, float *out_depth, float *out_acc, float *out_flow, int32_t *out_idx, uint32_t *qlist, uint32_t *qcount,
    bool has_flow, hipStream_t stream)
{
    const int gx = (prm.W + EX4D_TILE - 1) / EX4D_TILE, gy = (prm.H + EX4D_TILE - 1) / EX4D_TILE;
    const int T = gx * gy;
    const int slots = 8 * ((T + 7) / 8);
#define FWD_LAUNCH(FLOW) hipLaunchKernelGGL((composite_fwd_kernel<FLOW, ASM>), dim3(4 * slots), dim3(64), 0, stream, \
        prm.W, prm.H, gx, T, ranges, point_list, subpixel_offset, records, bg, \
        prm.max_depth, final_T, n_contrib, out_color, out_depth, out_acc, out_flow, out_idx, qlist, qcount, g_alpha_clamp_always.load(std::memory_order_relaxed))
    if (has_flow) { constexpr bool ASM = false; FWD_LAUNCH(true); }
    else if (g_fwd_asm.load(std::memory_order_relaxed)) { constexpr bool ASM = true; FWD_LAUNCH(false); }
    else { constexpr bool ASM = false; FWD_LAUNCH(false); }
#undef FWD_LAUNCH
    return hipGetLastError();
}

// variant: 4 = (Gaussian, pixel-slot) lanes with register accumulation, streaming the forward kernel's per-quadrant compacted lists (default);
//          8 = 4 + developer statistics (g_bwd_stats)
hipError_t ex4d_launch_composite_bwd(const Ex4dParams &prm, const uint2 *ranges, const uint32_t *point_list,
    const float *subpixel_offset, const float *bg, const float4 *records, const float *out_depth, const float *out_acc,
    const float *final_T, const uint32_t *n_contrib, const float *dL_dpix, const float *dL_ddepth,
    const float *dL_dflow, const float *dL_dacc, float *acc16, const uint32_t *qlist, const uint32_t *qcount,
    int variant, hipStream_t stream)
{
    const int gx = (prm.W + EX4D_TILE - 1) / EX4D_TILE, gy = (prm.H + EX4D_TILE - 1) / EX4D_TILE;
    const int T = gx * gy;
    const int Tpad = 8 * ((T + 7) / 8);
#define BWD_ARGS prm.W, prm.H, gx, T, ranges, point_list, subpixel_offset, bg, records, out_depth, out_acc, prm.min_depth, \
                 final_T, n_contrib, dL_dpix, dL_ddepth, dL_dflow, dL_dacc, acc16
    if (variant == 8) hipLaunchKernelGGL((composite_bwd_scan_kernel<1, true>), dim3(4 * Tpad), dim3(64), 0, stream, BWD_ARGS, qlist, qcount, 0);
    else hipLaunchKernelGGL((composite_bwd_scan_kernel<1, false>), dim3(4 * Tpad), dim3(64), 0, stream, BWD_ARGS, qlist, qcount, g_bwd_pairs.load(std::memory_order_relaxed));
#undef BWD_ARGS
    return hipGetLastError();
}
