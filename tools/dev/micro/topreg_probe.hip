// Developer probe (part 3) -- stand-alone reproducer of a gfx950 operand fault.
// tools/dev/spill_asm_variants.py traced the wrong gradients of the 128-register preprocess_bwd build to ONE instruction,
// `v_lshrrev_b64 v[22:23], v127, s[4:5]`: a 64-bit shift whose 32-bit shift amount sits in the LAST vector register of the wave's
// allocation.  This program shows it without the rest of that kernel (256 threads, __launch_bounds__(256, 4) = 128 registers, 26 KB of
// LDS, with and without scratch):
//   probe       the window of the original kernel (a 16-byte load in flight, v126 / v127 written, read by compares, a multiply-add,
//               plain moves AND the 64-bit shift): only the shift is ever wrong -- it shifts by `threadIdx.x & 63`, i.e. by VGPR0 (what
//               the ISA prescribes for an out-of-range source) -- never in blocks < 256 (the first wave of a SIMD), never with the same
//               amount in v125, never when the kernel allocates 136 registers;
//   ops_probe   sixteen instructions with a 32-bit operand in v127 against the same operand in v125: v_lshlrev_b64, v_lshrrev_b64 and
//               v_ashrrev_i64 differ, v_mad_u64_u32 / v_mad_i64_i32 / v_ldexp_f64 / v_cvt_f64_* / v_trig_preop_f64 / 32-bit ops never;
//   base_probe  per wave: HW_REG_GPR_ALLOC.VGPR_BASE and whether any of its shifts was wrong -- with four waves per SIMD every wave at
//               base 48 (registers 384..511, the top of the file) is wrong, the others sometimes.
// Output of the round-4 run: profiles/r04_topreg_probe.txt.  ex4dgs_amd/isa_check.py refuses objects that hold such an instruction.
//      hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage topreg_probe.hip -o topreg_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <bool SCRATCH, int NREGS>
__global__ __launch_bounds__(256, 4) void probe(const float4 *__restrict__ src, unsigned *__restrict__ stat, unsigned *__restrict__ samples,
                                                int iters, int nrows, unsigned long long need)
{
    __shared__ float4 lds[1664];                        // 26 KB per workgroup, like the kernel under test
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    volatile unsigned priv[4];
    if (SCRATCH) { for (int i = 0; i < 4; i++) priv[i] = threadIdx.x * 4u + (unsigned)i; }
    const float4 *base = src + (size_t)((blockIdx.x * 4 + wave) & 1023) * 768;      // a wave's block: 64 rows x 12 chunks of 16 bytes
    unsigned wrong_q = 0, wrong_g = 0, wrong_pred = 0, wrong_sh = 0, wrong_sh2 = 0;
    float4 keep = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int it = 0; it < iters; it++) {
        const float4 t0 = base[384 + lane];              // chunk 0 of the second half: in flight while the top registers are used
        unsigned q, g, pred;
        unsigned long long sh, sh2;
        if (NREGS == 128) {
            asm volatile(
                "v_or_b32_e32 v126, 0x1c0, %[lane]\n\t"
                "v_mul_u32_u24_e32 %[q], 0x1556, v126\n\t"
                "v_lshrrev_b32_e32 v127, 16, %[q]\n\t"
                "v_cmp_gt_i32_e32 vcc, %[nrows], v127\n\t"
                "v_cndmask_b32_e64 %[pred], 0, 1, vcc\n\t"
                "v_mad_i32_i24 %[q], v127, -12, v126\n\t"
                "v_cmp_gt_i32_e32 vcc, 12, %[q]\n\t"
                "v_cndmask_b32_e64 %[g], 0, 2, vcc\n\t"
                "v_or_b32_e32 %[pred], %[pred], %[g]\n\t"
                "v_lshrrev_b64 %[sh], v127, %[need]\n\t"
                "v_mov_b32_e32 v125, v127\n\t"
                "v_lshrrev_b64 %[sh2], v125, %[need]\n\t"
                "v_mov_b32_e32 %[g], v127\n\t"
                "v_mov_b32_e32 %[q], v126\n\t"
                : [q] "=&v"(q), [g] "=&v"(g), [pred] "=&v"(pred), [sh] "=&v"(sh), [sh2] "=&v"(sh2)
                : [lane] "v"(lane), [nrows] "v"(nrows), [need] "s"(need) : "vcc", "v125", "v126", "v127");
        } else {
            asm volatile(
                "v_or_b32_e32 v126, 0x1c0, %[lane]\n\t"
                "v_mul_u32_u24_e32 %[q], 0x1556, v126\n\t"
                "v_lshrrev_b32_e32 v127, 16, %[q]\n\t"
                "v_cmp_gt_i32_e32 vcc, %[nrows], v127\n\t"
                "v_cndmask_b32_e64 %[pred], 0, 1, vcc\n\t"
                "v_mad_i32_i24 %[q], v127, -12, v126\n\t"
                "v_cmp_gt_i32_e32 vcc, 12, %[q]\n\t"
                "v_cndmask_b32_e64 %[g], 0, 2, vcc\n\t"
                "v_or_b32_e32 %[pred], %[pred], %[g]\n\t"
                "v_lshrrev_b64 %[sh], v127, %[need]\n\t"
                "v_mov_b32_e32 v125, v127\n\t"
                "v_lshrrev_b64 %[sh2], v125, %[need]\n\t"
                "v_mov_b32_e32 %[g], v127\n\t"
                "v_mov_b32_e32 %[q], v126\n\t"
                : [q] "=&v"(q), [g] "=&v"(g), [pred] "=&v"(pred), [sh] "=&v"(sh), [sh2] "=&v"(sh2)
                : [lane] "v"(lane), [nrows] "v"(nrows), [need] "s"(need) : "vcc", "v125", "v126", "v127", "v135");
        }
        const unsigned q_ok = 0x1c0u | (unsigned)lane, g_ok = (q_ok * 0x1556u) >> 16;
        const bool bq = q != q_ok, bg = g != g_ok, bp = pred != 3u, bs = sh != (need >> g_ok), bs2 = sh2 != (need >> g_ok);
        wrong_q += bq; wrong_g += bg; wrong_pred += bp; wrong_sh += bs; wrong_sh2 += bs2;
        if (bq || bg || bp || bs || bs2) {
            const unsigned slot = atomicAdd(&stat[7], 1u);
            if (slot < 64) { samples[4 * slot] = blockIdx.x; samples[4 * slot + 1] = threadIdx.x | ((unsigned)it << 16); samples[4 * slot + 2] = (unsigned)sh; samples[4 * slot + 3] = g | (pred << 24); }
        }
        keep.x += t0.x; keep.y += t0.y; keep.z += t0.z; keep.w += t0.w;
        if (SCRATCH) priv[it & 3] += 1u;                // scratch traffic inside the loop, like the spill reloads
    }
    lds[threadIdx.x] = keep;
    __syncthreads();
    const float4 o = lds[(threadIdx.x + 64) & 255];
    unsigned fold = (o.x + o.y + o.z + o.w == 12345.678f) ? 1u : 0u;
    if (SCRATCH) fold += (priv[0] + priv[1] + priv[2] + priv[3] == 0xffffffffu) ? 1u : 0u;
    if (wrong_q | wrong_g | wrong_pred | wrong_sh | wrong_sh2 | fold) {
        atomicAdd(&stat[0], wrong_q); atomicAdd(&stat[1], wrong_g); atomicAdd(&stat[2], wrong_pred); atomicAdd(&stat[3], fold);
        atomicAdd(&stat[8], wrong_sh); atomicAdd(&stat[9], wrong_sh2);
        atomicAdd(&stat[blockIdx.x < 256 ? 4 : 5], wrong_q + wrong_g + wrong_pred + wrong_sh + wrong_sh2);
    }
}


// Which instructions are affected?  Each one is issued twice on the same operand values, once with its 32-bit operand in the wave's LAST
// register (v127 of a 128-register kernel) and once with it in v125; stat[16 + i] counts the lanes whose two results differ.
#define OPS 16
#define PAIR_OP(i, text) { unsigned long long d1, d2; \
    asm volatile("v_mov_b32_e32 v127, %[x]\n\tv_mov_b32_e32 v125, %[x]\n\ts_nop 1\n\t" text \
                 : [d1] "=&v"(d1), [d2] "=&v"(d2) : [x] "v"(x), [p] "v"(pairv), [n] "s"(need), [y] "v"(y) : "vcc", "v125", "v127"); \
    wrong[i] += d1 != d2; }
#define WORD_OP(i, text) { unsigned d1, d2; \
    asm volatile("v_mov_b32_e32 v127, %[x]\n\tv_mov_b32_e32 v125, %[x]\n\ts_nop 1\n\t" text \
                 : [d1] "=&v"(d1), [d2] "=&v"(d2) : [x] "v"(x), [p] "v"(pairv), [n] "s"(need), [y] "v"(y) : "vcc", "v125", "v127"); \
    wrong[i] += d1 != d2; }
__global__ __launch_bounds__(256, 4) void ops_probe(unsigned *__restrict__ stat, int iters, unsigned long long need)
{
    __shared__ float4 lds[1664];
    unsigned wrong[OPS] = { 0 };
    const unsigned lane = threadIdx.x & 63;
    for (int it = 0; it < iters; it++) {
        const unsigned x = 37u + ((lane + (unsigned)it) % 5u);                   // never equal to the thread index (what VGPR0 holds)
        const unsigned y = 0x9e3779b9u * (threadIdx.x + 1u) + (unsigned)it;
        const unsigned long long pairv = 0x3ff0000000000000ull + ((unsigned long long)y << 20);     // a double near 1
        PAIR_OP(0, "v_lshrrev_b64 %[d1], v127, %[n]\n\tv_lshrrev_b64 %[d2], v125, %[n]")
        PAIR_OP(1, "v_lshlrev_b64 %[d1], v127, %[n]\n\tv_lshlrev_b64 %[d2], v125, %[n]")
        PAIR_OP(2, "v_ashrrev_i64 %[d1], v127, %[n]\n\tv_ashrrev_i64 %[d2], v125, %[n]")
        PAIR_OP(3, "v_lshrrev_b64 %[d1], v127, %[p]\n\tv_lshrrev_b64 %[d2], v125, %[p]")
        PAIR_OP(4, "v_mad_u64_u32 %[d1], vcc, v127, %[y], %[p]\n\tv_mad_u64_u32 %[d2], vcc, v125, %[y], %[p]")
        PAIR_OP(5, "v_mad_u64_u32 %[d1], vcc, %[y], v127, %[p]\n\tv_mad_u64_u32 %[d2], vcc, %[y], v125, %[p]")
        PAIR_OP(6, "v_ldexp_f64 %[d1], %[p], v127\n\tv_ldexp_f64 %[d2], %[p], v125")
        PAIR_OP(7, "v_cvt_f64_u32_e32 %[d1], v127\n\tv_cvt_f64_u32_e32 %[d2], v125")
        PAIR_OP(8, "v_cvt_f64_i32_e32 %[d1], v127\n\tv_cvt_f64_i32_e32 %[d2], v125")
        PAIR_OP(9, "v_cvt_f64_f32_e32 %[d1], v127\n\tv_cvt_f64_f32_e32 %[d2], v125")
        PAIR_OP(10, "v_mad_i64_i32 %[d1], vcc, v127, %[y], %[p]\n\tv_mad_i64_i32 %[d2], vcc, v125, %[y], %[p]")
        PAIR_OP(11, "v_trig_preop_f64 %[d1], %[p], v127\n\tv_trig_preop_f64 %[d2], %[p], v125")
        WORD_OP(12, "v_mul_hi_u32 %[d1], v127, %[y]\n\tv_mul_hi_u32 %[d2], v125, %[y]")
        WORD_OP(13, "v_mul_lo_u32 %[d1], v127, %[y]\n\tv_mul_lo_u32 %[d2], v125, %[y]")
        WORD_OP(14, "v_lshrrev_b32_e32 %[d1], v127, %[y]\n\tv_lshrrev_b32_e32 %[d2], v125, %[y]")
        WORD_OP(15, "v_frexp_exp_i32_f64_e32 %[d1], %[p]\n\tv_mov_b32_e32 %[d2], %[d1]")      // (control: no 32-bit source)
    }
    lds[threadIdx.x] = make_float4((float)wrong[0], 0.f, 0.f, 0.f);
    __syncthreads();
    for (int i = 0; i < OPS; i++) if (wrong[i]) atomicAdd(&stat[16 + i], wrong[i]);
    if (blockIdx.x >= 256) { unsigned t = 0; for (int i = 0; i < OPS; i++) t += wrong[i]; if (t) atomicAdd(&stat[15], t); }
    if (lds[(threadIdx.x + 1) & 255].x == -1.f) stat[0] = 1;
}

// Where in the SIMD's register file do the failing waves live?  Every wave records its HW_REG_GPR_ALLOC (VGPR base / size) and how many
// of its 64-bit shifts by v127 came out different from the same shifts by v125.
__global__ __launch_bounds__(256, 4) void base_probe(unsigned *__restrict__ table, int iters, unsigned long long need)
{
    __shared__ float4 lds[1664];
    unsigned wrong = 0;
    const unsigned lane = threadIdx.x & 63;
    for (int it = 0; it < iters; it++) {
        const unsigned x = 37u + ((lane + (unsigned)it) % 5u);
        unsigned long long d1, d2;
        asm volatile("v_mov_b32_e32 v127, %[x]\n\tv_mov_b32_e32 v125, %[x]\n\ts_nop 1\n\t"
                     "v_lshrrev_b64 %[d1], v127, %[n]\n\tv_lshrrev_b64 %[d2], v125, %[n]"
                     : [d1] "=&v"(d1), [d2] "=&v"(d2) : [x] "v"(x), [n] "s"(need) : "v125", "v127");
        wrong += d1 != d2;
    }
    unsigned alloc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_GPR_ALLOC)" : "=s"(alloc));
    lds[threadIdx.x] = make_float4((float)wrong, 0.f, 0.f, 0.f);
    __syncthreads();
    const unsigned any = __popcll(__ballot(wrong != 0));
    if (lane == 0) { const unsigned w = blockIdx.x * 4 + (threadIdx.x >> 6); table[2 * w] = alloc; table[2 * w + 1] = any; }
    if (lds[(threadIdx.x + 1) & 255].x == -1.f) table[0] = 1;
}

template <bool SCRATCH, int NREGS>
static void run(const char *name, const float4 *src, unsigned *stat, unsigned *samples, int grid, int iters)
{
    hipMemset(stat, 0, 64); hipMemset(samples, 0, 1024);
    hipLaunchKernelGGL((probe<SCRATCH, NREGS>), dim3(grid), dim3(256), 0, 0, src, stat, samples, iters, 64, 0xF0F0A5A5C3C3FFFFull ^ ((unsigned long long)grid << 20));
    hipDeviceSynchronize();
    unsigned h[16], s[256];
    hipMemcpy(h, stat, 64, hipMemcpyDeviceToHost); hipMemcpy(s, samples, 1024, hipMemcpyDeviceToHost);
    printf("%-22s grid %5d iters %4d: wrong q %u, row %u, predicate %u, 64-bit shift by v127 %u, by v125 %u; in blocks < 256: %u, >= 256: %u", name, grid, iters, h[0], h[1], h[2], h[8], h[9], h[4], h[5]);
    for (unsigned i = 0; i < (h[7] < 4 ? h[7] : 4); i++)
        printf("  [block %u thread %u it %u: shift result (low) %#x row %#x pred %u]", s[4 * i], s[4 * i + 1] & 0xffff, s[4 * i + 1] >> 16, s[4 * i + 2], s[4 * i + 3] & 0xffffff, s[4 * i + 3] >> 24);
    printf("\n");
}

int main()
{
    float4 *src; unsigned *stat, *samples;
    hipMalloc(&src, 1024 * 768 * sizeof(float4)); hipMalloc(&stat, 256); hipMalloc(&samples, 1024);
    hipMemset(src, 0, 1024 * 768 * sizeof(float4));
    const int grids[3] = { 390, 512, 2048 }, its[3] = { 1, 8, 256 };
    for (int rep = 0; rep < 1; rep++)
        for (int gi = 0; gi < 3; gi++)
            for (int ii = 0; ii < 3; ii++) {
                run<false, 128>("128 regs", src, stat, samples, grids[gi], its[ii]);
                run<true, 128>("128 regs + scratch", src, stat, samples, grids[gi], its[ii]);
                run<true, 136>("136 regs + scratch", src, stat, samples, grids[gi], its[ii]);
            }
    static const char *names[OPS] = { "v_lshrrev_b64 (sgpr pair)", "v_lshlrev_b64", "v_ashrrev_i64", "v_lshrrev_b64 (vgpr pair)", "v_mad_u64_u32 src0", "v_mad_u64_u32 src1",
                                      "v_ldexp_f64 src1", "v_cvt_f64_u32", "v_cvt_f64_i32", "v_cvt_f64_f32", "v_mad_i64_i32 src0", "v_trig_preop_f64 src1",
                                      "v_mul_hi_u32", "v_mul_lo_u32", "v_lshrrev_b32", "(control)" };
    for (int gi = 1; gi < 3; gi++) {
        hipMemset(stat, 0, 256);
        hipLaunchKernelGGL(ops_probe, dim3(grids[gi]), dim3(256), 0, 0, stat, 64, 0xF0F0A5A5C3C3FFFFull);
        hipDeviceSynchronize();
        unsigned h[64];
        hipMemcpy(h, stat, 256, hipMemcpyDeviceToHost);
        printf("32-bit operand in the last register (v127) against the same operand in v125, grid %d, 64 iterations; lanes whose results differ (all in blocks >= 256: %s):\n",
               grids[gi], [&] { unsigned t = 0; for (int i = 0; i < OPS; i++) t += h[16 + i]; return t == h[15] ? "yes" : "NO"; }());
        for (int i = 0; i < OPS; i++) printf("  %-28s %u\n", names[i], h[16 + i]);
    }
    for (int gi = 0; gi < 3; gi++) {
        const int waves = grids[gi] * 4;
        unsigned *table; hipMalloc(&table, waves * 8); hipMemset(table, 0, waves * 8);
        hipLaunchKernelGGL(base_probe, dim3(grids[gi]), dim3(256), 0, 0, table, 64, 0xF0F0A5A5C3C3FFFFull);
        hipDeviceSynchronize();
        std::vector<unsigned> t(2 * waves);
        hipMemcpy(t.data(), table, waves * 8, hipMemcpyDeviceToHost);
        unsigned n[64] = { 0 }, bad[64] = { 0 }, size_seen = 0;
        for (int w = 0; w < waves; w++) { const unsigned base = t[2 * w] & 63u; size_seen = (t[2 * w] >> 8) & 63u; n[base]++; bad[base] += t[2 * w + 1] != 0; }
        printf("grid %d: waves by HW_REG_GPR_ALLOC.VGPR_BASE (VGPR_SIZE field %u) -- waves / waves with wrong 64-bit shifts by v127:", grids[gi], size_seen);
        for (int b = 0; b < 64; b++) if (n[b]) printf("  base %d: %u / %u", b, n[b], bad[b]);
        printf("\n");
        hipFree(table);
    }
    return 0;
}
