// gs_blend.hip -- k_blend: per-pixel front-to-back alpha blend, one wave per 8x8 pixel quadrant.
//
// Part of libgs3d_hip.so (gfx950 only).  Built with -ffp-contract=off: the floating-point contract of this path is "IEEE
// binary32, one rounding per operation, in the order the reference shader writes it" (DESIGN.md section 3); fused
// multiply-adds appear only where written explicitly.
// Reference restated (paths relative to /root/reference/src/shaders): render.comp:30-99
#include "gs_device.h"

namespace gs {

#ifndef GS_BLEND_ASM_ACCUMULATE
#define GS_BLEND_ASM_ACCUMULATE 1  // the guarded mode's accumulate as one hand-written exec-masked block (A/B: 0 = the compiler's form;
                                   // the same block for the nine-instruction accumulate of the other modes measured +-0: not kept)
#endif
#ifndef GS_BLEND_ASM_LOOP
#define GS_BLEND_ASM_LOOP 1  // round 5: the guarded mode's pair loop as ONE hand-written loop that keeps `alive` in exec (A/B: 0 = the loop below)
#endif
#ifndef GS_BLEND_SALU_DIET
#define GS_BLEND_SALU_DIET 2  // 2: the pair loop's tail hand-written as well (round 4: blend -3..4 %: the scalar unit is a co-bottleneck)
#endif

// ---------------------------------------------------------------------------------------
// blend.  One wave per 8x8 pixel quadrant of a 16x16 tile (4 waves = one workgroup per tile), and the
// four waves share nothing: their LDS slabs, tables and lists are wave-private, so a quadrant whose pixels have
// saturated retires at once.  The ONE workgroup barrier is optional and sits at the top of the chunk loop
// (`lockstep`, blend_walk): the waves of a tile that are still in that loop take every chunk together.  A wave leaves
// the loop for good when its quadrant has saturated or the guard abandons it, and s_barrier counts only the waves
// of the workgroup that have not ENDED -- so: NO barrier may follow the chunk loop (the exact re-render of an
// abandoned quadrant, the pixel stores): a wave waiting there would wait for siblings that wait at the loop's top
// for it.  tests/test_gpu_guard.py::test_lockstep_with_quadrants_that_end_early_abandon_and_walk_on mixes the three.
//
// Each wave walks its tile's depth-sorted list in chunks of 64 entries: lane l fetches entry l's record
// (one 64-byte line, gathered through the sorted Gaussian id; the next chunk is prefetched while the current one
// is blended), tests it against the wave's quadrant, parks it in a wave-private LDS slab, and a 64-bit
// ballot of the survivors drives a scalar loop that evaluates only those entries, in list order, with
// broadcast LDS reads.  Pixel predicates live in 64-bit scalar masks; the accumulate runs under the exec mask.
//
// render.comp:61-98 semantics.  Floating-point contract of the DEFAULT arithmetic (CONTRACT = false): every product and
// sum of :66 and :87 rounded on its own, in the order the shader writes them -- what the reference's text compiled for the
// CPU evaluates.  CONTRACT = true: the three multiply-adds GLSL lets a compiler contract, written as explicit FMAs (opt-in).
//
// The two data-dependent cuts of the loop, and why no exp mode can flip the first:
//   :78  `if (alpha < 1/255) continue`  is decided on `power`, against the entry's ALPHA CUT (gs_device.h: alpha_cut; a
//        function of the opacity, computed at load, carried in the record): power >= cut  <=>  the reference's alpha, with
//        libm's expf, is >= 1/255 -- bit for bit, by monotonicity.  Lanes below the cut never evaluate exp.
//   :82  `if (T (1 - alpha) < 1e-4) break`  depends on the accumulated T.  With EXP = 2 (libm's expf restated) alpha and T are
//        the reference's own bits.  With a fast exp they drift by a few ULP per step; GUARD = true makes the decision safe:
//        a wave in which a pixel's T (1 - alpha) comes within a PROVEN window of 1e-4 (blend_walk, kGuard*) abandons its
//        fast pass and re-renders its quadrant with EXP = 2.  Everything that is not re-rendered has taken exactly the
//        reference's decisions and differs from it by rounding noise only.
//
// Exactness of the culling: an entry contributes to a pixel only if power >= cut, and -power = q(d) = 0.5 d^T C d
// (C = conic) is a convex quadratic of d = uv - pixel.  If the minimum of q over the quadrant's pixel rectangle exceeds
// -cut (plus 16 ULP of the quadratic's largest TERMS over the quadrant, against the cancellation error of `power` and of
// the bound itself for thin diagonal splats), every pixel of the quadrant executes `continue` in the shader, so skipping
// the entry for that wave changes nothing.
// ---------------------------------------------------------------------------------------

// min over the pixel rectangle [xa,xb] x [ya,yb] of q(d) = 0.5 (c00 dx^2 + c11 dy^2) + c01 dx dy,
// d = uv - pixel.  q is convex with its minimum 0 at d = 0: inside the rectangle the answer is 0,
// otherwise the minimum lies on an edge facing the centre, where q is a 1-D parabola.
__device__ __forceinline__ float min_q_rect(float c00, float c01, float c11, float u, float v, float xa,
                                            float xb, float ya, float yb) {
    const float dx_lo = u - xb, dx_hi = u - xa, dy_lo = v - yb, dy_hi = v - ya;
    const bool in_x = !(dx_lo > 0.0f) && !(dx_hi < 0.0f), in_y = !(dy_lo > 0.0f) && !(dy_hi < 0.0f);  // NaN -> inside
    // This is a bound, not part of the pipeline's arithmetic: FMAs are welcome (the caller's slack covers rounding).
    // q(a, t) = h00 a^2 + t (h11 t + c01 a) with h = c / 2.  Only the edges that face the centre can hold the minimum
    // (a segment from the centre to a point of a far edge crosses a near edge, where the convex q is smaller): at most
    // one vertical and one horizontal edge; on an edge one coordinate is fixed and the other is the parabola's
    // minimiser r * fixed, clamped to the edge.
    const float h00 = 0.5f * c00, h11 = 0.5f * c11;
    const float r11 = -c01 * __builtin_amdgcn_rcpf(c11), r00 = -c01 * __builtin_amdgcn_rcpf(c00);
    const float a = dx_lo > 0.0f ? dx_lo : dx_hi;  // the vertical edge nearer to the centre: dx fixed, parabola in dy
    const float t = fminf(fmaxf(r11 * a, dy_lo), dy_hi);
    const float qv = __builtin_fmaf(t, __builtin_fmaf(h11, t, c01 * a), h00 * a * a);
    const float b = dy_lo > 0.0f ? dy_lo : dy_hi;  // the horizontal edge nearer to the centre
    const float s = fminf(fmaxf(r00 * b, dx_lo), dx_hi);
    const float qh = __builtin_fmaf(s, __builtin_fmaf(h00, s, c01 * b), h11 * b * b);
    // The cull `mq > lim` needs mq to be a LOWER bound of q over the quadrant, and evaluating the parabola at an inexact
    // minimiser (v_rcp_f32: 1 ULP) OVER-estimates its minimum -- by a second-order amount: q(t* + dt) - q(t*) = h dt^2 with
    // dt/t* ~ 2^-23, i.e. ~1e-14 relative, which the caller's slack (9.6e-7 x the quadratic's largest terms) absorbs
    // many times over.  A coarser reciprocal or a smaller slack must revisit this.
    return in_x ? (in_y ? 0.0f : qh) : (in_y ? qv : fminf(qv, qh));
}

#if defined(GS_BLEND_STATS) || defined(GS_BLEND_CLOCK)
// debug instrumentation (separate builds, never the shipped library).  GS_BLEND_STATS: the work counters (they live in the compiler's
// form of the pair loop); GS_BLEND_CLOCK: only the shader clock the kernel ran at, measured around the SHIPPED loops
__device__ unsigned long long g_blend_stats[12];
// per wave of the last launch: shader cycles (s_memtime) and constant-rate ticks (s_memrealtime) between its first and its last instruction
// -- plain stores into the wave's own slot (two atomics per wave on two addresses serialise at 12 ns each: 0.8 ms for config B's 32 640 waves)
constexpr uint32_t kClockSlots = 1u << 18;
__device__ uint2 g_blend_clock[kClockSlots];
#endif
#ifdef GS_BLEND_STATS
#define STAT_ADD(i, v) do { const unsigned long long v_ = (unsigned long long)(v); const bool first_ = (__ffsll((unsigned long long)__ballot(true)) - 1) == lane; if (first_) atomicAdd(&g_blend_stats[i], v_); } while (0)
#else
#define STAT_ADD(i, v) do { } while (0)
#endif

struct BlendEntry {
    float4 co;  // c00 c01 c11 opacity
    float4 uv;  // u v r g
    float2 bc;  // b, alpha cut
};

__device__ __forceinline__ void blend_fetch(BlendEntry& e, uint32_t g, const AttrRecord* __restrict__ rec) {
    const AttrRecord* r = rec + g;  // one 64-byte line
    e.co = r->conic_op;
    e.uv = r->uv_rg;
    const float4 t = r->b_depth_r;
    e.bc = make_float2(t.x, t.w);
}

// ---- the guard of render.comp:82-85 for a fast exp (GUARD) ---------------------------------------------------------
// Let a' = fl(o e') be the fast alpha (e' = v_exp_f32(fl(power log2e))) and a = fl(o e) the reference's (e = libm's expf).
// tests/test_gpu_expf.py scans every binary32 power in [-16, 0] ON THE DEVICE and asserts |e' - e| <= (E0e + E1 |power|) e;
// the two roundings of the product add 2^-23:  |a' - a| <= (kGuardE0 + kGuardE1 |power|) a,  kGuardE0 = E0e + 2^-23.
// One step  T <- fl(T fl(1 - alpha)):  the two chains' (1 - alpha) differ by |a' - a| plus one ULP, the products by one more
// ULP, so the relative distance of the chains grows per step by at most
//      a / (1 - a) (kGuardE0 + kGuardE1 |power|)  +  2^-22.
// With o <= 1, |power| <= ln(1/a) and a/(1 - a) ln(1/a) <= 1, so after n steps
//      |T'(1 - a') - T(1 - a)|  <=  W T(1 - a),      W = S kGuardE0 + n (kGuardE1 + 2^-22),   S = sum a_i / (1 - a_i)
// (n <= the (entry, wave) pairs evaluated so far, a scalar count).  S has two bounds:
//   * for any pixel S <= 297: while no break was taken prod (1 - a_i) >= 1e-4, and a/(1 - a) <= 21.5 (-ln(1 - a)) for a <= 0.99,
//     i.e. <= 198; the tested step adds <= 99.  With n <= kGuardMaxPairs this gives the COARSE window, two constants;
//   * per lane: over any run of steps  sum a_i/(1 - a_i) <= prod (1 + a_i/(1 - a_i)) - 1 = T_before / T_after - 1,  so a lane
//     that adds T_chunk_start / T_chunk_end - 1 to a running sum at every chunk end (one v_rcp per 64 entries) carries a bound
//     that is a few units for ordinary pixels (a dozen chunks, T falling by a factor of two or three in each).  This gives
//     the lane's OWN window, evaluated only for lanes inside the coarse one.
// A lane whose fast T(1 - alpha) lies outside [1e-4 (1 - W), 1e-4 (1 + W)) takes the reference's decision.  For a lane inside
// its own window, the reference's decision is COMPUTED:
// resolve_break replays that one pixel's list with the reference's arithmetic (64 entries per step across the lanes, libm's
// expf, then the chain of fl(T fl(1 - alpha)) products in list order) and returns what render.comp:83 sees.  A quadrant that
// needs more than kGuardMaxResolves of those, or one beyond the kGuardList entries its list of kept entries holds, is abandoned and re-rendered whole
// with EXP = 2.  (A scene holding an opacity > 1 -- outside the sigmoid's range, where |power| is not bounded by ln(1/a) -- is
// blended with EXP = 2 altogether: launch_blend's `unit_opacity`.)  The constants carry 5 % head-room for the
// second-order terms and the rounding of the thresholds themselves.
constexpr float kGuardE0 = 2.0e-7f, kGuardE1 = 8.0e-8f;  // measured on the device: E0e = 6.9e-8 (+ 2^-23 = 1.88e-7) with this E1
#ifndef GS_GUARD_SCALE
#define GS_GUARD_SCALE 1.0f  // experiments only (tools/r04_guard.py): the window scaled; below 1 the guarantee is gone
#endif
constexpr float kGuardUnit = GS_GUARD_SCALE * 1.05f * kGuardE0, kGuardBase = 297.0f * kGuardUnit, kGuardStep = GS_GUARD_SCALE * 1.05f * (kGuardE1 + 2.3841858e-7f);
constexpr float kGuardSMax = 297.0f;
constexpr uint32_t kGuardMaxResolves = 8, kGuardMaxPairs = 4096;
constexpr uint32_t kGuardList = 384;  // entries of a wave's list of kept entries (LDS, 1.5 KiB per wave: 19 KiB per workgroup, eight per CU)
// the coarse window, constant: S <= 297, n <= kGuardMaxPairs (a quadrant that keeps more entries is abandoned)
constexpr float kGuardCoarse = kGuardBase + (float)kGuardMaxPairs * kGuardStep;
constexpr float kGuardHi = 0.0001f * (1.0f + kGuardCoarse), kGuardLo = 0.0001f * (1.0f - kGuardCoarse);
// What the guard adds to EVERY pair is one compare: beside render.comp:83's own, an SDWA compare of the upper half of
// T (1 - alpha)'s bit pattern with 0x38D1 (1e-4 is 0x38D1B717): equal <=> the value lies in the slice [9.96590e-5, 1.004364e-4),
// 0.8 % wide, which contains the coarse window.  3 % of the pairs have a lane in the slice and go on to the window tests.
// (Measured alternatives, config B, serial blend: a second compare at break events only -- 40 % of the pairs have one -- 0.193 ms
// without any replay against 0.181 ms for this form and 0.178 ms for the unguarded loop.)
constexpr uint32_t kGuardSlice = 0x38D1u;
static_assert(kGuardHi < 1.0043e-4f && kGuardLo > 9.966e-5f, "the coarse window must lie inside the slice");

// render.comp:83 for ONE pixel (fxa, fya) at the entry in position `pos` of the wave's list of kept entries, as the reference
// evaluates it: is T (1 - alpha) < 1e-4 there?  Premise (the guard's induction): no earlier entry made this pixel break.
// klist: the Gaussian ids of the entries the quadrant's exact culling kept, in list order (what it dropped is skipped by
// every pixel of the quadrant, so the pixel's chain runs over these alone).  Every lane takes one entry per step -- two or
// three steps for an ordinary quadrant; the arithmetic is the pair loop's with EXP = 2 (the pre-scaled conic, the alpha cut,
// libm's expf); the T chain runs over the kept entries in list order through v_readlane.  All 64 lanes are active, every
// value that matters is wave-uniform, and no load is in flight when this returns (the caller's pair loop keeps the next
// chunk's prefetch outstanding: a possibly-pending load on its back-edge would make the compiler wait at every pair).
__device__ __forceinline__ bool resolve_break(const uint32_t* __restrict__ klist, const uint32_t pos, const AttrRecord* __restrict__ rec,
                                              const uint2* __restrict__ exptab, const int lane, const float fxa, const float fya) {
#ifdef GS_GUARD_STUB  // experiment: the replay compiled out (wrong decisions; timing only)
    return false;
#endif
    float T = 1.0f;
    bool brk = false;
    for (uint32_t p0 = 0; p0 <= pos; p0 += WAVE) {
        const uint32_t p = p0 + (uint32_t)lane;
        float4 co = make_float4(0, 0, 0, 0);
        float2 uv = make_float2(0, 0);
        float cut = __uint_as_float(0x7F800000u);
        if (p <= pos) {
            const AttrRecord* r = rec + klist[p];
            co = r->conic_op;
            uv = make_float2(r->uv_rg.x, r->uv_rg.y);
            cut = r->b_depth_r.w;
        }
        const float dx = uv.x - fxa, dy = uv.y - fya;
        const float cx = -0.5f * co.x, cy = -co.y, cz = -0.5f * co.z;
        const float s = cx * dx * dx + cz * dy * dy;
        const float power = s + cy * dx * dy;
        const bool kept = power <= 0.0f && !(power < cut);  // (cut = +inf past pos)
        const float alpha = fminf(0.99f, co.w * gs_expf_libm(power, exptab));
        const float oma = 1 - alpha;
        uint64_t m = __ballot(kept);
        while (m) {
            const int j = __ffsll((unsigned long long)m) - 1;
            m &= m - 1;
            const float f = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(oma), j));
            const float test_T = T * f;
            if (p0 + (uint32_t)j == pos) brk = test_T < 0.0001f;
            else T = test_T;
        }
    }
    return brk;
}

// ---- the guarded mode's pair loop, hand-written (round 5) -------------------------------------------------------------
// The compiler's form of the loop in blend_walk spends 26 vector and 18 scalar instructions per (entry, wave) pair, and the CU's
// one scalar unit is a co-bottleneck.  Most of those scalar instructions move lane masks between VCC,
// SGPR pairs and exec: `alive & ballot & ballot`, "is any lane left", save / restore exec around the accumulate, "did the last
// pixel die".  This loop keeps ALIVE IN EXEC for its whole life instead:
//   * the tests of render.comp:68,78 are v_cmpx (exec &= condition; round 5: two float compares, round 6: ONE unsigned compare, below): exec IS m2;
//   * under exec = m2 the break compare yields mk and the slice compare (into VCC: one s_cbranch_vccnz) yields sl directly;
//   * T <- T (1 - alpha) is written in place for every lane of m2 -- a lane that breaks here is dead from here on, its T is never
//     read again -- after the weight alpha T was formed from the old T; the accumulate runs under exec = m2 & ~mk (one s_andn2);
//   * the wave's kept entries are staged COMPACTED (rank order), so the loop walks an address and a count: no s_ff1 / s_bitset0.
// Round 5: 25 vector + 7 scalar instructions per pair on the common path, two branches (the event test and the back edge); round 6: 23.25 + 7 and
// 1.25 taken branches (below).  Same arithmetic, operation for operation, as the loop it
// replaces (every product and sum of render.comp:66 rounded on its own, v_exp_f32, fl(o e), min, 1 - alpha, T (1 - alpha), the
// fused accumulate of the guarded mode): the images of the two loops are bit-identical (tools/ab_image_check.py).
// The loop returns to C++ (event = 1) only where the guard needs it: a lane of m2 inside the guard's COARSE window around 1e-4 (the
// slice test, one instruction on every pair, sends ~3 % of the pairs to the two compares of the coarse window, still inside the loop;
// a third of those leave it); the pending pair is then finished by the caller (own window, resolve_break, accumulate) and the loop
// re-entered (leaving costs a few hundred cycles: the slab is re-read on both sides).
// Loaded records live in v[54:63] (a 128-bit operand's components cannot be named in inline asm): clobbered, the allocator
// keeps out.  Hazards inside the string: v_exp_f32 -> its consumer (trans op, 1 state: the s_nop); none of the others apply
// (no DPP / readlane / VMEM here; SALU and branch reads of VALU-written VCC / EXEC / SGPRs are interlocked).
// Round 6: three things less per pair.  (1) ONE compare decides render.comp:68 and :78: the conic is staged with its signs as
// they are ((c00/2, c01, c11/2) instead of the negated triple), so the chain yields pn = -power -- the same magnitudes, every
// rounding mirrored -- and the slab holds ncut = -cut >= +0.  power <= 0 and power >= cut  <=>  +0 <= pn <= ncut, and for binary32
// values that is ONE unsigned compare of the bit patterns: a negative pn (power > 0) has the sign bit set and compares above every
// ncut <= +inf, so does every NaN, and pn is never -0 (it is a sum whose first operand, c00' dx dx + c11' dy dy, is >= +0 for the
// c00, c11 >= +0 that preprocess produces from a > 0, c > 0, det > 0; x + y is -0 only for -0 + -0).  (2) The loop is unrolled by
// four over immediate ds_read offsets: the address moves once per four pairs and three of four back edges are not-taken
// branches.  (3) the exp's argument is pn * (-log2 e): the sign costs nothing.
// 24 - 0.75 vector + 7 scalar instructions per pair (round 5: 25 + 7); same arithmetic: frames bit-identical (tools/ab_image_check.py).
struct PairLoopEvent {
    uint64_t m2, mk, sl;   // lanes that evaluated exp; of those: T (1 - alpha) < 1e-4 with the fast exp; inside the slice
    float w;               // alpha * T(before) -- valid in the lanes of m2 (the entry's colour is re-read from the slab: 3 % of the pairs)
};
#define GS_STR_(x) #x
#define GS_STR(x) GS_STR_(x)
// one pair of the unrolled loop: J = 0..3, the three planes of entry J behind the address at J*16, 1024 + J*16, 2048 + J*16
#define GS_PAIR_HEAD(J)                                                                                                       \
        ".Lgs_pair" GS_STR(J) "_%=:\n\t"                                                                                      \
        "ds_read_b128 v[54:57], %[addr] offset:" GS_STR(J) "*16\n\t"                                                          \
        "ds_read_b128 v[58:61], %[addr] offset:1024+" GS_STR(J) "*16\n\t"                                                     \
        "ds_read_b64 v[62:63], %[addr] offset:2048+" GS_STR(J) "*16\n\t"                                                      \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                            \
        "v_sub_f32 v58, v58, %[fx]\n\t"           /* dx = u - x   (the loaded registers double as temporaries) */             \
        "v_sub_f32 v59, v59, %[fy]\n\t"           /* dy = v - y */                                                            \
        "v_mul_f32 v54, v54, v58\n\t"             /* c00' dx */                                                               \
        "v_mul_f32 v56, v56, v59\n\t"             /* c11' dy */                                                               \
        "v_mul_f32 v54, v58, v54\n\t"             /* c00' dx dx */                                                            \
        "v_mul_f32 v56, v59, v56\n\t"             /* c11' dy dy */                                                            \
        "v_mul_f32 v55, v55, v58\n\t"             /* c01' dx */                                                               \
        "v_add_f32 v54, v54, v56\n\t"             /* -s */                                                                    \
        "v_mul_f32 v55, v55, v59\n\t"             /* c01' dx dy */                                                            \
        "v_add_f32 v54, v55, v54\n\t"             /* pn = -power                          (render.comp:66) */                 \
        "v_cmpx_le_u32 vcc, v54, v63\n\t"         /* +0 <= pn <= ncut: power <= 0 and power >= the alpha cut (render.comp:68,78) */
#define GS_PAIR_GUARDED(J)                                                                                                    \
        GS_PAIR_HEAD(J)                                                                                                       \
        "v_mul_f32 v56, 0xbfb8aa3b, v54\n\t"      /* power log2 e */                                                          \
        "v_exp_f32 v56, v56\n\t"                                                                                              \
        "s_nop 0\n\t"                                                                                                         \
        "v_mul_f32 v56, v57, v56\n\t"             /* o e' */                                                                  \
        "v_min_f32 v56, 0x3f7d70a4, v56\n\t"      /* alpha = min(0.99, .)             (render.comp:77) */                     \
        "v_sub_f32 v55, 1.0, v56\n\t"             /* 1 - alpha */                                                             \
        "v_mul_f32 %[w], v56, %[T]\n\t"           /* alpha T: the weight of this entry (the T of before) */                   \
        "v_mul_f32 %[T], %[T], v55\n\t"           /* test_T = T (1 - alpha), in place */                                      \
        "v_cmp_gt_f32 %[mk], %[k1e4], %[T]\n\t"   /* test_T < 1e-4                     (render.comp:83) */                    \
        "v_cmp_eq_u32_sdwa vcc, %[T], %[slice] src0_sel:WORD_1 src1_sel:DWORD\n\t"  /* ... any of them inside the guard's slice? */ \
        "s_cbranch_vccnz .Lgs_slice" GS_STR(J) "_%=\n"                                                                        \
        ".Lgs_cont" GS_STR(J) "_%=:\n\t"                                                                                      \
        "s_andn2_b64 exec, exec, %[mk]\n\t"                                                                                   \
        "v_fmac_f32 %[c0], v60, %[w]\n\t"         /* render.comp:87 (the guarded mode's fused form) */                        \
        "v_fmac_f32 %[c1], v61, %[w]\n\t"                                                                                     \
        "v_fmac_f32 %[c2], v62, %[w]\n\t"                                                                                     \
        "s_andn2_b64 %[alive], %[alive], %[mk]\n\t"                                                                           \
        "s_cselect_b32 %[rem], %[rem], 0\n\t"     /* the quadrant's last pixel has saturated: this was the last pair */       \
        "s_mov_b64 exec, %[alive]\n\t"                                                                                        \
        "s_add_u32 %[rem], %[rem], -1\n\t"        /* carry = there was another pair */
// a lane of m2 inside the slice: inside the guard's COARSE window as well?  (under exec = m2: bits of m2's lanes only.)  No: every
// decision of this pair is the fast arithmetic's, carry on (T(6e6): most slice hits end here).  Yes: leave with the pending pair --
// exec restored FIRST (the address register belongs to every lane), then the address put past the pair.
#define GS_PAIR_SLICE(J)                                                                                                      \
        ".Lgs_slice" GS_STR(J) "_%=:\n\t"                                                                                     \
        "v_cmp_le_f32 %[sl], %[klo], %[T]\n\t"                                                                                \
        "v_cmp_gt_f32 %[m2], %[khi], %[T]\n\t"                                                                                \
        "s_and_b64 %[sl], %[sl], %[m2]\n\t"                                                                                   \
        "s_cbranch_scc0 .Lgs_cont" GS_STR(J) "_%=\n\t"                                                                        \
        "s_mov_b64 %[m2], exec\n\t"                                                                                           \
        "s_mov_b64 exec, %[sv]\n\t"                                                                                           \
        "v_add_u32 %[addr], 16+" GS_STR(J) "*16, %[addr]\n\t"                                                                 \
        "s_branch .Lgs_event_%=\n"
// slab_addr: LDS byte address of the next entry's {c00' c01' c11' o} (the other two planes 1024 and 2048 bytes behind it);
// rem: pairs left in the chunk MINUS ONE.  Returns 0 when the chunk is exhausted or alive == 0, 1 with `ev` filled (rem then
// still counts the pending pair, slab_addr is already past it).
__device__ __forceinline__ uint32_t blend_pair_loop(uint32_t& slab_addr, uint32_t& rem, uint64_t& alive, const float fx, const float fy, float& T,
                                                    float& c0, float& c1, float& c2, PairLoopEvent& ev) {
    uint32_t event;
    uint64_t saved;
    const float k1e4 = 0.0001f, klo = kGuardLo, khi = kGuardHi;
    // (the count is wave-uniform, but the compiler may carry it in a VGPR -- it feeds VALU code too -- and an inline-asm SGPR operand
    //  is not legalised for it: "illegal VGPR to SGPR copy")
    uint32_t left = (uint32_t)__builtin_amdgcn_readfirstlane((int)rem);
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, %[alive]\n"
        GS_PAIR_GUARDED(0)
        "s_cbranch_scc0 .Lgs_done_%=\n"
        GS_PAIR_GUARDED(1)
        "s_cbranch_scc0 .Lgs_done_%=\n"
        GS_PAIR_GUARDED(2)
        "s_cbranch_scc0 .Lgs_done_%=\n"
        GS_PAIR_GUARDED(3)
        "v_add_u32 %[addr], 64, %[addr]\n\t"
        "s_cbranch_scc1 .Lgs_pair0_%=\n"
        ".Lgs_done_%=:\n\t"
        "s_mov_b32 %[ev], 0\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "s_branch .Lgs_exit_%=\n"
        GS_PAIR_SLICE(0)
        GS_PAIR_SLICE(1)
        GS_PAIR_SLICE(2)
        GS_PAIR_SLICE(3)
        ".Lgs_event_%=:\n\t"
        "s_mov_b32 %[ev], 1\n"
        ".Lgs_exit_%=:"
        : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [T] "+v"(T), [addr] "+v"(slab_addr), [alive] "+s"(alive), [rem] "+s"(left),
          [ev] "=&s"(event), [mk] "=&s"(ev.mk), [m2] "=&s"(ev.m2), [sl] "=&s"(ev.sl), [sv] "=&s"(saved),
          [w] "=&v"(ev.w)
        : [fx] "v"(fx), [fy] "v"(fy), [k1e4] "s"(k1e4), [slice] "s"(kGuardSlice), [klo] "s"(klo), [khi] "s"(khi)
        : "vcc", "scc", "memory", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
    rem = left;
    return event;
}

// The same loop for the EXACT mode (EXP = 2, no guard: libm's expf restated in binary64, gs_expf_libm, and render.comp:87 as written --
// every product of color * alpha * T and every sum rounded on its own).  No events: the loop ends with the chunk or with the quadrant's last
// pixel.  Operation for operation the arithmetic of the compiler's form (the exact mode's frames are the reference's bit for bit either
// way: the parity tests run in this mode); what goes is its mask traffic -- 17 scalar instructions and three taken branches per pair down to
// 7 and one.  Registers: v[52:63] clobbered -- the record in v[54:63]; binary64 temporaries in the even-aligned pairs v[52:53] (the table
// entry, then y), v[54:55] (kd, then the cubic), v[58:59] (x, r, r s) as the record's fields retire.
#ifndef GS_BLEND_ASM_LOOP_EXACT
#define GS_BLEND_ASM_LOOP_EXACT 1  // 0 (A/B builds): the compiler's form of the exact mode's pair loop
#endif
// Round 6: the single unsigned compare on pn = -power and the unroll by four of the guarded loop (see there); gs_expf_libm takes
// power = -pn through the conversion's sign modifier.  The exp section is thirty issue slots here: pairs no pixel keeps branch over it.
#define GS_PAIR_EXACT(J)                                                                                                      \
        GS_PAIR_HEAD(J)                                                                                                       \
        "s_cbranch_execz .Lgs_xskip" GS_STR(J) "_%=\n\t"                                                                      \
        "v_cvt_f64_f32 v[58:59], -v54\n\t"                         /* gs_expf_libm(power):  xd */                             \
        "v_fma_f64 v[54:55], %[iln], v[58:59], %[shift]\n\t"       /* kd = fma(InvLn2N, xd, SHIFT): k in the low mantissa bits */ \
        "v_and_b32 v63, 31, v54\n\t"                                                                                          \
        "v_lshl_add_u32 v63, v63, 3, %[tab]\n\t"                                                                              \
        "ds_read_b64 v[52:53], v63\n\t"                            /* t = tab[k & 31] */                                      \
        "v_lshlrev_b32 v56, 15, v54\n\t"                           /* k << 47, its upper word */                              \
        "v_add_f64 v[54:55], v[54:55], -%[shift]\n\t"              /* kd - SHIFT */                                           \
        "v_fma_f64 v[58:59], v[58:59], %[iln], -v[54:55]\n\t"      /* r = fma(InvLn2N, xd, -kd) */                            \
        "v_fma_f64 v[54:55], %[C0], v[58:59], %[C1]\n\t"           /* q = (C0 r + C1) r + C2 */                               \
        "v_fma_f64 v[54:55], v[54:55], v[58:59], %[C2]\n\t"                                                                   \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                            \
        "v_add_u32 v53, v56, v53\n\t"                              /* s = t + (k << 47) */                                    \
        "v_mul_f64 v[58:59], v[58:59], v[52:53]\n\t"               /* r s */                                                  \
        "v_fma_f64 v[52:53], v[54:55], v[58:59], v[52:53]\n\t"     /* y = fma(q, r s, s) */                                   \
        "v_cvt_f32_f64 v54, v[52:53]\n\t"                          /* the one rounding to binary32 */                         \
        "v_mul_f32 v54, v57, v54\n\t"             /* o e */                                                                   \
        "v_min_f32 v54, 0x3f7d70a4, v54\n\t"      /* alpha = min(0.99, .)             (render.comp:77) */                     \
        "v_sub_f32 v55, 1.0, v54\n\t"             /* 1 - alpha */                                                             \
        "v_mul_f32 v55, %[T], v55\n\t"            /* test_T = T (1 - alpha) */                                                \
        "v_cmp_gt_f32 %[mk], %[k1e4], v55\n\t"    /* test_T < 1e-4: the pixel is done   (render.comp:83; under exec = m2) */  \
        "s_andn2_b64 exec, exec, %[mk]\n\t"                                                                                   \
        "v_mul_f32 v60, v60, v54\n\t"             /* color * alpha * T, left to right    (render.comp:87) */                  \
        "v_mul_f32 v61, v61, v54\n\t"                                                                                         \
        "v_mul_f32 v62, v62, v54\n\t"                                                                                         \
        "v_mul_f32 v60, %[T], v60\n\t"                                                                                        \
        "v_mul_f32 v61, %[T], v61\n\t"                                                                                        \
        "v_mul_f32 v62, %[T], v62\n\t"                                                                                        \
        "v_add_f32 %[c0], %[c0], v60\n\t"                                                                                     \
        "v_add_f32 %[c1], %[c1], v61\n\t"                                                                                     \
        "v_add_f32 %[c2], %[c2], v62\n\t"                                                                                     \
        "v_mov_b32 %[T], v55\n\t"                                                                                             \
        "s_andn2_b64 %[alive], %[alive], %[mk]\n\t"                                                                           \
        "s_cselect_b32 %[rem], %[rem], 0\n"       /* the quadrant's last pixel is done: this was the last pair */             \
        ".Lgs_xskip" GS_STR(J) "_%=:\n\t"                                                                                     \
        "s_mov_b64 exec, %[alive]\n\t"                                                                                        \
        "s_add_u32 %[rem], %[rem], -1\n\t"        /* carry = there was another pair */
__device__ __forceinline__ void blend_pair_loop_exact(uint32_t slab_addr, const uint32_t rem, uint64_t& alive, const float fx, const float fy,
                                                      float& T, float& c0, float& c1, float& c2, const uint2* __restrict__ tab) {
    const double InvLn2N = 0x1.71547652b82fep+0 * 32.0, SHIFT = 0x1.8p+52;  // gs_expf_libm's constants
    const double C0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0, C1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0, C2 = 0x1.62e42ff0c52d6p-1 / 32.0;
    const float k1e4 = 0.0001f;
    uint64_t saved, mk;
    uint32_t left = (uint32_t)__builtin_amdgcn_readfirstlane((int)rem);
    const uint32_t tab_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)tab);  // (wave-uniform: the wave's copy in LDS)
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, %[alive]\n"
        GS_PAIR_EXACT(0)
        "s_cbranch_scc0 .Lgs_xdone_%=\n"
        GS_PAIR_EXACT(1)
        "s_cbranch_scc0 .Lgs_xdone_%=\n"
        GS_PAIR_EXACT(2)
        "s_cbranch_scc0 .Lgs_xdone_%=\n"
        GS_PAIR_EXACT(3)
        "v_add_u32 %[addr], 64, %[addr]\n\t"
        "s_cbranch_scc1 .Lgs_pair0_%=\n"
        ".Lgs_xdone_%=:\n\t"
        "s_mov_b64 exec, %[sv]"
        : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [T] "+v"(T), [addr] "+v"(slab_addr), [alive] "+s"(alive), [rem] "+s"(left),
          [mk] "=&s"(mk), [sv] "=&s"(saved)
        : [fx] "v"(fx), [fy] "v"(fy), [k1e4] "s"(k1e4), [tab] "s"(tab_addr), [iln] "s"(InvLn2N), [shift] "v"(SHIFT), [C0] "s"(C0), [C1] "v"(C1),
          [C2] "v"(C2)
        : "vcc", "scc", "memory", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
}

// One pass of a wave over its tile's list: returns false if the quadrant has to be re-rendered exactly (GUARD only; c0..c2
// are then meaningless).  slab: this wave's three planes of 64 float4 {c00' c01' c11' o} {u v r g} {b, cut, -, -} --
// plane-major keeps the staging ds_write_b128 conflict-free (lane stride 16 B); one scalar-derived address + constant offsets
// serve the broadcast reads.  resolved: GUARD, how many break decisions resolve_break took.
// EXP: 0 the pipeline-defined polynomial (gs_exp_blend), 1 the hardware's v_exp_f32, 2 libm's expf restated (gs_expf_libm).
template <int EXP, bool CONTRACT, bool GUARD>
__device__ __forceinline__ bool blend_walk(const uint2 range, const uint32_t* __restrict__ sorted_gid,
                                           const AttrRecord* __restrict__ rec, float4 (*__restrict__ slab)[WAVE],
                                           uint2* __restrict__ exptab_rw, uint32_t* __restrict__ klist, const int lane, const float fx, const float fy,
                                           const float rx0, const float ry0, uint64_t alive, float& c0, float& c1, float& c2,
                                           uint32_t& resolved, bool& table_ready, const bool lockstep) {
    const uint2* __restrict__ exptab = exptab_rw;
#if defined(GS_BLEND_STATS)
    constexpr bool kAsmLoop = false, kAsmLoopExact = false;  // (the work counters live in the compiler's form of the loop)
#else
    constexpr bool kAsmLoop = GUARD && EXP == 1 && !CONTRACT && GS_BLEND_ASM_LOOP != 0;
    constexpr bool kAsmLoopExact = !GUARD && EXP == 2 && !CONTRACT && GS_BLEND_ASM_LOOP_EXACT != 0;
#endif
    float T = 1.0f;
    c0 = c1 = c2 = 0.0f;
    uint32_t npairs = 0;                       // GUARD: (entry, wave) pairs evaluated so far (bounds every lane's step count)
    bool abandon = false;                      // GUARD: re-render the quadrant exactly
    float S = 0.0f, T_cs = 1.0f;               // GUARD: the lane's bound on sum a/(1 - a) over the finished chunks; T at the chunk's start
    resolved = 0;
    // software pipeline over 64-entry chunks: ids two chunks ahead, records one chunk ahead
    BlendEntry nxt;
    nxt.co = make_float4(0, 0, 0, 0);
    nxt.uv = make_float4(0, 0, 0, 0);
    nxt.bc = make_float2(0, 0);
    uint32_t g_next = 0, g_nxt = 0;  // ids of chunk +2 (in flight); GUARD: the ids `nxt` was fetched through
    {
        const uint32_t i0 = range.x + lane;
        if (i0 < range.y) {
            g_nxt = sorted_gid[i0];
            blend_fetch(nxt, g_nxt, rec);
        }
        const uint32_t i1 = i0 + WAVE;
        if (i1 < range.y) g_next = sorted_gid[i1];
    }
    for (uint32_t base = range.x; base < range.y; base += WAVE) {
        // LOCKSTEP (a launch parameter, wave-uniform): the tile's four waves -- those still running: a wave that has ended no longer counts
        // for s_barrier -- take every chunk together, so that their gathers of the same 64 records meet in the CU's L1 (one fetch from
        // L2 instead of up to four).  Where the blend is bound by the L1-miss traffic (trained-like scenes: long lists of which a
        // quadrant keeps one entry in seven, T(6e6): 5.3 TB/s of L2 reads either way) this is worth 25 % of the kernel; where it is
        // bound by the pair loop (S scenes) the waves wait for the slowest of four at every chunk and lose 9 %.  The renderer
        // measures both and keeps the faster (gs_renderer.cpp: BlendTuner); the frames are bit-identical.
        if (lockstep) __builtin_amdgcn_s_barrier();
        const BlendEntry cur = nxt;
        const uint32_t g_cur = g_nxt;  // GUARD: this chunk's Gaussian ids (for the wave's list of kept entries)
        const bool have = base + lane < range.y;
        {   // prefetch: records of chunk +1 (ids already here), ids of chunk +2
            const uint32_t i1 = base + WAVE + lane;
            if (GUARD) g_nxt = g_next;
            if (i1 < range.y) blend_fetch(nxt, g_next, rec);
            const uint32_t i2 = i1 + WAVE;
            if (i2 < range.y) g_next = sorted_gid[i2];
        }
        // classify entry `lane` of this chunk against the wave's quadrant.  cut = +inf: no power <= 0 is kept (never
        // enters); -inf (NaN or infinite opacity: min(0.99, NaN) is 0.99 in the pipeline's definition): never culled
        const float cut = cur.bc.y;
        bool keep = have && cut <= 0.0f;
        if (keep) {
            const float mq = min_q_rect(cur.co.x, cur.co.y, cur.co.z, cur.uv.x, cur.uv.y, rx0, rx0 + 7.0f, ry0, ry0 + 7.0f);
            // The rounding error of the shader's `power` (and of mq) is relative to the TERMS c00 dx^2, c11 dy^2,
            // c01 dx dy, not to their sum: a thin diagonal splat far from its centre has terms ~1e5 cancelling to
            // q ~ 5.  The slack therefore grows with the terms at the quadrant's corner farthest from the centre
            // (16 roundings of 2^-24 each, generously).
            const float ax = fmaxf(fabsf(cur.uv.x - rx0), fabsf(cur.uv.x - (rx0 + 7.0f)));
            const float ay = fmaxf(fabsf(cur.uv.y - ry0), fabsf(cur.uv.y - (ry0 + 7.0f)));
            const float mag = __builtin_fmaf(0.5f * fabsf(cur.co.x) * ax, ax,
                                             __builtin_fmaf(0.5f * fabsf(cur.co.z) * ay, ay, fabsf(cur.co.y) * ax * ay));
            keep = !(mq > __builtin_fmaf(mag, 9.6e-7f, -cut));  // NaN -> keep
        }
        uint64_t bm = __ballot(keep);
        STAT_ADD(0, 1);
        STAT_ADD(6, __popcll(__ballot(have)));
        STAT_ADD(1, __popcll(bm));
        if (bm == 0) continue;
        const uint64_t bm0 = bm;      // GUARD: the chunk's kept entries (bm is consumed below)
        const uint32_t kbase = npairs;  // GUARD: how many entries the list of kept entries held before this chunk
        uint32_t rank = 0;              // GUARD: this lane's entry is the chunk's rank-th kept one
        if (GUARD) {
            // abandon: set where a break decision had to be resolved (see there); more pairs than the coarse window allows for
            npairs += (uint32_t)__popcll(bm);
            if (abandon || npairs > kGuardMaxPairs) {
                abandon = true;
                break;
            }
            // the wave's list of kept entries, for resolve_break: the Gaussian id of the chunk's r-th kept entry at kbase + r
            rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
            const uint32_t at = kbase + rank;
            if (keep && at < kGuardList) klist[at] = g_cur;
            __builtin_amdgcn_wave_barrier();
        }
        if constexpr (kAsmLoop) {
            // the kept entries staged in rank order: the hand-written loop walks an address and a count
            if (keep) {  // (the conic scaled by a power of two, signs as they are: the loop computes pn = -power; ncut = -cut)
                slab[0][rank] = make_float4(0.5f * cur.co.x, cur.co.y, 0.5f * cur.co.z, cur.co.w);
                slab[1][rank] = cur.uv;
                slab[2][rank] = make_float4(cur.bc.x, -cut, 0.0f, 0.0f);
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t n_kept = (uint32_t)__popcll(bm0);
            uint32_t rem = n_kept - 1u;
            uint32_t slab_addr = (uint32_t)(uintptr_t)&slab[0][0];  // (the low half of a flat LDS pointer is the LDS offset)
            PairLoopEvent ev;
            while (blend_pair_loop(slab_addr, rem, alive, fx, fy, T, c0, c1, c2, ev) != 0) {
                // a lane of ev.m2 has its T (1 - alpha) -- now in T -- inside the slice around 1e-4: the pending pair is finished here
                const float test_T = T;
                uint64_t mk = ev.mk;
                uint64_t amb = ev.sl & __builtin_amdgcn_ballot_w64(test_T >= kGuardLo && test_T < kGuardHi);  // the coarse window
                if (amb != 0) {  // ... and the lane's own?  (S: the finished chunks; T_cs / test_T - 1: this chunk, this step included)
                    const float S_now = fminf(kGuardSMax, S + __builtin_fmaf(T_cs * 1.00001f, __builtin_amdgcn_rcpf(test_T), -1.0f));
                    const float W = __builtin_fmaf((float)npairs, kGuardStep, S_now * kGuardUnit);
                    const bool inside_own = test_T >= __builtin_fmaf(-0.0001f, W, 0.0001f) && test_T < __builtin_fmaf(0.0001f, W, 0.0001f);
                    amb &= __builtin_amdgcn_ballot_w64(inside_own);
                }
                if (amb != 0) {  // ask the reference
                    if (!table_ready) {  // the wave's copy of libm's table, on first use
                        if (lane < 32) {
                            const uint64_t v = kExpfTab[lane];
                            exptab_rw[lane] = make_uint2((uint32_t)v, (uint32_t)(v >> 32));
                        }
                        __builtin_amdgcn_wave_barrier();
                        table_ready = true;
                    }
                    mk &= ~amb;
                    const uint32_t pos = kbase + (n_kept - 1u - rem);  // the tested entry in the list of kept entries
                    do {
                        const int a = __ffsll((unsigned long long)amb) - 1;
                        amb &= amb - 1;
                        const float fxa = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(fx), a));
                        const float fya = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(fy), a));
                        // past the list's end, or one pixel too many: the quadrant is abandoned at the next chunk and re-rendered exactly
                        if (pos >= kGuardList || resolved >= kGuardMaxResolves) abandon = true;
                        else if (resolve_break(klist, pos, rec, exptab, lane, fxa, fya)) mk |= 1ull << a;
                        ++resolved;
                    } while (amb != 0);
                }
                {   // render.comp:87 for the lanes that are kept and do not break (T already holds T (1 - alpha))
                    // (wave-uniform, but after resolve_break the compiler no longer proves it: made scalar by hand)
                    mk = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(mk >> 32)) << 32) |
                         (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)mk);
                    uint64_t updm = ev.m2 & ~mk;
                    updm = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(updm >> 32)) << 32) |
                           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)updm);
                    const uint32_t at = n_kept - 1u - rem;  // the pending pair's slot in the slab
                    const float er = slab[1][at].z, eg = slab[1][at].w, eb = slab[2][at].x;
                    uint64_t saved;
                    asm volatile("s_and_saveexec_b64 %[sv], %[um]\n\t"
                                 "v_fmac_f32 %[c0], %[r], %[w]\n\t"
                                 "v_fmac_f32 %[c1], %[g], %[w]\n\t"
                                 "v_fmac_f32 %[c2], %[b], %[w]\n\t"
                                 "s_or_b64 exec, exec, %[sv]"
                                 : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [sv] "=&s"(saved)
                                 : [um] "s"(updm), [w] "v"(ev.w), [r] "v"(er), [g] "v"(eg), [b] "v"(eb)
                                 : "scc");
                }
                alive &= ~mk;
                if (alive == 0 || rem == 0) break;
                --rem;
            }
        } else if constexpr (kAsmLoopExact) {
            const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(bm0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm0, 0u));
            if (keep) {  // staged in rank order, the conic pre-scaled by a power of two (signs as they are: pn = -power), ncut = -cut
                slab[0][r] = make_float4(0.5f * cur.co.x, cur.co.y, 0.5f * cur.co.z, cur.co.w);
                slab[1][r] = cur.uv;
                slab[2][r] = make_float4(cur.bc.x, -cut, 0.0f, 0.0f);
            }
            __builtin_amdgcn_wave_barrier();
            blend_pair_loop_exact((uint32_t)(uintptr_t)&slab[0][0], (uint32_t)__popcll(bm0) - 1u, alive, fx, fy, T, c0, c1, c2, exptab);
        } else {
        // conic pre-scaled once per entry: (-c00/2, -c01, -c11/2).  Scaling by a power of two commutes with every
        // rounding below, so power is bit-identical to render.comp:66 evaluated as written while the per-pixel body
        // loses the -0.5 multiply
        slab[0][lane] = make_float4(-0.5f * cur.co.x, -cur.co.y, -0.5f * cur.co.z, cur.co.w);
        slab[1][lane] = cur.uv;
        slab[2][lane] = make_float4(cur.bc.x, cut, 0.0f, 0.0f);

        while (bm) {
            const int k = __ffsll((unsigned long long)bm) - 1;
#if GS_BLEND_SALU_DIET
            asm("s_bitset0_b64 %0, %1" : "+s"(bm) : "s"(k));  // bm &= bm - 1 costs three scalar instructions
#else
            bm &= bm - 1;
#endif
            STAT_ADD(2, 1);                       // (entry, wave) pairs evaluated
            STAT_ADD(3, __popcll(alive));         // lanes alive
            float4 co = slab[0][k];
            float4 uv = slab[1][k];
            float4 bp = slab[2][k];
            // all ten floats in one LDS round trip: without this the compiler sinks the loads of o, r, g, b
            // behind the exp() branch, where 94 % of the pairs then pay a second LDS latency.  (Reading the NEXT pair's record
            // while this one is evaluated -- two register sets taking turns -- was measured in round 4: 0.203 against 0.183 ms
            // for the unguarded loop, 0.260 against 0.242 ms for the exact one: the loop is not waiting for LDS.)
            asm volatile("" : "+v"(co.w), "+v"(uv.z), "+v"(uv.w), "+v"(bp.x));
            const float dx = uv.x - fx;
            const float dy = uv.y - fy;
            // :66  -0.5 * (co.x*dx*dx + co.z*dy*dy) - co.y*dx*dy
            float power;
            if (CONTRACT) {
                const float s = __builtin_fmaf(co.z * dy, dy, co.x * dx * dx);    // FMA  (= -0.5 * the shader's sum)
                power = __builtin_fmaf(co.y * dx, dy, s);                         // FMA
            } else {  // -0.5 * (c00 dx dx + c11 dy dy) - c01 dx dy, every product and sum rounded (the conic is pre-scaled)
                const float s = co.x * dx * dx + co.z * dy * dy;
                power = s + co.y * dx * dy;
            }
            // :68 and :78.  power <= 0 is false for NaN: a NaN power skips the entry (the pipeline's definition);
            // power >= cut  <=>  alpha >= 1/255 with the reference's exp (the alpha cut)
            const uint64_t m2 = alive & __builtin_amdgcn_ballot_w64(power <= 0.0f) &
                                __builtin_amdgcn_ballot_w64(!(power < bp.y));
            if (m2 != 0) {
                STAT_ADD(4, 1);                   // pairs reaching exp
                STAT_ADD(5, __popcll(m2));        // lanes needing exp
                // :77.  EXP 1: the hardware's v_exp_f32 (what a Vulkan driver emits for exp()); 2: libm's expf, what the
                // reference's text compiled for the CPU calls; 0: the pipeline-defined polynomial
                const float ex = EXP == 1   ? __builtin_amdgcn_exp2f(power * 1.44269502162933349609375f)
                                 : EXP == 2 ? gs_expf_libm(power, exptab)
                                            : gs_exp_blend(power);
                const float alpha = fminf(0.99f, co.w * ex);
                const float test_T = T * (1 - alpha);
                // :82-85 break
                uint64_t mk;
                if (GUARD) {
                    mk = m2 & __builtin_amdgcn_ballot_w64(test_T < 0.0001f);
                    {
                        uint64_t eq;  // upper half of the bit pattern == 0x38D1: inside the slice around 1e-4
                        asm("v_cmp_eq_u32_sdwa %0, %1, %2 src0_sel:WORD_1 src1_sel:DWORD" : "=s"(eq) : "v"(test_T), "s"(kGuardSlice));  // (the constant in an SGPR: as a VGPR it was re-materialised at every pair)
                        const uint64_t sl = m2 & eq;
                        if (sl != 0) {  // (3 % of the pairs)
                            uint64_t amb = sl & __builtin_amdgcn_ballot_w64(test_T >= kGuardLo && test_T < kGuardHi);  // the coarse window
                            if (amb != 0) {  // ... and the lane's own?  (S: the finished chunks; T_cs / test_T - 1: this chunk, this step included)
                                const float S_now = fminf(kGuardSMax, S + __builtin_fmaf(T_cs * 1.00001f, __builtin_amdgcn_rcpf(test_T), -1.0f));
                                const float W = __builtin_fmaf((float)npairs, kGuardStep, S_now * kGuardUnit);
                                const bool inside_own = test_T >= __builtin_fmaf(-0.0001f, W, 0.0001f) && test_T < __builtin_fmaf(0.0001f, W, 0.0001f);
                                amb &= __builtin_amdgcn_ballot_w64(inside_own);
                            }
                            if (amb != 0) {  // ask the reference
                                if (!table_ready) {  // the wave's copy of libm's table, on first use
                                    if (lane < 32) {
                                        const uint64_t v = kExpfTab[lane];
                                        exptab_rw[lane] = make_uint2((uint32_t)v, (uint32_t)(v >> 32));
                                    }
                                    __builtin_amdgcn_wave_barrier();
                                    table_ready = true;
                                }
                                mk &= ~amb;
                                do {
                                    const int a = __ffsll((unsigned long long)amb) - 1;
                                    amb &= amb - 1;
                                    const float fxa = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(fx), a));
                                    const float fya = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(fy), a));
                                    const uint32_t pos = kbase + (uint32_t)__popcll(bm0 & ((1ull << k) - 1ull));  // the tested entry in the list
                                    // past the list's end, or one pixel too many: the quadrant is abandoned at the next chunk and
                                    // re-rendered exactly (necessary / cheaper)
                                    if (pos >= kGuardList || resolved >= kGuardMaxResolves) abandon = true;
                                    else if (resolve_break(klist, pos, rec, exptab, lane, fxa, fya)) mk |= 1ull << a;
                                    ++resolved;
                                } while (amb != 0);
                            }
                        }
                    }
                } else {
                    mk = m2 & __builtin_amdgcn_ballot_w64(test_T < 0.0001f);
                }
                STAT_ADD(8, __popcll(m2 & ~mk));   // (pixel, entry) pairs that contribute (alpha >= 1/255, before the break)
                // the reference's loop walks a pixel's list up to and including the entry it breaks at (render.comp:60-85)
                STAT_ADD(7, (unsigned long long)__popcll(mk) * ((base - range.x) + (uint32_t)k + 1u));
#if GS_BLEND_ASM_ACCUMULATE
                if (GUARD) {
                    // :87 c += color * alpha * T; T = T (1 - alpha) for the lanes that are kept and do not break -- under the exec
                    // mask, as ONE block of seven instructions: written by the compiler the same region costs two more branches (a
                    // skip for "no lane updates", which never pays here, and the jump back from the out-of-line block it places
                    // the region in).  The guarded mode owes the reference its DECISIONS bit for bit (they hang on `power` and on
                    // the T chain, both evaluated as written) and its pixels to rounding noise: the weight alpha * T is formed once
                    // and each channel takes one fused multiply-add (every contribution within an ULP of the reference's, no
                    // cancellation anywhere in this sum)
                    const uint64_t updm = m2 & ~mk;
                    uint64_t saved;
                    float wgt;
                    asm volatile("s_and_saveexec_b64 %[sv], %[um]\n\t"
                                 "v_mul_f32 %[w], %[a], %[T]\n\t"
                                 "v_fmac_f32 %[c0], %[r], %[w]\n\t"
                                 "v_fmac_f32 %[c1], %[g], %[w]\n\t"
                                 "v_fmac_f32 %[c2], %[b], %[w]\n\t"
                                 "v_mov_b32 %[T], %[tt]\n\t"
                                 "s_or_b64 exec, exec, %[sv]"
                                 : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [T] "+v"(T), [w] "=&v"(wgt), [sv] "=&s"(saved)
                                 : [um] "s"(updm), [a] "v"(alpha), [r] "v"(uv.z), [g] "v"(uv.w), [b] "v"(bp.x), [tt] "v"(test_T)
                                 : "scc");
                } else {
#endif
                const bool upd = __builtin_amdgcn_inverse_ballot_w64(m2 & ~mk);
                if (upd) {  // the accumulate runs under the exec mask: no selects
                    if (GUARD) {  // (the compiler's form of the block above)
                        const float wgt = alpha * T;
                        c0 = __builtin_fmaf(uv.z, wgt, c0);
                        c1 = __builtin_fmaf(uv.w, wgt, c1);
                        c2 = __builtin_fmaf(bp.x, wgt, c2);
                    } else if (CONTRACT) {
                        c0 = __builtin_fmaf(uv.z * alpha, T, c0);  // :87  FMA
                        c1 = __builtin_fmaf(uv.w * alpha, T, c1);
                        c2 = __builtin_fmaf(bp.x * alpha, T, c2);
                    } else {  // c += color * alpha * T
                        c0 = c0 + uv.z * alpha * T;
                        c1 = c1 + uv.w * alpha * T;
                        c2 = c2 + bp.x * alpha * T;
                    }
                    T = test_T;
                }
#if GS_BLEND_ASM_ACCUMULATE
                }
#endif
#if GS_BLEND_SALU_DIET == 2
                // alive &= ~mk; if (alive == 0) bm = 0 -- the pair loop then ends at its own test of bm.  Three scalar
                // instructions, written out: left to itself the compiler spends six on "did the last pixel just saturate" (it
                // materialises the condition as a lane mask before branching on it), and the CU's one scalar unit is a
                // co-bottleneck of this loop (0.64 scalar instructions per vector one)
                asm volatile("s_andn2_b64 %0, %0, %2\n\ts_cmp_eq_u64 %0, 0\n\ts_cselect_b64 %1, 0, %1" : "+s"(alive), "+s"(bm) : "s"(mk) : "scc");
#elif GS_BLEND_SALU_DIET
                // alive &= ~mk, kept opaque: left to itself the compiler turns "did the last pixel just saturate" into
                // seven scalar instructions of boolean materialisation
                asm volatile("s_andn2_b64 %0, %0, %1" : "+s"(alive) : "s"(mk) : "scc");
                if (alive == 0) break;  // every pixel of the quadrant has saturated (the outer loop ends below)
#else
                alive &= ~mk;
                if (alive == 0) bm = 0;
#endif
            }
        }
        }  // (the compiler's form of the pair loop)
        if (alive == 0) break;
        if (GUARD) {  // chunk end: sum a/(1 - a) over the chunk's steps <= T_start / T_end - 1 (1e-5: v_rcp_f32's ULP)
            S += __builtin_fmaf(T_cs * 1.00001f, __builtin_amdgcn_rcpf(T), -1.0f);
            T_cs = T;
        }
    }
    STAT_ADD(7, (unsigned long long)__popcll(alive) * (range.y - range.x));  // pixels that never broke walk the whole list
    return !abandon;
}

// EXP / CONTRACT: see blend_walk.  GUARD (with a fast exp, uncontracted): break decisions inside the guard's window are resolved
// exactly, pixel by pixel; a quadrant the fast pass abandons is re-rendered at once, by the same wave, with EXP = 2.
template <int EXP, bool CONTRACT, bool GUARD>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_blend(const uint2* __restrict__ ranges,
                                                 const uint32_t* __restrict__ sorted_gid,
                                                 const uint32_t* __restrict__ tile_order,
                                                 const AttrRecord* __restrict__ rec,
                                                 uint32_t width, uint32_t height, uint32_t tiles_x,
                                                 float4* __restrict__ rgba, uchar4* __restrict__ bgra,
                                                 Counters* __restrict__ counters, Counters* host_counters,
                                                 const FrameParams* __restrict__ fp, uint32_t* __restrict__ vis_count, uint32_t lockstep,
                                                 uint64_t* __restrict__ stamps) {
    frame_stamp(stamps, ST_BLEND);
#if defined(GS_BLEND_STATS) || defined(GS_BLEND_CLOCK)
    const long long stat_c0 = clock64();                 // s_memtime: shader cycles
    const unsigned long long stat_w0 = wall_clock64();   // s_memrealtime: the constant-rate clock
#endif
    static_assert(!GUARD || (EXP != 2 && !CONTRACT), "the guard belongs to a fast exp on the uncontracted arithmetic");
    if (fp) {  // graph replay: this frame's targets come from the parameter block
        rgba = reinterpret_cast<float4*>(fp->rgba);
        bgra = reinterpret_cast<uchar4*>(fp->bgra);
        host_counters = fp->host_counters;
    }
    // wave-private slabs (no cross-wave sharing; the only workgroup barrier is the lockstep one at the top of blend_walk's chunk loop)
    __shared__ float4 s_rec[4][3][WAVE];
    __shared__ uint2 s_exptab[(EXP == 2 || GUARD) ? 4 : 1][32];  // wave-private copies of kExpfTab: filled and read by their own wave, wave_barrier only
    __shared__ uint32_t s_klist[GUARD ? 4 : 1][GUARD ? kGuardList : 1];  // GUARD: each wave's list of kept entries (resolve_break)

    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    if (EXP == 2) {
        if (lane < 32) {
            const uint64_t v = kExpfTab[lane];
            s_exptab[w][lane] = make_uint2((uint32_t)v, (uint32_t)(v >> 32));
        }
        __builtin_amdgcn_wave_barrier();
    }
    // last kernel of the frame: hand V, D, E1 and the overflow flag to the host (pinned memory; visible to it once
    // the frame's completion event, which carries the system-scope release, has fired) -- no copy node in the stream
    if (host_counters && blockIdx.x == 0 && tid == 0) *host_counters = *counters;
    // ... and k_preprocess of the next frame on these buffers appends to the dense lists of visible Gaussians from zero again
    // (the counts of THIS frame stay readable in the word behind each counter: the stage taps rebuild the per-Gaussian planes from the lists)
    if (vis_count && blockIdx.x == 0 && (uint32_t)tid < kVisRegions) {
        vis_count[(uint32_t)tid * kVisCounterStride + 1] = vis_count[(uint32_t)tid * kVisCounterStride];
        vis_count[(uint32_t)tid * kVisCounterStride] = 0;
    }
    // XCD-aware, load-balanced tile order: a host-built table (gs_renderer.cpp, ensure_tile_order)
    const uint32_t tile = tile_order[blockIdx.x];
    const uint32_t tile_x = tile % tiles_x, tile_y = tile / tiles_x;
    const uint32_t qx0 = tile_x * kTile + (w & 1) * 8, qy0 = tile_y * kTile + (w >> 1) * 8;
    const uint32_t px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool inside = px < width && py < height;  // render.comp:36-39
    const float fx = (float)px, fy = (float)py;
    const float rx0 = (float)qx0, ry0 = (float)qy0;

    const uint2 range = ranges[tile];
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    // Pixel predicates live in 64-bit scalar masks (one bit per lane): combining them is scalar-unit work
    // and testing "any lane" is one s_cmp, where bool-typed code would spend VALU instructions on it.
    const uint64_t alive = __builtin_amdgcn_ballot_w64(inside);  // pixels still accumulating

    if (alive != 0 && range.x < range.y) {
        uint32_t resolved = 0, unused;
        bool table_ready = EXP == 2;  // GUARD: the wave copies libm's table into LDS when it first needs it
        const bool done = blend_walk<EXP, CONTRACT, GUARD>(range, sorted_gid, rec, s_rec[w], s_exptab[(EXP == 2 || GUARD) ? w : 0],
                                                           s_klist[GUARD ? w : 0], lane, fx, fy, rx0, ry0, alive, c0, c1, c2, resolved,
                                                           table_ready, lockstep != 0);
        if (GUARD && resolved != 0 && lane == 0) atomicAdd(&counters->blend_resolved, resolved);
        if (GUARD && !done) {  // wave-uniform: the whole quadrant again, with the reference's arithmetic
            if (!table_ready) {
                if (lane < 32) {
                    const uint64_t v = kExpfTab[lane];
                    s_exptab[w][lane] = make_uint2((uint32_t)v, (uint32_t)(v >> 32));
                }
                __builtin_amdgcn_wave_barrier();
                table_ready = true;
            }
            if (lane == 0) atomicAdd(&counters->blend_redo, 1u);
            // (alone: the other waves of the tile are in their own walks, or done.  lockstep = false HERE, always: this wave has left the
            // lockstepped loop, its siblings may still be in it -- a barrier in this walk would deadlock the tile)
            (void)blend_walk<2, false, false>(range, sorted_gid, rec, s_rec[w], s_exptab[w], nullptr, lane, fx, fy, rx0, ry0, alive, c0, c1, c2,
                                              unused, table_ready, false);
        }
    }
#if defined(GS_BLEND_STATS) || defined(GS_BLEND_CLOCK)
    {   // the shader clock this kernel actually ran at: cycles and constant-rate ticks per wave, summed (tools/blend_stats.py divides)
        const uint32_t slot = blockIdx.x * (BLOCK / WAVE) + threadIdx.x / WAVE;
        if ((threadIdx.x & (WAVE - 1)) == 0 && slot < kClockSlots)
            g_blend_clock[slot] = make_uint2((uint32_t)(clock64() - stat_c0), (uint32_t)(wall_clock64() - stat_w0));
    }
#endif
    // (no __syncthreads / s_barrier from here to the end: see the header)
    if (inside) {
        const size_t p = (size_t)py * width + px;
        if (rgba) rgba[p] = make_float4(c0, c1, c2, 1.0f);  // :98
        if (bgra) {
            // imageStore to B8G8R8A8_UNORM: clamp to [0,1], round to nearest
            const float r = fminf(fmaxf(c0, 0.0f), 1.0f), g = fminf(fmaxf(c1, 0.0f), 1.0f),
                        b = fminf(fmaxf(c2, 0.0f), 1.0f);
            bgra[p] = make_uchar4((unsigned char)(int)__builtin_rintf(b * 255.0f),
                                  (unsigned char)(int)__builtin_rintf(g * 255.0f),
                                  (unsigned char)(int)__builtin_rintf(r * 255.0f), 255);
        }
    }
}

template <int EXP, bool CONTRACT, bool GUARD>
static void launch_blend_as(const uint32_t* ranges, const uint32_t* sorted_gid, const uint32_t* tile_order, const AttrView& av,
                            uint32_t width, uint32_t height, uint32_t tx, uint32_t ty, float* rgba, uint8_t* bgra,
                            Counters* counters, Counters* host_counters, const FrameParams* fp, bool lockstep, uint64_t* stamps, hipStream_t s) {
    hipLaunchKernelGGL((k_blend<EXP, CONTRACT, GUARD>), dim3(tx * ty), dim3(BLOCK), 0, s, reinterpret_cast<const uint2*>(ranges),
                       sorted_gid, tile_order, av.rec, width, height, tx, reinterpret_cast<float4*>(rgba),
                       reinterpret_cast<uchar4*>(bgra), counters, host_counters, fp, av.vis_count, lockstep ? 1u : 0u, stamps);
}

// The frame's end on its timeline: one wave behind the blend (a dependent dispatch starts when the last workgroup of the kernel before it
// has retired) stamps the clock and hands the stamps to the host -- pinned memory, visible to it once the frame's completion event has fired.
__global__ __launch_bounds__(WAVE) void k_frame_end(uint64_t* __restrict__ stamps, uint64_t* host_stamps, const FrameParams* __restrict__ fp) {
    if (fp) host_stamps = fp->host_stamps;
    const uint64_t now = wall_clock64();
    const int t = threadIdx.x;
    if (t < ST_COUNT) {
        const uint64_t v = t == ST_END ? now : stamps[t];
        if (t == ST_END) stamps[t] = now;
        if (host_stamps) host_stamps[t] = v;
    }
}
void launch_frame_end(uint64_t* stamps, uint64_t* host_stamps, const FrameParams* fp, hipStream_t s) {
    if (!stamps) return;
    hipLaunchKernelGGL(k_frame_end, dim3(1), dim3(WAVE), 0, s, stamps, host_stamps, fp);
}

void launch_blend(const uint32_t* ranges, const uint32_t* sorted_gid, const uint32_t* tile_order, const AttrView& av,
                  uint32_t width,
                  uint32_t height, float* rgba, uint8_t* bgra, Counters* counters,
                  Counters* host_counters, int exp_mode, bool contract, const FrameParams* fp, bool lockstep, uint64_t* stamps, hipStream_t s) {
    if (width == 0 || height == 0) return;
    const uint32_t tx = (width + kTile - 1) / kTile, ty = (height + kTile - 1) / kTile;
    if (exp_mode == 3 && !contract)  // the guarded hardware exp; with the contractions on there is nothing to guard: mode 1
        return launch_blend_as<1, false, true>(ranges, sorted_gid, tile_order, av, width, height, tx, ty, rgba, bgra, counters,
                                               host_counters, fp, lockstep, stamps, s);
    if (exp_mode == 3) exp_mode = 1;
#define GS_BLEND_CASE(E, C)                                                                                              \
    if (exp_mode == E && contract == C)                                                                                  \
        return launch_blend_as<E, C, false>(ranges, sorted_gid, tile_order, av, width, height, tx, ty, rgba, bgra,       \
                                            counters, host_counters, fp, lockstep, stamps, s)
    GS_BLEND_CASE(0, true);
    GS_BLEND_CASE(0, false);
    GS_BLEND_CASE(1, true);
    GS_BLEND_CASE(1, false);
    GS_BLEND_CASE(2, true);
    GS_BLEND_CASE(2, false);
#undef GS_BLEND_CASE
}


// ---- test hook: the blend's exp() implementations evaluated ON THE DEVICE over ranges of binary32 bit patterns -------------
// (tests/test_gpu_expf.py: gs_expf_libm against the host's libm on every binary32 <= 0, and the constants of the guard)
__global__ __launch_bounds__(BLOCK) void k_expf_scan(uint32_t first_bits, uint64_t count, unsigned long long* __restrict__ block_sums,
                                                     unsigned long long* __restrict__ guard_max /* [2] bits of non-negative doubles */) {
    // block b of the grid covers 2^20 consecutive patterns: one 64-bit checksum per block
    const uint64_t base = (uint64_t)blockIdx.x << 20;
    unsigned long long sum = 0;
    double worst = 0.0, worst_near = 0.0;
    for (uint32_t j = threadIdx.x; j < (1u << 20); j += BLOCK) {
        const uint64_t at = base + j;
        if (at >= count) break;
        const uint32_t xb = first_bits + (uint32_t)at;
        const float x = __uint_as_float(xb);
        const float y = gs_expf_libm_full(x, reinterpret_cast<const uint2*>(kExpfTab));
        sum += (unsigned long long)__float_as_uint(y) * (unsigned long long)((xb * 0x9E3779B1u) | 1u);
        if (x >= -16.0f && x <= 0.0f) {  // the guard's premise: |v_exp_f32(fl(x log2e)) - expf(x)| <= (E0e + kGuardE1 |x|) expf(x)
            const float fast = __builtin_amdgcn_exp2f(x * 1.44269502162933349609375f);
            const double rel = fabs((double)fast - (double)y) / (double)y;
            worst = fmax(worst, rel - (double)kGuardE1 * fabs((double)x));
            if (x >= -1.0f) worst_near = fmax(worst_near, rel);
        }
    }
    // wave reduction, then one atomic per wave
    for (int d = 32; d >= 1; d >>= 1) {
        sum += __shfl_xor(sum, d, WAVE);
        worst = fmax(worst, __shfl_xor(worst, d, WAVE));
        worst_near = fmax(worst_near, __shfl_xor(worst_near, d, WAVE));
    }
    if ((threadIdx.x & (WAVE - 1)) == 0) {
        atomicAdd(&block_sums[blockIdx.x], sum);
        atomicMax(&guard_max[0], (unsigned long long)__double_as_longlong(worst));       // non-negative doubles order like their bits
        atomicMax(&guard_max[1], (unsigned long long)__double_as_longlong(worst_near));
    }
}

extern "C" int gs_debug_expf_scan(int device, uint32_t first_bits, uint64_t count, uint64_t* block_sums, uint64_t blocks_capacity,
                                  double* guard /* nullable [4] */) {
    const uint64_t blocks = (count + (1u << 20) - 1) >> 20;
    if (blocks == 0 || blocks > blocks_capacity || !block_sums || blocks > 0x7FFFFFFFull) return GS_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return GS_ERR_DEVICE;
    unsigned long long* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), (blocks + 2) * sizeof(unsigned long long)) != hipSuccess) return GS_ERR_NOMEM;
    int rc = GS_OK;
    if (hipMemset(d, 0, (blocks + 2) * sizeof(unsigned long long)) != hipSuccess) rc = GS_ERR_DEVICE;
    if (rc == GS_OK) {
        hipLaunchKernelGGL(k_expf_scan, dim3((uint32_t)blocks), dim3(BLOCK), 0, nullptr, first_bits, count, d, d + blocks);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = GS_ERR_DEVICE;
    }
    unsigned long long g[2] = {0, 0};
    if (rc == GS_OK && (hipMemcpy(block_sums, d, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess ||
                        hipMemcpy(g, d + blocks, sizeof g, hipMemcpyDeviceToHost) != hipSuccess))
        rc = GS_ERR_DEVICE;
    (void)hipFree(d);
    if (rc == GS_OK && guard) {
        guard[0] = __builtin_bit_cast(double, g[0]);  // max over x in [-16, 0] of  rel(x) - kGuardE1 |x|   (must stay below kGuardE0 - 2^-23)
        guard[1] = __builtin_bit_cast(double, g[1]);  // max rel(x) over x in [-1, 0]
        guard[2] = (double)kGuardE0;
        guard[3] = (double)kGuardE1;
    }
    return rc;
}

#if defined(GS_BLEND_STATS) || defined(GS_BLEND_CLOCK)
extern "C" int gs_debug_blend_stats(unsigned long long* out, int reset) {
    unsigned long long z[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_blend_stats), sizeof z) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_blend_stats), z, sizeof z) != hipSuccess) return -1;
    return 0;
}
// the last launch's per-wave (cycles, ticks) of the first `waves` waves, summed: out[0] cycles, out[1] ticks
extern "C" int gs_debug_blend_clock(unsigned long long* out, unsigned waves) {
    if (waves > kClockSlots) waves = kClockSlots;
    uint2* h = new uint2[waves];
    const hipError_t e = hipMemcpyFromSymbol(h, HIP_SYMBOL(g_blend_clock), sizeof(uint2) * waves);
    out[0] = out[1] = 0;
    for (unsigned i = 0; e == hipSuccess && i < waves; ++i) {
        out[0] += h[i].x;
        out[1] += h[i].y;
    }
    delete[] h;
    return e == hipSuccess ? 0 : -1;
}
#endif

}  // namespace gs
