// gs_blend.hip -- k_blend: per-pixel front-to-back alpha blend, one wave per 8x8 pixel quadrant.
//
// Part of libgs3d_hip.so (gfx950 only).  Built with -ffp-contract=off: the floating-point contract of this path is "IEEE
// binary32, one rounding per operation, in the order the reference shader writes it" (DESIGN.md section 3); fused
// multiply-adds appear only where written explicitly.
// Reference restated (paths relative to /root/reference/src/shaders): render.comp:30-99
#include "gs_device.h"

namespace gs {

#ifndef GS_BLEND_SALU_DIET
#define GS_BLEND_SALU_DIET 1
#endif

// ---------------------------------------------------------------------------------------
// blend.  One wave per 8x8 pixel quadrant of a 16x16 tile (4 waves = one workgroup per tile), and the
// four waves are fully independent: no workgroup barrier anywhere, so a quadrant whose pixels have
// saturated retires at once and a slow quadrant never stalls its neighbours.
//
// Each wave walks its tile's depth-sorted list in chunks of 64 entries: lane l fetches entry l's record
// (9 floats, gathered through the sorted Gaussian id; the next chunk is prefetched while the current one
// is blended), tests it against the wave's quadrant, parks it in a wave-private LDS slab, and a 64-bit
// ballot of the survivors drives a scalar loop that evaluates only those entries, in list order, with
// broadcast LDS reads.  The per-pixel body is predicated (selects) instead of branched: nested divergent
// branches cost ~40 scalar exec-mask instructions per entry and saturate the CU's scalar unit.
//
// render.comp:61-98 semantics.  Floating-point contract: the shader's expressions with the three
// multiply-adds that GLSL lets a compiler contract written as explicit FMAs (marked FMA below; the
// oracle makes the same choice), nothing reassociated.
//
// Exactness of the culling: an entry contributes to a pixel only if alpha = min(0.99, o*exp(power))
// >= 1/255, i.e. power >= -tau with tau = ln(255*o), and -power = q(d) = 0.5 d^T C d (C = conic) is
// a convex quadratic of d = uv - pixel.  If the minimum of q over the quadrant's pixel rectangle exceeds
// tau (with 0.1 % + 1e-3 slack against the rounding of exp/log, plus 8 ULP of the quadratic's largest TERMS over the
// quadrant against the cancellation error of `power` for thin diagonal splats), every pixel of the
// quadrant executes `continue` in the shader, so skipping the entry for that wave changes nothing.
// The same bound gives a per-entry lower limit on power below which exp() need not be evaluated.
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
    // dt/t* ~ 2^-23, i.e. ~1e-14 relative, which the caller's slack (4.8e-7 x the quadratic's largest terms + 0.1 %) absorbs
    // many times over.  A coarser reciprocal or a smaller slack must revisit this.
    return in_x ? (in_y ? 0.0f : qh) : (in_y ? qv : fminf(qv, qh));
}

#ifdef GS_BLEND_STATS
// debug instrumentation (separate build, never the shipped library)
__device__ unsigned long long g_blend_stats[12];
#define STAT_ADD(i, v) do { const unsigned long long v_ = (unsigned long long)(v); const bool first_ = (__ffsll((unsigned long long)__ballot(true)) - 1) == lane; if (first_) atomicAdd(&g_blend_stats[i], v_); } while (0)
#else
#define STAT_ADD(i, v) do { } while (0)
#endif

// gs_exp without the lower clamp: every lane whose result is used has power in [-7, 0].
__device__ __forceinline__ float gs_exp_blend(float x) {
    const float L2E = 1.44269502162933349609375f;
    const float MAGIC = 12582912.0f;
    float tm = __builtin_fmaf(x, L2E, MAGIC);
    float n = tm - MAGIC;
    float f = __builtin_fmaf(x, L2E, -n);
    float p = 0x1.41d332p-13f;
    p = __builtin_fmaf(p, f, 0x1.5f456ap-10f);
    p = __builtin_fmaf(p, f, 0x1.3b2dbcp-7f);
    p = __builtin_fmaf(p, f, 0x1.c6aed4p-5f);
    p = __builtin_fmaf(p, f, 0x1.ebfbdap-3f);
    p = __builtin_fmaf(p, f, 0x1.62e430p-1f);
    p = __builtin_fmaf(p, f, 1.0f);
    return __uint_as_float(__float_as_uint(p) + (__float_as_uint(tm) << 23));
}

// exp() as glibc's expf evaluates it (glibc >= 2.27, sysdeps/ieee754/flt-32/e_expf.c = ARM optimized-routines expf: x 32/ln2 split
// into k + r in binary64, 2^(k/32) from a 32-entry table, a cubic in r, ONE rounding to binary32 at the end), operation by
// operation in binary64 with the fused operations of the x86-64 FMA build -- so that the blend can be bit-identical to the
// reference's shader text compiled for the CPU (the test suite's checker), whose exp() is libm's.  Restated from the published algorithm,
// the table generated (2^(i/32) correctly rounded, exponent pre-subtracted), and PINNED by tests/test_expf_libm.py: equal to
// this container's libm expf on every binary32 <= 0 (2.1e9 values).  9 binary64 operations (half rate on gfx950) + one
// LDS read: ~23 issue slots against 10 for the polynomial.  Valid for the blend's range (x <= 0, results used for x >= -7).
__device__ const uint64_t kExpfTab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull,
    0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull,
    0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull,
    0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull,
    0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
__device__ __forceinline__ float gs_expf_libm(float x, const uint2* __restrict__ tab /* LDS copy of kExpfTab */) {
    const double InvLn2N = 0x1.71547652b82fep+0 * 32.0, SHIFT = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0, C1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0, C2 = 0x1.62e42ff0c52d6p-1 / 32.0;
    const double xd = (double)x;
    double kd = __builtin_fma(InvLn2N, xd, SHIFT);          // k = round(x 32/ln2) in the low mantissa bits
    const uint32_t ki = (uint32_t)__double_as_longlong(kd);
    kd = kd - SHIFT;
    const double r = __builtin_fma(InvLn2N, xd, -kd);
    uint2 t = tab[ki & 31u];
    t.y += ki << 15;                                        // t += ki << 47: the exponent of 2^(k/32)
    const double sc = __longlong_as_double((long long)(((uint64_t)t.y << 32) | t.x));
    // glibc evaluates  z = C0 r + C1;  y = C2 r + 1;  y = z r^2 + y;  y = y s  (five operations).  Here the same cubic times the
    // same s in four:  q = (C0 r + C1) r + C2;  y = q (r s) + s.  The two differ in the last bits of the binary64 value, never
    // in its rounding to binary32: tests/test_expf_libm.py runs this very sequence against libm's expf on every binary32 <= 0
    // (IEEE binary64 operations give the same bits on the host as on the device).  q's first fma is written as one VOP3
    // v_fma_f64 (left to itself the compiler copies C1 and uses the two-address v_fmac_f64).
    double q0, q;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(q0) : "s"(C0), "v"(r), "v"(C1));
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(q) : "v"(q0), "v"(r), "v"(C2));
    const double rs = r * sc;
    const double y = __builtin_fma(q, rs, sc);
    return (float)y;
}

struct BlendEntry {
    float4 co;  // c00 c01 c11 opacity
    float4 uv;  // u v r g
    float b;
};

__device__ __forceinline__ void blend_fetch(BlendEntry& e, uint32_t g, const AttrRecord* __restrict__ rec) {
    const AttrRecord* r = rec + g;  // one 64-byte line
    e.co = r->conic_op;
    e.uv = r->uv_rg;
    e.b = r->b_depth_r.x;
}

// CONTRACT: the pipeline's three contractions of render.comp:66,87 (default) or the uncontracted reading, one rounding per
// operation exactly as the shader is written -- what the reference's text compiled for the CPU evaluates (gs_set_blend_contraction).
// EXP: 0 the pipeline-defined polynomial (gs_exp_blend), 1 the hardware's v_exp_f32, 2 libm's expf restated (gs_expf_libm).
template <int EXP, bool CONTRACT>
__global__ __launch_bounds__(BLOCK) void k_blend(const uint2* __restrict__ ranges,
                                                 const uint32_t* __restrict__ sorted_gid,
                                                 const uint32_t* __restrict__ tile_order,
                                                 const AttrRecord* __restrict__ rec,
                                                 uint32_t width, uint32_t height, uint32_t tiles_x,
                                                 float4* __restrict__ rgba, uchar4* __restrict__ bgra,
                                                 const Counters* __restrict__ counters, Counters* host_counters,
                                                 const FrameParams* __restrict__ fp, uint32_t* __restrict__ vis_count) {
    if (fp) {  // graph replay: this frame's targets come from the parameter block
        rgba = reinterpret_cast<float4*>(fp->rgba);
        bgra = reinterpret_cast<uchar4*>(fp->bgra);
        host_counters = fp->host_counters;
    }
    // wave-private slabs (no cross-wave sharing, no barriers), three planes of 64 float4 per wave: {c00 c01 c11 o} {u v r g} {b, pmin, -, -}.  Plane-major keeps the staging
    // ds_write_b128 conflict-free (lane stride 16 B); one scalar-derived address + constant offsets serve the reads
    __shared__ float4 s_rec[4][3][WAVE];
    __shared__ uint2 s_exptab[EXP == 2 ? 4 : 1][32];  // wave-private copies of kExpfTab (no workgroup barrier in this kernel)

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
    if (vis_count && blockIdx.x == 0 && (uint32_t)tid < kVisRegions) vis_count[(uint32_t)tid * kVisCounterStride] = 0;
    // XCD-aware, load-balanced tile order: a host-built table (gs_capi.cpp, ensure_tile_order)
    const uint32_t tile = tile_order[blockIdx.x];
    const uint32_t tile_x = tile % tiles_x, tile_y = tile / tiles_x;
    const uint32_t qx0 = tile_x * kTile + (w & 1) * 8, qy0 = tile_y * kTile + (w >> 1) * 8;
    const uint32_t px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool inside = px < width && py < height;  // render.comp:36-39
    const float fx = (float)px, fy = (float)py;
    const float rx0 = (float)qx0, ry0 = (float)qy0;

    const uint2 range = ranges[tile];
    float T = 1.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    // Pixel predicates live in 64-bit scalar masks (one bit per lane): combining them is scalar-unit work
    // and testing "any lane" is one s_cmp, where bool-typed code would spend VALU instructions on it.
    uint64_t alive = __builtin_amdgcn_ballot_w64(inside);  // pixels still accumulating

    if (alive != 0 && range.x < range.y) {
        // software pipeline over 64-entry chunks: ids two chunks ahead, records one chunk ahead
        BlendEntry nxt;
        nxt.co = make_float4(0, 0, 0, 0);
        nxt.uv = make_float4(0, 0, 0, 0);
        nxt.b = 0;
        uint32_t g_next = 0;
        {
            const uint32_t i0 = range.x + lane;
            if (i0 < range.y) blend_fetch(nxt, sorted_gid[i0], rec);
            const uint32_t i1 = i0 + WAVE;
            if (i1 < range.y) g_next = sorted_gid[i1];
        }
        for (uint32_t base = range.x; base < range.y; base += WAVE) {
            const BlendEntry cur = nxt;
            const bool have = base + lane < range.y;
            {   // prefetch: records of chunk +1 (ids already here), ids of chunk +2
                const uint32_t i1 = base + WAVE + lane;
                if (i1 < range.y) blend_fetch(nxt, g_next, rec);
                const uint32_t i2 = i1 + WAVE;
                if (i2 < range.y) g_next = sorted_gid[i2];
            }
            // classify entry `lane` of this chunk against the wave's quadrant
            const float tau = __logf(255.0f * cur.co.w);
            // NaN opacity: min(0.99, NaN) is 0.99 in the pipeline's definition -> the entry is never culled
            const float lim = tau == tau ? fmaxf(tau, 0.0f) * 1.001f + 1e-3f : 3.0e38f;
            bool keep = have && !(tau <= -1e-3f);  // tau <= 0: o*exp(p) < 1/255 for every p <= 0
            if (keep) {
                const float mq = min_q_rect(cur.co.x, cur.co.y, cur.co.z, cur.uv.x, cur.uv.y, rx0, rx0 + 7.0f,
                                            ry0, ry0 + 7.0f);
                // The rounding error of the shader's `power` (and of mq) is relative to the TERMS c00 dx^2, c11 dy^2,
                // c01 dx dy, not to their sum: a thin diagonal splat far from its centre has terms ~1e5 cancelling to
                // q ~ 5.  The slack therefore grows with the terms at the quadrant's corner farthest from the centre
                // (8 roundings of 2^-24 each, generously).
                const float ax = fmaxf(fabsf(cur.uv.x - rx0), fabsf(cur.uv.x - (rx0 + 7.0f)));
                const float ay = fmaxf(fabsf(cur.uv.y - ry0), fabsf(cur.uv.y - (ry0 + 7.0f)));
                const float mag = __builtin_fmaf(0.5f * fabsf(cur.co.x) * ax, ax,
                                                 __builtin_fmaf(0.5f * fabsf(cur.co.z) * ay, ay, fabsf(cur.co.y) * ax * ay));
                keep = !(mq > __builtin_fmaf(mag, 4.8e-7f, lim));  // NaN -> keep
            }
            uint64_t bm = __ballot(keep);
            STAT_ADD(0, 1);
            STAT_ADD(6, __popcll(__ballot(have)));
            STAT_ADD(1, __popcll(bm));
            if (bm == 0) continue;
            // conic pre-scaled once per entry: (-c00/2, -c01, -c11/2).  Scaling by a power of two commutes with every
            // rounding below, so power is bit-identical to render.comp:66 evaluated as written (with its three
            // contractions) while the per-pixel body loses the -0.5 multiply
            s_rec[w][0][lane] = make_float4(-0.5f * cur.co.x, -cur.co.y, -0.5f * cur.co.z, cur.co.w);
            s_rec[w][1][lane] = cur.uv;
            s_rec[w][2][lane] = make_float4(cur.b, -lim, 0.0f, 0.0f);

            while (bm) {
                const int k = __ffsll((unsigned long long)bm) - 1;
#if GS_BLEND_SALU_DIET
                asm("s_bitset0_b64 %0, %1" : "+s"(bm) : "s"(k));  // bm &= bm - 1 costs three scalar instructions
#else
                bm &= bm - 1;
#endif
                STAT_ADD(2, 1);                       // (entry, wave) pairs evaluated
                STAT_ADD(3, __popcll(alive));         // lanes alive
                float4 co = s_rec[w][0][k];
                float4 uv = s_rec[w][1][k];
                float4 bp = s_rec[w][2][k];
                // all ten floats in one LDS round trip: without this the compiler sinks the loads of o, r, g, b
                // behind the exp() branch, where 94 % of the pairs then pay a second LDS latency
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
                // power <= 0 is false for NaN: a NaN power skips the entry (the pipeline's definition)
                const uint64_t m1 = alive & __builtin_amdgcn_ballot_w64(power <= 0.0f) &
                                    __builtin_amdgcn_ballot_w64(!(power < bp.y));
                if (m1 != 0) {
                    STAT_ADD(4, 1);                   // pairs reaching exp
                    STAT_ADD(5, __popcll(m1));        // lanes needing exp
                    // :77.  EXP 1: the hardware's v_exp_f32 (what a Vulkan driver emits for exp()); 2: libm's expf, what the
                    // reference's text compiled for the CPU calls; 0: the pipeline-defined polynomial.  The oracle reproduces
                    // 0 and 2 bit for bit
                    const float ex = EXP == 1   ? __builtin_amdgcn_exp2f(power * 1.44269502162933349609375f)
                                     : EXP == 2 ? gs_expf_libm(power, s_exptab[EXP == 2 ? w : 0])
                                                : gs_exp_blend(power);
                    const float alpha = fminf(0.99f, co.w * ex);
                    const uint64_t m2 = m1 & __builtin_amdgcn_ballot_w64(!(alpha < 1.0f / 255.0f));
                    const float test_T = T * (1 - alpha);
                    const uint64_t mk = m2 & __builtin_amdgcn_ballot_w64(test_T < 0.0001f);  // :82-85 break
                    STAT_ADD(8, __popcll(m2 & ~mk));   // (pixel, entry) pairs that contribute (alpha >= 1/255, before the break)
                    // the reference's loop walks a pixel's list up to and including the entry it breaks at (render.comp:60-85)
                    STAT_ADD(7, (unsigned long long)__popcll(mk) * ((base - range.x) + (uint32_t)k + 1u));
                    const bool upd = __builtin_amdgcn_inverse_ballot_w64(m2 & ~mk);
                    if (upd) {  // the accumulate runs under the exec mask: no selects
                        if (CONTRACT) {
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
#if GS_BLEND_SALU_DIET
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
            if (alive == 0) break;
        }
    }
    STAT_ADD(7, (unsigned long long)__popcll(alive) * (range.y - range.x));  // pixels that never broke walk the whole list
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

template <int EXP, bool CONTRACT>
static void launch_blend_as(const uint32_t* ranges, const uint32_t* sorted_gid, const uint32_t* tile_order, const AttrView& av,
                            uint32_t width, uint32_t height, uint32_t tx, uint32_t ty, float* rgba, uint8_t* bgra,
                            const Counters* counters, Counters* host_counters, const FrameParams* fp, hipStream_t s) {
    hipLaunchKernelGGL((k_blend<EXP, CONTRACT>), dim3(tx * ty), dim3(BLOCK), 0, s, reinterpret_cast<const uint2*>(ranges),
                       sorted_gid, tile_order, av.rec, width, height, tx, reinterpret_cast<float4*>(rgba),
                       reinterpret_cast<uchar4*>(bgra), counters, host_counters, fp, av.vis_count);
}

void launch_blend(const uint32_t* ranges, const uint32_t* sorted_gid, const uint32_t* tile_order, const AttrView& av,
                  uint32_t width,
                  uint32_t height, float* rgba, uint8_t* bgra, const Counters* counters,
                  Counters* host_counters, int exp_mode, bool contract, const FrameParams* fp, hipStream_t s) {
    if (width == 0 || height == 0) return;
    const uint32_t tx = (width + kTile - 1) / kTile, ty = (height + kTile - 1) / kTile;
#define GS_BLEND_CASE(E, C)                                                                                              \
    if (exp_mode == E && contract == C)                                                                                  \
        return launch_blend_as<E, C>(ranges, sorted_gid, tile_order, av, width, height, tx, ty, rgba, bgra, counters,    \
                                     host_counters, fp, s)
    GS_BLEND_CASE(0, true);
    GS_BLEND_CASE(0, false);
    GS_BLEND_CASE(1, true);
    GS_BLEND_CASE(1, false);
    GS_BLEND_CASE(2, true);
    GS_BLEND_CASE(2, false);
#undef GS_BLEND_CASE
}

#ifdef GS_BLEND_STATS
extern "C" int gs_debug_blend_stats(unsigned long long* out, int reset) {
    unsigned long long z[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_blend_stats), sizeof z) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_blend_stats), z, sizeof z) != hipSuccess) return -1;
    return 0;
}
#endif

}  // namespace gs
