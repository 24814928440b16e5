/*
 * wdpm_stencil.h — the per-neighbour water transfer, shared by every stencil kernel.
 *
 * flow_add  : one neighbour step of runoffs()  (reference src/WDPMCL.c:1945-1959), add + subtract
 * flow_drain: one neighbour step of runoffd()'s non-outlet branch (WDPMCL.c:1988-2000)
 *
 * Both are written branch-free (v_cndmask selects) so a wave never diverges; the values computed
 * are exactly the reference's: same operands, same operation order, fp64, no contraction.
 * `x * 0.125` is the correctly rounded x/8 — identical to the reference's `x/8.0`.
 * The reference's min/max macros are `a<b?a:b` / `a>b?a:b` (WDPMCL.c:19-20) and are restated as
 * such (not fmin/fmax) so NaN and signed-zero behaviour is the same.
 *
 * The caller has already established that the neighbour is a valid cell (bigdem > missingvalue,
 * WDPMCL.c:1944) — or, in the fused kernel, encodes invalid cells as dem = +inf, which makes
 * ht_diff = -inf or NaN and therefore `ht_diff > 0` false, i.e. no transfer, with no extra test.
 */
#ifndef WDPM_STENCIL_H
#define WDPM_STENCIL_H

#include <hip/hip_runtime.h>

/* DEM code -> elevation (wdpm_kernels.h::DemCode).  n = q + k0 is an exact integer in fp64; n * rD is
 * within an ulp of n / D and the two fused steps (a Newton correction on the residual n - q0 * D,
 * which the FMA computes exactly) land on the correctly rounded quotient — but nothing rests on
 * that argument: the encoder runs this very function on every cell and compares bits. */
__device__ __forceinline__ double dem32_decode(const int q, const double k0, const double D, const double rD) {
  const double n = (double)q + k0;
  const double q0 = n * rD;
  const double r = __builtin_fma(-q0, D, n);
  const double v = __builtin_fma(r, rD, q0);
  return q == (int)0x80000000 ? __builtin_inf() : v;
}

/* The same for the iteration kernels, one instruction shorter: a NODATA code comes out as a NaN (only the high word is replaced)
 * instead of +inf.  Every use of an elevation in the kernels that stream codes treats the two alike.
 * add / subtract (flow_add_nz): as a neighbour, dem + w is NaN / inf, ht is NaN / -inf and nothing moves; as a centre, `dem < inf` is
 * false (the gate, the max-diff validity test), and in the gate-free variants the centre depth is +0.0 and the flow comes out as 0
 * through `NaN > en` = false, x = ht = NaN, max(NaN / 8, -0.0) = -0.0 (clamped: +0.0).
 * drain (flow_drain_nz; it streams the codes since late in round 4, ADVICE r4 asked for the argument): NaN NEIGHBOUR - nwe, ht, s and
 * big = ldexp(ht) are NaN, m = v_min(NaN, NaN) = NaN, `dem_c > nwe` is false, x = m = NaN and f = max(NaN / 8, -0.0) = -0.0 (v_max
 * returns its other operand; clamped: NaN -> +0.0): no transfer, as with +inf (nwe = inf, ht = s = big = m = -inf, f = -0.0).  NaN
 * CENTRE - gated variants: `d11 < inf` is false, the centre runs as -inf with a local depth of 0; gate-free variants: its depth is +0.0
 * exactly, ht, s, big, m are NaN, `NaN > nwe` is false, x = NaN, f = +-0.  The OUTLET's sink tests `dn < inf` (false for NaN as for
 * +inf: a NODATA cell is never drained into) and the outlet itself is found among cells with dem > 0 on the fp64 DEM, never on the
 * codes; drain()'s owed sum reads the fp64 DEM.  tests/test_hip_parity.py::test_drain_on_codes_with_nodata_around_the_outlet puts
 * NODATA at a centre, at a neighbour and on three sides of the outlet and forces the codes on the marching and the relay kernel. */
__device__ __forceinline__ double dem32_decode_nan(const int q, const double k0, const double D, const double rD) {
  const double n = (double)q + k0;
  const double q0 = n * rD;
  const double r = __builtin_fma(-q0, D, n);
  const double v = __builtin_fma(r, rD, q0);
  const int hi = q == (int)0x80000000 ? 0x7ff80000 : __double2hiint(v);
  return __hiloint2double(hi, __double2loint(v));
}

/* ... and from a 16-bit offset and its group's base (wdpm_kernels.h::DemCode::h, ::gb): q = gb + h, NODATA is h == 0xFFFF */
__device__ __forceinline__ double dem16_decode_nan(const int h, const int gb, const double k0, const double D, const double rD) {
  const double n = (double)(gb + h) + k0;
  const double q0 = n * rD;
  const double r = __builtin_fma(-q0, D, n);
  const double v = __builtin_fma(r, rD, q0);
  const int hi = h == 0xFFFF ? 0x7ff80000 : __double2hiint(v);
  return __hiloint2double(hi, __double2loint(v));
}

/* neighbour k = 0..7 in the reference's visiting order: rowloc outer -1..+1, colloc inner -1..+1,
 * centre skipped (WDPMCL.c:1940-1943) */
__host__ __device__ constexpr int nb_dr(int k) { return (k < 3) ? -1 : (k < 5 ? 0 : 1); }
__host__ __device__ constexpr int nb_dc(int k) { return (k < 3) ? k - 1 : (k == 3 ? -1 : (k == 4 ? 1 : k - 6)); }

/* `gate` folds the centre test of the sweep loops (WDPMCL.c:1099) into the transfer condition */
/* v_min_f64 / v_max_f64 without the canonicalising v_max x,x the compiler adds in IEEE mode.
 * Same value as the reference's a<b?a:b / a>b?a:b whenever the operands are not NaN and are not
 * zeros of opposite sign — which holds where the result is used: flow is +0 or positive when a
 * transfer happens and the centre water is positive (the other case, flow = -inf/NaN next to an
 * invalid neighbour, is discarded by the `go` select). */
__device__ __forceinline__ double vmin_f64(const double a, const double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double vmax_f64(const double a, const double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ void flow_add(const double dem_c, double &w_c, const double dem_n, double &w_n,
                                         const bool gate = true) {
  const double en = dem_n + w_n;                 // :1946
  const double ht = (dem_c + w_c) - en;          // :1945-1946
  const bool go = gate & (ht > 0);               // :1947
  const double x = (dem_c > en) ? w_c : ht;      // :1948 selects w_c/8 (:1949) or ht/8 (:1955)
  double flow = x * 0.125;
  flow = vmin_f64(flow, w_c);                    // :1957 min(flow, w_c)
  // One select instead of two: when no transfer happens the flow becomes -0.0, and
  //   w_n + (-0.0) == w_n   and   w_c - |-0.0| == w_c - (+0.0) == w_c
  // hold bit-for-bit for EVERY double (signed zeros, infinities and NaN payloads included), so
  // the unconditional updates below equal the reference's conditional ones (:1958-1959).  When a
  // transfer happens flow >= +0, hence |flow| == flow.  |.| is a free VOP3 source modifier.
  const double fsel = go ? flow : -0.0;
  w_c = w_c - __builtin_fabs(fsel);              // :1958
  w_n = w_n + fsel;                              // :1959
}

/* flow_add_nz — the same neighbour step in 10 VALU instructions instead of 13, bit-identical to
 * flow_add whenever the water raster holds no negative zero (the caller guarantees it: the
 * library scans every upload, and no operation of the loop can create a -0.0 depth).
 *
 * `dem_c` must already carry the centre gate: the caller passes -inf for a centre that may not
 * give water (dry, NODATA or outside the slab), which makes ht_diff -inf for every neighbour.
 *
 *   reference (:1947-1959)                     here
 *   if (ht > 0) { flow = ...; }                f = max(x/8, -0.0)     no compare, no mask
 *   flow = min(flow, w_c)                      dropped: a no-op, see (b)
 *
 * (a) no transfer (ht <= 0, -inf or NaN): dem_c > en is false [dem_c > en implies ht > 0 for
 *     w_c >= 0, because fl(dem_c+w_c) >= dem_c > en and distinct doubles never subtract to 0],
 *     so x = ht and f = max(ht/8, -0.0) is -0.0 (or +0.0 when ht == +0): w_n + f == w_n for every
 *     w_n except -0.0 (+0.0 case) — excluded — and w_c - |f| == w_c always.
 * (b) transfer (ht > 0, w_c > 0): x/8 <= w_c, so the reference's min() returns x/8:
 *     for x = w_c trivially; for x = ht (dem_c <= en): with a = fl(dem_c + w_c),
 *     ht = fl(a - en) <= fl(a - dem_c) and a - dem_c <= w_c + ulp(a)/2; ht > 0 needs a != dem_c,
 *     i.e. w_c >= ulp(a)/2 (or w_c dominates), hence ht <= 2 w_c (1 + 2^-53) < 8 w_c.
 * tests/test_stencil_forms.py checks (a),(b) on ~10^8 adversarial operand tuples against the
 * reference form, and the GPU parity tests run this variant on every golden vector. */
/* CLAMP (round 4): `max(x / 8, -0.0)` as ONE instruction.  The VOP3 clamp bit on an fp64 result clamps to [0, 1]
 * after rounding, NaN -> +0.0 (DX10_CLAMP, the HSA default), subnormal quotients kept: tools/clamp_probe.hip ran it against
 * v_ldexp_f64 + v_max_f64 on 4 * 10^6 adversarial operands on the chip - equal for every x <= 8 up to the sign of a zero result
 * (+0.0 here where the two-instruction form gives -0.0; `w + (+-0)` and `w - |+-0|` return w either way for every w that is not
 * -0.0, which the _nz forms exclude).  For x > 8 the result is 1.0 instead of x / 8, so a caller may use it only where no flow
 * of a neighbour step can exceed 1 m - the kernels establish that per wave and step from the depths they hold
 * (wdpm_fused.hip: `deep`): a flow is at most (centre depth + half an ulp of its elevation) / 8, and a depth grows by at most
 * that in each of the eight passes of an iteration in which its cell receives. */
__device__ __forceinline__ double eighth_clamped(const double x) {
  double r;
  asm("v_ldexp_f64 %0, %1, -3 clamp" : "=v"(r) : "v"(x));
  return r;
}

template <bool CLAMP = false>
__device__ __forceinline__ void flow_add_nz(const double dem_c, double &w_c, const double dem_n, double &w_n) {
  const double en = dem_n + w_n;                 // :1946
  const double ht = (dem_c + w_c) - en;          // :1945-1946
  const double x = (dem_c > en) ? w_c : ht;      // :1948-1955
  const double f = CLAMP ? eighth_clamped(x) : vmax_f64(x * 0.125, -0.0);    // :1947 + :1949/:1955 (+ :1957, a no-op)
  w_c = w_c - __builtin_fabs(f);                 // :1958
  w_n = w_n + f;                                 // :1959
}

__device__ __forceinline__ void flow_drain(const double dem_c, double &w_c, const double dem_n, double &w_n,
                                           const bool gate = true) {
  const double cwe = dem_c + w_c;                // :1977
  const double nwe = dem_n + w_n;                // :1978
  const double ht = cwe - nwe;                   // :1988
  const bool go = gate & (ht > 0);               // :1989
  const double alt = ((dem_c - dem_n) + (w_c - w_n)) * 0.125;   // :1995-1996
  double flow = (dem_c > nwe) ? w_c * 0.125 : alt;              // :1990-1991
  flow = vmax_f64(flow, 0.0);                    // :1998 max(flow, 0.0)
  flow = vmin_f64(flow, w_c);                    // :1998 min(.., w_c)
  double wc2 = w_c - flow;
  wc2 = vmax_f64(wc2, 0.0);                      // :1999 max(w_c - flow, 0.0)
  const double wn2 = w_n + flow;                 // :2000
  w_c = go ? wc2 : w_c;
  w_n = go ? wn2 : w_n;
}

/* flow_drain_nz — the non-outlet neighbour step of runoffd() in 15 VALU instructions (14 clamped) instead of
 * 21, bit-identical to flow_drain under flow_add_nz's precondition (no -0.0 depth in the raster;
 * `dem_c` = -inf for a centre that may not give water, NODATA neighbours at dem = +inf).
 *
 *   reference (:1988-2000)                          here
 *   if (ht > 0) {                                   big = ldexp(ht, 2200): +inf / 0 / -inf by sign
 *     flow = dem_c > nwe ? w_c/8 : s/8              x = dem_c > nwe ? w_c : min(s, big);  x/8
 *     flow = min(max(flow, 0.0), w_c)               f = max(x/8, -0.0)    (the min is a no-op: see the end of this note)
 *     w_c = max(w_c - flow, 0.0)                    w_c - |f|      (the max is a no-op: f <= w_c)
 *     w_n = w_n + flow }                            w_n + f
 *   with s = (dem_c - dem_n) + (w_c - w_n).
 *
 * `w_c` must be >= +0: the caller runs a block whose centre may not give water (dry, negative or NaN
 * depth from an odd input file, NODATA) on a local centre depth of 0.0 and keeps the cell's value.
 * ht > 0: big = +inf, so x is the reference's operand; max(.., -0.0) differs from max(.., 0.0) only
 *   by the sign of a zero flow, and w - |+-0| == w, w_n + (+-0) == w_n (w_n is not -0.0).
 * NaN anywhere (an odd input file): x is NaN, v_max returns its other operand, f = -0.0: no transfer,
 *   as with the reference's `ht > 0` test.
 * ht <= 0, -inf: dem_c > nwe is impossible [it implies ht > 0 for w_c >= 0, see flow_add_nz (a)], so
 *   x = min(s, big) <= 0 whatever the sign of s (s and ht are rounded differently and may disagree
 *   about the sign of a difference near zero: the reference tests ht, so must we), f = +-0, and
 *   both updates return their inputs.
 * tests/test_stencil_forms.py checks this against the reference form on adversarial operands.
 *
 * Round 4: the reference's min(flow, w_c) of :1998 is a no-op here too, as :1957's is in the add step - it was kept for three
 * rounds on the belief that s, "rounded differently from ht", might exceed 8 w_c.  It cannot.  Where water moves on the else
 * branch: w_c > 0, a = fl(dem_c + w_c) > nwe = fl(dem_n + w_n) >= dem_c.  (i) a > nwe means dem_c + w_c reaches the midpoint above
 * nwe: w_c >= gap/2 + (nwe - dem_c), gap = the spacing of doubles above nwe.  (ii) In real numbers dem_c - dem_n - w_n =
 * (dem_c - nwe) + (nwe - (dem_n + w_n)) <= -(nwe - dem_c) + gap/2 <= w_c - 2 (nwe - dem_c), so (dem_c - dem_n) + (w_c - w_n) <= 2 w_c.
 * (iii) The three roundings of s add little on the high side: fl(w_c - w_n) lies above w_c - w_n by at most min(w_c, half a
 * spacing) - the difference sits on w_n's grid, and w_c is either absorbed (the error is -w_c) or at least half a spacing -
 * and fl(dem_c - dem_n) <= fl(nwe - dem_n), which is w_n plus the rounding error of nwe (the Fast2Sum identity) plus at most half
 * a spacing of that; those spacings are gap-sized, i.e. <= 2 w_c by (i).  s <= ~5 w_c in the worst accounting, s / 8 < w_c.
 * tests/test_step_floor.py hunts for a counter-example over 2 * 10^8 operand tuples built for it (binade boundaries, exact
 * cancellations of a deep neighbour, subnormals, negative depths): the largest flow / w_c it finds is 0.25 - (ii)'s bound. */
template <bool CLAMP = false>
__device__ __forceinline__ void flow_drain_nz(const double dem_c, double &w_c, const double dem_n, double &w_n) {
  const double nwe = dem_n + w_n;                               // :1978
  const double ht = (dem_c + w_c) - nwe;                        // :1977,1988
  const double s = (dem_c - dem_n) + (w_c - w_n);               // :1995-1996
  const double big = __builtin_ldexp(ht, 2200);                 // :1989 the sign of ht as +inf / 0 / -inf
  // computed on both sides of the select below: with the inline-asm minimum inside one arm of the
  // ternary the compiler cannot speculate it and builds a divergent branch (s_and_saveexec /
  // s_cbranch_execz) around EVERY neighbour step - 72 branches per window step, no scheduling across them
  const double m = vmin_f64(s, big);
  const double x = (dem_c > nwe) ? w_c : m;                     // :1990-1996
  const double f = CLAMP ? eighth_clamped(x) : vmax_f64(x * 0.125, -0.0);   // :1998 max(flow, 0.0); NaN -> +-0
  // :1998 min(.., w_c) is dead (round 4; 16 -> 14 instructions with the clamp): see the note above
  w_c = w_c - __builtin_fabs(f);                                // :1999
  w_n = w_n + f;                                                // :2000
}

#endif
