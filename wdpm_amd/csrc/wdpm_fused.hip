/*
 * wdpm_fused.hip — one whole WDPM iteration (all 9 colour passes, reference src/WDPMCL.c:1094-1106,
 * kernels runoff.cl:137-183) in ONE launch, reading each raster from HBM once and writing the
 * water raster once: 24 B per cell-update, the algorithmic minimum.
 *
 * Design (DESIGN.md §4 has the derivation and the numpy model tests/fused_model.py checks the
 * schedule against the oracle):
 *
 *  - A wave64 is the unit of work; waves never synchronise with each other (no barriers; LDS is
 *    only a wave-private transpose buffer for the stores).  Lane m owns raster columns
 *    c0+3m .. c0+3m+2, so a wave covers a 192-column strip and a 64-lane row load is one
 *    contiguous 1536-byte segment.  Waves per SIMD: two wherever chunks of 18 (add / subtract on the DEM codes) or 12 rows
 *    (drain) fill every slot, one otherwise.  With two, a workgroup is eight waves and the two waves of a SIMD tell each other
 *    their step through LDS: whoever is behind is served first (see the marching loop).  Chunk heights follow what each XCD
 *    delivers (wdpm_kernels.h::XcdBalance).  Round 4's counters: fp64 VALU issue in ~89 % of the kernel's cycles, ~4.7 TB/s of
 *    HBM traffic - between both ceilings.
 *  - The wave marches down a chunk of rows with a 7-row window held in registers (dem + water,
 *    3 columns per lane).  Each step loads 3 new rows and applies, in this order, row alignment
 *    oi=1 to rows 3n..3n+2, oi=2 to rows 3n-2..3n, oi=3 to rows 3n-4..3n-2 — a skew that respects
 *    every dependency of the reference's pass order, because a 3x3 block of pass (oi,oj) only
 *    needs its nine cells to have finished all earlier passes.  Rows 3n-4..3n-2 are then final
 *    and are stored.
 *  - Within a row alignment the three column alignments oj=1,2,3 shift the blocks right by one
 *    column each: the lane borrows column 0 (then column 1) of lane m+1 with a DPP wave_shl:1
 *    register move, updates it, and hands both back with wave_shr:1.  No LDS traffic.
 *  - Information moves at most 8 columns left / 12 right and 2 rows up / 4 down per iteration, so
 *    a wave's 192 x (H+6) input trapezoid yields a 171 x H exact output; neighbouring waves
 *    recompute the overlap (redundant, bit-identical work) instead of communicating.
 *  - Cells with bigdem <= missingvalue and cells outside the slab are held as dem = +inf: as a
 *    neighbour this makes ht_diff -inf/NaN, so `ht_diff > 0` is false with no extra test
 *    (WDPMCL.c:1944); as a centre the gate `dem < +inf` replaces `bigdem > missingvalue` (:1099).
 *
 *  - DEM32: the static DEM can be streamed as 32-bit codes q with dem == (q + k0) / 10^e bit for bit
 *    (verified per cell at upload by the decoder used here, wdpm_stencil.h::dem32_decode): 12 bytes
 *    per lane and row in one global_load_dwordx3, 20 B of HBM traffic per cell-update instead of 24 - on the largest launches
 *    as 16-bit offsets from one 32-bit base per 48 columns (18.1 B; wdpm_kernels.h::DemCode).
 *  - Memory operations are unconditional and fixed in number per step (prefetch by inline-asm
 *    loads one step ahead, exact s_waitcnt; stores transposed through LDS to 512-byte contiguous
 *    writes, saddr-form inline asm, issued at the top of the following step).
 *  - The neighbour step is 9 (add) / 14 (drain) VALU instructions where the depths a wave holds allow the one-instruction
 *    clamped max(x / 8, -0.0) (wdpm_stencil.h; `deep` below), 10 / 15 otherwise.
 *
 * Results are bit-identical to the serial reference: same operands, same order, fp64, no FMA in
 * the stencil (the DEM decode uses two, on values it has been verified to reproduce exactly).
 *
 * Three kernels share the per-block arithmetic (block_update / blocks_lockstep) and the trapezoid:
 *   fused_iteration_kernel  the marching window above: rasters that fill the chip (from ~2300^2 add, ~2900^2 drain)
 *   relay_iteration_kernel  workgroups of four / eight waves, one row block per wave and row alignment, shared rows passed
 *                           through LDS: small and mid-size rasters (round 3; 482^2 add 7.3 -> 5.2 us, drain 12.1 -> 8.4)
 *   tri_iteration_kernel    one wave, nine (twelve) rows in, three (six) out, row blocks in lockstep: what is between the two,
 *                           and a small raster's last launch of a block (max diff folded in)
 * wdpm_launch_fused_rows picks by size, module and what is known about the raster.
 */
#include "wdpm_kernels.h"
#include "wdpm_stencil.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace {

constexpr int kLanes = 64;
constexpr int kStripIn = 3 * kLanes;                     // 192 columns loaded per wave
// Per fused iteration an error at a strip's edge travels at most 8 columns inward from the left and
// 12 from the right (cell-level dependency simulation, tests/test_rowblock.py); 13 are given up on
// the right so that the strip pitch, 171, is a multiple of 3 and every strip starts on a block edge.
constexpr int kHaloL = 8, kHaloR = 13;
constexpr int kStripOut = kStripIn - kHaloL - kHaloR;    // 171 columns stored per wave

#define WDPM_INF (__builtin_inf())

/* value held by lane+1; lane 63 receives 0.0 (DPP bound_ctrl zero fill, so no register has to
 * be preset).  v_mov_b32_dpp wave_shl:1.  Lane 63's borrowed columns lie beyond the strip: its
 * blocks of passes oj=2,3 are inside the right halo whose results are never stored, so
 * any finite fill is as good as the true value there. */
__device__ __forceinline__ double lane_next(const double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

/* value held by lane-1; lane 0 keeps `keep`.  v_mov_b32_dpp wave_shr:1 */
__device__ __forceinline__ double lane_prev(const double v, const double keep) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(__double2loint(keep), lo, 0x138, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(__double2hiint(keep), hi, 0x138, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_read(const double v, const int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

/* drain-module bookkeeping carried through the passes (compiles away for add/subtract) */
struct DrainState {
  double td;      // running totaldrain (wave-uniform value, one copy per lane)
  bool hit;       // this lane drained in the current pass
};

/* one neighbour step of runoffd() including its outlet branch (WDPMCL.c:1975-2002) */
__device__ __forceinline__ void step_drain(const double dc, double &wc, const double dn, double &wn,
                                           const bool gate, const bool is_outlet, DrainState &ds) {
  const bool hit = gate & is_outlet & (dn < WDPM_INF);   // :1976,1980
  const double td2 = (ds.td + wn) + wc;                   // :1982
  ds.td = hit ? td2 : ds.td;
  ds.hit = ds.hit | hit;
  wn = hit ? 0.0 : wn;                                    // :1983
  wc = hit ? 0.0 : wc;                                    // :1984
  flow_drain(dc, wc, dn, wn, gate & !hit);                // :1988-2000
}

/* one 3x3 block of one colour pass: centre (1,1), neighbours in row-major order.  OUTLET (drain
 * only): this block may touch the outlet cell, run runoffd()'s outlet branch too; the caller knows
 * wave-uniformly that it cannot for all but a handful of the raster's blocks. */
/* PLAIN (only with !SZ_SAFE): the caller knows that every cell which may not give water - dry, NODATA, outside the slab - holds
 * +0.0 exactly (wdpm_kernels.h::wdpm_launch_scan_water; no operation of the loop leaves that state).  The centre test of the
 * sweep (WDPMCL.c:1099) then costs nothing: for a centre depth of +0.0 the neighbour steps below come out as f = +-0 by
 * themselves, whatever the elevations - finite centre: dem_c > en gives x = w_c = 0, otherwise ht = dem_c - en <= 0 and
 * max(ht / 8, -0.0) = +-0; NODATA centre (dem = +inf): inf > en gives x = w_c = 0, and next to a NODATA neighbour ht is NaN,
 * which v_max turns into -0.0 - and w + (+-0) == w, w - |+-0| == w for every w that is not -0.0.  (drain: the same through
 * min(max(x / 8, -0.0), w_c = 0).)  Four instructions per block less for add, eight for drain. */
/* FLAGS: bit 0 = PLAIN, bit 1 = CLAMP (wdpm_stencil.h::eighth_clamped: the caller knows that no depth of this block exceeds 8 m
 * during the pass - see `deep` in the marching kernel; never on the outlet's path) */
template <int MODULE, bool SZ_SAFE, bool OUTLET = true, int FLAGS = 0>
__device__ __forceinline__ void block_update(
    double &w00, double &w01, double &w02, double &w10, double &w11, double &w12, double &w20, double &w21,
    double &w22, const double d00, const double d01, const double d02, const double d10, const double d11,
    const double d12, const double d20, const double d21, const double d22,
    const bool rd0, const bool rd1, const bool rd2, const bool cd0, const bool cd1, const bool cd2,
    DrainState &ds) {
  constexpr bool PLAIN = (FLAGS & 1) != 0, CL = (FLAGS & 2) != 0;
  double wc = w11;
  bool gate = (wc > 0.0) & (d11 < WDPM_INF);            // WDPMCL.c:1099
  if (MODULE == 2 && !OUTLET && !SZ_SAFE && PLAIN) {
    flow_drain_nz<CL>(d11, wc, d00, w00);
    flow_drain_nz<CL>(d11, wc, d01, w01);
    flow_drain_nz<CL>(d11, wc, d02, w02);
    flow_drain_nz<CL>(d11, wc, d10, w10);
    flow_drain_nz<CL>(d11, wc, d12, w12);
    flow_drain_nz<CL>(d11, wc, d20, w20);
    flow_drain_nz<CL>(d11, wc, d21, w21);
    flow_drain_nz<CL>(d11, wc, d22, w22);
  } else if (MODULE == 2 && !OUTLET && !SZ_SAFE) {
    const double dce = gate ? d11 : -WDPM_INF;           // see flow_drain_nz
    wc = gate ? wc : 0.0;                                // its clamp needs a centre depth >= +0
    flow_drain_nz<CL>(dce, wc, d00, w00);
    flow_drain_nz<CL>(dce, wc, d01, w01);
    flow_drain_nz<CL>(dce, wc, d02, w02);
    flow_drain_nz<CL>(dce, wc, d10, w10);
    flow_drain_nz<CL>(dce, wc, d12, w12);
    flow_drain_nz<CL>(dce, wc, d20, w20);
    flow_drain_nz<CL>(dce, wc, d21, w21);
    flow_drain_nz<CL>(dce, wc, d22, w22);
    wc = gate ? wc : w11;
  } else if (MODULE == 2 && !OUTLET) {
    flow_drain(d11, wc, d00, w00, gate);
    flow_drain(d11, wc, d01, w01, gate);
    flow_drain(d11, wc, d02, w02, gate);
    flow_drain(d11, wc, d10, w10, gate);
    flow_drain(d11, wc, d12, w12, gate);
    flow_drain(d11, wc, d20, w20, gate);
    flow_drain(d11, wc, d21, w21, gate);
    flow_drain(d11, wc, d22, w22, gate);
  } else if (MODULE == 2 && !SZ_SAFE) {
    // A block that may hold the outlet, no -0.0 in the raster: the plain neighbour step (flow_drain_nz) plus,
    // for the ONE neighbour position that can be the outlet in this wave, runoffd()'s sink (:1980-1985).
    // Which of the block's rows is the outlet's row is wave-uniform (rd*), whether ANY lane has the outlet's
    // column at block column c takes a ballot (anyc*): the sink's few instructions sit behind a scalar branch
    // that all but a handful of the raster's blocks never take.  After the sink the centre holds 0, and a
    // centre without water moves nothing in the remaining steps (f = min(.., 0)), exactly as in the reference,
    // whose loop also runs on with w[centre] = 0.
    gate = gate & !(rd1 & cd1);                          // :1082 the outlet is never a centre
    const double dce = gate ? d11 : -WDPM_INF;
    wc = gate ? wc : 0.0;
    const bool any0 = __ballot(cd0) != 0, any1 = __ballot(cd1) != 0, any2 = __ballot(cd2) != 0;
    bool hit_any = false;
    auto nb = [&](const double dn, double &wn, const bool rd, const bool anyc, const bool cd) {
      if (rd & anyc) {                                   // wave-uniform
        const bool hit = gate & cd & (dn < WDPM_INF);    // :1976,1980
        const double td2 = (ds.td + wn) + wc;            // :1982
        ds.td = hit ? td2 : ds.td;
        wn = hit ? 0.0 : wn;                             // :1983
        wc = hit ? 0.0 : wc;                             // :1984
        hit_any = hit_any | hit;
      }
      flow_drain_nz<false>(dce, wc, dn, wn);                 // :1988-2000 (never clamped: the outlet's path is off the hot path)
    };
    nb(d00, w00, rd0, any0, cd0);
    nb(d01, w01, rd0, any1, cd1);
    nb(d02, w02, rd0, any2, cd2);
    nb(d10, w10, rd1, any0, cd0);
    nb(d12, w12, rd1, any2, cd2);
    nb(d20, w20, rd2, any0, cd0);
    nb(d21, w21, rd2, any1, cd1);
    nb(d22, w22, rd2, any2, cd2);
    wc = gate ? wc : w11;
    // at most one lane of the wave holds the outlet in this pass: make its totaldrain the wave's
    const unsigned long long m = __ballot(hit_any);
    if (m) ds.td = wave_read(ds.td, __ffsll((long long)m) - 1);
  } else if (MODULE == 2) {
    gate = gate & !(rd1 & cd1);                          // :1082 the outlet is never a centre
    ds.hit = false;
    step_drain(d11, wc, d00, w00, gate, rd0 & cd0, ds);
    step_drain(d11, wc, d01, w01, gate, rd0 & cd1, ds);
    step_drain(d11, wc, d02, w02, gate, rd0 & cd2, ds);
    step_drain(d11, wc, d10, w10, gate, rd1 & cd0, ds);
    step_drain(d11, wc, d12, w12, gate, rd1 & cd2, ds);
    step_drain(d11, wc, d20, w20, gate, rd2 & cd0, ds);
    step_drain(d11, wc, d21, w21, gate, rd2 & cd1, ds);
    step_drain(d11, wc, d22, w22, gate, rd2 & cd2, ds);
    // at most one lane of the wave holds the outlet in this pass: make its totaldrain the wave's
    const unsigned long long m = __ballot(ds.hit);
    if (m) ds.td = wave_read(ds.td, __ffsll((long long)m) - 1);
  } else if (!SZ_SAFE) {
    // no -0.0 in the raster: the gate rides on the centre elevation (see flow_add_nz) - or is not needed at all (PLAIN)
    const double dce = PLAIN ? d11 : (gate ? d11 : -WDPM_INF);
    flow_add_nz<CL>(dce, wc, d00, w00);
    flow_add_nz<CL>(dce, wc, d01, w01);
    flow_add_nz<CL>(dce, wc, d02, w02);
    flow_add_nz<CL>(dce, wc, d10, w10);
    flow_add_nz<CL>(dce, wc, d12, w12);
    flow_add_nz<CL>(dce, wc, d20, w20);
    flow_add_nz<CL>(dce, wc, d21, w21);
    flow_add_nz<CL>(dce, wc, d22, w22);
  } else {
    flow_add(d11, wc, d00, w00, gate);
    flow_add(d11, wc, d01, w01, gate);
    flow_add(d11, wc, d02, w02, gate);
    flow_add(d11, wc, d10, w10, gate);
    flow_add(d11, wc, d12, w12, gate);
    flow_add(d11, wc, d20, w20, gate);
    flow_add(d11, wc, d21, w21, gate);
    flow_add(d11, wc, d22, w22, gate);
  }
  w11 = wc;
}

/* the three column alignments oj = 1,2,3 of one row alignment, on window slots S0..S0+2 */
template <int MODULE, bool SZ_SAFE, int S0, bool OUTLET, int PLAIN = 0>
__device__ __forceinline__ void stage_impl(double (&W)[7][3], const double (&D)[7][3], const int row_s0,
                                           const int drain_row, const bool (&cdr)[5], DrainState &ds) {
  const bool rd0 = MODULE == 2 && OUTLET && row_s0 == drain_row;
  const bool rd1 = MODULE == 2 && OUTLET && row_s0 + 1 == drain_row;
  const bool rd2 = MODULE == 2 && OUTLET && row_s0 + 2 == drain_row;
  constexpr int a = S0, b = S0 + 1, c = S0 + 2;
  // oj = 1: own columns 0,1,2
  block_update<MODULE, SZ_SAFE, OUTLET, PLAIN>(W[a][0], W[a][1], W[a][2], W[b][0], W[b][1], W[b][2], W[c][0], W[c][1], W[c][2],
                       D[a][0], D[a][1], D[a][2], D[b][0], D[b][1], D[b][2], D[c][0], D[c][1], D[c][2],
                       rd0, rd1, rd2, cdr[0], cdr[1], cdr[2], ds);
  // oj = 2: own columns 1,2 + column 0 of the next lane
  double wa0 = lane_next(W[a][0]), wb0 = lane_next(W[b][0]), wc0 = lane_next(W[c][0]);
  const double da0 = lane_next(D[a][0]), db0 = lane_next(D[b][0]),
               dc0 = lane_next(D[c][0]);
  block_update<MODULE, SZ_SAFE, OUTLET, PLAIN>(W[a][1], W[a][2], wa0, W[b][1], W[b][2], wb0, W[c][1], W[c][2], wc0,
                       D[a][1], D[a][2], da0, D[b][1], D[b][2], db0, D[c][1], D[c][2], dc0,
                       rd0, rd1, rd2, cdr[1], cdr[2], cdr[3], ds);
  // oj = 3: own column 2 + columns 0,1 of the next lane
  double wa1 = lane_next(W[a][1]), wb1 = lane_next(W[b][1]), wc1 = lane_next(W[c][1]);
  const double da1 = lane_next(D[a][1]), db1 = lane_next(D[b][1]),
               dc1 = lane_next(D[c][1]);
  block_update<MODULE, SZ_SAFE, OUTLET, PLAIN>(W[a][2], wa0, wa1, W[b][2], wb0, wb1, W[c][2], wc0, wc1,
                       D[a][2], da0, da1, D[b][2], db0, db1, D[c][2], dc0, dc1,
                       rd0, rd1, rd2, cdr[2], cdr[3], cdr[4], ds);
  // hand the borrowed columns back to lane+1; lane 0 keeps its own (nothing to its left)
  W[a][0] = lane_prev(wa0, W[a][0]);  W[a][1] = lane_prev(wa1, W[a][1]);
  W[b][0] = lane_prev(wb0, W[b][0]);  W[b][1] = lane_prev(wb1, W[b][1]);
  W[c][0] = lane_prev(wc0, W[c][0]);  W[c][1] = lane_prev(wc1, W[c][1]);
}

/* Drain: the outlet branch of runoffd() costs ten more instructions per neighbour step, and exactly
 * one cell of the raster needs it.  Whether any of the seven rows of this step's window is the outlet's
 * row is wave-uniform (computed on the scalar unit), so all other steps run the plain neighbour step -
 * as ONE basic block of three stages, which lets the scheduler overlap the stages' tails and heads
 * (a test per stage cut the step into three blocks). */
/* NSTAGES < 3: the first two steps of a chunk.  Step 0 holds real rows only in slots 4..6: its oi = 2 block has its centre
 * on an empty slot (dem = +inf: nothing moves), its oi = 3 block lies on empty slots altogether.  Step 1's oi = 3 block
 * (rows A-1 .. A+1, centre row A) touches rows A and A+1 only, after the last pass that reads them in this iteration, and
 * those rows are never stored (a chunk stores from row A+2; for A = 0 the centre row is the raster's border or a slab's
 * first halo row, which is wrong from the first iteration after a refresh on anyway): dead work.  Leaving those three stage
 * executions out saves 3 of 3 (H/3 + 2): 9 % of a 27-row chunk, 5.6 % at 48 rows, 0.4 % at 780. */
template <int MODULE, bool SZ_SAFE, int NSTAGES = 3, int PLAIN = 0>
__device__ __forceinline__ void three_stages(double (&W)[7][3], const double (&D)[7][3], const int rbase,
                                             const int drain_row, const bool (&cdr)[5], DrainState &ds) {
  if (MODULE == 2 && drain_row >= rbase && drain_row <= rbase + 6) {           // wave-uniform, rare
    stage_impl<MODULE, SZ_SAFE, 4, true>(W, D, rbase + 4, drain_row, cdr, ds);   // oi = 1 on rows 3n   .. 3n+2
    if (NSTAGES >= 2) stage_impl<MODULE, SZ_SAFE, 2, true>(W, D, rbase + 2, drain_row, cdr, ds);   // oi = 2 on rows 3n-2 .. 3n
    if (NSTAGES >= 3) stage_impl<MODULE, SZ_SAFE, 0, true>(W, D, rbase + 0, drain_row, cdr, ds);   // oi = 3 on rows 3n-4 .. 3n-2
  } else {
    stage_impl<MODULE, SZ_SAFE, 4, false, PLAIN>(W, D, rbase + 4, drain_row, cdr, ds);
    if (NSTAGES >= 2) stage_impl<MODULE, SZ_SAFE, 2, false, PLAIN>(W, D, rbase + 2, drain_row, cdr, ds);
    if (NSTAGES >= 3) stage_impl<MODULE, SZ_SAFE, 0, false, PLAIN>(W, D, rbase + 0, drain_row, cdr, ds);
  }
}

#ifdef WDPM_WAVE_TIMES   /* timing builds only (tools/wave_times.py): when each wave of the marching kernel starts and ends, and where */
static __device__ unsigned long long g_wave_times[4 * 8192];   /* one per translation unit (WDPM_TU): each has its accessor */
/* relay kernel: eight stamps per wave (tools/relay_times.py); the wait in front of each makes the stamp mean "everything before is done" */
#define WDPM_RSTAMP(k) do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); rst[k] = wall_clock64(); __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define WDPM_RSTAMP(k) do { } while (0)
#endif
#ifndef WDPM_FUSED_MIN_WAVES
#define WDPM_FUSED_MIN_WAVES 2   /* waves per SIMD the register allocator must leave room for */
#endif
#ifndef WDPM_FUSED_CODES_WAVES
#define WDPM_FUSED_CODES_WAVES 2 /* the same for the add / subtract instances that stream the DEM as codes */
#endif

/* raw registers of the row prefetch.  DEM32: the dem rows arrive as 32-bit codes, a lane's three columns
 * in one 12-byte load (qi).  Only the members an instantiation uses exist. */
typedef int wdpm_i3 __attribute__((ext_vector_type(3)));
struct Prefetched {
  double NW[3][3], ND[3][3];
  wdpm_i3 qi[3];
  int qh[3][3];      /* DEM32 == 2: 16-bit offsets (zero-extended by the load) ... */
  int gbv[3];        /* ... and the lane's group base per row */
};

/* waves per SIMD an instantiation is built for.  The drain variant for rasters that hold a -0.0 depth (SZ_SAFE: the reference's
 * conditional 21-instruction step) does not fit two waves' 256 VGPRs - it spilled 12 bytes in round 3 - and is built for one. */
template <int MODULE, bool SZ_SAFE, int DEM32, bool MD>
constexpr int fused_built_for() {
  return (MODULE == 2 && SZ_SAFE) ? 1 : (MODULE != 2 && DEM32 != 0 && !MD && !SZ_SAFE) ? WDPM_FUSED_CODES_WAVES : WDPM_FUSED_MIN_WAVES;
}

/* DEM32: 0 = the fp64 DEM, 1 = verified 32-bit codes, 2 = those as 16-bit offsets + group bases (wdpm_kernels.h::DemCode) */
template <int MODULE, bool SZ_SAFE, int DEM32, bool FLUSH = false, bool MD = false, bool PLAIN = false>
__global__ void __launch_bounds__((fused_built_for<MODULE, SZ_SAFE, DEM32, MD>() >= 2 ? 512 : 256), (fused_built_for<MODULE, SZ_SAFE, DEM32, MD>()))
fused_iteration_kernel(const double *__restrict__ win, double *__restrict__ wout,
                       const double *__restrict__ dem, const DemCode code, const SlabGeom g, const int nstrips,
                       const int nitems, const int H, const int A0, const int out_last,
                       double *__restrict__ totaldrain, const double thres, const int drain_owed,
                       const TileFlags tf, const MaxDiffArgs md, const int prio, const int no_clamp, const BalanceArgs bal) {
  const int lane = threadIdx.x & 63;
  // Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2; placement is only a
  // speed matter, never correctness): give each XCD a contiguous run of work items so that the
  // halo columns shared by neighbouring strips hit in its L2 (+2 % measured).  gridDim.x is a
  // multiple of 8, so b -> (b % 8) * (gridDim.x / 8) + b / 8 is a permutation of the blocks.
  // ... and the share a workgroup takes is that of the PHYSICAL XCD it is expected to run on (BalanceArgs::rot): blockIdx % 8 turned by
  // the rotation the previous launch observed - one value for the whole launch, so this stays a permutation of the blocks.
  const int xcc_phys = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7);          // XCC_ID
  int rot_prev = 0;
  if (bal.rot) {
    rot_prev = __builtin_amdgcn_readfirstlane((int)(bal.rot[bal.parity] & 7));
    if (blockIdx.x == 0 && threadIdx.x == 0) bal.rot[bal.parity ^ 1] = (unsigned long long)xcc_phys;   // workgroup 0's XCD = this launch's rotation
  }
  const int share = (int)((blockIdx.x + (unsigned)rot_prev) % 8);
#ifdef WDPM_XCD_REVERSE   /* timing experiments: XCD x takes the raster's band 7 - x (does a slow XCD stay slow, or the band?) */
  const int vb = (7 - share) * (gridDim.x / 8) + blockIdx.x / 8;
#elif defined(WDPM_ORDER_REVERSE)   /* timing experiments: an XCD's workgroups take its band from the bottom up (is it the first rows of the raster that are slow, or the workgroups dispatched first?) */
  const int vb = share * (gridDim.x / 8) + (gridDim.x / 8 - 1 - blockIdx.x / 8);
#else
  const int vb = share * (gridDim.x / 8) + blockIdx.x / 8;
#endif
  // the wave number is the same in all 64 lanes: say so, and everything derived from it (strip,
  // chunk, row bases, loop bounds) lives in SGPRs and is computed on the scalar unit
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // A workgroup is eight waves where the launch puts two waves on every SIMD (one workgroup per CU: the two waves of a SIMD can
  // then see each other's progress in LDS - see `prio` in the marching loop), four where it leaves half of the slots empty.
  const int item = vb * (int)(blockDim.x >> 6) + wave;
  // Which two waves of this workgroup share a SIMD (see `prio` in the marching loop): settled by the waves themselves.
  // Rounds 3 - 4 took "SIMD id x 2 + the low bit of the wave slot" from HW_ID, which pairs two waves only while they sit in slots
  // that differ in bit 0 - true on an empty chip, not when other kernels hold slots (RCCL's send / recv kernels during the overlapped
  // iteration of an N-GPU run, the tail of the previous launch): slots 0 / 2 then share a word, each wave reads its own step
  // number back and the mechanism switches itself off.  Now every wave takes a seat at its SIMD (an LDS counter per SIMD id);
  // seats 0 and 1 of a SIMD are a pair, anything else (a SIMD with one or three of this workgroup's waves) runs without priorities.
  __shared__ int progress_all[8];                        // a pair's two step numbers: words 2 * SIMD + seat
  __shared__ int simd_seats[4];
  int my_word = 0;
  bool paired = false;
  if (prio) {                                            // the same in every wave of the launch: all of them reach the barriers
    if (threadIdx.x < 8) progress_all[threadIdx.x] = 0;
    if (threadIdx.x < 4) simd_seats[threadIdx.x] = 0;
    __syncthreads();
    const int simd = (int)((__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 4) & 3);      // HW_ID: SIMD [5:4]
    int seat = 0;
    if (lane == 0) seat = atomicAdd(&simd_seats[simd], 1);
    seat = __builtin_amdgcn_readfirstlane(seat);
    __syncthreads();
    paired = seat < 2 && simd_seats[simd] == 2;
    my_word = 2 * simd + (seat & 1);
  }
  const int use_prio = paired ? 1 : 0;
  if (item >= nitems) return;                       // wave-uniform
#ifdef WDPM_WAVE_TIMES
  const unsigned long long wt0 = wall_clock64();
#endif
  const int strip = item % nstrips, chunk = item / nstrips;
  const int c0 = kStripOut * strip;
  const int oc_lo = strip == 0 ? 0 : c0 + kHaloL;
  const int oc_hi = c0 + kStripIn - 1 - kHaloR;
  // a launch produces the rows [A0 + 2 (0 when A0 == 0), out_last] of the slab; A0 % 3 == 0
  int A = A0 + H * chunk, A_next = A + H;
  if (bal.table) {
    // chunk heights that follow what each XCD delivers (wdpm_kernels.h::BalanceArgs): this strip's own row boundaries.  Whatever
    // the table holds, no access can leave the slab: A is brought into it, waves near its end take the clamping loads
    // (`edge`), and rows outside [or_lo, or_hi] - or_hi never beyond out_last - are stored to the dump area.
    A = bal.table[chunk * nstrips + strip];
    A_next = bal.table[(chunk + 1) * nstrips + strip];
    A = A < 0 ? 0 : (A > g.rows ? g.rows : A);
    const int h = A_next - A;
    A_next = A + (h < 0 ? 0 : (h > 30000 ? 30000 : h));
  }
  const unsigned long long bal_t0 = bal.acc ? wall_clock64() : 0;
  const int nsteps = (A_next - A) / 3 + 2;
  const int or_lo = A == 0 ? 0 : A + 2;
  int or_hi = A_next + 1;
  if (or_hi > out_last) or_hi = out_last;

  // Dry tiles (wdpm_kernels.h::TileFlags): if this tile and its eight neighbours hold nothing but +0.0 in `win`,
  // the whole input window of this wave is dry - it lies inside those nine exact output blocks, because
  // H >= 6 and the halo is 8 / 13 columns - no centre has water to give, and the output block is all +0.0.
  // Then nothing is loaded, and nothing is stored either if `wout`'s block is known to hold zeros already.
  const int zpitch = nstrips + 2;                                      // flag arrays carry a border of 1s ("dry")
  const int tile = (chunk + 1) * zpitch + strip + 1;
  if (tf.zout) {                                                       // wave-uniform
    bool dry = false;
    if (tf.zin) {
      // nine unconditional loads in flight together (a bounds test per neighbour made nine dependent round trips)
      const unsigned char *z = tf.zin + tile;
      const unsigned f = z[-zpitch - 1] & z[-zpitch] & z[-zpitch + 1] & z[-1] & z[0] & z[1] & z[zpitch - 1] & z[zpitch] &
                         z[zpitch + 1];
      dry = f != 0;
    }
    if (dry) {
      if (!(tf.zout_known && tf.zout[tile] != 0)) {
        const int hi = oc_hi < g.ncp - 1 ? oc_hi : g.ncp - 1;
        for (int r = or_lo; r <= or_hi; r++)
#pragma unroll
          for (int k = 0; k < 3; k++) {
            const int c = oc_lo + 64 * k + lane;
            if (c <= hi) wout[(size_t)r * g.ncp + c] = 0.0;
          }
        if (lane == 0) tf.zout[tile] = 1;
      }
      return;
    }
  }
  unsigned long long nzmask = 0;        // lanes that staged a value other than 0.0 (no -0.0 exists where tiles are tracked)
  // CLAMP (round 4; wdpm_stencil.h::eighth_clamped): one instruction less per neighbour step, exact while no flow of the step
  // exceeds 1 m.  A flow is at most (the centre's depth + half an ulp of its elevation) / 8, a cell receives in at most eight
  // of an iteration's nine passes (it is the centre of the ninth), and within one pass in at most one block: if every depth of
  // the window was <= M when it was loaded, every depth the window holds during the iteration is <= M (9/8)^8 + 8 ulp < 2.566 M + 8 ulp.
  // With M < 3.000002 m (the high words of the nine values a step loads, compared as integers: a negative or NaN depth counts as
  // deep) and elevations below 2^30 m in magnitude (the host's part: `no_clamp`) that is < 7.7 m.  The seven rows of a step's
  // window were loaded by this step and the two before it: bits 0..2 of `deep`; bit 3 = no_clamp.  A step with any of them set
  // runs the unclamped stages - same results, the round-3 instruction count.  Wave-uniform: one scalar branch per step.
  int deep = (!SZ_SAFE && !no_clamp) ? 0 : 8;
  double md_max = 0.0;                  // MD: this lane's max |w - oldw| over the cells of its output block (WDPMCL.c:1239-1254)
  const int colb = c0 + 3 * lane;
  // store side: after the LDS transpose, store instruction k = 0,1,2 writes the strip-relative
  // columns lo + 64k + lane of the exact output range [lo, hi].  Lanes past hi are clamped to hi:
  // they read the same LDS word and write the same address with the same value as the last valid
  // lane, so every store is unconditional (fixed instruction count) and nothing is written twice
  // with different data.
  int scol[3];
  unsigned soff[3];                     // the same as byte offsets: a store's address is a wave-uniform row base (SGPRs) + this
  {
    const int lo = oc_lo - c0;
    const int hi = (oc_hi < g.ncp - 1 ? oc_hi : g.ncp - 1) - c0;
#pragma unroll
    for (int k = 0; k < 3; k++) { scol[k] = lo + 64 * k + lane < hi ? lo + 64 * k + lane : hi; soff[k] = 8u * (unsigned)scol[k]; }
  }
  __shared__ double stage_all[8][3 * kStripIn];          // 4.5 KiB per wave, private to the wave
  double *const stage_lds = stage_all[wave];
  volatile int *const my_progress = progress_all + my_word;
  volatile int *const partner_progress = progress_all + (my_word ^ 1);
  bool cdr[5];
#pragma unroll
  for (int j = 0; j < 5; j++) cdr[j] = MODULE == 2 && colb + j == g.dc;
  DrainState ds;
  ds.td = MODULE == 2 ? *totaldrain : 0.0;
  ds.hit = false;
  // exactly one wave holds the outlet cell in its exact output region; it sees every pass that touches
  // the outlet, in the reference's order, so its totaldrain is the raster's
  bool owner = false;
  if (MODULE == 2) {
    const int dcol_hi = oc_hi < g.ncp - 1 ? oc_hi : g.ncp - 1;
    owner = g.dr >= or_lo && g.dr <= or_hi && g.dc >= oc_lo && g.dc <= dcol_hi;
    // drain_owed: the previous iteration's drain() (WDPMCL.c:1089 -> :1859-1897) has not been applied to `win`
    // yet - there is no launch of its own for it.  Its sum enters totaldrain here (row-major from 0.0, valid
    // cells with water, :1877-1884), its zeroing of the nine cells (:1885-1889) happens as rows are loaded.
    if (drain_owed && owner && g.dr >= 1 && g.dr <= g.rows - 2 && g.dc >= 1 && g.dc <= g.ncp - 2) {
      double sum = 0.0;
#pragma unroll
      for (int i = -1; i <= 1; i++)
#pragma unroll
        for (int j = -1; j <= 1; j++) {
          const size_t k = (size_t)(g.dr + i) * g.ncp + (g.dc + j);
          const double wk = win[k];
          if (dem[k] < WDPM_INF && wk > 0) sum += wk;
        }
      ds.td = ds.td + sum;
    }
  }
  const bool owed_here = MODULE == 2 && drain_owed && g.dr >= 1 && g.dr <= g.rows - 2 && g.dc >= 1 && g.dc <= g.ncp - 2;
  bool owed_col[3];
#pragma unroll
  for (int j = 0; j < 3; j++) owed_col[j] = owed_here && colb + j >= g.dc - 1 && colb + j <= g.dc + 1;

  double W[7][3], D[7][3];
#pragma unroll
  for (int k = 0; k < 7; k++)
#pragma unroll
    for (int j = 0; j < 3; j++) { W[k][j] = 0.0; D[k][j] = WDPM_INF; }

  // Only waves of the last strip / last chunk can touch cells outside the slab; every other wave
  // loads without any predicate.  `edge` is wave-uniform (the margin lets interior waves prefetch
  // past their last step unconditionally).
  // Round 5: a strip that reaches past the raster's right edge is NOT an edge case.  Its lanes beyond column ncp - 1 hold whatever
  // the clamped loads bring - the border cell and the first cells of the next row - and never touch the raster: column ncp - 1 is the
  // reference's 1-cell border (bigdem = missingvalue, WDPMCL.c:796-807; +inf / NaN here), which neither gives (its centre test fails;
  // in the gate-free variants its depth is +0.0, or they would not run) nor receives (as a neighbour dem + w is inf / NaN and the
  // flow +-0), every path from a cell beyond the edge to a cell of the raster leads through it, and no lane beyond it is ever
  // stored (the store columns are clamped to the border, which holds what it held).  The kernel relied on that border before - it
  // runs centres on every column, the border's included.  Rounds 1 - 4 sent those waves down the masking instantiation of the
  // marching loop: one wave per workgroup, alone on a code path nobody shared its instruction-cache misses with - they ended 3.5 %
  // (8192^2 drain) to 12 % (the 8-GPU drain slab) after everybody else, the last waves of almost every launch
  // (profiles/r05/first_chunk_row.txt, wave_times_coledge.txt).  -DWDPM_COL_EDGE_MASKS: as before (A/B).
#ifdef WDPM_COL_EDGE_MASKS
  const bool edge = (c0 + kStripIn > g.ncp) || (A + 3 * (nsteps + 1) > g.rows);
#else
  const bool edge = (A + 3 * (nsteps + 1) > g.rows);
#endif
  const size_t pitch = (size_t)g.ncp;
  // Cells a wave must not write (outside its exact output block) are redirected to a 64-double
  // dump area behind the raster instead of being branched around: the loop then issues the same
  // number of loads and stores on every trip, so the compiler can wait with exact vmcnt counts
  // (with conditional memory operations it falls back to vmcnt(0) at the top of each step, which
  // exposes the full latency of the stores just issued — measured 27 % of the kernel).
  // (192 doubles behind the raster, allocated by wdpm_create: a dumped row goes there at the same lane offsets as a stored one, so
  // that every store is "uniform base + lane offset" - the saddr form, no 64-bit address arithmetic on the vector unit: nine
  // v_lshl_add_u64 per step less, round 4)
  char *const dump = reinterpret_cast<char *>(wout + (size_t)g.rows * pitch);

  // The marching loop, instantiated for interior (EDGE = false) and edge waves.
  auto march = [&](auto edge_tag) {
    constexpr bool EDGE = decltype(edge_tag)::value;
    // lane byte offsets inside a row: interior waves use one offset + immediates, edge waves clamp
    // every column into the raster (values of clamped cells are masked on use)
    // Edge waves clamp the row (scalar) and the lane's first column into the raster and read their three columns with the same
    // one-offset loads as everybody else: a lane whose columns straddle the row's end reads up to two cells past it - the next
    // row's first cells, or, in the slab's last row, the 192 spare cells behind every raster wdpm_create allocates (16 codes behind
    // the DEM codes) - and masks them on use, as it masks whole lanes beyond the raster.  (Rounds 1 - 3 clamped column by column:
    // nine loads per row instead of three, and the edge waves were the last of every launch to end, round 4.)
    const int voff0 = 8 * (colb < g.ncp ? colb : g.ncp - 1);      // loop-invariant: the clamp costs the interior waves nothing
    const int qoff0 = voff0 / 2;                     // the same for the 4-byte dem codes,
    const int hoff0 = voff0 / 4;                     // the 2-byte offsets
    const int goff0 = 4 * ((voff0 / 8) / kDemGroup); // and the 4-byte group bases (one per kDemGroup columns)

    // Prefetch of the three rows starting at r0 into raw registers.  The loads are issued with
    // inline asm (saddr form: wave-uniform row base in SGPRs + a per-lane byte offset) so that
    // they stay exactly here, one whole step ahead of their use, and are waited for with an exact
    // `s_waitcnt vmcnt(9)` (the 9 stores of the step are the only younger memory operations).
    // Left to the compiler the loads get sunk next to their consumer or guarded by vmcnt(0), which
    // exposes a full memory round trip per step.  The registers are not read before wait_rows().
    auto prefetch = [&](Prefetched &P, const int r0) {
#pragma unroll
      for (int i = 0; i < 3; i++) {
        int r = r0 + i;
        if (EDGE) r = r < g.rows ? r : g.rows - 1;
#ifdef WDPM_ABLATE_HBM   /* timing experiments only (tools/build_variant.sh ablateN -DWDPM_ABLATE_HBM=N, then tools/ab_interleaved.sh): what the kernel costs when its rows come from / go to the caches
                            instead of HBM.  Bits: 1 = the water loads come from the raster's first 48 rows, 2 = the stores land there,
                            4 = the DEM loads come from there.  Same instructions, same number of memory operations, wrong results. */
        const int rw = (WDPM_ABLATE_HBM & 1) ? r % 48 : r, rdm = (WDPM_ABLATE_HBM & 4) ? r % 48 : r;
#else
        const int rw = r, rdm = r;
#endif
        const double *bw = win + (size_t)rw * pitch;     // wave-uniform
        const double *bd = dem + (size_t)rdm * pitch;
        const int *bq = code.q + (size_t)rdm * pitch;
        const unsigned short *bh = code.h + (size_t)rdm * pitch;
        const int *bg = code.gb + (size_t)rdm * code.ngroups;
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(P.NW[i][0]) : "v"(voff0), "s"(bw) : "memory");
        asm volatile("global_load_dwordx2 %0, %1, %2 offset:8" : "=v"(P.NW[i][1]) : "v"(voff0), "s"(bw) : "memory");
        asm volatile("global_load_dwordx2 %0, %1, %2 offset:16" : "=v"(P.NW[i][2]) : "v"(voff0), "s"(bw) : "memory");
        if (DEM32 == 2) {
          asm volatile("global_load_ushort %0, %1, %2" : "=v"(P.qh[i][0]) : "v"(hoff0), "s"(bh) : "memory");
          asm volatile("global_load_ushort %0, %1, %2 offset:2" : "=v"(P.qh[i][1]) : "v"(hoff0), "s"(bh) : "memory");
          asm volatile("global_load_ushort %0, %1, %2 offset:4" : "=v"(P.qh[i][2]) : "v"(hoff0), "s"(bh) : "memory");
          asm volatile("global_load_dword %0, %1, %2" : "=v"(P.gbv[i]) : "v"(goff0), "s"(bg) : "memory");
        } else if (DEM32) {
          asm volatile("global_load_dwordx3 %0, %1, %2" : "=v"(P.qi[i]) : "v"(qoff0), "s"(bq) : "memory");
        } else {
          asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(P.ND[i][0]) : "v"(voff0), "s"(bd) : "memory");
          asm volatile("global_load_dwordx2 %0, %1, %2 offset:8" : "=v"(P.ND[i][1]) : "v"(voff0), "s"(bd) : "memory");
          asm volatile("global_load_dwordx2 %0, %1, %2 offset:16" : "=v"(P.ND[i][2]) : "v"(voff0), "s"(bd) : "memory");
        }
      }
    };
    // the wait that makes the prefetched registers readable; every register is an in/out operand so
    // no use can be scheduled above it.  YOUNGER = memory operations issued after the loads.
#define WDPM_WAIT_W "+v"(P.NW[0][0]), "+v"(P.NW[0][1]), "+v"(P.NW[0][2]), "+v"(P.NW[1][0]), "+v"(P.NW[1][1]), \
                    "+v"(P.NW[1][2]), "+v"(P.NW[2][0]), "+v"(P.NW[2][1]), "+v"(P.NW[2][2])
#define WDPM_WAIT_ROWS(YOUNGER)                                                                        \
  do {                                                                                                 \
    if (!DEM32)                                                                                        \
      asm volatile("s_waitcnt vmcnt(" #YOUNGER ")"                                                     \
                   : WDPM_WAIT_W, "+v"(P.ND[0][0]), "+v"(P.ND[0][1]), "+v"(P.ND[0][2]), "+v"(P.ND[1][0]), \
                     "+v"(P.ND[1][1]), "+v"(P.ND[1][2]), "+v"(P.ND[2][0]), "+v"(P.ND[2][1]),           \
                     "+v"(P.ND[2][2])                                                                  \
                   :                                                                                   \
                   : "memory");                                                                        \
    else if (DEM32 == 2)                                                                               \
      asm volatile("s_waitcnt vmcnt(" #YOUNGER ")"                                                     \
                   : WDPM_WAIT_W, "+v"(P.qh[0][0]), "+v"(P.qh[0][1]), "+v"(P.qh[0][2]), "+v"(P.qh[1][0]), "+v"(P.qh[1][1]), \
                     "+v"(P.qh[1][2]), "+v"(P.qh[2][0]), "+v"(P.qh[2][1]), "+v"(P.qh[2][2]), "+v"(P.gbv[0]), "+v"(P.gbv[1]), \
                     "+v"(P.gbv[2])                                                                    \
                   :                                                                                   \
                   : "memory");                                                                        \
    else                                                                                               \
      asm volatile("s_waitcnt vmcnt(" #YOUNGER ")"                                                     \
                   : WDPM_WAIT_W, "+v"(P.qi[0]), "+v"(P.qi[1]), "+v"(P.qi[2])                          \
                   :                                                                                   \
                   : "memory");                                                                        \
  } while (0)

    // the three staged rows [rb, rb+2] from LDS to HBM: 9 unconditional stores.  In two halves: the LDS reads go out at
    // the very top of a step, ahead of the decode work, so that their round trip is over when the stores want the values
    // (issued right in front of the stores, each step stalled on it: +0.4 % at 8192^2, profiles/r02/early_lds_ab.txt)
    double staged[3][3];
    auto read_staged = [&]() {
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) staged[i][k] = stage_lds[i * kStripIn + scol[k]];
    };
    auto write_staged = [&](const int rb) {
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const int r = rb + i;
        const bool row_ok = r >= or_lo && r <= or_hi;             // wave-uniform
        // rows outside the chunk's output range (first / last trips only) go to the dump area
#ifdef WDPM_ABLATE_HBM
        char *const orow = row_ok ? reinterpret_cast<char *>(wout + (size_t)((WDPM_ABLATE_HBM & 2) ? r % 48 : r) * pitch + c0) : dump;
#else
        char *const orow = row_ok ? reinterpret_cast<char *>(wout + (size_t)r * pitch + c0) : dump;   // wave-uniform
#endif
        // Stores as inline asm, saddr form: a wave-uniform row base in SGPRs + the lane's byte offset, no address arithmetic on
        // the vector unit.  Ordinary stores, not non-temporal ones (round 4).  Rounds 1 - 3 believed they were choosing between
        // the two by size: left to the compiler, the two arms of that branch (the same store with and without !nontemporal)
        // had been merged into ONE plain store with a 64-bit VALU address - round 3's final ISA has no `nt` store in this
        // kernel at all.  With real `nt` stores measured against plain ones (profiles/r04/stores_shapes_ab.txt): add 4096^2
        // 108.0 against 103.1 us, 3000^2 65.9 against 62.9, drain 8192^2 401.6 against 397.8, the 1053-row drain slab 77.7
        // against 75.2; a tie at 8192^2 add, on the 2116-row add slab and at 16384^2; drain 4096^2 128.9 against 130.3.
#pragma unroll
        for (int k = 0; k < 3; k++) asm volatile("global_store_dwordx2 %0, %1, %2" : : "v"(soff[k]), "v"(staged[i][k]), "s"(orow) : "memory");
      }
      __builtin_amdgcn_wave_barrier();
    };

    auto step = [&](const int n, Prefetched &P, auto nstages_tag) {
      constexpr int NSTAGES = decltype(nstages_tag)::value;
      int partner_at = 0;
      if (use_prio) {                                    // wave-uniform
        *my_progress = n;                                // every lane the same word
        partner_at = *partner_progress;                  // read back after the step's arithmetic
      }
      read_staged();
      // consume the prefetched rows into window slots 4..6; the device DEM already holds +inf for
      // NODATA cells, so only edge waves have anything to mask (outside the slab: dem=+inf, w=0)
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
          // FLUSH: the block's threshold flush (WDPMCL.c:1059-1062) applied to the water as it arrives
          W[4 + i][j] = FLUSH && P.NW[i][j] < thres ? 0.0 : P.NW[i][j];
          if (DEM32 == 2) D[4 + i][j] = dem16_decode_nan(P.qh[i][j], P.gbv[i], code.k0, code.D, code.rD);
          else if (DEM32) D[4 + i][j] = dem32_decode_nan(P.qi[i][j], code.k0, code.D, code.rD);
          else D[4 + i][j] = P.ND[i][j];
        }
      if (MODULE == 2 && owed_here && A + 3 * n + 2 >= g.dr - 1 && A + 3 * n <= g.dr + 1) {   // wave-uniform, rare
#pragma unroll
        for (int i = 0; i < 3; i++) {
          const bool owed_row = A + 3 * n + i >= g.dr - 1 && A + 3 * n + i <= g.dr + 1;
#pragma unroll
          for (int j = 0; j < 3; j++) W[4 + i][j] = owed_row && owed_col[j] ? 0.0 : W[4 + i][j];
        }
      }
      if (EDGE) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
          const bool row_ok = A + 3 * n + i < g.rows;
#pragma unroll
          for (int j = 0; j < 3; j++) {
            const bool ok = row_ok & (colb + j < g.ncp);
            W[4 + i][j] = ok ? W[4 + i][j] : 0.0;
            D[4 + i][j] = ok ? D[4 + i][j] : WDPM_INF;
          }
        }
      }
      if constexpr (!SZ_SAFE) {
        auto hi = [](const double v) { return (unsigned)__double2hiint(v); };
        auto max3 = [](const unsigned a, const unsigned b, const unsigned c) { const unsigned m = a > b ? a : b; return m > c ? m : c; };
        unsigned hm = max3(hi(W[4][0]), hi(W[4][1]), hi(W[4][2]));
        hm = max3(hm, hi(W[5][0]), hi(W[5][1]));
        hm = max3(hm, hi(W[5][2]), hi(W[6][0]));
        hm = max3(hm, hi(W[6][1]), hi(W[6][2]));
        deep = (deep & 8) | ((deep & 3) << 1) | (__ballot(hm > 0x40080000u) != 0 ? 1 : 0);   // 0x40080000'00000000 = 3.0
      }
      // always issued (the last trips re-read clamped / following rows and drop them)
      prefetch(P, A + 3 * (n + 1));
      // the rows the previous step staged in LDS go out now, behind the new loads and ahead of a
      // whole step of arithmetic (+1 % over storing at the end of the step); n = 0 has none: dump
      write_staged(A + 3 * (n - 1) - 4);

      const int rbase = A + 3 * n - 4;                 // slab row of window slot 0
#ifndef WDPM_ABLATE_COMPUTE                            /* timing experiments only: memory pattern alone */
      if (SZ_SAFE || deep) three_stages<MODULE, SZ_SAFE, NSTAGES, (PLAIN ? 1 : 0)>(W, D, rbase, g.dr, cdr, ds);
      else three_stages<MODULE, SZ_SAFE, NSTAGES, (PLAIN ? 1 : 0) | 2>(W, D, rbase, g.dr, cdr, ds);
#else
#pragma unroll
      for (int j = 0; j < 3; j++) W[0][j] += D[0][j] + D[1][j] + D[2][j];   // keep the dem loads alive
#endif

      // rows 3n-4 .. 3n-2 have now seen all nine passes.  The wave transposes each row through its
      // private LDS slice (no barrier: a wave's LDS operations complete in order) so that every
      // store instruction writes 512 contiguous bytes; the stores themselves are issued by write_staged()
      // at the top of the next step.
      if (MD) {
        // rows 3n-4 .. 3n-2 are final: their part of the block's max-change reduction, against the snapshot
        // (which may still be owed the block's flush: applied as it is read).  Only cells of this wave's own
        // output block, of the rows asked for, with bigdem > missingvalue - plus the reference's seed cell [0][0].
        const int hi = oc_hi < g.ncp - 1 ? oc_hi : g.ncp - 1;
#pragma unroll
        for (int i = 0; i < 3; i++) {
          const int r = rbase + i;
          const bool row_in = r >= or_lo && r <= or_hi && r >= md.row_lo && r < md.row_hi;     // wave-uniform
          const int rc = r < 0 ? 0 : (r < g.rows ? r : g.rows - 1);
#pragma unroll
          for (int j = 0; j < 3; j++) {
            const int c = colb + j;
            const int cc = c < g.ncp ? c : g.ncp - 1;
            double o = md.old[(size_t)rc * pitch + cc];
            o = o < md.thres ? 0.0 : o;                                                         // :1059-1062
            const double dd = __builtin_fabs(W[i][j] - o);                                      // :1241
            const bool cell = row_in & (c >= oc_lo) & (c <= hi) & ((D[i][j] < WDPM_INF) | ((r == 0) & (c == 0)));
            md_max = (cell & (dd > md_max)) ? dd : md_max;                                        // :1245-1254
          }
        }
      }
      unsigned any_lo = 0, any_hi = 0;
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
          stage_lds[i * kStripIn + 3 * lane + j] = W[i][j];
          // "is any staged value something other than +0.0?": the OR of the nine bit images (v_or3_b32: nine instructions and one
          // compare per step, where a compare per value took twenty)
          any_lo |= (unsigned)__double2loint(W[i][j]);
          any_hi |= (unsigned)__double2hiint(W[i][j]);
        }
      nzmask |= __ballot(((any_lo | any_hi) != 0) & (colb < g.ncp));      // (lanes beyond the raster's right edge hold other rows' cells)
      __builtin_amdgcn_wave_barrier();
      if (use_prio) {
        const int p = __builtin_amdgcn_readfirstlane(partner_at);
        if (p > n) __builtin_amdgcn_s_setprio(3);        // the other wave of this SIMD is ahead: this one is served first
        else if (p < n) __builtin_amdgcn_s_setprio(0);
        else __builtin_amdgcn_s_setprio(1);
      }
      // The rows requested at the top of this step must have landed before the compiler may touch
      // their registers (it copies them around the loop back-edge): wait here, where the only
      // younger memory operations are the 9 stores issued right after them.
      WDPM_WAIT_ROWS(9);
      // slide the window down three rows
#pragma unroll
      for (int k = 0; k < 4; k++)
#pragma unroll
        for (int j = 0; j < 3; j++) { W[k][j] = W[k + 3][j]; D[k][j] = D[k + 3][j]; }
    };

    // The SIMD's arbiter serves the OLDER of its two waves first whenever both have an instruction ready.  Left alone, the older
    // wave of every SIMD gets ~65 % of the issue slots, finishes its chunk at 0.65 of the launch's time, and the younger runs
    // the last third alone with nobody to hide its latencies behind (tools/wave_times.py; 4096^2 in round 4: slot 0 ends at
    // 55.7 us, slot 1 at 85.2 us of a 90.4 us launch, wave-time in flight / (waves x span) = 0.77).
    // Round 3 answered with priorities that FALL as a wave advances through its chunk (four levels at fixed fractions, the
    // fractions different for the two slots, from a two-wave model of the arbiter): 0.96 in flight at 16384^2 (+5.6 %), but nothing
    // or a loss on chunks of fewer than ~40 steps - 4096^2, every 8-GPU slab - and an instantiation of its own.
    // Round 4: the two waves TELL each other where they are.  With two waves per SIMD a workgroup is eight waves, one workgroup
    // per CU, so the two waves of a SIMD share LDS: each step a wave writes its step number to its word (SIMD id x 2 + the
    // seat it took there at the start of the kernel) and reads the other's; whoever is behind gets priority 3, whoever is ahead 0,
    // level 1 when they are level (the older then leads by a step, and is overtaken).  Two LDS operations and a few scalar
    // instructions per step, no assumption about chunk heights or about who started first; a partner that has finished keeps its
    // last step number and reads as "ahead" until it is passed - either way only who issues first is decided here, never what is computed.
    // `prio` = 0 (WDPM_PRIO=0, or a launch of four-wave workgroups): no priorities.
    Prefetched P;
    prefetch(P, A);
    WDPM_WAIT_ROWS(0);
    // The dead stages of a chunk's first two steps are left out (see three_stages): 3 of 3 (H/3 + 2) stage executions.
    step(0, P, std::integral_constant<int, 1>{});      // nsteps >= 3: H >= 3
    step(1, P, std::integral_constant<int, 2>{});
    for (int n = 2; n < nsteps; n++) step(n, P, std::integral_constant<int, 3>{});
    read_staged();
    write_staged(A + 3 * (nsteps - 1) - 4);    // the last step's rows
#undef WDPM_WAIT_ROWS
#undef WDPM_WAIT_W
  };
  if (edge) march(std::true_type{});
  else march(std::false_type{});

  if (MODULE == 2 && owner && lane == 0) *totaldrain = ds.td;
  if (MD) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double o = __shfl_xor(md_max, off, 64);
      md_max = o > md_max ? o : md_max;
    }
    if (lane == 0 && md_max > 0.0) atomicMax(md.bits, (unsigned long long)__double_as_longlong(md_max));   // >= 0: order-preserving
  }
  if (tf.zout && lane == 0) {
    // every staged row went into the mask, the warm-up rows above the block included: a flag of 0 only says "unknown"
    tf.zout[tile] = nzmask ? 0 : 1;
    atomicAdd(tf.active, 1u);
  }
  if (bal.acc && lane == 0) {
    // class 0 .. 7: the XCD (workgroups are dealt round-robin over them); class 8: the waves of a strip's last chunk, whatever
    // their XCD - they take the clamping loads and the masks of the slab's lower edge on every step and end ~5 % late
    // (with `rot`: by physical XCD, and only where the share this workgroup took is the one sized for the XCD it ran on)
    const int x = chunk == tf.nchunks - 1 ? 8 : chunk == 0 ? 9 : (bal.rot ? xcc_phys : (int)(blockIdx.x & 7));
    if (!bal.rot || share == xcc_phys) {
      atomicAdd(bal.acc + x, wall_clock64() - bal_t0);
      atomicAdd(bal.acc + kBalClasses + x, 1ull);
    }
  }
#ifdef WDPM_WAVE_TIMES
  if (lane == 0 && item < 8192) {
    __builtin_amdgcn_s_waitcnt(0);   // the wave's stores have left
    g_wave_times[4 * item] = wt0;
    g_wave_times[4 * item + 1] = wall_clock64();
    g_wave_times[4 * item + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32);  // HW_ID, XCC_ID
    g_wave_times[4 * item + 3] = ((unsigned long long)strip << 48) | ((unsigned long long)(nsteps & 0xffff) << 32) | (unsigned)chunk;   /* (the marching kernel) */
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// Small rasters: the "triangle" kernel.
//
// A raster of a few hundred rows cannot fill the chip, and the marching kernel above then runs at its
// latency floor: every wave needs H/3 + 2 window steps of nine DEPENDENT 3x3 blocks each (one lane works
// on one block at a time), about 10 us per launch on a 482 x 471 raster however many SIMDs idle.  Here a
// wave takes NINE rows at once and produces the three middle ones in one go:
//       oi = 1 on rows 0-2, 3-5, 6-8      three independent row blocks ...
//       oi = 2 on rows 1-3, 4-6           ... two ...
//       oi = 3 on rows 2-4                ... one: rows 2-4 have then seen all nine passes
// Six stages instead of the marching kernel's nine for a 3-row chunk, and the row blocks of a stage are
// advanced in LOCKSTEP - neighbour step k of block a, of block b, of block c, then step k+1 ... - so the
// fp64 pipeline sees up to three independent dependency chains instead of one.
// Same trapezoid, same pass order per cell, same arithmetic (stage_impl's blocks): bit-identical.
// ---------------------------------------------------------------------------------------------
/* NB independent 3x3 blocks of one colour pass, advanced in lockstep (non-outlet form; block b is
 * w[b][row][col] with the centre at [1][1], neighbours in row-major order) */
template <int MODULE, int NB, int FLAGS = 0>   /* FLAGS as block_update's: bit 0 PLAIN, bit 1 CLAMP */
__device__ __forceinline__ void blocks_lockstep(double (&w)[NB][3][3], const double (&d)[NB][3][3]) {
  constexpr bool PLAIN = (FLAGS & 1) != 0, CL = (FLAGS & 2) != 0;
  double wc[NB], dce[NB];
  bool gate[NB];
#pragma unroll
  for (int b = 0; b < NB; b++) {
    wc[b] = w[b][1][1];
    gate[b] = PLAIN || ((wc[b] > 0.0) & (d[b][1][1] < WDPM_INF));  // WDPMCL.c:1099 (PLAIN: see block_update)
    dce[b] = gate[b] ? d[b][1][1] : -WDPM_INF;                      // see flow_add_nz / flow_drain_nz
    if (MODULE == 2) wc[b] = gate[b] ? wc[b] : 0.0;
  }
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int r = 1 + nb_dr(k), c = 1 + nb_dc(k);
#pragma unroll
    for (int b = 0; b < NB; b++) {
      if (MODULE == 2) flow_drain_nz<CL>(dce[b], wc[b], d[b][r][c], w[b][r][c]);
      else flow_add_nz<CL>(dce[b], wc[b], d[b][r][c], w[b][r][c]);
    }
  }
#pragma unroll
  for (int b = 0; b < NB; b++) w[b][1][1] = MODULE == 2 ? (gate[b] ? wc[b] : w[b][1][1]) : wc[b];
}

/* one row alignment (three column alignments oj = 1,2,3) on the NB row blocks at window slots
 * S0 + 3b .. S0 + 3b + 2 of a window of NR rows; the lockstep twin of stage_impl */
template <int MODULE, int NB, int S0, int NR, int FLAGS = 0>
__device__ __forceinline__ void stage_lockstep(double (&W)[NR][3], const double (&D)[NR][3]) {
  double w[NB][3][3], d[NB][3][3];
  double n0[NB][3], n1[NB][3], e0[NB][3], e1[NB][3];      // columns 0 and 1 of the next lane: water, elevation
  // oj = 1: own columns 0,1,2
#pragma unroll
  for (int b = 0; b < NB; b++)
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) { w[b][r][c] = W[S0 + 3 * b + r][c]; d[b][r][c] = D[S0 + 3 * b + r][c]; }
  blocks_lockstep<MODULE, NB, FLAGS>(w, d);
  // oj = 2: own columns 1,2 + column 0 of the next lane
#pragma unroll
  for (int b = 0; b < NB; b++)
#pragma unroll
    for (int r = 0; r < 3; r++) {
      n0[b][r] = lane_next(w[b][r][0]);
      e0[b][r] = lane_next(d[b][r][0]);
      W[S0 + 3 * b + r][0] = w[b][r][0];                  // column 0 is done for this row alignment (until handed back)
      w[b][r][0] = w[b][r][1]; w[b][r][1] = w[b][r][2]; w[b][r][2] = n0[b][r];
      d[b][r][0] = d[b][r][1]; d[b][r][1] = d[b][r][2]; d[b][r][2] = e0[b][r];
    }
  blocks_lockstep<MODULE, NB, FLAGS>(w, d);
  // oj = 3: own column 2 + columns 0,1 of the next lane
#pragma unroll
  for (int b = 0; b < NB; b++)
#pragma unroll
    for (int r = 0; r < 3; r++) {
      n1[b][r] = lane_next(W[S0 + 3 * b + r][1] = w[b][r][0]);   // own column 1 as updated by oj = 1,2 ...
      e1[b][r] = lane_next(D[S0 + 3 * b + r][1]);                // ... the next lane's is the one we borrow
      w[b][r][0] = w[b][r][1]; w[b][r][1] = w[b][r][2]; w[b][r][2] = n1[b][r];
      d[b][r][0] = d[b][r][1]; d[b][r][1] = d[b][r][2]; d[b][r][2] = e1[b][r];
    }
  blocks_lockstep<MODULE, NB, FLAGS>(w, d);
  // own column 2 back into the window; the borrowed columns back to lane+1 (lane 0 keeps its own)
#pragma unroll
  for (int b = 0; b < NB; b++)
#pragma unroll
    for (int r = 0; r < 3; r++) {
      W[S0 + 3 * b + r][2] = w[b][r][0];
      W[S0 + 3 * b + r][0] = lane_prev(w[b][r][1], W[S0 + 3 * b + r][0]);
      W[S0 + 3 * b + r][1] = lane_prev(w[b][r][2], W[S0 + 3 * b + r][1]);
    }
}

template <int MODULE, bool FLUSH, int K = 1, bool PLAIN = false, bool MD = false>
__global__ void __launch_bounds__(256, 2)
tri_iteration_kernel(const double *__restrict__ win, double *__restrict__ wout, const double *__restrict__ dem,
                     const SlabGeom g, const int nstrips, const int nitems, const int A0, const int out_last,
                     double *__restrict__ totaldrain, const double thres, const int drain_owed, const MaxDiffArgs md) {
  static_assert(!MD || (K == 1 && MODULE != 2), "the max-diff variant exists for three rows per wave, add / subtract");
  // K row blocks out per wave: 3K + 6 rows in, oi = 1 on K + 2 row blocks, oi = 2 on K + 1, oi = 3 on K.  K = 1 is the kernel
  // described above; K = 2 (add / subtract only) does 9 block stages for six rows instead of 12 - for rasters whose waves no
  // longer fit on the chip in one round, where the launch is bound by instruction issue, not by a single wave's latency.
  static_assert(K == 1 || MODULE != 2, "the outlet's permutation below is written for K = 1");
  constexpr int NR = 3 * K + 6;
  const int lane = threadIdx.x & 63;
  const int vb = (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8;       // XCD-contiguous items
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int item = vb * 4 + wave;
  if (item >= nitems) return;                                               // wave-uniform
  const int strip = item % nstrips, chunk = item / nstrips;
  const int c0 = kStripOut * strip;
  const int oc_lo = strip == 0 ? 0 : c0 + kHaloL;
  int oc_hi = c0 + kStripIn - 1 - kHaloR;
  if (oc_hi > g.ncp - 1) oc_hi = g.ncp - 1;
  const int A = A0 + 3 * K * chunk;               // window rows A .. A+NR-1; exact output rows A+2 .. A+3K+1
  const int or_lo = A == 0 ? 0 : A + 2;
  int or_hi = A + 3 * K + 1;
  if (or_hi > out_last) or_hi = out_last;
  const int colb = c0 + 3 * lane;
  const size_t pitch = (size_t)g.ncp;

  DrainState ds;
  ds.td = MODULE == 2 ? *totaldrain : 0.0;
  ds.hit = false;
  bool owner = false;
  const bool outlet_inside = g.dr >= 1 && g.dr <= g.rows - 2 && g.dc >= 1 && g.dc <= g.ncp - 2;
  if (MODULE == 2) {
    owner = g.dr >= or_lo && g.dr <= or_hi && g.dc >= oc_lo && g.dc <= oc_hi;
    if (drain_owed && owner && outlet_inside) {                              // the previous iteration's drain(), see above
      double sum = 0.0;
#pragma unroll
      for (int i = -1; i <= 1; i++)
#pragma unroll
        for (int j = -1; j <= 1; j++) {
          const size_t k = (size_t)(g.dr + i) * g.ncp + (g.dc + j);
          const double wk = win[k];
          if (dem[k] < WDPM_INF && wk > 0) sum += wk;
        }
      ds.td = ds.td + sum;
    }
  }

  // nine rows x three columns per lane; cells outside the slab: dem = +inf, water = 0.  Row bases are
  // wave-uniform (scalar registers), the lane contributes a 32-bit byte offset: saddr-form loads, no 64-bit
  // address arithmetic on the vector unit.  Only waves at the slab's right / lower edge mask anything.
  const bool edge = (c0 + kStripIn > g.ncp) || (A + NR > g.rows);                  // wave-uniform
  unsigned voff[3];
#pragma unroll
  for (int j = 0; j < 3; j++) voff[j] = 8u * (unsigned)(colb + j < g.ncp ? colb + j : g.ncp - 1);
  double W[NR][3], D[NR][3];
#pragma unroll
  for (int i = 0; i < NR; i++) {
    const int rc = A + i < g.rows ? A + i : g.rows - 1;
    const char *bw = reinterpret_cast<const char *>(win + (size_t)rc * pitch);
    const char *bd = reinterpret_cast<const char *>(dem + (size_t)rc * pitch);
#pragma unroll
    for (int j = 0; j < 3; j++) {
      W[i][j] = *reinterpret_cast<const double *>(bw + voff[j]);
      D[i][j] = *reinterpret_cast<const double *>(bd + voff[j]);
    }
  }
  // MD (round 3): the block's last launch of a small raster stays with this kernel - the snapshot's rows behind the rows this
  // wave will store, requested with the window (the snapshot may still be owed the block's flush: applied as it is compared)
  double O[3 * K + 2][3];
  if (MD) {
#pragma unroll
    for (int i = 0; i < 3 * K + 2; i++) {
      const int rc = A + i < g.rows ? A + i : g.rows - 1;
      const char *bo = reinterpret_cast<const char *>(md.old + (size_t)rc * pitch);
#pragma unroll
      for (int j = 0; j < 3; j++) O[i][j] = *reinterpret_cast<const double *>(bo + voff[j]);
    }
  }
  if (FLUSH) {
#pragma unroll
    for (int i = 0; i < NR; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) W[i][j] = W[i][j] < thres ? 0.0 : W[i][j];       // WDPMCL.c:1059-1062
  }
  if (MODULE == 2 && drain_owed && outlet_inside && A + NR - 1 >= g.dr - 1 && A <= g.dr + 1) {   // wave-uniform, rare
#pragma unroll
    for (int i = 0; i < NR; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const int r = A + i, c = colb + j;
        W[i][j] = (r >= g.dr - 1 && r <= g.dr + 1 && c >= g.dc - 1 && c <= g.dc + 1) ? 0.0 : W[i][j];   // :1885-1889
      }
  }
  if (edge) {
#pragma unroll
    for (int i = 0; i < NR; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const bool ok = (A + i < g.rows) & (colb + j < g.ncp);
        W[i][j] = ok ? W[i][j] : 0.0;
        D[i][j] = ok ? D[i][j] : WDPM_INF;
      }
  }

  bool outlet_here = false;
  if constexpr (MODULE == 2) outlet_here = g.dr >= A && g.dr <= A + 8;
  if constexpr (MODULE == 2) if (outlet_here) {
    // The outlet's row is in this window (a handful of waves).  In every row alignment at most ONE row block
    // holds that row (row blocks of an alignment are disjoint, so totaldrain still accumulates in pass order):
    // it runs the marching kernel's stage with runoffd()'s sink, the other blocks of the alignment stay in
    // lockstep.  Which block it is is wave-uniform but only known at run time: the window is permuted with
    // selects so that one copy of the code serves every position.
    bool cdr[5];
#pragma unroll
    for (int j = 0; j < 5; j++) cdr[j] = colb + j == g.dc;
    const int off = g.dr - A;                                   // 0 .. 8
    auto sel = [](const bool c, const double a, const double b) { return c ? a : b; };
    double P[9][3], Q[9][3];
    {  // oi = 1: row blocks at slots 0, 3, 6; the outlet's is block ob = off / 3
      const int ob = off / 3;
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
          P[r][j] = sel(ob == 0, W[r][j], sel(ob == 1, W[3 + r][j], W[6 + r][j]));
          Q[r][j] = sel(ob == 0, D[r][j], sel(ob == 1, D[3 + r][j], D[6 + r][j]));
          P[3 + r][j] = sel(ob == 0, W[3 + r][j], W[r][j]);         Q[3 + r][j] = sel(ob == 0, D[3 + r][j], D[r][j]);
          P[6 + r][j] = sel(ob == 2, W[3 + r][j], W[6 + r][j]);     Q[6 + r][j] = sel(ob == 2, D[3 + r][j], D[6 + r][j]);
        }
      double Wm[7][3], Dm[7][3];
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int j = 0; j < 3; j++) { Wm[r][j] = P[r][j]; Dm[r][j] = Q[r][j]; }
      stage_impl<2, false, 0, true>(Wm, Dm, A + 3 * ob, g.dr, cdr, ds);
      stage_lockstep<2, 2, 3, 9>(P, Q);
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
          const double o = Wm[r][j], x = P[3 + r][j], y = P[6 + r][j];   // outlet block, first other, second other
          W[r][j] = sel(ob == 0, o, x);
          W[3 + r][j] = sel(ob == 1, o, sel(ob == 0, x, y));
          W[6 + r][j] = sel(ob == 2, o, y);
        }
    }
    {  // oi = 2: row blocks at slots 1 and 4 (rows 1-3, 4-6); the outlet's row is in one of them for off = 1 .. 6
      const int ob = off >= 1 && off <= 6 ? (off - 1) / 3 : -1;
      if (ob < 0) {
        stage_lockstep<2, 2, 1, 9>(W, D);
      } else {
        double Wm[7][3], Dm[7][3], Wo[3][3], Do[3][3];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int j = 0; j < 3; j++) {
            Wm[r][j] = sel(ob == 0, W[1 + r][j], W[4 + r][j]);   Dm[r][j] = sel(ob == 0, D[1 + r][j], D[4 + r][j]);
            Wo[r][j] = sel(ob == 0, W[4 + r][j], W[1 + r][j]);   Do[r][j] = sel(ob == 0, D[4 + r][j], D[1 + r][j]);
          }
        stage_impl<2, false, 0, true>(Wm, Dm, A + 1 + 3 * ob, g.dr, cdr, ds);
        stage_lockstep<2, 1, 0, 3>(Wo, Do);
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int j = 0; j < 3; j++) {
            W[1 + r][j] = sel(ob == 0, Wm[r][j], Wo[r][j]);
            W[4 + r][j] = sel(ob == 0, Wo[r][j], Wm[r][j]);
          }
      }
    }
    // oi = 3: the one row block at slot 2 (rows 2-4)
    if (off >= 2 && off <= 4) {
      double Wm[7][3], Dm[7][3];
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int j = 0; j < 3; j++) { Wm[r][j] = W[2 + r][j]; Dm[r][j] = D[2 + r][j]; }
      stage_impl<2, false, 0, true>(Wm, Dm, A + 2, g.dr, cdr, ds);
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int j = 0; j < 3; j++) W[2 + r][j] = Wm[r][j];
    } else {
      stage_lockstep<2, 1, 2, 9>(W, D);
    }
  }
  if (!outlet_here) {
    // (no clamped neighbour step here - see `deep` in the marching kernel: this kernel sits at 256 VGPRs, and a second copy of
    // the stages behind a test of the window's depths made five of its instantiations spill, 8 - 164 bytes, round 4)
    constexpr int F = PLAIN ? 1 : 0;
    stage_lockstep<MODULE, K + 2, 0, NR, F>(W, D);   // oi = 1 on rows 0-2, 3-5, 6-8 (, 9-11)
    stage_lockstep<MODULE, K + 1, 1, NR, F>(W, D);   // oi = 2 on rows 1-3, 4-6 (, 7-9)
    stage_lockstep<MODULE, K, 2, NR, F>(W, D);       // oi = 3 on rows 2-4 (, 5-7)
  }

  // rows or_lo .. or_hi, columns oc_lo .. oc_hi (slots 0 and 1 only for the raster's first rows)
#pragma unroll
  for (int i = 0; i < 3 * K + 2; i++) {
    const int r = A + i;
    if (r < or_lo || r > or_hi) continue;                                    // wave-uniform
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int c = colb + j;
      if (c >= oc_lo && c <= oc_hi) __builtin_nontemporal_store(W[i][j], wout + (size_t)r * pitch + c);
    }
  }
  if (MODULE == 2 && owner && lane == 0) *totaldrain = ds.td;
  if (MD) {
    // max |w - oldw| over the cells this wave stored, of the rows asked for, with bigdem > missingvalue - plus the reference's
    // seed cell [0][0] (WDPMCL.c:1239-1254); `if (d > m)` per lane, wave maximum, one atomicMax
    double md_max = 0.0;
#pragma unroll
    for (int i = 0; i < 3 * K + 2; i++) {
      const int r = A + i;
      const bool row_in = r >= or_lo && r <= or_hi && r >= md.row_lo && r < md.row_hi;           // wave-uniform
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const int c = colb + j;
        double o = O[i][j];
        o = o < md.thres ? 0.0 : o;                                                               // :1059-1062
        const double dd = __builtin_fabs(W[i][j] - o);                                            // :1241
        const bool cell = row_in & (c >= oc_lo) & (c <= oc_hi) & ((D[i][j] < WDPM_INF) | ((r == 0) & (c == 0)));
        md_max = (cell & (dd > md_max)) ? dd : md_max;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double o = __shfl_xor(md_max, off, 64);
      md_max = o > md_max ? o : md_max;
    }
    if (lane == 0 && md_max > 0.0) atomicMax(md.bits, (unsigned long long)__double_as_longlong(md_max));
  }
}

// ---------------------------------------------------------------------------------------------
// Smallest rasters: the "relay" kernel (round 3).  A workgroup of four waves takes twelve rows of one strip and
// produces the middle six; each wave carries ONE row block through a row alignment - a single dependent chain per lane, which
// runs at its latency, where the triangle kernel's wave issues three, two, one blocks in lockstep and is bound by instruction
// issue - and the rows two alignments share pass from wave to wave through LDS:
//       oi = 1   wave w on rows 3w .. 3w+2                      (w = 0..3)
//       row 3w+3 (wave w+1's first row) -> wave w               LDS + s_barrier
//       oi = 2   wave w on rows 3w+1 .. 3w+3                    (w = 0..2)
//       row 3w+4 (wave w+1's second row) -> wave w              LDS + s_barrier
//       oi = 3   wave w on rows 3w+2 .. 3w+4, stored            (w = 0, 1)
// Nine block stages for six rows, as in the six-row triangle wave, but three in a row per wave instead of nine.  Same trapezoid,
// same pass order per cell, same arithmetic (stage_lockstep with one block): bit-identical.
// ---------------------------------------------------------------------------------------------
template <int MODULE, bool FLUSH, bool PLAIN, int NW = 4, bool DEM32 = false>
__global__ void __launch_bounds__(64 * NW, 2)
relay_iteration_kernel(const double *__restrict__ win, double *__restrict__ wout, const double *__restrict__ dem,
                       const SlabGeom g, const int nstrips, const int nwg, const int A0, const int out_last, const double thres,
                       double *__restrict__ totaldrain, const int drain_owed, const DemCode code, const int store_plain) {
  static_assert(!DEM32 || (MODULE != 2 && NW == 8), "the DEM as 32-bit codes: add / subtract launches of several rounds");
  const int lane = threadIdx.x & 63;
  const int vb = (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8;       // XCD-contiguous workgroups
  if (vb >= nwg) return;                                                    // workgroup-uniform: nobody is left at a barrier
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int strip = vb % nstrips, chunk = vb / nstrips;
  const int c0 = kStripOut * strip;
  const int oc_lo = strip == 0 ? 0 : c0 + kHaloL;
  int oc_hi = c0 + kStripIn - 1 - kHaloR;
  if (oc_hi > g.ncp - 1) oc_hi = g.ncp - 1;
  // NW waves: rows A .. A + 3 NW - 1 in, the middle 3 (NW - 2) out (NW = 4: twelve in, six out; NW = 8: 24 in, 18 out - fewer
  // waves and less re-reading per row where the waves no longer fit on the chip at once)
  constexpr int kOut = 3 * (NW - 2);
  const int A = A0 + kOut * chunk;                // exact output rows A+2 .. A+kOut+1
  const int or_lo = A == 0 ? 0 : A + 2;
  int or_hi = A + kOut + 1;
  if (or_hi > out_last) or_hi = out_last;
  const int R0 = A + 3 * wave;                    // this wave's rows R0 .. R0+2, later up to R0+4
  const int colb = c0 + 3 * lane;
  const size_t pitch = (size_t)g.ncp;
#ifdef WDPM_WAVE_TIMES
  unsigned long long rst[8];
  rst[0] = wall_clock64();
#endif
  const bool rprio = (store_plain & 2) != 0;      // wave-uniform: see the stages
  if (rprio) __builtin_amdgcn_s_setprio(3);       // the loads of a new workgroup go out ahead of an older one's arithmetic
  unsigned voff[3];
#pragma unroll
  for (int j = 0; j < 3; j++) voff[j] = 8u * (unsigned)(colb + j < g.ncp ? colb + j : g.ncp - 1);
  double W[7][3], D[7][3];                        // slots 0 .. 4 are used (seven: stage_impl's window type)
  // Every wave fetches its own three rows of water and of elevations.  The elevations of the two rows it takes over later are
  // static data the wave below holds anyway: they come with the water through LDS (LDSDEM) where the launch is more than a
  // round of waves - five rows of DEM fetched per wave made those memory-bound (add 2000^2 32.9 -> 30.0 us, 3000^2 66.7 -> 63.0,
  // drain 1200^2 15.0 -> 14.3) - and for drain; add / subtract workgroups of four waves, the one-round sizes, fetch all five
  // themselves: there the extra LDS traffic sits on the latency path (482^2 5.23 against 5.38 us; profiles/r03/relay_ldsdem_ab.txt)
  constexpr bool LDSDEM = NW == 8 || MODULE == 2;
#pragma unroll
  for (int i = 0; i < (LDSDEM ? 3 : 5); i++) {
    const int rc = R0 + i < g.rows ? R0 + i : g.rows - 1;
    const char *bw = reinterpret_cast<const char *>(win + (size_t)rc * pitch);
    const char *bd = reinterpret_cast<const char *>(dem + (size_t)rc * pitch);
    const char *bq = reinterpret_cast<const char *>(code.q + (size_t)rc * pitch);
#pragma unroll
    for (int j = 0; j < 3; j++) {
      // DEM32: the elevations as verified-lossless 32-bit codes (wdpm_kernels.h::DemCode): launches of several rounds are bound
      // by the memory system, and four bytes per cell less are worth nine decodes per wave there
      if (DEM32) D[i][j] = dem32_decode_nan(*reinterpret_cast<const int *>(bq + voff[j] / 2), code.k0, code.D, code.rD);
      else D[i][j] = *reinterpret_cast<const double *>(bd + voff[j]);
      W[i][j] = i < 3 ? *reinterpret_cast<const double *>(bw + voff[j]) : 0.0;
    }
  }
#pragma unroll
  for (int i = (LDSDEM ? 3 : 5); i < 7; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) { W[i][j] = 0.0; D[i][j] = WDPM_INF; }

  // Drain: totaldrain is carried from row alignment to row alignment by whichever wave holds the outlet's block (at most one
  // per alignment and workgroup: wave-uniform arithmetic every wave can do), through LDS; the workgroup whose stored block
  // holds the outlet has seen every pass that touches it, in the reference's order, and writes it back.
  DrainState ds;
  ds.td = 0.0;
  ds.hit = false;
  bool owner = false;
  int wo[3] = {-1, -1, -1};                       // wave that holds the outlet's row in its block of alignment oi = 1, 2, 3
  bool cdr[5];
#pragma unroll
  for (int j = 0; j < 5; j++) cdr[j] = MODULE == 2 && colb + j == g.dc;
  if (MODULE == 2) {
    const bool outlet_inside = g.dr >= 1 && g.dr <= g.rows - 2 && g.dc >= 1 && g.dc <= g.ncp - 2;
    const bool col_here = g.dc >= c0 && g.dc <= c0 + kStripIn + 1;      // the strip and the two columns lane 63 borrows
#pragma unroll
    for (int st = 0; st < 3; st++) {
      const int rel = g.dr - A - st;              // alignment st + 1: wave w on rows A + 3w + st .. + 2
      if (col_here && rel >= 0 && rel / 3 <= NW - 1 - st) wo[st] = rel / 3;
    }
    owner = g.dr >= or_lo && g.dr <= or_hi && g.dc >= oc_lo && g.dc <= oc_hi;
    ds.td = *totaldrain;
    if (drain_owed && outlet_inside) {
      if (owner) {                                // the previous iteration's drain() (see fused_iteration_kernel): every wave of the
        double sum = 0.0;                         // owning workgroup adds it up for itself
#pragma unroll
        for (int i = -1; i <= 1; i++)
#pragma unroll
          for (int j = -1; j <= 1; j++) {
            const size_t k = (size_t)(g.dr + i) * g.ncp + (g.dc + j);
            const double wk = win[k];
            if (dem[k] < WDPM_INF && wk > 0) sum += wk;
          }
        ds.td = ds.td + sum;
      }
      if (R0 + 2 >= g.dr - 1 && R0 <= g.dr + 1) {                                   // wave-uniform, rare
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = 0; j < 3; j++) {
            const int r = R0 + i, c = colb + j;
            W[i][j] = (r >= g.dr - 1 && r <= g.dr + 1 && c >= g.dc - 1 && c <= g.dc + 1) ? 0.0 : W[i][j];   // :1885-1889
          }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < (LDSDEM ? 3 : 5); i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      if (FLUSH && i < 3) W[i][j] = W[i][j] < thres ? 0.0 : W[i][j];               // WDPMCL.c:1059-1062
      const bool ok = (R0 + i < g.rows) & (colb + j < g.ncp);                      // outside the slab: dem = +inf, water = 0
      W[i][j] = ok ? W[i][j] : 0.0;
      D[i][j] = ok ? D[i][j] : WDPM_INF;
    }
  __shared__ double xch[2][NW][3 * kLanes];
  __shared__ double xdem[2][LDSDEM ? NW : 1][LDSDEM ? 3 * kLanes : 1];   // elevations of a wave's first two rows, for the wave above
  __shared__ double td_sh[3];
  __shared__ int deep_sh[NW];
  // The clamped neighbour step where the depths are shallow enough for it to be exact (see `deep` in the marching kernel: the same
  // bound).  A wave tests the three rows it loaded: enough for the first alignment, which stays inside them; the rows it takes
  // over afterwards have been through other waves' blocks, so from the first barrier on the workgroup's OR decides.  Wave-uniform.
  bool deep = (store_plain & 4) != 0;              // the host's part: elevations too large (WDPM_LAUNCH_CLAMP_OK not given)
  {
    unsigned hm = 0;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) { const unsigned h = (unsigned)__double2hiint(W[i][j]); hm = h > hm ? h : hm; }
    deep = deep || __ballot(hm > 0x40080000u) != 0;
    if (lane == 0) deep_sh[wave] = deep ? 1 : 0;
  }
  // one row alignment on this wave's block at slots S0 .. S0+2 (st = S0): the outlet's block takes block_update's outlet form
#define WDPM_RELAY_STAGE(S0)                                                                               \
  do {                                                                                                     \
    if (MODULE == 2 && wo[S0] == wave) {                                                                   \
      stage_impl<2, false, S0, true>(W, D, R0 + S0, g.dr, cdr, ds);                                        \
      if (lane == 0) td_sh[S0] = ds.td;                                                                    \
    } else if (MODULE == 2 || deep) {      /* drain: never clamped here - the second copy of the stages took the  */ \
      /* drain instantiations from 112 to 150 VGPRs, one eight-wave workgroup per CU instead of two: -17 % (round 4) */   \
      stage_lockstep<MODULE, 1, S0, 7, PLAIN ? 1 : 0>(W, D);                                               \
    } else {                                                                                               \
      stage_lockstep<MODULE, 1, S0, 7, (PLAIN ? 1 : 0) | 2>(W, D);                                         \
    }                                                                                                      \
  } while (0)
  // rprio (launches of a few rounds of workgroups: the host decides): a workgroup's issue priority falls from stage to stage, so that
  // of the workgroups sharing a CU's SIMDs the one behind is served first (the arbiter's own rule is oldest first: see the marching
  // loop).  profiles/r03/relay_prio_ab.txt: 1000^2 add 9.72 -> 9.11 us, drain 13.5 -> 12.0; 1200^2 and 1400^2 add +4.5 %; from five
  // rounds of workgroups on nothing or a loss (1800^2 drain -4 %, add -10 %; 2400^2 add -10 %: there the old workgroups' early exit is
  // what feeds the next round), so only launches of up to four rounds carry the flag
  if (rprio) __builtin_amdgcn_s_setprio(2);
  WDPM_RSTAMP(1);                                                             // rows loaded
  WDPM_RELAY_STAGE(0);                                                        // oi = 1
  WDPM_RSTAMP(2);
#pragma unroll
  for (int j = 0; j < 3; j++) {
    xch[0][wave][j * kLanes + lane] = W[0][j];
    if (LDSDEM) {
      xdem[0][wave][j * kLanes + lane] = D[0][j];
      xdem[1][wave][j * kLanes + lane] = D[1][j];
    }
  }
  __syncthreads();
  WDPM_RSTAMP(3);
#pragma unroll
  for (int w = 0; w < NW; w++) deep = deep || deep_sh[w] != 0;
  if (MODULE == 2 && wo[0] >= 0) ds.td = td_sh[0];
  if (wave < NW - 1) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
      W[3][j] = xch[0][wave + 1][j * kLanes + lane];
      if (LDSDEM) {
        D[3][j] = xdem[0][wave + 1][j * kLanes + lane];
        D[4][j] = xdem[1][wave + 1][j * kLanes + lane];
      }
    }
    if (rprio) __builtin_amdgcn_s_setprio(1);
    WDPM_RELAY_STAGE(1);                                                      // oi = 2
  }
  WDPM_RSTAMP(4);
#pragma unroll
  for (int j = 0; j < 3; j++) xch[1][wave][j * kLanes + lane] = W[1][j];
  __syncthreads();
  WDPM_RSTAMP(5);
#ifdef WDPM_WAVE_TIMES
  if (wave >= NW - 2 && lane == 0) {
    const int wid = ((int)blockIdx.x * NW + wave);
    if (wid < 4096) { for (int k = 0; k < 6; k++) g_wave_times[8 * wid + k] = rst[k]; g_wave_times[8 * wid + 6] = rst[5]; g_wave_times[8 * wid + 7] = rst[5]; }
  }
#endif
  if (wave >= NW - 2) return;
  if (MODULE == 2 && wo[1] >= 0) ds.td = td_sh[1];
#pragma unroll
  for (int j = 0; j < 3; j++) W[4][j] = xch[1][wave + 1][j * kLanes + lane];
  if (rprio) __builtin_amdgcn_s_setprio(0);
  WDPM_RELAY_STAGE(2);                                                        // oi = 3
  WDPM_RSTAMP(6);
#undef WDPM_RELAY_STAGE
  // wave w stores rows A+3w+2 .. A+3w+4 (wave 0 also rows 0, 1 of the raster's first chunk)
  if (LDSDEM) {
    // Through the wave's LDS slices (free since the second barrier), so that one store instruction writes 512 contiguous
    // bytes - as the marching kernel does.  Lane by lane a wave's three stores per row interleave at 24-byte strides, and as
    // non-temporal stores they reach the memory as partial lines: 52 MB written per 2000^2 launch for 32 MB of raster
    // (profiles/r03/add2000_pmc_summary.json; add 2000^2 28.9 -> 26.1 us, 3000^2 58.2 -> 50.8, relay_stores_ab.txt).  Lanes past
    // the exact output range are clamped onto its last column.  store_plain bit 0: ordinary instead of non-temporal stores (the
    // host sets it for launches of six rounds and more: relay_stores_ab.txt).
    double *const t[3] = {&xdem[0][wave][0], &xdem[1][wave][0], &xch[0][wave][0]};
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) t[i][3 * lane + j] = W[2 + i][j];
    __builtin_amdgcn_wave_barrier();
    const int lo = oc_lo - c0, hi = oc_hi - c0;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const int r = R0 + 2 + i;
      if (r < or_lo || r > or_hi) continue;                                         // wave-uniform
#pragma unroll
      for (int kk = 0; kk < 3; kk++) {
        const int c = lo + 64 * kk + lane < hi ? lo + 64 * kk + lane : hi;
        double *const q = wout + (size_t)r * pitch + c0 + c;
        if (store_plain & 1) *q = t[i][c];
        else __builtin_nontemporal_store(t[i][c], q);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < (LDSDEM ? 2 : 5); i++) {
    const int r = R0 + i;
    if (r < or_lo || r > or_hi || (i < 2 && !(wave == 0 && A == 0))) continue;    // wave-uniform
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int c = colb + j;
      if (c >= oc_lo && c <= oc_hi) __builtin_nontemporal_store(W[i][j], wout + (size_t)r * pitch + c);
    }
  }
  // the wave that ran the last alignment's outlet block holds the final value; if that alignment had none here, wave 0 does
  if (MODULE == 2 && owner && lane == 0 && wave == (wo[2] >= 0 ? wo[2] : 0)) *totaldrain = ds.td;
#ifdef WDPM_WAVE_TIMES
  WDPM_RSTAMP(7);
  if (lane == 0) {
    const int wid = ((int)blockIdx.x * NW + wave);
    if (wid < 4096) for (int k = 0; k < 8; k++) g_wave_times[8 * wid + k] = rst[k];
  }
#endif
}

/* wdpm_kernels.h::XcdBalance: (update) move the weights towards equal mean wave durations - weight 0 .. 7: the relative height of
 * the chunks whose work items run on that XCD (the physical one: BalanceArgs::rot), weight 8: a factor on a strip's last chunk (its
 * waves are the slab's lower edge and take the masking loop), weight 9: one on its first (round 5: the slab's first chunk row ends late);
 * `from_uniform`: the measured launches ran on equal heights, whatever the weights say - then rebuild the table of row boundaries
 * for the launch geometry given: per strip, heights in proportion to the weights of its chunks, in whole row triples, at least two
 * per chunk, the last boundary at the launch's last triple.  One workgroup; the sums are cleared for the next measurement. */
__global__ void __launch_bounds__(256)
xcd_rebalance_kernel(float *__restrict__ weight, unsigned long long *__restrict__ acc, int *__restrict__ table, const int nstrips,
                     const int nchunks, const int A0, const int out_last, const int ipx, const int update, const int from_uniform,
                     const int pair) {
  __shared__ float w[kBalClasses];
  if (threadIdx.x == 0) {
    float mean[kBalClasses], m = 0.f;
    bool all = true;
    for (int x = 0; x < kBalClasses; x++) {
      const float v = weight[x];
      w[x] = (v > 0.5f && v < 2.0f) ? v : 1.0f;
      const unsigned long long n = acc[kBalClasses + x];
      mean[x] = n ? (float)((double)acc[x] / (double)n) : 0.f;
      if (x < 8) { all = all && n > 0 && mean[x] > 0.f; m += mean[x]; }
    }
    if (update && all) {
      m *= 0.125f;
      float sum = 0.f;
      for (int x = 0; x < kBalClasses; x++) {
        if (x >= 8 && !(mean[x] > 0.f)) continue;            // (a launch of two chunk rows has no waves in a class of its own)
        const float base = from_uniform ? 1.0f : w[x];
        float r = m / mean[x];                               // > 1: these waves end early, they can take taller chunks
        r = r < 0.8f ? 0.8f : (r > 1.25f ? 1.25f : r);
        w[x] = base * (1.0f + 0.75f * (r - 1.0f));           // damped
        if (x < 8) sum += w[x];
      }
      for (int x = 0; x < 8; x++) {
        const float v = w[x] * 8.0f / sum;
        w[x] = v < 0.7f ? 0.7f : (v > 1.4f ? 1.4f : v);
      }
      for (int x = 8; x < kBalClasses; x++) w[x] = w[x] < 0.75f ? 0.75f : (w[x] > 1.2f ? 1.2f : w[x]);
    }
    for (int x = 0; x < kBalClasses; x++) { weight[x] = w[x]; acc[x] = 0; acc[kBalClasses + x] = 0; }
  }
  __syncthreads();
  const int T = (out_last - 1 - A0 + 2) / 3;                 // row triples the chunks of a strip share: A0 + 3 T >= out_last - 1
  auto wgt = [&](const int c, const int s) {
    const int x = (c * nstrips + s) / ipx;
    return w[x < 7 ? x : 7] * (c == nchunks - 1 ? w[8] : c == 0 ? w[9] : 1.0f);
  };
  for (int s = (int)threadIdx.x; s < nstrips; s += (int)blockDim.x) {
    float total = 0.f;
    for (int c = 0; c < nchunks; c++) total += wgt(c, s);
    float cum = 0.f;
    int prev = 0;
    table[s] = A0;
    // `pair` (round 5): heights come in whole row triples, so where a chunk is a few triples tall (an 8-GPU slab: 8.4) some chunks are
    // a triple taller than others - and a launch of one resident round ends with its busiest SIMD, i.e. with the pair of waves that
    // holds two tall chunks.  The two waves of a SIMD are waves w and w + 4 of an eight-wave workgroup, work items four apart: strips
    // s and s + 4 of one chunk row.  Rounding the boundaries of those two strips half a triple apart puts the tall chunks of one
    // beside the short chunks of the other (exactly so for equal weights: with a fractional height f <= 1/2 two tall chunks never
    // meet, with f > 1/2 two short ones never do), and every SIMD gets the same number of steps to within one.
    const float phase = pair ? (((s >> 2) & 1) ? 0.75f : 0.25f) : 0.5f;
    for (int c = 0; c < nchunks; c++) {
      cum += wgt(c, s);
      int t = c == nchunks - 1 ? T : (int)((float)T * cum / total + phase);
      const int lo = prev + 2, hi = T - 2 * (nchunks - 1 - c);
      t = t < lo ? lo : t;
      t = t > hi ? hi : t;
      table[(c + 1) * nstrips + s] = A0 + 3 * t;
      prev = t;
    }
  }
}

__global__ void dpp_probe_kernel(int *out) {
  const int lane = threadIdx.x;
  const double v = (double)lane;
  out[lane] = (int)lane_next(v);
  out[64 + lane] = (int)lane_prev(v, -2.0);
}

}  // namespace

/* Number of waves of the fused kernel the whole chip holds at once (CUs x blocks/CU x 4 waves),
 * from the occupancy API; cached per module.  All work items of a launch are made resident
 * together — one round, no tail — so the item count is sized to this. */
template <int MODULE, bool SZ_SAFE, int DEM32 = 0>
static int resident_waves() {
  static std::atomic<int> cached{0};      // rank threads of one process launch concurrently: no plain statics
  if (cached.load(std::memory_order_relaxed)) return cached.load(std::memory_order_relaxed);
  int dev = 0, cus = 256, blocks = 2;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, fused_iteration_kernel<MODULE, SZ_SAFE, DEM32>, 256, 0) != hipSuccess || blocks < 1)
    blocks = 2;
  // never more than the kernel was built for: the chunk heights and every threshold of the dispatch were measured at that
  // occupancy, and the register count an instantiation ends up with may allow more from one compiler run to the next
  constexpr int built_for = fused_built_for<MODULE, SZ_SAFE, DEM32, false>();
  if (blocks > built_for) blocks = built_for;
  cached.store(cus * blocks * 4, std::memory_order_relaxed);
  return cus * blocks * 4;
}

/* chunk height in rows (multiple of 3): the rows are cut into as many chunks as keep
 * strips x chunks within one resident round; each chunk pays a 6-row warm-up. */
static int pick_chunk_rows(const int rows, const int nstrips, const int override_rows, const int slots) {
  static std::atomic<int> env_h{-1};
  if (env_h < 0) {
    const char *e = getenv("WDPM_CHUNK_ROWS");
    env_h = e ? atoi(e) : 0;
  }
  int H;
  if (override_rows >= 3) {
    H = override_rows / 3 * 3;
  } else if (env_h >= 3) {
    H = env_h / 3 * 3;
  } else {
    int nch = slots / nstrips;
    if (nch < 1) nch = 1;
    H = ((rows - 2 + nch - 1) / nch + 2) / 3 * 3;
    // small rasters cannot fill the chip: short chunks then cost redundant warm-up rows on CUs that
    // would idle anyway and cut the launch's critical path (a wave's march) to a few steps - down to
    // one 3-row step of output per wave (basin5-sized rasters: 12.3 us per iteration against 14.3 with
    // 6-row chunks; the formula only gets there when all chunks still fit in one resident round)
  }
  if (H > rows) H = (rows + 2) / 3 * 3;
  if (H < 3) H = 3;
  return H;
}

static hipError_t dpp_selfcheck(hipStream_t s) {
  static std::atomic<int> state{0};   // 0 unknown, 1 ok, -1 bad
  if (state == 1) return hipSuccess;
  if (state == -1) return hipErrorUnknown;
  int *d = nullptr;
  int h[128];
  hipError_t e = hipMalloc(&d, sizeof h);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(dpp_probe_kernel, dim3(1), dim3(64), 0, s, d);
  e = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(d);
  if (e != hipSuccess) return e;
  bool ok = true;
  for (int l = 0; l < 64; l++) {
    ok = ok && h[l] == (l < 63 ? l + 1 : 0);
    ok = ok && h[64 + l] == (l > 0 ? l - 1 : -2);
  }
  state = ok ? 1 : -1;
  return ok ? hipSuccess : hipErrorUnknown;
}

/* The launches of small and mid-size rasters (relay and triangle kernels): *taken says whether one was queued.  Compiled as a
 * translation unit of its own (WDPM_TU == 2; the marching kernel and everything else: WDPM_TU == 1), because the two want different
 * instruction schedulers: the marching kernel is 2 % faster under the compiler's max-ILP strategy (-mllvm -amdgpu-sched-strategy=max-ilp:
 * add 16384^2 1.1425 -> 1.1189 ms), the eight-wave relay instantiations 2.4 % slower (profiles/r03/sched_strategy_ab.txt). */
hipError_t wdpm_launch_small_rows(int module, const double *w_in, double *w_out, const double *dem, const DemCode &code,
                                  const SlabGeom &g, int A0, int out_last, int chunk_rows, int signed_zero_safe, bool flush,
                                  double thres, int drain_owed, double *totaldrain, hipStream_t s, TilePlan *tiles,
                                  const MaxDiffArgs *md, bool fold_md, bool plain, int no_clamp, bool *taken, bool dry)
#if !defined(WDPM_TU) || WDPM_TU == 2
{
  *taken = false;
    // Small launches: if every 3-row chunk of the window fits on the chip at once, the triangle kernel's six
    // lockstep stages beat the marching kernel's nine dependent ones (482 x 471: DESIGN.md §4.1c).
    // WDPM_TRI=0 keeps the marching kernel (A/B runs), WDPM_TRI=2 forces the triangle kernel on any size.
    static std::atomic<int> env_tri{-1}, tri_slots{0};
    if (env_tri < 0) { const char *t = getenv("WDPM_TRI"); env_tri = t ? atoi(t) : 1; }
    if (!tri_slots) {
      int dev = 0, cus = 256, blocks = 1;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, tri_iteration_kernel<0, false>, 256, 0) != hipSuccess || blocks < 1) blocks = 1;
      tri_slots = cus * blocks * 4;
    }
    int nstr = 1;
    if (g.ncp > kStripIn - kHaloR) nstr = (g.ncp - (kStripIn - kHaloR) + kStripOut - 1) / kStripOut + 1;
    int nch = (out_last - A0 - 1 + 2) / 3;
    if (nch < 1) nch = 1;
    long long items = (long long)nstr * nch;
    static std::atomic<int> env_k{-1};      // WDPM_TRI_K=1|2 forces the triangle kernel's height (tuning, tests); 0 = automatic
    if (env_k < 0) { const char *t = getenv("WDPM_TRI_K"); env_k = t ? atoi(t) : 0; }
    // ... and somewhat beyond: up to 2.7 rounds of (three-row) triangle waves still beat the marching kernel, whose chunks are
    // only a few steps high at these sizes - with six rows per wave (K = 2) for add / subtract once there is more than one round
    // (1200^2: 24.5 -> 14.7 us per add iteration, 1600^2: 30.2 -> 24.6, 2000^2: 35.8 -> 33.7; drain 1200^2: 31.3 -> 20.8,
    // 2000^2: 48.6 -> 46.8; profiles/r02/tri_sweep.txt)
    // ... unless dry-tile flags are being kept and have not (yet, or lately) said that most of the raster works: the triangle
    // kernel keeps no flags, and a mostly dry raster of this size is better off with the marching kernel skipping its dry tiles
    const bool wide = !tiles || tiles->wide_tri_ok;
    const long long slots_now = tri_slots.load(std::memory_order_relaxed);
    const long long tri_limit = wide ? slots_now * 27 / 10 : slots_now;
    // more than one round of waves: the launch is bound by instruction issue, and six rows per wave (K = 2: 9 block stages
    // instead of 12 for them) are the cheaper way through; one round: three rows per wave is the shorter critical path
    const bool two = module != 2 && (env_k == 2 || (env_k == 0 && items > slots_now));
    {
      // the relay kernel (four or eight waves per strip of six / eighteen rows): add 482^2 7.3 -> 5.2 us per iteration, 700^2
      // 10.0 -> 7.3, 1200^2 14.8 -> 11.8, 1600^2 24.9 -> 23.2; drain 482^2 12.1 -> 8.2, 1000^2 16.3 -> 13.1, 1600^2 34.3 -> 25.5,
      // 2400^2 51.6 -> 45.3, 3000^2 77.5 -> 67.3 (profiles/r03/relay_nw_sweep.txt)
      static std::atomic<int> env_relay{-1};
      if (env_relay < 0) { const char *t = getenv("WDPM_RELAY"); env_relay = t ? atoi(t) : 1; }
      static std::atomic<int> env_nw{-1};     // WDPM_RELAY_NW=4|8 forces the workgroup's height (tuning); 0 = automatic
      if (env_nw < 0) { const char *t = getenv("WDPM_RELAY_NW"); env_nw = t ? atoi(t) : 0; }
      static std::atomic<int> ncus{0};
      if (!ncus) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        ncus = cus;
      }
      const long long cus = ncus.load(std::memory_order_relaxed);
      const int rows_out = out_last - A0 - 1;
      const long long nwg4 = (long long)nstr * ((rows_out + 5) / 6 > 0 ? (rows_out + 5) / 6 : 1);        // four waves: six rows out of twelve
      const long long nwg8 = (long long)nstr * ((rows_out + 17) / 18 > 0 ? (rows_out + 17) / 18 : 1);    // eight waves: 18 out of 24
      // rounds of workgroups at two waves per SIMD (a CU holds two workgroups of four waves or one of eight); the taller workgroup
      // where it needs fewer rounds: fewer waves and less re-reading per row, but twice as many rows behind one barrier
      const long long r4 = (nwg4 + 2 * cus - 1) / (2 * cus), r8 = (nwg8 + cus - 1) / cus;
      const bool tall = env_nw == 8 || (env_nw == 0 && r8 < r4);
      long long nwg = tall ? nwg8 : nwg4;
      // With every wave on a SIMD of its own: always.  Beyond that, while the sweeps (profiles/r03/relay_nw_sweep.txt) have it ahead
      // of the triangle / marching kernels - add / subtract up to eight rounds (1600^2 20.2 against 24.8 us, 2000^2 28.6 against 33.9,
      // 2400^2 39.0 against 42.4, 3000^2 a tie), drain up to fourteen (3000^2 67 against 78 us, 3600^2 behind) - and where the triangle kernel would run in one
      // round anyway or the raster is known to be mostly wet: like the triangle kernel this one keeps no dry-tile flags (`wide`)
      // Round 4, after the marching kernel's gains (profiles/r04/relay_breakeven.txt, relay against marching, us per iteration): add
      // 2000^2 27.5 / 37.8, 2200^2 (7 rounds) 32.6 / 33.3, 2400^2 (8) 38.4 / 37.4, 2700^2 47.8 / 47.1; drain 2400^2 (8) 51.6 / 57.4,
      // 2700^2 (10) 62.7 / 66.3, 3000^2 (12) 75.9 / 73.6, the 1053 x 8190 slab (12) 74.4 / 72.2: seven and ten rounds.
      const bool relay_ok = env_tri && !signed_zero_safe && chunk_rows < 3 &&
                            (nwg4 * 4 <= 4 * cus || ((wide || items <= slots_now) && (tall ? r8 : r4) <= (module == 2 ? 10 : 7)) ||
                             env_relay == 2);
      if (env_relay && !fold_md && relay_ok) {
        if (dry) { *taken = true; return hipSuccess; }      /* (wdpm_small_rows_take: the caller only asks) */
        const dim3 rgrid(((unsigned)nwg + 7) / 8 * 8), rblock(tall ? 512 : 256);
        int relay_plain = (module != 2 && tall && r8 >= 6) ? 1 : 0;   // 2000^2 25.8 -> 25.1 us, 3000^2 50.1 -> 46.1
        // bit 1: stage priorities, where workgroups share SIMDs and the launch is a few rounds long (see the kernel; WDPM_RELAY_PRIO=0/2: never / always)
        static std::atomic<int> env_rprio{-1};
        if (env_rprio < 0) { const char *t = getenv("WDPM_RELAY_PRIO"); env_rprio = t ? atoi(t) : 1; }
        if (env_rprio == 2 || (env_rprio == 1 && nwg > cus && (tall ? r8 : r4) <= 4)) relay_plain |= 2;
        if (no_clamp) relay_plain |= 4;        // bit 2: the clamped neighbour step is not exact on this DEM (see the kernel)
#define WDPM_RELAY_LAUNCH(...) hipLaunchKernelGGL((relay_iteration_kernel<__VA_ARGS__>), rgrid, rblock, 0, s, w_in, w_out, dem, g, nstr, (int)nwg, A0, out_last, thres, totaldrain, module == 2 ? drain_owed : 0, code, relay_plain)
#define WDPM_RELAY_PICK(NW)                                                                                        \
        do {                                                                                                       \
          if (module == 2) { if (flush) WDPM_RELAY_LAUNCH(2, true, false, NW); else if (plain) WDPM_RELAY_LAUNCH(2, false, true, NW); else WDPM_RELAY_LAUNCH(2, false, false, NW); } \
          else if (flush) WDPM_RELAY_LAUNCH(0, true, false, NW); else if (plain) WDPM_RELAY_LAUNCH(0, false, true, NW); else WDPM_RELAY_LAUNCH(0, false, false, NW); \
        } while (0)
        static std::atomic<int> env_r32{-1};     // WDPM_RELAY_DEM32=0: the fp64 DEM in the relay kernel (A/B)
        if (env_r32 < 0) { const char *t = getenv("WDPM_RELAY_DEM32"); env_r32 = t ? atoi(t) : 1; }
        // from two rounds of workgroups on (700^2, one round: 7.05 against 7.5 us with the decode on the latency path; 1200^2 11.7 ->
        // 10.95, 1600^2 22.0 -> 20.2, 2400^2 42.9 -> 39.0; profiles/r03/relay_dem32_ab.txt)
        if (tall && module != 2 && code.q != nullptr && env_r32 && (r8 >= 2 || code.force)) {
          if (flush) WDPM_RELAY_LAUNCH(0, true, false, 8, true); else if (plain) WDPM_RELAY_LAUNCH(0, false, true, 8, true); else WDPM_RELAY_LAUNCH(0, false, false, 8, true);
        } else if (tall) WDPM_RELAY_PICK(8); else WDPM_RELAY_PICK(4);
#undef WDPM_RELAY_PICK
#undef WDPM_RELAY_LAUNCH
        *taken = true;
        return hipGetLastError();
      }
    }
    // the block's last launch (max diff folded in) stays here where three rows per wave do (round 3); six-row waves have no
    // registers left for the snapshot's rows: those launches go to the marching kernel as before
    if (env_tri && !signed_zero_safe && !(fold_md && two) && chunk_rows < 3 && (items <= tri_limit || env_tri == 2)) {
      if (dry) { *taken = true; return hipSuccess; }
      if (two) {
        nch = (out_last - A0 - 1 + 5) / 6;
        if (nch < 1) nch = 1;
        items = (long long)nstr * nch;
      }
      const dim3 tgrid(((unsigned)((items + 3) / 4) + 7) / 8 * 8), tblock(256);
      const MaxDiffArgs tmd = fold_md ? *md : MaxDiffArgs{nullptr, 0.0, 0, 0, nullptr};
#define WDPM_TRI_LAUNCH(...) hipLaunchKernelGGL((tri_iteration_kernel<__VA_ARGS__>), tgrid, tblock, 0, s, w_in, w_out, dem, g, nstr, (int)items, A0, out_last, totaldrain, thres, module == 2 ? drain_owed : 0, tmd)
      if (fold_md) { if (flush) WDPM_TRI_LAUNCH(0, true, 1, false, true); else WDPM_TRI_LAUNCH(0, false, 1, false, true); }
      else if (module == 2) { if (flush) WDPM_TRI_LAUNCH(2, true); else if (plain) WDPM_TRI_LAUNCH(2, false, 1, true); else WDPM_TRI_LAUNCH(2, false); }
      else if (two) { if (flush) WDPM_TRI_LAUNCH(0, true, 2); else if (plain) WDPM_TRI_LAUNCH(0, false, 2, true); else WDPM_TRI_LAUNCH(0, false, 2); }
      else { if (flush) WDPM_TRI_LAUNCH(0, true); else if (plain) WDPM_TRI_LAUNCH(0, false, 1, true); else WDPM_TRI_LAUNCH(0, false); }
#undef WDPM_TRI_LAUNCH
      *taken = true;
      return hipGetLastError();
    }
    return hipSuccess;
}
#else
;
#endif

#if !defined(WDPM_TU) || WDPM_TU == 1
hipError_t wdpm_launch_fused(int module, const double *w_in, double *w_out, const double *dem, const DemCode &code,
                             const SlabGeom &g, int chunk_rows, int signed_zero_safe, const double *flush,
                             int drain_owed, double *totaldrain, hipStream_t s, TilePlan *tiles, const MaxDiffArgs *md,
                             int plain_water, XcdBalance *bal) {
  return wdpm_launch_fused_rows(module, w_in, w_out, dem, code, g, 0, g.rows - 1, chunk_rows, signed_zero_safe, flush,
                                drain_owed, totaldrain, s, tiles, md, 0, plain_water, bal);
}

/* would a whole-slab steady launch (no flush, no folded max diff) of this context go to the relay / triangle kernels of small rasters?
 * Those launches keep no state on the host between iterations (no tile flags, no balance table), which is what lets the caller
 * replay a run of them as a HIP graph (wdpm_capi.hip: GraphCache).  Asks the dispatch itself; launches nothing. */
bool wdpm_small_rows_take(int module, const SlabGeom &g, int chunk_rows, int signed_zero_safe, TilePlan *tiles) {
  bool taken = false;
  const DemCode none{nullptr, 0.0, 1.0, 1.0, 0, nullptr, nullptr, 0};
  if (wdpm_launch_small_rows(module, nullptr, nullptr, nullptr, none, g, 0, g.rows - 1, chunk_rows, signed_zero_safe, false, 0.0, 0, nullptr,
                             nullptr, tiles, nullptr, false, false, 0, &taken, true) != hipSuccess) return false;
  return taken;
}

/* one iteration restricted to the output rows [A0 + 2 (0 when A0 == 0), out_last]; A0 % 3 == 0 */
hipError_t wdpm_launch_fused_rows(int module, const double *w_in, double *w_out, const double *dem,
                                  const DemCode &code, const SlabGeom &g, int A0, int out_last, int chunk_rows,
                                  int signed_zero_safe, const double *flush, int drain_owed, double *totaldrain,
                                  hipStream_t s, TilePlan *tiles, const MaxDiffArgs *md, int leave_cus, int plain_water,
                                  XcdBalance *bal) {
  if (tiles) tiles->maintained = 0;
  /* The gate-free variants (PLAIN, see block_update) run for the launches between a block's first (flush on load) and last
   * (max diff), every module and kernel family, whenever the water kinds allow it (drain 8192^2 +4.5 %, 482^2 +3 %, add 16384^2
   * +1.3 % once the two waves of a SIMD ran in step: profiles/r03/plain_water_ab.txt, plain_add_prio_ab.txt). */
  const bool plain = (plain_water & WDPM_LAUNCH_PLAIN) && !signed_zero_safe && !flush && !(md && md->old);
  const int no_clamp = (plain_water & WDPM_LAUNCH_CLAMP_OK) ? 0 : 1;     /* see `deep` in the marching kernel */
  const bool fold_md = md && md->old && module != 2 && !signed_zero_safe;
  if (md && md->old && !fold_md) return hipErrorInvalidValue;     /* the caller asks only where a folding variant exists */
  hipError_t e = dpp_selfcheck(s);
  if (e != hipSuccess) return e;
  if (A0 < 0 || A0 % 3 != 0 || out_last > g.rows - 1 || out_last < A0) return hipErrorInvalidValue;
  if (flush && signed_zero_safe) return hipErrorInvalidValue;   /* the caller flushes in place for that variant */
  const double thres = flush ? *flush : 0.0;
  {
    bool taken = false;
    e = wdpm_launch_small_rows(module, w_in, w_out, dem, code, g, A0, out_last, chunk_rows, signed_zero_safe, flush != nullptr, thres,
                               drain_owed, totaldrain, s, tiles, md, fold_md, plain, no_clamp, &taken, false);
    if (taken || e != hipSuccess) return e;
  }
  int nstrips = 1;
  if (g.ncp > kStripIn - kHaloR) nstrips = (g.ncp - (kStripIn - kHaloR) + kStripOut - 1) / kStripOut + 1;
  const bool fast = !signed_zero_safe;
  // add / subtract with an encodable DEM: 20 B per cell-update instead of 24 (drain: see big_drain below)
  // ... and only pays on launches big enough for two waves per SIMD (see below): at one wave per SIMD
  // the wave's own latency chain is the limit and the nine decodes per step cost 3-5 % (size sweep in
  // profiles/r01: 512^2 - 3072^2 slower with codes, 4096^2 and up 5-12 % faster).
  // Chunk height at two waves per SIMD from which a launch fills every slot: 18 rows (add / subtract), 12 (drain).  Rounds 1 - 3 had
  // 36 / 18; with the two waves of a SIMD keeping each other in step and chunk heights following the XCDs, the second wave pays on
  // shorter chunks: add 2700^2 +3.4 %, 3000^2 +6.7 %, 3300^2 +9.7 %, 3600^2 +2.6 %; drain 2400^2 +2.0 %, 3000^2 +1.1 %
  // (profiles/r04/tall_rows_sweep.txt).
  constexpr long long tall_add = 18, tall_drain = 12;
  const bool big = (long long)(out_last - A0 + 1) * nstrips >= (long long)tall_add * resident_waves<0, false>();
  // Drain with the codes (round 4, late): "latency-bound, the decode would only add instructions" was round 1's reading of a kernel
  // at one wave per SIMD; at two, and with the card at its power cap whatever the kernel does, four bytes per
  // cell-update less are worth more than the nine decodes of a step - on launches that fill every slot at two waves per SIMD.
  const bool big_drain = module == 2 && fast &&
                         (long long)(out_last - A0 + 1) * nstrips >= (long long)tall_drain * resident_waves<2, false, 1>();
  const bool dem32 = fast && code.q != nullptr && ((module == 2 ? big_drain : big) || code.force);
  bool two_per_simd = false;      /* every slot filled: two waves per SIMD, workgroups of eight waves (one per CU) */
  int slots = module == 2 ? (!fast ? resident_waves<2, true>() : dem32 ? resident_waves<2, false, 1>() : resident_waves<2, false>())
              : dem32     ? resident_waves<0, false, 1>()
              : fast      ? resident_waves<0, false>()
                          : resident_waves<0, true>();
  {
    // (Round 1, fp64 DEM:) the kernel is bound by the memory system with ONE wave per SIMD already; a second wave per SIMD only adds
    // concurrent DRAM row streams and, on rasters too small to fill the chip, makes the dispatcher
    // double up waves on some SIMDs while others idle.  Filling half of the resident slots measured
    // +4 % at 16384^2, +13 % at 6000^2, x1.9 at 1500^2, x2.3 at 1024^2 (-4 % at 4096^2).
    // The drain variant is different: 16 instructions per neighbour step on one dependent chain per
    // lane leave it latency-bound (59 % VALU issue at one wave per SIMD), and a second wave per SIMD
    // fills the bubbles: +21 % at 8192^2, +30 % at 16384^2 - as long as the chunks stay tall enough
    // for the 6-row warm-up of each not to eat the gain.
    // With the DEM as 32-bit codes the add kernel is in the same position: fewer bytes, nine decodes
    // more per step - one wave per SIMD 1.34 ms per 16384^2 launch (no gain), two waves 1.23 ms.
    // Thresholds from tools/threshold_sweep.sh and tools/slab_sweep.sh (profiles/r01): chunk height at
    // two waves per SIMD >= 36 rows for the DEM-code add kernel (3072^2 still loses, 3600^2 gains 12 %),
    // >= 18 rows for drain (+11 % on a 1055 x 8190 slab, +15 % at 3072^2, a wash at 2048^2).
    if (dem32 && big) two_per_simd = true;
    if (module == 2 && fast && (long long)(out_last - A0 + 1) * nstrips >= (long long)tall_drain * slots) two_per_simd = true;
    if (!two_per_simd && !(module == 2 && !fast)) slots = slots / 2;      /* (the -0.0-safe drain variant is built for one wave per SIMD) */
    if (leave_cus > 0) {                               // room for somebody else's kernels (see wdpm_kernels.h)
      int dev = 0, cus = 256;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      if (leave_cus < cus / 2) slots = (int)((long long)slots * (cus - leave_cus) / cus);
    }
  }
  const int wrows = out_last - A0 + 1;                       // rows of this launch's window
  const int H = pick_chunk_rows(wrows, nstrips, chunk_rows, slots);
  // chunk i stores rows [A0+H*i+2 (0 for A0+H*i = 0), A0+H*(i+1)+1]; the last must reach out_last
  int nchunks = (out_last - A0 - 1 + H - 1) / H;
  if (nchunks < 1) nchunks = 1;
  int nitems = nstrips * nchunks;
  const int wpb = two_per_simd ? 8 : 4;                            // waves per workgroup (see the kernel: `prio`)
  dim3 grid(((nitems + wpb - 1) / wpb + 7) / 8 * 8), block(64 * wpb);   // multiple of 8: see the XCD remap
  TileFlags tf{nullptr, nullptr, 0, nullptr, nchunks};
  if (tiles && tiles->zout && fast && H >= 6 && A0 == 0 && out_last == g.rows - 1 &&
      (nstrips + 2) * (nchunks + 2) <= tiles->capacity) {
    // the flags describe one tiling: another chunk height or strip count makes the old ones meaningless
    const bool same = tiles->nstrips == nstrips && tiles->H == H && tiles->nchunks == nchunks;
    if (!same) {   /* new pitch: the output raster's flag array needs its border of 1s before the kernel fills the interior */
      e = hipMemsetAsync(tiles->zout, 1, (size_t)tiles->capacity, s);
      if (e != hipSuccess) return e;
    }
    tf.zin = same && tiles->zin_valid ? tiles->zin : nullptr;
    tf.zout = tiles->zout;
    tf.zout_known = same && tiles->zout_valid;
    tf.active = tiles->active;
    tiles->nstrips = nstrips; tiles->H = H; tiles->nchunks = nchunks;
    tiles->maintained = 1;
  }
  const MaxDiffArgs mda = fold_md ? *md : MaxDiffArgs{nullptr, 0.0, 0, 0, nullptr};
  if (fold_md) tf = TileFlags{nullptr, nullptr, 0, nullptr, nchunks};   /* every wave must look at its block: no skipping in this launch */
  if (fold_md && tiles) tiles->maintained = 0;
  // Chunk heights by what each XCD delivers (wdpm_kernels.h::XcdBalance): whole-slab launches of two waves per SIMD that keep no
  // dry-tile flags.  Every launch may be measured (the block's first and last are other instantiations: not those); the table
  // needs chunks of a dozen rows at least, so that whole row triples can follow weights a few per cent apart.
  BalanceArgs ba{nullptr, nullptr, nullptr, 0};
  const bool bal_forced = bal && bal->mode == 2;          /* tests: the table on launches of any size, from skewed weights */
  if (bal && bal->mode && (two_per_simd || bal_forced) && A0 == 0 && out_last == g.rows - 1 && (chunk_rows < 3 || bal_forced) && nchunks >= 2) {
    const bool can_table = !tf.zout && H >= (bal_forced ? 6 : 12) && (nchunks + 1) * nstrips <= bal->capacity;
    // Round 5: with the table in charge a strip's chunks need not be equally tall, so a strip is cut into as many chunks as the
    // resident round has slots for - equal heights in whole triples left slots empty (the 8-GPU drain slab, 1055 x 8190: 27-row
    // chunks, 39 x 48 = 1872 waves on 2048 slots) - and the table pairs tall chunks with short ones on every SIMD (`pair` in
    // xcd_rebalance_kernel).  WDPM_PAIR=0: round 4's geometry (A/B, tests).
    static std::atomic<int> env_pair{-1};
    if (env_pair < 0) { const char *t = getenv("WDPM_PAIR"); env_pair = t ? atoi(t) : 1; }
    // (WDPM_PAIR=2, tests: also on forced tables and launches of one wave per SIMD, down to the table's minimum of two triples a chunk)
    const int pair_mode = env_pair.load(std::memory_order_relaxed);
    const int pair = ((wpb == 8 && !bal_forced && pair_mode != 0) || pair_mode == 2) ? 1 : 0;
    if (can_table && pair) {
      const int T = (out_last - 1 - A0 + 2) / 3, nc = slots / nstrips;
      if (nc > nchunks && T / nc >= (pair_mode == 2 ? 2 : 4) && (nc + 1) * nstrips <= bal->capacity) {
        nchunks = nc;
        nitems = nstrips * nchunks;
        grid = dim3(((nitems + wpb - 1) / wpb + 7) / 8 * 8);
        tf.nchunks = nchunks;
      }
    }
    const int ipx = wpb * (int)(grid.x / 8);
    const bool steady = !flush && !fold_md;
    if (can_table) {
      const bool same = bal->nstrips == nstrips && bal->nchunks == nchunks && bal->A0 == A0 && bal->out_last == out_last && bal->ipx == ipx;
      const bool update = bal->measured >= 3 && (same || bal->measured_uniform);
      if (!same || update) {
        hipLaunchKernelGGL(xcd_rebalance_kernel, dim3(1), dim3(256), 0, s, bal->weight, bal->acc, bal->table, nstrips, nchunks, A0,
                           out_last, ipx, update ? 1 : 0, bal->measured_uniform, pair);
        bal->measured_uniform = 0;
        bal->nstrips = nstrips; bal->nchunks = nchunks; bal->A0 = A0; bal->out_last = out_last; bal->ipx = ipx;
        bal->measured = 0;
        if (update) bal->updates++;
      }
      ba.table = bal->table;
      static std::atomic<int> env_rot{-1};         // WDPM_ROT=0: shares and measurements by blockIdx % 8, as in round 4 (A/B)
      if (env_rot < 0) { const char *t = getenv("WDPM_ROT"); env_rot = t ? atoi(t) : 1; }
      if (env_rot.load(std::memory_order_relaxed) != 0) {
        ba.rot = bal->acc + 2 * kBalClasses;          // shares and measurements by physical XCD (BalanceArgs::rot)
        ba.parity = bal->seq++ & 1;
      }
      // measured: while the weights are young, every steady launch (an update every three); afterwards three launches in 256
      if (steady && (bal->updates < 6 || (bal->launches & 255) < 3)) {
        if (bal->measured == 0 || !bal->measured_uniform) { ba.acc = bal->acc; bal->measured_uniform = 0; bal->measured++; }
      }
      bal->launches++;
    } else if (steady && bal->updates == 0 && H >= 12 && (bal->measured == 0 || bal->measured_uniform)) {
      // equal heights (a block that keeps dry-tile flags): what the XCDs deliver can be learnt here already
      ba.acc = bal->acc;
      ba.rot = bal->acc + 2 * kBalClasses;
      ba.parity = bal->seq++ & 1;
      bal->measured_uniform = 1;
      bal->measured++;
    }
  }
  // Workgroups per CU, whatever the register allocator ends up with (an instantiation at 166 VGPRs would let the dispatcher stack
  // three four-wave workgroups on some CUs and one on others): unused dynamic LDS beside the 36 KiB of staging - four-wave
  // workgroups 36 KiB (two fit a CU's 160 KiB, three do not), eight-wave workgroups 48 KiB (one fits).
  // The pad follows the CU's LDS (not a constant tied to one architecture): an eight-wave workgroup takes just over half
  // of it, a four-wave one just over a third; 160 KiB on gfx950: 45 KiB and 18 KiB beside the 36 KiB of staging.
  static std::atomic<int> cu_lds{0};
  if (!cu_lds) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) != hipSuccess || v < 65536)
      v = 163840;
    cu_lds = v;
  }
  const int staging = 8 * 3 * kStripIn * (int)sizeof(double) + 64;
  const int share = cu_lds.load(std::memory_order_relaxed) / (wpb == 8 ? 2 : 3) + 1024;
  const unsigned lds_pad = share > staging ? (unsigned)(share - staging) : 0u;
  // the two waves of a SIMD keep each other in step (see the marching loop); WDPM_PRIO=0: no priorities (A/B, tests)
  static std::atomic<int> env_prio{-1};
  if (env_prio < 0) { const char *e = getenv("WDPM_PRIO"); env_prio = e ? atoi(e) : 1; }
  const int prio = (wpb == 8 && env_prio.load(std::memory_order_relaxed) != 0) ? 1 : 0;
#define WDPM_LAUNCH(...) hipLaunchKernelGGL((fused_iteration_kernel<__VA_ARGS__>), grid, block, lds_pad, s, w_in, w_out, dem, code, g, nstrips, nitems, H, A0, out_last, totaldrain, thres, module == 2 ? drain_owed : 0, tf, mda, prio, no_clamp, ba)
  // <module, -0.0-safe, DEM codes, flush on load, max diff folded in, gate-free>: which instantiation runs is decided here and
  // nowhere else (DESIGN.md §4 has the table)
#define WDPM_LAUNCH_ADD(D32) do { if (fold_md) { if (flush) WDPM_LAUNCH(0, false, D32, true, true); else WDPM_LAUNCH(0, false, D32, false, true); } \
                                  else if (plain) WDPM_LAUNCH(0, false, D32, false, false, true);                                                   \
                                  else if (flush) WDPM_LAUNCH(0, false, D32, true, false); else WDPM_LAUNCH(0, false, D32, false, false); } while (0)
#define WDPM_LAUNCH_DRAIN(D32) do { if (plain) WDPM_LAUNCH(2, false, D32, false, false, true);                                                      \
                                    else if (flush) WDPM_LAUNCH(2, false, D32, true, false); else WDPM_LAUNCH(2, false, D32, false, false); } while (0)
  // the codes as 16-bit offsets (18.1 B of HBM traffic per cell-update) on launches of 10^8 cells and more: wdpm_kernels.h::wdpm_dem16_pays
  const bool dem16 = dem32 && code.h != nullptr && wdpm_dem16_pays((long long)wrows * g.ncp, code.force);
  if (module == 2 && !fast) WDPM_LAUNCH(2, true, 0, false, false);
  else if (module == 2 && dem16) WDPM_LAUNCH_DRAIN(2);
  else if (module == 2 && dem32) WDPM_LAUNCH_DRAIN(1);
  else if (module == 2) WDPM_LAUNCH_DRAIN(0);
  else if (!fast) WDPM_LAUNCH(0, true, 0, false, false);
  else if (dem16) WDPM_LAUNCH_ADD(2);
  else if (dem32) WDPM_LAUNCH_ADD(1);
  else WDPM_LAUNCH_ADD(0);
#undef WDPM_LAUNCH_DRAIN
#undef WDPM_LAUNCH_ADD
#undef WDPM_LAUNCH
  return hipGetLastError();
}

#endif   /* WDPM_TU 1 */

#ifdef WDPM_WAVE_TIMES
#if !defined(WDPM_TU) || WDPM_TU == 1
extern "C" int wdpm_debug_wave_times(unsigned long long *out, int nwaves) {
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_times), (size_t)nwaves * 32) != hipSuccess;
}
#endif
#if !defined(WDPM_TU) || WDPM_TU == 2
extern "C" int wdpm_debug_relay_times(unsigned long long *out, int nwaves) {
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_times), (size_t)nwaves * 32) != hipSuccess;
}
#endif
#endif
