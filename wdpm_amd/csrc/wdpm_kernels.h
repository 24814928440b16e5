/*
 * wdpm_kernels.h — launch interface between the C-ABI layer (wdpm_capi.hip) and the gfx950
 * kernels (wdpm_kernels.hip, wdpm_fused.hip).  Internal; not part of the drop-in boundary.
 */
#ifndef WDPM_KERNELS_H
#define WDPM_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

/* A raster slab resident in HBM: rows x ncp doubles, row-major, row 0 of the slab is padded
 * row `row0` of the whole raster (row0 % 3 == 0). */
struct SlabGeom {
  int rows;      /* padded rows held */
  int ncp;       /* padded columns (ncols + 2) */
  int row0;      /* global padded row of slab row 0 */
  int R, C;      /* file raster size (centres are global rows 1..R, cols 1..C) */
  int dr, dc;    /* drain cell, slab-local row / column (drain module; may lie outside the slab) */
  double miss;   /* NODATA value */
};

/* The static DEM as 32-bit codes (4 B per cell instead of 8 in the iteration kernel's HBM traffic):
 * dem[i] == dem32_decode(q[i]) BIT FOR BIT for every cell, verified on the device when the DEM is
 * uploaded (wdpm_launch_dem_encode); NODATA is INT32_MIN.  Real DEMs are decimal text, i.e.
 * v = k / 10^e for an integer k, and the decode is the correctly rounded quotient; a raster that
 * is not of that form (any cell fails the check for e = 0..6) simply keeps the fp64 DEM. */
struct DemCode {
  const int *q;      /* rows x ncp codes, or nullptr when the raster is not encodable */
  double k0;         /* integer offset: k = q + k0 */
  double D, rD;      /* 10^e and its correctly rounded reciprocal */
  int force;         /* use the codes on launches of any size (tests); normally only where they pay */
  /* Second level (tried in round 3 at +0.7 % when the kernel was bound by instruction issue alone; kept in round 4, when it
   * is as close to the memory system's roof): the same codes as 16-bit offsets from one 32-bit base per group of kDemGroup
   * columns of a row - q[r][c] == gb[r][c / kDemGroup] + h[r][c] for every valid cell, h == 0xFFFF for NODATA - 2.08 B per
   * cell of HBM traffic instead of 4 (18.1 B per cell-update instead of 20).  Possible when no group of 48 neighbouring cells
   * of a row spans more than 65 534 quanta (6.5 m of relief at 1e-4 m, 655 m at 1e-2 m); checked for the whole raster when
   * the DEM is uploaded, else the 32-bit codes stay in charge.  An exact integer identity with the verified 32-bit codes.
   * A lane's three columns start at a multiple of 3, so they never straddle a group: one base per lane and row. */
  const unsigned short *h;   /* rows x ncp offsets, or nullptr */
  const int *gb;             /* rows x ngroups bases */
  int ngroups;
};
constexpr int kDemGroup = 48;
/* ... and only where it pays: rasters far beyond the 256 MB Infinity Cache, where the bytes are what a launch waits for (16384^2
 * +2.5 %, 8192^2 +0.3 %, 4096^2 -2.2 %: nine integer adds more per step; profiles/r04/dem16_bench*_ab.txt) - or when forced (tests) */
inline bool wdpm_dem16_pays(long long cells_of_the_launch, int force) { return cells_of_the_launch >= 100000000LL || force != 0; }

/* smallest valid (finite) dem value, as an order-preserving uint64 key in key[0] (all ones: none), and the bit image of the
 * largest |dem| over the valid cells in key[1] (0: none) */
hipError_t wdpm_launch_dem_min(const double *dem, size_t cells, unsigned long long *key, hipStream_t s);
double wdpm_dem_key_to_double(unsigned long long key);
/* q[i] = code of dem[i]; *bad |= 1 if any cell does not decode to exactly dem[i] */
hipError_t wdpm_launch_dem_encode(const double *dem, size_t cells, double k0, double D, double rD, int *q,
                                  unsigned long long *bad, hipStream_t s);

/* h / gb of DemCode from the verified 32-bit codes q (rows x ncp); *bad |= 1 if some group spans more than 65 534 quanta */
hipError_t wdpm_launch_dem16_encode(const int *q, int rows, int ncp, int ngroups, unsigned short *h, int *gb,
                                    unsigned long long *bad, hipStream_t s);

/* The reference's SEQUENTIAL volume sum (WDPMCL.c:1259-1266) evaluated in parallel, see
 * tests/seqsum_model.py: chunks of kSeqSumChunk cells in row-major order.
 *   pass A: approx[c] = ordinary sum of the valid cells of chunk c, dirty[c] = a negative / non-finite term
 *   pass B: for chunks with kexp[c] != INT32_MIN, isum[c] = sum of round(x / 2^(kexp-52)) as int64,
 *           tie[c] = some term lies exactly half way between two multiples of 2^(kexp-52) */
constexpr int kSeqSumChunk = 65536;
hipError_t wdpm_launch_seqsum_a(const double *w, const double *dem, size_t n, double *approx, unsigned *dirty,
                                hipStream_t s);
hipError_t wdpm_launch_seqsum_b(const double *w, const double *dem, size_t n, const int *kexp, long long *isum,
                                unsigned *tie, hipStream_t s);

/* Dry-tile skipping (the reference skips dry centres cell by cell, WDPMCL.c:1099; basins are mostly dry land).
 * A tile is the exact output block of one wave of the marching kernel (strip x chunk).  Per water raster one
 * byte per tile: 1 = every cell of the block holds +0.0, 0 = unknown.  A wave whose tile and eight neighbours
 * are flagged in the input raster has an all-dry input window: it loads nothing and its output block is all
 * +0.0 - stored only if the output raster's own flag does not already say so.  Every other wave works as
 * usual and leaves the flag of what it stored.  Only where no -0.0 depth exists (signed_zero_safe == 0).
 * Layout: (nchunks + 2) x (nstrips + 2) bytes, tile (chunk, strip) at [(chunk+1) * (nstrips+2) + strip + 1], the
 * border always 1 (outside the slab there is no water), so that the nine flags are read without bounds tests. */
struct TileFlags {               /* kernel argument */
  const unsigned char *zin;      /* flags of w_in for this tiling, or nullptr (unknown: nobody skips) */
  unsigned char *zout;           /* flags of w_out, written for every tile; nullptr = tiles not tracked in this launch */
  int zout_known;                /* zout's current content describes w_out's current content */
  unsigned *active;              /* += 1 per wave that did not skip */
  int nchunks;
};
struct TilePlan {                /* host side of it, kept by the context per launch */
  const unsigned char *zin;      /* in: flag arrays of the input / output raster */
  unsigned char *zout;
  int zin_valid, zout_valid;     /* in: they describe those rasters' current content (for the tiling below) */
  unsigned *active;
  int capacity;                  /* in: tiles each flag array has room for */
  int nstrips, H, nchunks;       /* in: the tiling the flags were made for; out: the tiling of this launch */
  int maintained;                /* out: this launch read / wrote the flags (marching kernel, whole slab, H >= 6) */
  int wide_tri_ok;               /* in: the last block with flags found most tiles working: the triangle kernel (which keeps no
                                  * flags) may also take rasters of a few rounds of its waves (wdpm_launch_fused_rows) */
};

/* max |w - oldw| folded into an iteration launch (the last one of a block): the waves have the final values in
 * registers anyway; one more raster read (the snapshot) instead of a pass over three.  old == nullptr: off. */
struct MaxDiffArgs {
  const double *old;             /* the snapshot, possibly still owed the threshold flush `thres` (applied as read) */
  double thres;
  int row_lo, row_hi;            /* slab rows [row_lo, row_hi) */
  unsigned long long *bits;      /* atomicMax of the bit image of the (non-negative) maximum; zeroed by the caller */
};

/* Chunk heights that follow what each XCD delivers (round 4).  A marching launch is ONE round of waves, eight equal shares of
 * work items dealt to the eight XCDs - and the XCDs of a chip are not equally fast: tools/wave_times.py finds their median
 * wave durations 7 % apart at 16384^2, 11 % on an 8-GPU slab, 19 % at 4096^2 (profiles/r04/wave_times_coop.txt), the same XCDs
 * slow in every launch on a box and different ones from box to box; the launch ends with the slowest.  So the row boundaries
 * of the chunks come from a table in device memory - per strip its own - in which a chunk's height is proportional to a
 * weight of the XCD its work item runs on; waves of measured launches add their duration to a per-XCD sum, and a one-workgroup
 * kernel on the same stream (xcd_rebalance_kernel: no host round trip) moves the weights towards equal durations and rebuilds the
 * table.  Results do not depend on where chunks begin (the parity suites run with skewed weights forced: WDPM_BALANCE=2).
 * Only whole-slab launches of two waves per SIMD without dry-tile flags (a tiling of its own: mostly wet rasters). */
constexpr int kBalClasses = 10;  /* eight XCDs, a strip's last chunk (the slab's lower edge), and (round 5) its first: the waves of the slab's
                                  * first chunk row end 2 - 11 % after everybody else's (tools/wave_times.py, profiles/r05/first_chunk_row.txt:
                                  * the rows, not the workgroups dispatched first - it stays with chunk row 0 when the dispatch order is reversed) */
struct BalanceArgs {             /* kernel argument */
  const int *table;              /* (nchunks + 1) x nstrips slab rows: chunk c of strip s marches from [c][s] to [c+1][s]; nullptr: equal heights */
  unsigned long long *acc;       /* [x] += wave duration, [kBalClasses + x] += 1; x = the XCD - or 8 for a strip's last chunk, 9 for its first; nullptr: not measured */
  /* Round 5: WHICH XCD.  Workgroups are dealt round-robin over the eight XCDs - but not from XCD 0: the dispatcher carries on where the
   * previous dispatch stopped, so every kernel whose workgroup count is not a multiple of eight (a one-workgroup rebalance, a blit
   * kernel behind a small copy, RCCL's send / recv kernels) turns the mapping of blockIdx % 8 to physical XCDs by a few places
   * (tools/wave_times.py: physical - logical XCD is one value for all waves of a launch, and another in the next launch but one).
   * Round 4 keyed weights and shares by blockIdx % 8: after its own rebalance kernel they sat one XCD off.  Now a launch leaves
   * the rotation it ran under in rot[parity ^ 1] (the physical XCD of workgroup 0), the next launch reads rot[parity] - one value
   * for all its workgroups, so the block remap stays a permutation whatever it holds - and takes its share by PHYSICAL XCD; durations
   * are booked under the physical XCD, and not at all in a launch whose rotation was not the expected one.  nullptr: blockIdx % 8. */
  unsigned long long *rot;
  int parity;
};
struct XcdBalance {              /* host side, per context */
  int *table;                    /* device */
  unsigned long long *acc;       /* device: 2 * kBalClasses cells, and two more behind them: BalanceArgs::rot */
  int seq;                       /* marching launches that were handed `rot` so far (its parity) */
  float *weight;                 /* device: 8 relative chunk heights, mean 1, the factor on a strip's last chunk and the one on its first */
  int capacity;                  /* ints `table` has room for */
  int nstrips, nchunks, A0, out_last, ipx;   /* the launch geometry the table was built for (nstrips == 0: none yet) */
  int measured;                  /* launches measured since the last rebalance */
  int measured_uniform;          /* ... and they ran on equal heights (the weights in effect were 1) */
  int updates;                   /* rebalances so far */
  long long launches;            /* balanced launches so far (measurement schedule) */
  int mode;                      /* 0 off, 1 adaptive, 2 adaptive from deliberately skewed weights (tests) */
};

/* in place: dem <= miss (or NaN) -> +inf.  Every other kernel expects the DEM in this form. */
hipError_t wdpm_launch_mark_nodata(double *dem, size_t cells, double miss, hipStream_t s);
/* one colour pass, in place (reference kernels add/subtract/ddrain, runoff.cl:137-183) */
hipError_t wdpm_launch_pass(int module, double *w, const double *dem, const SlabGeom &g, int oi, int oj,
                            double *totaldrain, hipStream_t s);
/* one whole iteration (9 passes) fused in one launch: w_in -> w_out (distinct buffers) */
/* signed_zero_safe = 0 selects the faster add/subtract variant that is exact when the water raster
 * holds no -0.0 (wdpm_stencil.h::flow_add_nz) */
/* drain_owed (drain module): the previous iteration's drain() has not been applied to w_in; this launch does it
 * (sum into totaldrain by the wave that owns the outlet, the nine cells read as 0).
 * flush != nullptr: every water value is replaced by 0 when it is < *flush as it is loaded (the block's
 * threshold flush, WDPMCL.c:1055-1065, riding on the first iteration; only with signed_zero_safe == 0) */
hipError_t wdpm_launch_fused(int module, const double *w_in, double *w_out, const double *dem, const DemCode &code,
                             const SlabGeom &g, int chunk_rows, int signed_zero_safe, const double *flush,
                             int drain_owed, double *totaldrain, hipStream_t s, TilePlan *tiles = nullptr,
                             const MaxDiffArgs *md = nullptr, int plain_water = 0, XcdBalance *bal = nullptr);
/* whether a steady whole-slab launch of this geometry goes to the small-raster kernels (relay / triangle): see wdpm_fused.hip */
bool wdpm_small_rows_take(int module, const SlabGeom &g, int chunk_rows, int signed_zero_safe, TilePlan *tiles);
hipError_t wdpm_launch_fused_rows(int module, const double *w_in, double *w_out, const double *dem,
                                  const DemCode &code, const SlabGeom &g, int A0, int out_last, int chunk_rows,
                                  int signed_zero_safe, const double *flush, int drain_owed, double *totaldrain,
                                  hipStream_t s, TilePlan *tiles = nullptr, const MaxDiffArgs *md = nullptr,
                                  int leave_cus = 0, int plain_water = 0, XcdBalance *bal = nullptr);
/* plain_water, bit 0 (WDPM_LAUNCH_PLAIN): the caller knows (wdpm_launch_scan_water, and nothing written since that could change
 * it) that every cell of w_in that may not give water holds +0.0: launches that have such a variant then run without the centre gate.
 * bit 1 (WDPM_LAUNCH_CLAMP_OK): every valid elevation is below 2^30 m in magnitude (wdpm_launch_dem_min), so that half an ulp of an
 * elevation is nothing against a depth: the kernels may take a flow's `max(x / 8, -0.0)` as one clamped instruction wherever
 * the depths they hold are shallow enough for it to be exact (wdpm_stencil.h::eighth_clamped, `deep` in wdpm_fused.hip). */
enum { WDPM_LAUNCH_PLAIN = 1, WDPM_LAUNCH_CLAMP_OK = 2 };
/* leave_cus: size the launch as if the chip had that many compute units fewer - the interior launch of an overlapped
 * iteration leaves room for the RCCL send/recv kernels queued beside it (a launch otherwise fills every slot for its
 * whole duration, and the transfer would start only when the first waves retire) */
/* What kinds of depth do the n cells at p hold (dem: the device DEM of the same cells, NODATA = +inf)?
 *   *flag |= 1  a -0.0                       (the exact-zero stencil variant must run, see signed_zero_safe)
 *   *flag |= 2  a negative depth             (gone once a threshold flush with thres >= 0 has been applied)
 *   *flag |= 4  NaN, a depth above 1e290, or water on a NODATA cell   (stays)
 * With none of them, every cell that may not give water (dry, NODATA, outside the slab) holds +0.0 exactly - and then
 * the centre gate of the reference's sweep (WDPMCL.c:1099) needs no instructions: see PLAIN in wdpm_fused.hip. */
enum { WDPM_WATER_NEGZERO = 1, WDPM_WATER_NEGATIVE = 2, WDPM_WATER_ODD = 4 };
hipError_t wdpm_launch_scan_water(const double *p, const double *dem, size_t n, unsigned long long *flag, hipStream_t s);
/* drain() (WDPMCL.c:1859-1897) on the device */
hipError_t wdpm_launch_drain_outlet(double *w, const double *dem, const SlabGeom &g, double *totaldrain,
                                    hipStream_t s);
/* threshold flush + snapshot (WDPMCL.c:1055-1073); old == w: flush in place only */
hipError_t wdpm_launch_flush_snapshot(double *w, double *old, size_t cells, double thres, hipStream_t s);
/* max |w-flush(old)| over valid cells of rows [row_lo,row_hi) (+ seed cell 0), result as uint64 bits;
 * flush(v) = v < old_thres ? 0 : v is the block's threshold flush the snapshot may still be owed (-inf: none) */
hipError_t wdpm_launch_max_diff(const double *w, const double *old, double old_thres, const double *dem, const SlabGeom &g,
                                int row_lo, int row_hi, unsigned long long *result_bits, hipStream_t s);

/* set-up and final statistics on the device, see wdpm_kernels.hip */
hipError_t wdpm_launch_pad_setup(const double *fdem, const double *fwater, double *dem, double *w, const SlabGeom &g,
                                 int op, double add, double rof, double sub, hipStream_t s);
hipError_t wdpm_launch_count_stats(const double *w, const double *dem, size_t first, size_t last, double miss,
                                   unsigned long long *out3, hipStream_t s);
hipError_t wdpm_launch_find_drain(const double *dem, size_t first, size_t last, unsigned long long *key_and_index,
                                  hipStream_t s);
hipError_t wdpm_launch_unpad(const double *w, const double *dem, const SlabGeom &g, int frow, int nrows, int mask,
                             double *out, hipStream_t s);

#endif
