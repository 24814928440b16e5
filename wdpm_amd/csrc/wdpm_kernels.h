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

/* in place: dem <= miss (or NaN) -> +inf.  Every other kernel expects the DEM in this form. */
hipError_t wdpm_launch_mark_nodata(double *dem, size_t cells, double miss, hipStream_t s);
/* one colour pass, in place (reference kernels add/subtract/ddrain, runoff.cl:137-183) */
hipError_t wdpm_launch_pass(int module, double *w, const double *dem, const SlabGeom &g, int oi, int oj,
                            double *totaldrain, hipStream_t s);
/* one whole iteration (9 passes) fused in one launch: w_in -> w_out (distinct buffers) */
/* signed_zero_safe = 0 selects the faster add/subtract variant that is exact when the water raster
 * holds no -0.0 (wdpm_stencil.h::flow_add_nz) */
hipError_t wdpm_launch_fused(int module, const double *w_in, double *w_out, const double *dem,
                             const SlabGeom &g, int chunk_rows, int signed_zero_safe, double *totaldrain,
                             hipStream_t s);
/* two whole iterations of add / subtract in one launch (12 B of HBM traffic per cell-update) */
hipError_t wdpm_launch_fused2(const double *w_in, double *w_out, const double *dem, const SlabGeom &g,
                              int chunk_rows, int signed_zero_safe, hipStream_t s);
hipError_t wdpm_launch_fused2w(const double *w_in, double *w_out, const double *dem, const SlabGeom &g,
                               int chunk_rows, int signed_zero_safe, hipStream_t s);
hipError_t wdpm_launch_fused_rows(int module, const double *w_in, double *w_out, const double *dem,
                                  const SlabGeom &g, int A0, int out_last, int chunk_rows,
                                  int signed_zero_safe, double *totaldrain, hipStream_t s);
/* *flag |= 1 if any of the n doubles at p is -0.0 */
hipError_t wdpm_launch_scan_negzero(const double *p, size_t n, unsigned long long *flag, hipStream_t s);
/* drain() (WDPMCL.c:1859-1897) on the device */
hipError_t wdpm_launch_drain_outlet(double *w, const double *dem, const SlabGeom &g, double *totaldrain,
                                    hipStream_t s);
/* threshold flush + snapshot (WDPMCL.c:1055-1073) */
hipError_t wdpm_launch_flush_snapshot(double *w, double *old, size_t cells, double thres, hipStream_t s);
/* max |w-old| over valid cells of rows [row_lo,row_hi) (+ seed cell 0), result as uint64 bits */
hipError_t wdpm_launch_max_diff(const double *w, const double *old, const double *dem, const SlabGeom &g,
                                int row_lo, int row_hi, unsigned long long *result_bits, hipStream_t s);

#endif
