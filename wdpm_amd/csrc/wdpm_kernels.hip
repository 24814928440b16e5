/*
 * wdpm_kernels.hip — gfx950 kernels for the WDPM block loop that are not the fused iteration:
 *   - one colour pass per launch (the unit the reference launches, runoff.cl:137-183)
 *   - drain() outlet emptying (WDPMCL.c:1859-1897)
 *   - threshold flush + snapshot (WDPMCL.c:1055-1073)
 *   - max |w - oldw| over valid cells (WDPMCL.c:1239-1254): wave64 shuffle reduction, one
 *     atomicMax per workgroup on the uint64 image of the (non-negative) double
 * Row-major padded rasters; all arithmetic fp64 in the reference's order; compiled with
 * -ffp-contract=off (there are no products to contract, x/8.0 is written x*0.125 which is the
 * same correctly-rounded value, subnormals included).
 */
#include "wdpm_kernels.h"
#include "wdpm_stencil.h"

/* The device copy of the DEM holds +inf where bigdem <= missingvalue (wdpm_launch_mark_nodata), so
 * "bigdem > missingvalue" (WDPMCL.c:1099,1944,1248,1880) is `dem < +inf` everywhere on the device. */
__device__ __forceinline__ bool cell_valid(const double dem) { return dem < __builtin_inf(); }

__global__ void __launch_bounds__(256)
mark_nodata_kernel(double *__restrict__ dem, size_t cells, double miss) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += stride) {
    const double d = dem[i];
    if (!(d > miss)) dem[i] = __builtin_inf();
  }
}

hipError_t wdpm_launch_mark_nodata(double *dem, size_t cells, double miss, hipStream_t s) {
  size_t blocks = (cells + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(mark_nodata_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dem, cells, miss);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// one colour pass: one thread per 3x3 block of the pass, in place
// ---------------------------------------------------------------------------------------------
template <int MODULE>
__global__ void __launch_bounds__(256)
pass_kernel(double *__restrict__ w, const double *__restrict__ dem, SlabGeom g, int oi, int oj,
            double *totaldrain) {
  const int bc = blockIdx.x * 64 + (threadIdx.x & 63);
  const int br = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int c = oj + 3 * bc;   // centre column (padded coords)
  // centre row, slab-local (row0 % 3 == 0 keeps the colour alignment); a slab that does not start at
  // the raster's first row also has centres in its local row 0 (global row row0 = oi + 3j for oi = 3)
  const int r = oi + 3 * br - (g.row0 > 0 && oi == 3 ? 3 : 0);
  if (c > g.C || r > g.rows - 1 || r + g.row0 > g.R) return;
  const size_t ic = (size_t)r * g.ncp + c;
  const double dc_ = dem[ic];
  double wc = w[ic];
  if (!(wc > 0.0 && cell_valid(dc_))) return;       // WDPMCL.c:1099
  if (MODULE == 2 && r == g.dr && c == g.dc) return; // WDPMCL.c:1082
  double td = 0.0;
  bool drained = false;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int rr = r + nb_dr(k), cc = c + nb_dc(k);
    if (rr < 0 || rr >= g.rows) continue;   // slab edge: rows outside the slab act as NODATA
    const size_t in = (size_t)rr * g.ncp + cc;
    const double dn = dem[in];
    if (!cell_valid(dn)) continue;
    double wn = w[in];
    if (MODULE == 2) {
      if (rr == g.dr && cc == g.dc) {
        // WDPMCL.c:1980-1985: everything in the centre and in the drain cell leaves the raster
        td = (*totaldrain + wn) + wc;
        drained = true;
        wn = 0.0;
        wc = 0.0;
      } else {
        flow_drain(dc_, wc, dn, wn);
      }
    } else {
      flow_add(dc_, wc, dn, wn);
    }
    w[in] = wn;
  }
  w[ic] = wc;
  if (MODULE == 2 && drained) *totaldrain = td;
}

hipError_t wdpm_launch_pass(int module, double *w, const double *dem, const SlabGeom &g, int oi, int oj,
                            double *totaldrain, hipStream_t s) {
  const int nbc = (g.C - oj) / 3 + 1;               // centres oj, oj+3, ... <= C
  int rmax = g.rows - 1;
  if (g.R - g.row0 < rmax) rmax = g.R - g.row0;
  const int rfirst = oi - (g.row0 > 0 && oi == 3 ? 3 : 0);
  if (rmax < rfirst || g.C < oj) return hipSuccess;
  const int nbr = (rmax - rfirst) / 3 + 1;
  dim3 grid((nbc + 63) / 64, (nbr + 3) / 4);
  if (module == 2)
    hipLaunchKernelGGL(pass_kernel<2>, grid, dim3(256), 0, s, w, dem, g, oi, oj, totaldrain);
  else
    hipLaunchKernelGGL(pass_kernel<0>, grid, dim3(256), 0, s, w, dem, g, oi, oj, totaldrain);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// drain(): totaldrain += sum of positive water on valid cells of the outlet's 3x3 (row-major
// order, starting from 0), then zero all nine cells.  One lane; once per iteration.
// ---------------------------------------------------------------------------------------------
__global__ void drain_outlet_kernel(double *__restrict__ w, const double *__restrict__ dem, SlabGeom g,
                                    double *totaldrain) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s = 0.0;
  for (int i = -1; i <= 1; i++)
    for (int j = -1; j <= 1; j++) {
      const size_t k = (size_t)(g.dr + i) * g.ncp + (g.dc + j);
      const double wk = w[k];
      if (cell_valid(dem[k]) && wk > 0) s += wk;   // WDPMCL.c:1877-1884
    }
  for (int i = -1; i <= 1; i++)
    for (int j = -1; j <= 1; j++) w[(size_t)(g.dr + i) * g.ncp + (g.dc + j)] = 0.0;  // :1885-1889
  *totaldrain = *totaldrain + s;                 // :1089
}

hipError_t wdpm_launch_drain_outlet(double *w, const double *dem, const SlabGeom &g, double *totaldrain,
                                    hipStream_t s) {
  if (g.dr < 1 || g.dr > g.rows - 2 || g.dc < 1 || g.dc > g.ncp - 2) return hipSuccess;
  hipLaunchKernelGGL(drain_outlet_kernel, dim3(1), dim3(64), 0, s, w, dem, g, totaldrain);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// flush + snapshot: w<thres -> 0 over the whole padded slab, then old = w.  16 B per lane.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
flush_snapshot_kernel(double *w, double *old, size_t cells, double thres) {   /* old == w: flush in place */
  const size_t npair = cells / 2;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  double2 *w2 = reinterpret_cast<double2 *>(w);
  double2 *o2 = reinterpret_cast<double2 *>(old);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npair; i += stride) {
    double2 v = w2[i];
    if (v.x < thres) v.x = 0;   // WDPMCL.c:1059-1062 (note: NaN and values >= thres are kept)
    if (v.y < thres) v.y = 0;
    w2[i] = v;
    o2[i] = v;
  }
  if ((cells & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    double v = w[cells - 1];
    if (v < thres) v = 0;
    w[cells - 1] = v;
    old[cells - 1] = v;
  }
}

hipError_t wdpm_launch_flush_snapshot(double *w, double *old, size_t cells, double thres, hipStream_t s) {
  size_t blocks = (cells / 2 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(flush_snapshot_kernel, dim3((unsigned)blocks), dim3(256), 0, s, w, old, cells, thres);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// DEM -> 32-bit codes (wdpm_kernels.h::DemCode), verified cell by cell with the decoder itself
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long ordered_key(const double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);      // monotone in v for all non-NaN doubles
}

double wdpm_dem_key_to_double(unsigned long long key) {
  const unsigned long long b = (key >> 63) ? (key & 0x7fffffffffffffffull) : ~key;
  union { unsigned long long u; double d; } pun;
  pun.u = b;
  return pun.d;
}

__global__ void __launch_bounds__(256)
dem_min_kernel(const double *__restrict__ dem, size_t n, unsigned long long *key) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned long long m = ~0ull, a = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double v = dem[i];
    if (cell_valid(v)) {
      const unsigned long long kx = ordered_key(v);
      m = kx < m ? kx : m;
      const unsigned long long ab = (unsigned long long)__double_as_longlong(v) & 0x7fffffffffffffffull;   // |v|: monotone as an integer
      a = ab > a ? ab : a;
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned long long o = __shfl_xor(m, off, 64), oa = __shfl_xor(a, off, 64);
    m = o < m ? o : m;
    a = oa > a ? oa : a;
  }
  if ((threadIdx.x & 63) == 0 && m != ~0ull) { atomicMin(key, m); atomicMax(key + 1, a); }
}

hipError_t wdpm_launch_dem_min(const double *dem, size_t cells, unsigned long long *key, hipStream_t s) {
  hipError_t e = hipMemsetAsync(key, 0xff, sizeof(unsigned long long), s);
  if (e == hipSuccess) e = hipMemsetAsync(key + 1, 0, sizeof(unsigned long long), s);
  if (e != hipSuccess || cells == 0) return e;
  size_t blocks = (cells + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(dem_min_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dem, cells, key);
  return hipGetLastError();
}

__global__ void __launch_bounds__(256)
dem_encode_kernel(const double *__restrict__ dem, size_t n, double k0, double D, double rD, int *__restrict__ q,
                  unsigned long long *bad) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  bool miss = false;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double v = dem[i];
    int code = (int)0x80000000;                              // NODATA (+inf on the device)
    if (cell_valid(v)) {
      const double kk = rint(v * D) - k0;
      const bool fits = fabs(kk) < 2147483647.0;
      code = fits ? (int)kk : 0;
      const bool same = __double_as_longlong(dem32_decode(code, k0, D, rD)) == __double_as_longlong(v);
      miss |= !(fits && same);
    }
    q[i] = code;
  }
  if (__ballot(miss) && (threadIdx.x & 63) == 0) atomicOr(bad, 1ull);
}

hipError_t wdpm_launch_dem_encode(const double *dem, size_t cells, double k0, double D, double rD, int *q,
                                  unsigned long long *bad, hipStream_t s) {
  hipError_t e = hipMemsetAsync(bad, 0, sizeof(unsigned long long), s);
  if (e != hipSuccess || cells == 0) return e;
  size_t blocks = (cells + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dem_encode_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dem, cells, k0, D, rD, q, bad);
  return hipGetLastError();
}

// the 32-bit codes as 16-bit offsets from one base per group of kDemGroup columns of a row (DemCode::h, ::gb); one thread
// per group - a one-off at upload
__global__ void __launch_bounds__(256)
dem16_encode_kernel(const int *__restrict__ q, int rows, int ncp, int ngroups, unsigned short *__restrict__ h,
                    int *__restrict__ gb, unsigned long long *bad) {
  const size_t total = (size_t)rows * ngroups, stride = (size_t)gridDim.x * blockDim.x;
  bool miss = false;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int r = (int)(i / ngroups), g = (int)(i % ngroups);
    const int c0 = g * kDemGroup, c1 = c0 + kDemGroup < ncp ? c0 + kDemGroup : ncp;
    const int *row = q + (size_t)r * ncp;
    long long lo = 0x7fffffffLL, hi = -0x80000000LL;
    for (int c = c0; c < c1; c++) {
      const int v = row[c];
      if (v != (int)0x80000000) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
    }
    if (hi < lo) lo = hi = 0;                                  // no valid cell in the group
    miss |= hi - lo > 65534;
    gb[i] = (int)lo;
    for (int c = c0; c < c1; c++) {
      const int v = row[c];
      h[(size_t)r * ncp + c] = v == (int)0x80000000 ? (unsigned short)0xFFFF : (unsigned short)((long long)v - lo);
    }
  }
  if (__ballot(miss) && (threadIdx.x & 63) == 0) atomicOr(bad, 1ull);
}

hipError_t wdpm_launch_dem16_encode(const int *q, int rows, int ncp, int ngroups, unsigned short *h, int *gb,
                                    unsigned long long *bad, hipStream_t s) {
  hipError_t e = hipMemsetAsync(bad, 0, sizeof(unsigned long long), s);
  if (e != hipSuccess || rows <= 0) return e;
  size_t blocks = ((size_t)rows * ngroups + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(dem16_encode_kernel, dim3((unsigned)blocks), dim3(256), 0, s, q, rows, ncp, ngroups, h, gb, bad);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// the sequential volume sum in parallel (tests/seqsum_model.py is the model and the proof sketch)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
seqsum_a_kernel(const double *__restrict__ w, const double *__restrict__ dem, size_t n, double *approx,
                unsigned *dirty) {
  __shared__ double part[4];
  const size_t lo = (size_t)blockIdx.x * kSeqSumChunk, hi = lo + kSeqSumChunk < n ? lo + kSeqSumChunk : n;
  double s = 0.0;
  bool bad = false;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256)
    if (cell_valid(dem[i])) {
      const double v = w[i];
      bad |= !(v >= 0.0) | (v == __builtin_inf());
      s += v;
    }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  const bool any_bad = __syncthreads_or(bad);
  if (threadIdx.x == 0) {
    approx[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
    dirty[blockIdx.x] = any_bad ? 1u : 0u;
  }
}

__global__ void __launch_bounds__(256)
seqsum_b_kernel(const double *__restrict__ w, const double *__restrict__ dem, size_t n, const int *__restrict__ kexp,
                long long *isum, unsigned *tie) {
  __shared__ long long part[4];
  const int k = kexp[blockIdx.x];
  if (k == (int)0x80000000) return;                        // this chunk is summed term by term on the host
  const size_t lo = (size_t)blockIdx.x * kSeqSumChunk, hi = lo + kSeqSumChunk < n ? lo + kSeqSumChunk : n;
  long long q = 0;
  bool half = false;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256)
    if (cell_valid(dem[i])) {
      const double t = __builtin_ldexp(w[i], 52 - k);        // exact scaling to units of u = 2^(k-52)
      const double m = floor(t);
      const double r = t - m;                                // exact
      q += (long long)m + (r > 0.5 ? 1 : 0);
      half |= r == 0.5;
    }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = q;
  const bool any_half = __syncthreads_or(half);
  if (threadIdx.x == 0) {
    isum[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
    tie[blockIdx.x] = any_half ? 1u : 0u;
  }
}

hipError_t wdpm_launch_seqsum_a(const double *w, const double *dem, size_t n, double *approx, unsigned *dirty,
                                hipStream_t s) {
  if (n == 0) return hipSuccess;
  const unsigned blocks = (unsigned)((n + kSeqSumChunk - 1) / kSeqSumChunk);
  hipLaunchKernelGGL(seqsum_a_kernel, dim3(blocks), dim3(256), 0, s, w, dem, n, approx, dirty);
  return hipGetLastError();
}

hipError_t wdpm_launch_seqsum_b(const double *w, const double *dem, size_t n, const int *kexp, long long *isum,
                                unsigned *tie, hipStream_t s) {
  if (n == 0) return hipSuccess;
  const unsigned blocks = (unsigned)((n + kSeqSumChunk - 1) / kSeqSumChunk);
  hipLaunchKernelGGL(seqsum_b_kernel, dim3(blocks), dim3(256), 0, s, w, dem, n, kexp, isum, tie);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// does the raster hold a negative zero?  (decides which add/subtract stencil variant is exact)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
scan_water_kernel(const double *__restrict__ p, const double *__restrict__ dem, size_t n, unsigned long long *flag) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned hit = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double w = p[i];
    const unsigned long long bits = (unsigned long long)__double_as_longlong(w);
    if (bits == 0x8000000000000000ull) hit |= 1u;                       // -0.0
    else if (w < 0.0) hit |= 2u;                                         // negative: gone after a flush with a threshold >= 0
    else if (!(w <= 1e290) || (w > 0.0 && !(dem[i] < __builtin_inf()))) hit |= 4u;   // NaN, absurdly large, or water on a NODATA cell
  }
#pragma unroll
  for (unsigned b = 1; b <= 4; b <<= 1)
    if (__ballot((hit & b) != 0) && (threadIdx.x & 63) == 0) atomicOr(flag, (unsigned long long)b);
}

hipError_t wdpm_launch_scan_water(const double *p, const double *dem, size_t n, unsigned long long *flag, hipStream_t s) {
  if (n == 0) return hipSuccess;
  size_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(scan_water_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, dem, n, flag);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// max_diff: per-lane running max with the reference's "if (d > m) m = d" (NaN never wins),
// wave64 xor-shuffle reduction, LDS across the 4 waves, one atomicMax per workgroup.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_max(double m) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double o = __shfl_xor(m, off, 64);
    if (o > m) m = o;
  }
  return m;
}

__global__ void __launch_bounds__(256)
max_diff_kernel(const double *__restrict__ w, const double *__restrict__ old, const double old_thres,
                const double *__restrict__ dem, size_t first, size_t last, int seed_cell0,
                unsigned long long *result_bits) {
  __shared__ double part[4];
  double m = 0.0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = first + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < last; i += stride) {
    if (cell_valid(dem[i]) || (seed_cell0 && i == 0)) {   // WDPMCL.c:1245 seeds with diff[0][0]
      double o = old[i];
      if (o < old_thres) o = 0;                           // the flush the snapshot is still owed (WDPMCL.c:1059-1062)
      const double d = fabs(w[i] - o);
      if (d > m) m = d;
    }
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; k++)
      if (part[k] > m) m = part[k];
    atomicMax(result_bits, (unsigned long long)__double_as_longlong(m));  // m >= 0: order-preserving
  }
}

hipError_t wdpm_launch_max_diff(const double *w, const double *old, double old_thres, const double *dem, const SlabGeom &g,
                                int row_lo, int row_hi, unsigned long long *result_bits, hipStream_t s) {
  hipError_t e = hipMemsetAsync(result_bits, 0, sizeof(unsigned long long), s);
  if (e != hipSuccess) return e;
  if (row_hi <= row_lo) return hipSuccess;
  const size_t first = (size_t)row_lo * g.ncp, last = (size_t)row_hi * g.ncp;
  size_t blocks = (last - first + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(max_diff_kernel, dim3((unsigned)blocks), dim3(256), 0, s, w, old, old_thres, dem, first, last,
                     row_lo == 0 ? 1 : 0, result_bits);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Set-up and final statistics on the device: what the reference does in host loops
// around the block loop (WDPMCL.c:643-650, :727-740, :796-807, :879-885, :1005-1017, :1379-1459).
// ---------------------------------------------------------------------------------------------
/* padded slab rows [0, rows) from the UNPADDED file rasters: border = (+inf for NODATA, 0) (:796-807), the
 * module's water adjustment on valid cells (add :727-740, subtract :879-885); fdem / fwater point at file row
 * (row0 - 1) of the staged file rasters (fwater may be null: zeros) */
__global__ void __launch_bounds__(256)
pad_setup_kernel(const double *__restrict__ fdem, const double *__restrict__ fwater, double *__restrict__ dem,
                 double *__restrict__ w, SlabGeom g, int op, double add, double rof, double sub) {
  const size_t n = (size_t)g.rows * g.ncp, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int r = (int)(i / g.ncp), c = (int)(i % g.ncp), gr = r + g.row0;     // padded coordinates
    double d = __builtin_inf(), v = 0.0;
    if (gr >= 1 && gr <= g.R && c >= 1 && c <= g.C) {
      const size_t k = (size_t)r * g.C + (c - 1);        // fdem already starts at file row row0 - 1
      d = fdem[k];
      v = fwater ? fwater[k] : 0.0;
      if (d > g.miss) {
        if (op == 1) {                                   // :727-740: wet cells += add, then cells <= 0 = add * rof
          if (v > 0) v += add;
          if (v <= 0) v = add * rof;
        } else if (op == 2) {                            // :879-885
          const double t = v - sub;
          v = t > 0 ? t : 0;
        }
      } else {
        d = __builtin_inf();
      }
    }
    dem[i] = d;
    w[i] = v;
  }
}

hipError_t wdpm_launch_pad_setup(const double *fdem, const double *fwater, double *dem, double *w, const SlabGeom &g,
                                 int op, double add, double rof, double sub, hipStream_t s) {
  size_t blocks = ((size_t)g.rows * g.ncp + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pad_setup_kernel, dim3((unsigned)blocks), dim3(256), 0, s, fdem, fwater, dem, w, g, op, add, rof, sub);
  return hipGetLastError();
}

/* out[0] += valid cells, out[1] += valid cells with water > 0.001 (:1397-1404), out[2] = max over cells of
 * (valid ? water : miss) as an ordered key (:1448-1457 after the masking of :1386-1391; `>` from -inf) */
__global__ void __launch_bounds__(256)
count_stats_kernel(const double *__restrict__ w, const double *__restrict__ dem, size_t first, size_t last, double miss,
                   unsigned long long *out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned long long nv = 0, nw = 0;
  double m = -__builtin_inf();
  for (size_t i = first + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < last; i += stride) {
    const bool ok = cell_valid(dem[i]);
    const double v = ok ? w[i] : miss;
    nv += ok;
    nw += ok && v > 0.001;
    if (v > m) m = v;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    nv += __shfl_xor(nv, off, 64);
    nw += __shfl_xor(nw, off, 64);
    const double o = __shfl_xor(m, off, 64);
    if (o > m) m = o;
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&out[0], nv);
    atomicAdd(&out[1], nw);
    atomicMax(&out[2], ordered_key(m));
  }
}

hipError_t wdpm_launch_count_stats(const double *w, const double *dem, size_t first, size_t last, double miss,
                                   unsigned long long *out3, hipStream_t s) {
  hipError_t e = hipMemsetAsync(out3, 0, 3 * sizeof(unsigned long long), s);   /* key 0 sorts below every double */
  if (e != hipSuccess || last <= first) return e;
  size_t blocks = (last - first + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(count_stats_kernel, dim3((unsigned)blocks), dim3(256), 0, s, w, dem, first, last, miss, out3);
  return hipGetLastError();
}

/* drain-cell search (:1005-1017): the smallest dem > 0 and, among the cells that hold it, the first in
 * row-major order: pass 1 the value, pass 2 the index */
__global__ void __launch_bounds__(256)
drain_min_kernel(const double *__restrict__ dem, size_t first, size_t last, unsigned long long *key) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned long long m = ~0ull;
  for (size_t i = first + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < last; i += stride) {
    const double v = dem[i];
    if (v > 0 && v < 100000000.0) {          // :1007,1011 (NODATA is +inf here; a NODATA value >= 0 is the caller's case)
      const unsigned long long kx = ordered_key(v);
      m = kx < m ? kx : m;
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned long long o = __shfl_xor(m, off, 64);
    m = o < m ? o : m;
  }
  if ((threadIdx.x & 63) == 0 && m != ~0ull) atomicMin(key, m);
}

__global__ void __launch_bounds__(256)
drain_first_kernel(const double *__restrict__ dem, size_t first, size_t last, const unsigned long long *key,
                   unsigned long long *index) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const unsigned long long want = *key;
  unsigned long long m = ~0ull;
  for (size_t i = first + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < last; i += stride) {
    const double v = dem[i];
    if (v > 0 && ordered_key(v) == want) m = i < m ? i : m;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned long long o = __shfl_xor(m, off, 64);
    m = o < m ? o : m;
  }
  if ((threadIdx.x & 63) == 0 && m != ~0ull) atomicMin(index, m);
}

hipError_t wdpm_launch_find_drain(const double *dem, size_t first, size_t last, unsigned long long *key_and_index,
                                  hipStream_t s) {
  hipError_t e = hipMemsetAsync(key_and_index, 0xff, 2 * sizeof(unsigned long long), s);
  if (e != hipSuccess || last <= first) return e;
  size_t blocks = (last - first + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(drain_min_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dem, first, last, key_and_index);
  hipLaunchKernelGGL(drain_first_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dem, first, last, key_and_index,
                     key_and_index + 1);
  return hipGetLastError();
}

/* the un-padded raster of file rows [frow, frow + nrows): out[(r - frow) * C + c] = water, or miss on NODATA
 * cells when mask is set (:1379-1392, the scratch variants :1336-1344) */
__global__ void __launch_bounds__(256)
unpad_kernel(const double *__restrict__ w, const double *__restrict__ dem, SlabGeom g, int frow, int nrows, int mask,
             double *__restrict__ out) {
  const size_t n = (size_t)nrows * g.C, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int r = (int)(i / g.C), c = (int)(i % g.C);
    const size_t k = (size_t)(frow + r + 1 - g.row0) * g.ncp + (c + 1);
    const double v = w[k];
    out[i] = mask && !cell_valid(dem[k]) ? g.miss : v;
  }
}

hipError_t wdpm_launch_unpad(const double *w, const double *dem, const SlabGeom &g, int frow, int nrows, int mask,
                             double *out, hipStream_t s) {
  if (nrows <= 0) return hipSuccess;
  size_t blocks = ((size_t)nrows * g.C + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(unpad_kernel, dim3((unsigned)blocks), dim3(256), 0, s, w, dem, g, frow, nrows, mask, out);
  return hipGetLastError();
}
