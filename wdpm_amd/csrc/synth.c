/*
 * synth.c — deterministic synthetic DEM generator for the roofline / scaling workloads
 * (SURVEY.md §8d, configs 3-5).  Host code, plain C, integer-seeded: the same (n, seed) gives
 * the same raster bit-for-bit on every host, whatever the evaluation order, because every
 * lattice point draws its random number from a hash of (seed, x, y) rather than from a stream.
 *
 * Diamond-square fractal on a (2^L+1)^2 lattice cropped to n x n; base 500 m, initial amplitude
 * 16 m, roughness 2^-0.75 per level, planar tilt 1e-4 m/cell falling toward the last row/column,
 * quantised to 1e-4 m like dem/basin5.asc.  All cells valid (no NODATA).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/wdpm.h"

static inline uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

/* uniform in [-1, 1), a function of (seed, x, y) only */
static inline double noise(uint64_t seed, uint32_t x, uint32_t y) {
  uint64_t h = splitmix64(seed ^ splitmix64(((uint64_t)x << 32) | (uint64_t)y));
  return (double)(h >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}

int wdpm_synth_dem(int32_t n, uint64_t seed, double *dem) {
  if (n < 2 || !dem) return 1;
  int L = 1;
  while ((1 << L) < n) L++;
  const size_t N = ((size_t)1 << L) + 1;
  double *z = (double *)malloc(N * N * sizeof(double));
  if (!z) return 1;
  const double base = 500.0, rough = 0.59460355750136051; /* 2^-0.75 */
  double amp = 16.0;
  const size_t last = N - 1;
  z[0] = base + amp * noise(seed, 0, 0);
  z[last] = base + amp * noise(seed, 0, (uint32_t)last);
  z[last * N] = base + amp * noise(seed, (uint32_t)last, 0);
  z[last * N + last] = base + amp * noise(seed, (uint32_t)last, (uint32_t)last);
  for (size_t step = last; step >= 2; step /= 2) {
    const size_t half = step / 2;
    amp *= rough;
    /* diamond: centre of each square */
    for (size_t i = half; i < N; i += step)
      for (size_t j = half; j < N; j += step) {
        double s = ((z[(i - half) * N + (j - half)] + z[(i - half) * N + (j + half)]) +
                    (z[(i + half) * N + (j - half)] + z[(i + half) * N + (j + half)])) * 0.25;
        z[i * N + j] = s + amp * noise(seed, (uint32_t)i, (uint32_t)j);
      }
    /* square: edge midpoints (rows alternate between the two phases) */
    for (size_t i = 0; i < N; i += half) {
      size_t j0 = ((i / half) % 2 == 0) ? half : 0;
      for (size_t j = j0; j < N; j += step) {
        double s = 0.0;
        int cnt = 0;
        if (i >= half) { s += z[(i - half) * N + j]; cnt++; }
        if (i + half < N) { s += z[(i + half) * N + j]; cnt++; }
        if (j >= half) { s += z[i * N + (j - half)]; cnt++; }
        if (j + half < N) { s += z[i * N + (j + half)]; cnt++; }
        z[i * N + j] = s / (double)cnt + amp * noise(seed, (uint32_t)i, (uint32_t)j);
      }
    }
  }
  for (int32_t i = 0; i < n; i++)
    for (int32_t j = 0; j < n; j++) {
      double v = z[(size_t)i * N + j] - 1e-4 * (double)(i + j);
      dem[(size_t)i * n + j] = rint(v * 1e4) / 1e4;
    }
  free(z);
  return 0;
}
