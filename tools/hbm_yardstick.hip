// Streaming yardstick for the MI355X HBM rate at the stencil's read/write mix (2 reads : 1 write of
// fp64 rasters): out[i] = a[i] + b[i] over 2^28 doubles, in several access shapes.  Not part of the
// product; prints GB/s of real traffic so that the fused kernel's 4.8 TB/s can be put in context.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int VEC, bool NT>
__global__ void __launch_bounds__(256) triad(const double *__restrict__ a, const double *__restrict__ b,
                                              double *__restrict__ o, size_t n) {
  size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * VEC, stride = (size_t)gridDim.x * 256 * VEC;
  for (; i + VEC <= n; i += stride) {
    double x[VEC], y[VEC];
#pragma unroll
    for (int k = 0; k < VEC; k++) { x[k] = a[i + k]; y[k] = b[i + k]; }
#pragma unroll
    for (int k = 0; k < VEC; k++) {
      if (NT) __builtin_nontemporal_store(x[k] + y[k], o + i + k); else o[i + k] = x[k] + y[k];
    }
  }
}
template <int VEC>
__global__ void __launch_bounds__(256) copy1(const double *__restrict__ a, double *__restrict__ o, size_t n) {
  size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * VEC, stride = (size_t)gridDim.x * 256 * VEC;
  for (; i + VEC <= n; i += stride) {
    double x[VEC];
#pragma unroll
    for (int k = 0; k < VEC; k++) x[k] = a[i + k];
#pragma unroll
    for (int k = 0; k < VEC; k++) o[i + k] = x[k];
  }
}
template <int VEC>
__global__ void __launch_bounds__(256) read2(const double *__restrict__ a, const double *__restrict__ b,
                                              double *__restrict__ o, size_t n) {
  size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * VEC, stride = (size_t)gridDim.x * 256 * VEC;
  double s = 0;
  for (; i + VEC <= n; i += stride) {
#pragma unroll
    for (int k = 0; k < VEC; k++) s += a[i + k] + b[i + k];
  }
  if (s == 12345.678) o[0] = s;
}

template <class F> static void run(const char *name, double bytes, F f) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) f();
  CK(hipEventRecord(e0));
  const int reps = 20;
  for (int i = 0; i < reps; i++) f();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-44s %8.3f ms  %7.0f GB/s\n", name, ms / reps, bytes / (ms / reps) / 1e6);
}

int main() {
  const size_t n = (size_t)1 << 28;                 // 2.147 GB per array, like a 16384^2 raster
  double *a, *b, *o;
  CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&o, n * 8));
  CK(hipMemset(a, 0, n * 8)); CK(hipMemset(b, 0, n * 8)); CK(hipMemset(o, 0, n * 8));
  const double B3 = 3.0 * n * 8, B2 = 2.0 * n * 8;
  for (int grid : {1024, 4096, 16384, 65536}) {
    char nm[96];
    snprintf(nm, 96, "triad 8B/lane  plain   grid %d", grid);  run(nm, B3, [&] { triad<1, false><<<grid, 256>>>(a, b, o, n); });
    snprintf(nm, 96, "triad 8B/lane  nt      grid %d", grid);  run(nm, B3, [&] { triad<1, true><<<grid, 256>>>(a, b, o, n); });
    snprintf(nm, 96, "triad 16B/lane plain   grid %d", grid);  run(nm, B3, [&] { triad<2, false><<<grid, 256>>>(a, b, o, n); });
    snprintf(nm, 96, "triad 16B/lane nt      grid %d", grid);  run(nm, B3, [&] { triad<2, true><<<grid, 256>>>(a, b, o, n); });
    snprintf(nm, 96, "copy  16B/lane         grid %d", grid);  run(nm, B2, [&] { copy1<2><<<grid, 256>>>(a, o, n); });
    snprintf(nm, 96, "read2 16B/lane         grid %d", grid);  run(nm, B2, [&] { read2<2><<<grid, 256>>>(a, b, o, n); });
  }
  // 256 workgroups of 256 threads = one wave per SIMD, like the fused kernel's residency
  run("triad 16B/lane nt  grid 256 (1 wave/SIMD)", B3, [&] { triad<2, true><<<256, 256>>>(a, b, o, n); });
  run("triad 32B/lane nt  grid 256 (1 wave/SIMD)", B3, [&] { triad<4, true><<<256, 256>>>(a, b, o, n); });
  return 0;
}
