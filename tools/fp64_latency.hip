// fp64 VALU issue vs dependent-chain latency on gfx950 (one wave on a SIMD): cycles per instruction for
// 1, 2, 4 independent chains of v_add_f64 / v_max_f64 / v_cndmask pairs.  Tooling, not product.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int CHAINS>
__global__ void chain_add(double *out, long long *cyc, double c, int iters) {
  double x[CHAINS];
  for (int k = 0; k < CHAINS; k++) x[k] = threadIdx.x + k;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 64 / CHAINS; u++)
#pragma unroll
      for (int k = 0; k < CHAINS; k++) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[k]) : "v"(c));
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int k = 0; k < CHAINS; k++) s += x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// the neighbour step's real shape: add, add, cmp, 2 cndmask, ldexp, max, add, add on one chain
template <int CHAINS>
__global__ void chain_step(double *out, long long *cyc, double dn, int iters) {
  double wc[CHAINS], wn[CHAINS];
  for (int k = 0; k < CHAINS; k++) { wc[k] = 1.0 + threadIdx.x + k; wn[k] = 0.5; }
  const double dc = 500.0;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
      for (int k = 0; k < CHAINS; k++) {
        const double en = dn + wn[k];
        const double ht = (dc + wc[k]) - en;
        const double x = (dc > en) ? wc[k] : ht;
        double f;
        asm("v_max_f64 %0, %1, %2" : "=v"(f) : "v"(x * 0.125), "v"(-0.0));
        wc[k] = wc[k] - __builtin_fabs(f);
        wn[k] = wn[k] + f;
      }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int k = 0; k < CHAINS; k++) s += wc[k] + wn[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  double *out; long long *cyc, h[4];
  CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&cyc, 4 * sizeof(long long)));
  const int iters = 2000;
#define RUN(K, NAME, NINSTR)                                                                        \
  K<<<1, 64>>>(out, cyc, 1.25, iters); CK(hipDeviceSynchronize()); K<<<1, 64>>>(out, cyc, 1.25, iters); \
  CK(hipDeviceSynchronize()); CK(hipMemcpy(h, cyc, sizeof(long long), hipMemcpyDeviceToHost));      \
  printf("%-44s %6.2f s_memtime ticks per instruction\n", NAME, (double)h[0] / ((double)iters * (NINSTR)));
  RUN(chain_add<1>, "v_add_f64, 1 dependent chain", 64)
  RUN(chain_add<2>, "v_add_f64, 2 independent chains", 64)
  RUN(chain_add<4>, "v_add_f64, 4 independent chains", 64)
  RUN(chain_add<8>, "v_add_f64, 8 independent chains", 64)
  RUN(chain_step<1>, "neighbour step (10 instr), 1 chain", 80)
  RUN(chain_step<2>, "neighbour step (10 instr), 2 chains", 160)
  RUN(chain_step<4>, "neighbour step (10 instr), 4 chains", 320)
  // s_memtime runs at a fixed 100 MHz on this part; report the ratio to a known 4-cycle stream instead
  return 0;
}
