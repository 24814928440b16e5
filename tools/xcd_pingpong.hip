// Cost of handing data from one wave to a wave on ANOTHER XCD through device memory (release store of a flag after a
// 1.5 KB payload, acquire load + payload read on the other side): the synchronisation a persistent small-raster kernel
// would pay once per iteration instead of the ~2 us between dependent launches.  Every spin is bounded.  Tooling, not product.
//   hipcc --offload-arch=gfx950 -O2 tools/xcd_pingpong.hip -o /tmp/xcd_pingpong && /tmp/xcd_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void pingpong(int *flag, double *payload, long long *cycles, int *fail, int rounds, int partner_block) {
  const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == partner_block ? 1 : -1);
  if (me < 0) return;
  const int lane = threadIdx.x;
  double acc = 0.0;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < rounds; r++) {
    const int want = 2 * r + me;              // me = 0 starts: waits for 2r (0 at first), writes 2r+1; me = 1 waits for 2r+1, writes 2r+2
    int spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
      if (++spins > (1 << 22)) { if (lane == 0) *fail = 1; return; }
    }
    for (int k = 0; k < 3; k++) acc += __builtin_nontemporal_load(payload + (1 - me) * 192 + 64 * k + lane);   // what the partner wrote
    for (int k = 0; k < 3; k++) __builtin_nontemporal_store(acc + r + k, payload + me * 192 + 64 * k + lane);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) __hip_atomic_store(flag, want + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) { cycles[me] = t1 - t0; payload[400 + me] = acc; }
}

int main() {
  int *flag, *fail; double *payload; long long *cycles;
  CK(hipMalloc(&flag, 4)); CK(hipMalloc(&fail, 4)); CK(hipMalloc(&payload, 512 * 8)); CK(hipMalloc(&cycles, 16));
  const int rounds = 2000;
  for (int partner : {1, 8, 9, 4}) {          // blocks 0 and 1: neighbouring XCDs; 0 and 8: the same XCD (dealt round-robin over 8)
    CK(hipMemset(flag, 0, 4)); CK(hipMemset(fail, 0, 4)); CK(hipMemset(payload, 0, 512 * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(pingpong, dim3(16), dim3(64), 0, 0, flag, payload, cycles, fail, rounds, partner);
    CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    int f; long long c[2]; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c, cycles, 16, hipMemcpyDeviceToHost));
    printf("blocks 0 <-> %d: %s  %.3f us per one-way hand-off (kernel %.3f ms for %d round trips; s_memtime %lld ticks)\n", partner,
           f ? "SPIN LIMIT HIT" : "ok", ms * 1e3 / (2.0 * rounds), ms, rounds, c[0]);
  }
  return 0;
}
