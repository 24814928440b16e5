// What does the VOP3 clamp bit do on an fp64 result on gfx950?  (round 4: the neighbour step's
// `max(x / 8, -0.0)` as ONE instruction, `v_ldexp_f64 f, x, -3 clamp`.)
// Prints, for adversarial x: the two-instruction form, the clamped form, and whether they agree up to the
// sign of a zero whenever x / 8 <= 1.  hipcc --offload-arch=gfx950 -O2 tools/clamp_probe.hip -o clamp_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void probe(const double *x, double *two, double *one, double *mulc, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double a, b, c, m;
  asm volatile("v_ldexp_f64 %0, %1, -3" : "=v"(a) : "v"(v));
  asm volatile("v_max_f64 %0, %1, %2" : "=v"(m) : "v"(a), "v"(-0.0));
  asm volatile("v_ldexp_f64 %0, %1, -3 clamp" : "=v"(b) : "v"(v));
  const double eighth = 0.125;
  asm volatile("v_mul_f64 %0, %1, %2 clamp" : "=v"(c) : "v"(v), "v"(eighth));
  two[i] = m; one[i] = b; mulc[i] = c;
}

static uint64_t bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static double from(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

int main() {
  std::vector<double> x;
  const double specials[] = {0.0, -0.0, 1.0, 8.0, 7.999999999999999, 8.000000000000002, 16.0, 1e300, -1.0, -1e-320, 1e-320,
                             4.9e-324, -4.9e-324, 2.2250738585072014e-308, 1.7800590868057611e-307, 3.9e-323, 4.4e-323,
                             INFINITY, -INFINITY, NAN, -NAN, 0.1, 0.3, 2.5, 1e-17, 7.5, 3.0, 24.0, 1e-310, 7e-323};
  for (double s : specials) x.push_back(s);
  x.push_back(from(0x7ff0000000000001ull));   // signalling NaN
  x.push_back(from(0xfff8000000000123ull));
  uint64_t st = 0x9e3779b97f4a7c15ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
  for (int i = 0; i < 4000000; i++) {
    const uint64_t r = rnd();
    const int kind = r & 7;
    double v;
    if (kind == 0) v = from(rnd());                                              // any bit pattern
    else if (kind == 1) v = from(rnd() & 0x000fffffffffffffull) * ((r & 8) ? -1 : 1);   // subnormals
    else if (kind == 2) v = from((rnd() & 0x000fffffffffffffull) | ((uint64_t)(1 + (r >> 8) % 6) << 52));   // just above the subnormals: x/8 becomes subnormal
    else if (kind == 3) v = ldexp((double)(rnd() >> 11) * 0x1p-53, (int)((r >> 8) % 8) - 2);   // depths up to 32 m
    else if (kind == 4) v = -ldexp((double)(rnd() >> 11) * 0x1p-53, (int)((r >> 8) % 40) - 30);
    else v = ldexp((double)(rnd() >> 11) * 0x1p-53, -(int)((r >> 8) % 60));     // small positive depths
    x.push_back(v);
  }
  const int n = (int)x.size();
  double *dx, *d2, *d1, *dm;
  hipMalloc(&dx, n * 8); hipMalloc(&d2, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&dm, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3((n + 255) / 256), dim3(256), 0, 0, dx, d2, d1, dm, n);
  std::vector<double> two(n), one(n), mulc(n);
  hipMemcpy(two.data(), d2, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(one.data(), d1, n * 8, hipMemcpyDeviceToHost);
  if (hipMemcpy(mulc.data(), dm, n * 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 2; }
  for (int i = 0; i < 34; i++)
    printf("x=%-24.17g (%016llx)  ldexp,max=%016llx  ldexp clamp=%016llx  mul clamp=%016llx\n", x[i], (unsigned long long)bits(x[i]),
           (unsigned long long)bits(two[i]), (unsigned long long)bits(one[i]), (unsigned long long)bits(mulc[i]));
  long bad_le = 0, bad_gt = 0, bad_mul = 0, zero_sign = 0, checked = 0;
  for (int i = 0; i < n; i++) {
    const bool small = !(x[i] > 8.0);            // NaN counts as small: no transfer
    const uint64_t t = bits(two[i]), o = bits(one[i]), m = bits(mulc[i]);
    if (small) {
      checked++;
      const bool same = t == o || ((t << 1) == 0 && (o << 1) == 0);
      if (t != o && same) zero_sign++;
      if (!same) { if (bad_le++ < 10) printf("MISMATCH x=%.17g two=%.17g one=%.17g\n", x[i], two[i], one[i]); }
      if (o != m) { if (bad_mul++ < 10) printf("MUL differs x=%.17g ldexp-clamp=%.17g mul-clamp=%.17g\n", x[i], one[i], mulc[i]); }
    } else if (one[i] != 1.0) bad_gt++;
  }
  printf("n=%d  checked (x <= 8 or NaN)=%ld  mismatches=%ld  (zero-sign-only differences=%ld)  x>8 not clamped to 1.0: %ld  mul-vs-ldexp differences: %ld\n",
         n, checked, bad_le, zero_sign, bad_gt, bad_mul);
  return bad_le ? 1 : 0;
}
