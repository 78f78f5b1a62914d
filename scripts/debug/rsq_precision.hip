// How accurate is v_rsq_f64 (and one / two Newton steps on top of it)?  hipcc -O3 --offload-arch=gfx950 scripts/debug/rsq_precision.hip -o /tmp/rsq && /tmp/rsq
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double d = x[i];
  double r = __builtin_amdgcn_rsq(d);
  r0[i] = r;
  r = r * (1.5 - 0.5 * d * r * r);
  r1[i] = r;
  r = r * (1.5 - 0.5 * d * r * r);
  r2[i] = r;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> h(n), a(n), b(n), c(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = std::ldexp(1.0 + (double)(s >> 11) / 9007199254740992.0, (int)(s % 40) - 20); }
  double *x, *r0, *r1, *r2;
  hipMalloc(&x, n * 8); hipMalloc(&r0, n * 8); hipMalloc(&r1, n * 8); hipMalloc(&r2, n * 8);
  hipMemcpy(x, h.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, x, r0, r1, r2, n);
  hipMemcpy(a.data(), r0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), r1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), r2, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < n; ++i) {
    const long double t = 1.0L / sqrtl((long double)h[i]);
    e0 = std::fmax(e0, (double)fabsl((a[i] - t) / t)); e1 = std::fmax(e1, (double)fabsl((b[i] - t) / t)); e2 = std::fmax(e2, (double)fabsl((c[i] - t) / t));
  }
  printf("max relative error of 1/sqrt: v_rsq_f64 %.3e (%.1f ulp), + one Newton step %.3e (%.2f ulp), + two %.3e (%.2f ulp)\n", e0, e0 / 1.11e-16, e1, e1 / 1.11e-16, e2, e2 / 1.11e-16);
  return 0;
}
