// Practical FP64 ceilings of the device: register-only v_mfma_f64_16x16x4_f64 and v_fma_f64 streams (no memory traffic),
// with the shader clock measured inside the kernel (s_memtime ticks per 100 MHz wall-clock tick).
//   hipcc -O3 --offload-arch=gfx950 scripts/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, long long* clk, int iters) {
  double4_t acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = double4_t{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
template <int NACC>
__global__ __launch_bounds__(256) void k_fma(double* out, long long* clk, int iters) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = i;
  double a = 1.0 + threadIdx.x * 1e-9, b = threadIdx.x * 1e-7;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
template <class F>
void run(const char* what, F launch, int blocks, double flops) {
  double* out; hipMalloc(&out, sizeof(double) * blocks * 256);
  long long* clk; hipMalloc(&clk, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(out, clk, true);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch(out, clk, false);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  printf("%-34s %8.2f ms  %6.1f TFLOP/s   shader clock during the loop: %.0f MHz\n", what, ms, flops / ms / 1e9, 100.0 * (double)h[0] / (double)h[1]);
  hipFree(out); hipFree(clk);
}
int main() {
  for (int wps : {1, 2, 4}) {
    const int blocks = 256 * wps, iters = 20000;
    char name[64]; snprintf(name, sizeof name, "mfma_f64_16x16x4, %d waves/SIMD", wps);
    run(name, [&](double* o, long long* c, bool warm) { hipLaunchKernelGGL(k_mfma<16>, dim3(blocks), dim3(256), 0, 0, o, c, warm ? 10 : iters); },
        blocks, (double)blocks * 4 * iters * 16 * 2048.0);
  }
  for (int wps : {1, 2, 4}) {
    const int blocks = 256 * wps, iters = 200000;
    char name[64]; snprintf(name, sizeof name, "v_fma_f64, %d waves/SIMD", wps);
    run(name, [&](double* o, long long* c, bool warm) { hipLaunchKernelGGL(k_fma<16>, dim3(blocks), dim3(256), 0, 0, o, c, warm ? 10 : iters); },
        blocks, (double)blocks * 256 * (double)iters * 16 * 2.0);
  }
  return 0;
}
