// Practical FP64 ceilings of the device: register-only v_mfma_f64_16x16x4_f64, v_mfma_f64_4x4x4_4b and v_fma_f64 streams, and
// MFMA + VALU-FMA mixes in one wave / in neighbouring waves (can the vector ALU add to the matrix pipe?) -- no memory traffic,
// with the shader clock measured inside the kernel (s_memtime ticks per 100 MHz wall-clock tick).
//   hipcc -O3 --offload-arch=gfx950 scripts/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
// NOTE on launch bounds: with __launch_bounds__(256) (512 registers per lane available) the compiler parks the accumulators
// in AGPRs and copies all of them to VGPRs and back EVERY loop iteration (256 v_accvgpr moves per 16 MFMAs); that stream
// measures 36 TFLOP/s and was round 1's "measured ceiling".  It is an artefact of the benchmark: with the accumulators in
// VGPRs (bounds 512 -> 256 registers) the same instruction stream issues one v_mfma_f64_16x16x4 per 64 cycles = 78 TFLOP/s,
// the spec-sheet figure.  Both forms are kept so that the difference stays on record.
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma_agpr(double* out, long long* clk, int iters) {
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
__global__ __launch_bounds__(512) void k_mfma(double* out, long long* clk, int iters) {
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
// v_mfma_f64_4x4x4_4b_f64: four 4x4x4 blocks per instruction, one accumulator double per lane, 512 flop per wave
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma4(double* out, long long* clk, int iters) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = 0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
// MFMA and VALU FMA in ONE wave, interleaved: NF v_fma_f64 per v_mfma_f64_16x16x4 (independent accumulators on both sides).
// If the matrix pipe runs beside the vector ALU the FMAs are free until they fill the 128-cycle MFMA slot (32 FMAs).
template <int NF>
__global__ __launch_bounds__(256) void k_mix(double* out, long long* clk, int iters) {
  double4_t acc[8];
  double f[NF > 0 ? NF : 1];
  for (int i = 0; i < 8; ++i) acc[i] = double4_t{0, 0, 0, 0};
  for (int i = 0; i < NF; ++i) f[i] = i;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4, fa = 1.0 + threadIdx.x * 1e-9, fb = threadIdx.x * 1e-7;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < NF; ++k) f[k] = __builtin_fma(f[k], fa, fb);
    }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < NF; ++i) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
// MFMA waves and FMA waves side by side on the same SIMD: workgroups of 8 waves (2 per SIMD), waves 0-3 issue MFMAs, 4-7 FMAs
__global__ __launch_bounds__(512) void k_split(double* out, long long* clk, int iters_mfma, int iters_fma) {
  double s = 0;
  const long long c0 = clock64(), w0 = wall_clock64();
  if ((threadIdx.x >> 8) == 0) {
    double4_t acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = double4_t{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters_mfma; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    double f[16];
    for (int i = 0; i < 16; ++i) f[i] = i;
    double fa = 1.0 + threadIdx.x * 1e-9, fb = threadIdx.x * 1e-7;
    for (int it = 0; it < iters_fma; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) f[i] = __builtin_fma(f[i], fa, fb);
    }
    for (int i = 0; i < 16; ++i) s += f[i];
  }
  const long long c1 = clock64(), w1 = wall_clock64();
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
  for (int wps : {1, 2}) {
    const int blocks = 256 * wps, iters = 20000;
    char name[64]; snprintf(name, sizeof name, "mfma_f64_16x16x4, %d waves/SIMD", wps);
    run(name, [&](double* o, long long* c, bool warm) { hipLaunchKernelGGL(k_mfma<16>, dim3(blocks), dim3(256), 0, 0, o, c, warm ? 10 : iters); },
        blocks, (double)blocks * 4 * iters * 16 * 2048.0);
  }
  {
    const int blocks = 256, iters = 20000;
    run("  same, accumulators via AGPRs (r1)", [&](double* o, long long* c, bool warm) { hipLaunchKernelGGL(k_mfma_agpr<16>, dim3(blocks), dim3(256), 0, 0, o, c, warm ? 10 : iters); },
        blocks, (double)blocks * 4 * iters * 16 * 2048.0);
  }
  for (int wps : {1, 2, 4}) {
    const int blocks = 256 * wps, iters = 200000;
    char name[64]; snprintf(name, sizeof name, "v_fma_f64, %d waves/SIMD", wps);
    run(name, [&](double* o, long long* c, bool warm) { hipLaunchKernelGGL(k_fma<16>, dim3(blocks), dim3(256), 0, 0, o, c, warm ? 10 : iters); },
        blocks, (double)blocks * 256 * (double)iters * 16 * 2.0);
  }
  for (int wps : {1, 2, 4}) {
    const int blocks = 256 * wps, iters = 100000;
    char name[64]; snprintf(name, sizeof name, "mfma_f64_4x4x4_4b, %d waves/SIMD", wps);
    run(name, [&](double* o, long long* c, bool warm) { hipLaunchKernelGGL(k_mfma4<16>, dim3(blocks), dim3(256), 0, 0, o, c, warm ? 10 : iters); },
        blocks, (double)blocks * 4 * iters * 16 * 512.0);
  }
  {
    const int blocks = 512, iters = 20000;   // 2 waves per SIMD
    auto mix = [&](const char* name, auto kern, int nf) {
      run(name, [&](double* o, long long* c, bool warm) { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, o, c, warm ? 10 : iters); },
          blocks, (double)blocks * 4 * iters * 8 * (2048.0 + nf * 128.0));
    };
    mix("1 wave: mfma + 0 fma each (ref)", k_mix<0>, 0);
    mix("1 wave: mfma + 4 fma each", k_mix<4>, 4);
    mix("1 wave: mfma + 8 fma each", k_mix<8>, 8);
    mix("1 wave: mfma + 16 fma each", k_mix<16>, 16);
    mix("1 wave: mfma + 24 fma each", k_mix<24>, 24);
  }
  {
    // equal wall time targets: an MFMA wave needs 16 x 128 cycles per iteration, an FMA wave 16 x 4
    const int blocks = 256, im = 10000;
    for (int ratio : {0, 8, 16, 32}) {
      const int ifm = im * ratio;
      char name[64]; snprintf(name, sizeof name, "split waves: 16 mfma | %d fma", 16 * ratio);
      run(name, [&](double* o, long long* c, bool warm) { hipLaunchKernelGGL(k_split, dim3(blocks), dim3(512), 0, 0, o, c, warm ? 10 : im, warm ? 10 : ifm); },
          blocks * 2, (double)blocks * 4 * ((double)im * 16 * 2048.0 + (double)ifm * 16 * 128.0));
    }
  }
  return 0;
}
