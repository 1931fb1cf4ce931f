// Microbench: sustained bf16 MFMA rate with every CU busy, as the GNT kernels use the pipe:
// W waves per SIMD, each wave a stream of MFMAs on NACC accumulators (NACC = 1: one dependent chain).
//   mfmabench            -> table over {16x16x32, 32x32x16} x waves/SIMD {1, 2} x NACC {1, 2, 4}
// Prints cycles per MFMA per SIMD (shader clock from s_memtime against the 100 MHz wall clock) and TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int BIG, int NACC>
__global__ void __launch_bounds__(512) k(int iters, float* sink, unsigned long long* clk) {
  const int lane = threadIdx.x & 63;
  bf8 a, b;
  for (int e = 0; e < 8; ++e) a[e] = (__bf16)(0.001f * (lane + e)), b[e] = (__bf16)(0.002f * (lane - e));
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  float r = 0.f;
  if (BIG) {
    f16v acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 12; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u % NACC], 0, 0, 0);
    }
    for (int n = 0; n < NACC; ++n) for (int e = 0; e < 16; ++e) r += acc[n][e];
  } else {
    f4 acc[NACC];
    for (int n = 0; n < NACC; ++n) acc[n] = (f4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 24; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[u % NACC], 0, 0, 0);
    }
    for (int n = 0; n < NACC; ++n) r += acc[n].x + acc[n].y + acc[n].z + acc[n].w;
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  if (r == 12345.678f) sink[0] = r;
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0, clk[1] = w1 - w0;
}

template <int BIG, int NACC>
void run(int waves_per_simd, float* sink, unsigned long long* clk) {
  const int threads = 256 * waves_per_simd, iters = 20000, grid = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  // 100 KB of dynamic LDS: one workgroup per CU
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<BIG, NACC>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipLaunchKernelGGL((k<BIG, NACC>), dim3(grid), dim3(threads), 100 * 1024, 0, 1000, sink, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<BIG, NACC>), dim3(grid), dim3(threads), 100 * 1024, 0, iters, sink, clk);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2];
  hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  const double mhz = (double)h[0] / ((double)h[1] / 100.0);
  const double per_simd = (double)iters * 12 * waves_per_simd;  // 32x32x16-equivalents (a 16x16x32 is half)
  const double cyc = ms * 1e-3 * mhz * 1e6 / per_simd;
  const double tf = per_simd * 1024.0 * 32768.0 / (ms * 1e-3) / 1e12;
  printf("%s  waves/SIMD %d  accumulators %d: %.3f ms, clock %.0f MHz, %.1f cycles per 32x32x16-equivalent, %.0f TFLOP/s\n",
         BIG ? "32x32x16" : "16x16x32", waves_per_simd, NACC, ms, mhz, cyc, tf);
}

int main() {
  float* sink;
  unsigned long long* clk;
  hipMalloc(&sink, 64), hipMalloc(&clk, 64);
  for (int w = 1; w <= 2; ++w) {
    run<1, 1>(w, sink, clk), run<1, 2>(w, sink, clk), run<1, 4>(w, sink, clk);
    run<0, 1>(w, sink, clk), run<0, 2>(w, sink, clk), run<0, 4>(w, sink, clk);
  }
  return 0;
}
