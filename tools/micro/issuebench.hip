// Microbench: what vector instructions cost next to MFMAs on one SIMD (gfx950), order pinned by inline asm.
// Per loop iteration a wave issues 64 MFMAs (4 accumulators, round robin) and 64 vector instructions
// (16 x {v_fma_f32, v_exp_f32, v_add_f32, v_fma_f32}, independent of the MFMAs), either
//   mode 0: MFMAs only            mode 1: vector instructions only
//   mode 2: MFMAs, then the 64 vector instructions in one block
//   mode 3: interleaved -- 4 vector instructions after every 4 MFMAs
//   mode 4: MFMAs with LDS reads as the kernels issue them for the weight fragments: one ds_read_b128
//           per MFMA (bf16) or per 4 MFMAs (fp32), conflict free, results waited for once per 4 MFMAs
//   mode 5: mode 4 + the 4 vector instructions after every 4 MFMAs
// with 1 or 2 waves per SIMD, for v_mfma_f32_16x16x4_f32 (fp32) and v_mfma_f32_16x16x32_bf16.
// Prints shader cycles per iteration per SIMD (all waves of the SIMD together).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

#define MF32(acc) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(af), "v"(bf))
#define MBF(acc) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(ab), "v"(bb))
#define LDSR(dst, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(laddr))
#define VEC4(t, u) asm volatile("v_fma_f32 %0, %0, %2, %1\n\tv_exp_f32 %1, %0\n\tv_add_f32 %1, %1, %0\n\tv_fma_f32 %0, %1, %2, %0" : "+v"(t), "+v"(u) : "v"(cst))

template <int BF, int MODE>
__global__ void __launch_bounds__(512) k(int iters, float* sink, unsigned long long* clk) {
  const int lane = threadIdx.x & 63;
  float af = 0.001f * lane, bf = 0.002f * lane, cst = 0.5f;
  bf8 ab, bb;
  for (int e = 0; e < 8; ++e) ab[e] = (__bf16)(0.001f * (lane + e)), bb[e] = (__bf16)(0.002f * (lane - e));
  f4 acc[4];
  for (int n = 0; n < 4; ++n) acc[n] = (f4){0, 0, 0, 0};
  float t[4] = {0.1f * lane, 0.2f, 0.3f, 0.4f}, u[4] = {0.f, 0.f, 0.f, 0.f};
  extern __shared__ float lds[];
  const unsigned laddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds + lane * 16;
  f4 w0r, w1r, w2r, w3r;
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (MODE >= 4) {
        if (BF) { LDSR(w0r, 0); LDSR(w1r, 1024); LDSR(w2r, 2048); LDSR(w3r, 3072); }
        else LDSR(w0r, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (BF) asm volatile("" :: "v"(w0r), "v"(w1r), "v"(w2r), "v"(w3r));
        else asm volatile("" :: "v"(w0r));
      }
      if (MODE != 1) {
        if (BF) { MBF(acc[0]); MBF(acc[1]); MBF(acc[2]); MBF(acc[3]); }
        else { MF32(acc[0]); MF32(acc[1]); MF32(acc[2]); MF32(acc[3]); }
      }
      if (MODE == 3 || MODE == 1 || MODE == 5) VEC4(t[s & 3], u[s & 3]);
    }
    if (MODE == 2) {
#pragma unroll
      for (int s = 0; s < 16; ++s) VEC4(t[s & 3], u[s & 3]);
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float r = 0.f;
  for (int n = 0; n < 4; ++n) r += acc[n].x + acc[n].y + acc[n].z + acc[n].w + t[n] + u[n];
  if (r == 12345.678f) sink[0] = r;
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0, clk[1] = w1 - w0;
}

template <int BF, int MODE>
void run(int wps, float* sink, unsigned long long* clk) {
  const int threads = 256 * wps, iters = 4000, grid = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<BF, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipLaunchKernelGGL((k<BF, MODE>), dim3(grid), dim3(threads), 100 * 1024, 0, 200, sink, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<BF, MODE>), dim3(grid), dim3(threads), 100 * 1024, 0, iters, sink, clk);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2];
  hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  const double mhz = (double)h[0] / ((double)h[1] / 100.0);
  const double cyc = ms * 1e-3 * mhz * 1e6 / iters;  // cycles per iteration of the SIMD (its waves run concurrently)
  static const char* names[] = {"MFMA only", "vector only", "MFMA then vector block", "interleaved 4:4", "MFMA + LDS reads", "MFMA + LDS + vector 4:4"};
  printf("%-5s waves/SIMD %d  %-24s %8.0f cycles per iteration per wave  (%6.0f per iteration and SIMD-wave pair: /%d)  clock %.0f MHz\n",
         BF ? "bf16" : "fp32", wps, names[MODE], cyc, cyc / wps, wps, mhz);
}

int main() {
  float* sink;
  unsigned long long* clk;
  hipMalloc(&sink, 64), hipMalloc(&clk, 64);
  for (int w = 1; w <= 2; ++w) {
    run<0, 0>(w, sink, clk), run<0, 1>(w, sink, clk), run<0, 2>(w, sink, clk), run<0, 3>(w, sink, clk);
    run<0, 4>(w, sink, clk), run<0, 5>(w, sink, clk);
    run<1, 0>(w, sink, clk), run<1, 1>(w, sink, clk), run<1, 2>(w, sink, clk), run<1, 3>(w, sink, clk);
    run<1, 4>(w, sink, clk), run<1, 5>(w, sink, clk);
  }
  return 0;
}
