// Microbench: HBM throughput of 256-byte-row streaming as a function of the request shape.
//  mode 0: lane (j,g) of a 16-row tile loads 16 B at 64*T + 16*g for T = 0..3  (4 instr, 64 B per row per instr)
//  mode 1: lane (p = lane>>3, c = lane&7) loads piece c of line h of row 8*half + p (4 instr, whole 128-B lines)
//  mode 2: lane loads 16 B at lane*16 of a contiguous 1 KiB chunk, 4 chunks (fully linear)
// Each wave sums what it loads (so loads are not eliminated) and optionally writes a 256-B row back per row.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE, bool WRITE, bool RANDOM>
__global__ void __launch_bounds__(512) k(const float* __restrict__ in, float* __restrict__ out, const int* __restrict__ perm, long long nrows, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const long long ntiles = nrows / 16;
  f4 acc = {0, 0, 0, 0};
  for (long long tile = (long long)blockIdx.x * 8 + wave; tile < ntiles; tile += (long long)gridDim.x * 8) {
    f4 x[4];
    if (MODE == 0) {
      long long row = tile * 16 + j;
      if (RANDOM) row = perm[row];
      const float* p = in + row * 64 + 4 * g;
#pragma unroll
      for (int T = 0; T < 4; ++T) x[T] = *(const f4*)(p + 16 * T);
    } else if (MODE == 1) {
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) {
        long long row = tile * 16 + (k2 & 1) * 8 + (lane >> 3);
        if (RANDOM) row = perm[row];
        x[k2] = *(const f4*)(in + row * 64 + (k2 >> 1) * 32 + (lane & 7) * 4);
      }
    } else {
      const float* p = in + tile * 16 * 64 + lane * 4;
#pragma unroll
      for (int T = 0; T < 4; ++T) x[T] = *(const f4*)(p + 256 * T);
    }
    acc += x[0] + x[1] + x[2] + x[3];
    if (WRITE) {
      long long row = tile * 16 + j;
      if (RANDOM) row = perm[row];
      float* q = out + row * 64 + 4 * g;
#pragma unroll
      for (int T = 0; T < 4; ++T) *(f4*)(q + 16 * T) = x[T];
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}
template <int MODE, bool WRITE, bool RANDOM>
void run(const char* name, const float* in, float* out, const int* perm, long long nrows, float* sink) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<MODE, WRITE, RANDOM>), dim3(256), dim3(512), 0, 0, in, out, perm, nrows, sink);
  hipEventRecord(a);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<MODE, WRITE, RANDOM>), dim3(256), dim3(512), 0, 0, in, out, perm, nrows, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
  const double bytes = (double)nrows * 256 * (WRITE ? 2 : 1);
  printf("%-46s %8.1f us  %7.2f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
}
int main() {
  const long long nrows = 4 << 20;  // 1 GiB of rows
  float *in, *out, *sink; int* perm;
  hipMalloc(&in, nrows * 256); hipMalloc(&out, nrows * 256); hipMalloc(&sink, 64); hipMalloc(&perm, nrows * 4);
  hipMemset(in, 0, nrows * 256);
  int* h = (int*)malloc(nrows * 4);
  for (long long i = 0; i < nrows; ++i) h[i] = (int)i;
  unsigned long long s = 88172645463325252ull;
  for (long long i = nrows - 1; i > 0; --i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; long long r = s % (i + 1); int t = h[i]; h[i] = h[r]; h[r] = t; }
  hipMemcpy(perm, h, nrows * 4, hipMemcpyHostToDevice);
  run<2, false, false>("linear 1 KiB chunks, read", in, out, perm, nrows, sink);
  run<0, false, false>("64 B per row per instr, read, sequential rows", in, out, perm, nrows, sink);
  run<1, false, false>("whole lines per instr, read, sequential rows", in, out, perm, nrows, sink);
  run<0, false, true>("64 B per row per instr, read, random rows", in, out, perm, nrows, sink);
  run<1, false, true>("whole lines per instr, read, random rows", in, out, perm, nrows, sink);
  run<2, true, false>("linear read + 64B-piece write, sequential", in, out, perm, nrows, sink);
  run<0, true, false>("64 B pieces read+write, sequential rows", in, out, perm, nrows, sink);
  run<1, true, false>("whole-line read + 64B-piece write, sequential", in, out, perm, nrows, sink);
  run<0, true, true>("64 B pieces read+write, random rows", in, out, perm, nrows, sink);
  run<1, true, true>("whole-line read + 64B-piece write, random rows", in, out, perm, nrows, sink);
  return 0;
}
