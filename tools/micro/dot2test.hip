// Checks what v_dot2c_f32_bf16 computes with constant and with register selectors (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, float* o, int sel0r, int sel1r) {
  const int i = threadIdx.x;
  const float x0 = x[2 * i], x1 = x[2 * i + 1];
  const bf2 h = {(__bf16)x0, (__bf16)x1};
  const bf2 c0 = {(__bf16)-1.0f, (__bf16)0.0f}, c1 = {(__bf16)0.0f, (__bf16)-1.0f};
  o[6 * i + 0] = x0 - (float)h[0];
  o[6 * i + 1] = x1 - (float)h[1];
  o[6 * i + 2] = __builtin_amdgcn_fdot2_f32_bf16(h, c0, x0, false);
  o[6 * i + 3] = __builtin_amdgcn_fdot2_f32_bf16(h, c1, x1, false);
  o[6 * i + 4] = __builtin_amdgcn_fdot2_f32_bf16(h, __builtin_bit_cast(bf2, sel0r), x0, false);
  o[6 * i + 5] = __builtin_amdgcn_fdot2_f32_bf16(h, __builtin_bit_cast(bf2, sel1r), x1, false);
}
int main() {
  float hx[128], ho[384], *dx, *dout;
  for (int i = 0; i < 128; ++i) hx[i] = sinf(i * 1.37f) * expf((i % 13) - 6.f);
  hipMalloc(&dx, sizeof hx); hipMalloc(&dout, sizeof ho);
  hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dx, dout, 0x0000bf80, (int)0xbf800000);
  hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
  int badc = 0, badr = 0;
  for (int i = 0; i < 64; ++i) {
    badc += ho[6*i+2] != ho[6*i] || ho[6*i+3] != ho[6*i+1];
    badr += ho[6*i+4] != ho[6*i] || ho[6*i+5] != ho[6*i+1];
  }
  for (int i = 0; i < 3; ++i) printf("x %g %g  ref %g %g  const %g %g  reg %g %g\n", hx[2*i], hx[2*i+1], ho[6*i], ho[6*i+1], ho[6*i+2], ho[6*i+3], ho[6*i+4], ho[6*i+5]);
  printf("mismatch const %d reg %d of 64\n", badc, badr);
  return 0;
}
