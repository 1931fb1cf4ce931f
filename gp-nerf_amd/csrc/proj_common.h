// Bit-critical projection / bilinear-tap arithmetic shared by the projector kernels and the fused
// deep-semantic aggregation.  Every float operation is rounded separately and in the order the
// reference's CPU ops apply them (contract(off) is forced per function), because masks and tap
// indices must match the reference bit for bit (SURVEY.md section 7 "hard parts"):
//   P.[x,y,z,1]  : k-ascending FMA chain  == torch CPU bmm           (projection.py:44-46)
//   pix          : correctly rounded divide by clamp(z, 1e-8), clamp +-1e6   (:48-51)
//   normalise    : 2*p/(w-1) - 1                                      (:24-29)
//   unnormalise  : ((n+1)/2)*(Wm-1)   (grid_sample, align_corners=True)
//   bilinear     : w = x-floor(x), e = 1-w; nw = s*e ...; out = nw_v*nw, then fma(ne), fma(sw), fma(se)
#pragma once
#include <hip/hip_runtime.h>

struct GpnPix {
  float px, py;  // clamped pixel coordinates in the source image
  bool front;    // projections.z > 0
};

__device__ __forceinline__ GpnPix gpn_project(const float* __restrict__ P, float x, float y, float z) {
#pragma clang fp contract(off)
  float X = P[0] * x;
  X = __builtin_fmaf(P[1], y, X);
  X = __builtin_fmaf(P[2], z, X);
  X = __builtin_fmaf(P[3], 1.0f, X);
  float Y = P[4] * x;
  Y = __builtin_fmaf(P[5], y, Y);
  Y = __builtin_fmaf(P[6], z, Y);
  Y = __builtin_fmaf(P[7], 1.0f, Y);
  float Z = P[8] * x;
  Z = __builtin_fmaf(P[9], y, Z);
  Z = __builtin_fmaf(P[10], z, Z);
  Z = __builtin_fmaf(P[11], 1.0f, Z);
  const float zc = Z < 1e-8f ? 1e-8f : Z;  // torch.clamp(min=1e-8)
  GpnPix r;
  r.px = fminf(fmaxf(X / zc, -1e6f), 1e6f);
  r.py = fminf(fmaxf(Y / zc, -1e6f), 1e6f);
  r.front = Z > 0.0f;
  return r;
}

__device__ __forceinline__ bool gpn_inbound(float px, float py, float h, float w) {
#pragma clang fp contract(off)
  return (px <= w - 1.0f) && (px >= 0.0f) && (py <= h - 1.0f) && (py >= 0.0f);
}

struct GpnTaps {
  int x0, y0;            // north-west tap
  float nw, ne, sw, se;  // bilinear weights
  bool vx0, vx1, vy0, vy1;
};

// Sampling position of pixel (px,py) of an (img_h,img_w) image on a (mh,mw) map.
__device__ __forceinline__ GpnTaps gpn_taps(float px, float py, float img_h, float img_w, int mh,
                                            int mw) {
#pragma clang fp contract(off)
  const float nx = (2.0f * px) / (img_w - 1.0f) - 1.0f;
  const float ny = (2.0f * py) / (img_h - 1.0f) - 1.0f;
  const float ix = ((nx + 1.0f) / 2.0f) * (float)(mw - 1);
  const float iy = ((ny + 1.0f) / 2.0f) * (float)(mh - 1);
  const float xf = floorf(ix), yf = floorf(iy);
  const float w = ix - xf, e = 1.0f - w;
  const float n = iy - yf, s = 1.0f - n;
  GpnTaps t;
  t.nw = s * e;
  t.ne = s * w;
  t.sw = n * e;
  t.se = n * w;
  t.x0 = (int)xf;
  t.y0 = (int)yf;
  t.vx0 = t.x0 >= 0 && t.x0 < mw;
  t.vx1 = t.x0 + 1 >= 0 && t.x0 + 1 < mw;
  t.vy0 = t.y0 >= 0 && t.y0 < mh;
  t.vy1 = t.y0 + 1 >= 0 && t.y0 + 1 < mh;
  return t;
}

__device__ __forceinline__ float gpn_bilerp(float vnw, float vne, float vsw, float vse,
                                            const GpnTaps& t) {
#pragma clang fp contract(off)
  float o = vnw * t.nw;
  o = __builtin_fmaf(vne, t.ne, o);
  o = __builtin_fmaf(vsw, t.sw, o);
  o = __builtin_fmaf(vse, t.se, o);
  return o;
}

// Same accumulation with the four weights passed separately (a zero weight stands for an
// out-of-map tap: value * 0 contributes the same exact zero as grid_sample's zero padding).
__device__ __forceinline__ float gpn_bilerp4(float vnw, float vne, float vsw, float vse, float wnw,
                                             float wne, float wsw, float wse) {
#pragma clang fp contract(off)
  float o = vnw * wnw;
  o = __builtin_fmaf(vne, wne, o);
  o = __builtin_fmaf(vsw, wsw, o);
  o = __builtin_fmaf(vse, wse, o);
  return o;
}

// compute_angle (reference projection.py:57-82) for one (point, view) pair: unit(o_q - x) - unit(o_v - x),
// normalised (clamped at 1e-6), and the dot product of the two unit vectors.  Every operation rounded
// separately, in the reference's order.
__device__ __forceinline__ float4 gpn_ray_diff(float x, float y, float z, float qx, float qy, float qz,
                                               float sx, float sy, float sz) {
#pragma clang fp contract(off)
  float ax = qx - x, ay = qy - y, az = qz - z;
  const float na = sqrtf(ax * ax + ay * ay + az * az) + 1e-6f;
  ax /= na, ay /= na, az /= na;
  float bx = sx - x, by = sy - y, bz = sz - z;
  const float nb = sqrtf(bx * bx + by * by + bz * bz) + 1e-6f;
  bx /= nb, by /= nb, bz /= nb;
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  float nd = sqrtf(dx * dx + dy * dy + dz * dz);
  const float dot = ax * bx + ay * by + az * bz;
  nd = nd < 1e-6f ? 1e-6f : nd;
  return make_float4(dx / nd, dy / nd, dz / nd, dot);
}
