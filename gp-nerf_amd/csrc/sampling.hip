// Wavefront-level samplers and compositing (north_star subsystem 3).
//   gpn_sample_along_ray : reference gpnerf/render_ray.py:58-102
//   gpn_sample_fine      : reference gpnerf/render_ray.py:162-191 + :9-55
//   gpn_raw2outputs      : reference gpnerf/render_ray.py:110-159
//   gpn_rays_single_image: reference gpnerf/sample_ray.py:108-132
// Built with -ffp-contract=off: every mul/add below is rounded separately, in the order the
// reference's elementwise torch ops apply them.
#include "common.h"

// depth of coarse sample i of S (reference :73-88): start + i*step, optionally through 1/x.
__device__ __forceinline__ float coarse_depth(int i, int S, float near_d, float far_d, int inv) {
  const float denom = (float)(S - 1);
  if (inv) {
    const float start = 1.0f / near_d;
    const float step = (1.0f / far_d - start) / denom;
    return 1.0f / (start + (float)i * step);
  }
  const float step = (far_d - near_d) / denom;
  return near_d + (float)i * step;
}

__global__ void k_sample_along_ray(const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                   float near_d, float far_d, int R, int S, int inv,
                                   const float* __restrict__ t_rand, float* __restrict__ pts,
                                   float* __restrict__ z_vals) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)R * S) return;
  const int r = (int)(idx / S), i = (int)(idx % S);
  float z = coarse_depth(i, S, near_d, far_d, inv);
  if (t_rand) {  // stratified jitter (:90-97)
    const float lower = (i == 0) ? z : 0.5f * (z + coarse_depth(i - 1, S, near_d, far_d, inv));
    const float upper = (i == S - 1) ? z : 0.5f * (coarse_depth(i + 1, S, near_d, far_d, inv) + z);
    z = lower + (upper - lower) * t_rand[idx];
  }
  z_vals[idx] = z;
  for (int c = 0; c < 3; ++c) pts[idx * 3 + c] = z * ray_d[r * 3 + c] + ray_o[r * 3 + c];
}

// pdf -> cdf (reference :21-23); ww already holds weights + 1e-5.  M <= 254, sequential like
// torch.cumsum.  Ends with a barrier.
__device__ __forceinline__ void build_cdf(const float* ww, float* cdf, int M, int lane) {
  if (lane == 0) {
    // torch.sum is a cascade (near-exact) reduction: accumulate in fp64 and round once
    double acc = 0.0;
    for (int i = 0; i < M; ++i) acc += (double)ww[i];
    const float total = (float)acc;
    float c = 0.0f;
    cdf[0] = 0.0f;
    for (int i = 0; i < M; ++i) {
      c += ww[i] / total;
      cdf[i + 1] = c;
    }
  }
  __syncthreads();
}

// inverse-CDF lookup of one uniform (reference :33-53)
__device__ __forceinline__ float invert_cdf(const float* bins, const float* cdf, int M, float uu) {
  int above = 0;
  for (int i = 0; i < M; ++i) above += (uu >= cdf[i]) ? 1 : 0;
  const int below = above - 1 > 0 ? above - 1 : 0;
  const float cb = cdf[below], ca = cdf[above];
  const float bb = bins[below], ba = bins[above];
  float denom = ca - cb;
  if (denom < 1e-5f) denom = 1.0f;
  const float t = (uu - cb) / denom;
  return bb + t * (ba - bb);
}

// sample_pdf on explicit bins: one wave per ray.
__global__ void __launch_bounds__(64)
k_sample_pdf(const float* __restrict__ bins_g, const float* __restrict__ weights,
             const float* __restrict__ u, int u_row_stride, int M, int NI,
             float* __restrict__ samples) {
  extern __shared__ float sm[];
  float* bins = sm;            // [M+1]
  float* ww = bins + (M + 1);  // [M]
  float* cdf = ww + M;         // [M+1]
  const int r = blockIdx.x, lane = threadIdx.x;
  for (int i = lane; i < M + 1; i += 64) bins[i] = bins_g[(size_t)r * (M + 1) + i];
  for (int i = lane; i < M; i += 64) ww[i] = weights[(size_t)r * M + i] + 1e-5f;
  __syncthreads();
  build_cdf(ww, cdf, M, lane);
  for (int n = lane; n < NI; n += 64)
    samples[(size_t)r * NI + n] = invert_cdf(bins, cdf, M, u[(size_t)r * u_row_stride + n]);
}

// One wave per ray; LDS holds bins / weights / cdf / the merged depth list.
__global__ void __launch_bounds__(64)
k_sample_fine(const float* __restrict__ ray_o, const float* __restrict__ ray_d,
              const float* __restrict__ weights, const float* __restrict__ z_vals,
              const float* __restrict__ u, int u_row_stride, int R, int S, int NI, int inv,
              float* __restrict__ pts, float* __restrict__ z_out) {
  extern __shared__ float sm[];
  const int M = S - 2, T = S + NI;
  float* bins = sm;            // [M+1]
  float* ww = bins + (S - 1);  // [M]
  float* cdf = ww + M;         // [M+1]
  float* zall = cdf + (S - 1); // [T]
  float* zsort = zall + T;     // [T]
  const int r = blockIdx.x, lane = threadIdx.x;
  const float* z = z_vals + (size_t)r * S;
  const float* w = weights + (size_t)r * S;

  for (int i = lane; i < S - 1; i += 64) {
    if (inv) {  // bins = flip(0.5*(1/z[1:] + 1/z[:-1]))  (:164-168)
      const int k = S - 2 - i;
      bins[i] = 0.5f * (1.0f / z[k + 1] + 1.0f / z[k]);
    } else {
      bins[i] = 0.5f * (z[i + 1] + z[i]);
    }
  }
  for (int i = lane; i < M; i += 64) ww[i] = (inv ? w[M - i] : w[1 + i]) + 1e-5f;  // (:19, :166-169)
  for (int i = lane; i < S; i += 64) zall[i] = z[i];
  __syncthreads();
  build_cdf(ww, cdf, M, lane);
  for (int n = lane; n < NI; n += 64) {
    const float smp = invert_cdf(bins, cdf, M, u[(size_t)r * u_row_stride + n]);
    zall[S + n] = inv ? 1.0f / smp : smp;
  }
  __syncthreads();
  for (int i = lane; i < T; i += 64) {  // rank sort == torch.sort values (:185)
    const float v = zall[i];
    int rank = 0;
    for (int j = 0; j < T; ++j) {
      const float vj = zall[j];
      rank += (vj < v || (vj == v && j < i)) ? 1 : 0;
    }
    zsort[rank] = v;
  }
  __syncthreads();
  const float ox = ray_o[r * 3 + 0], oy = ray_o[r * 3 + 1], oz = ray_o[r * 3 + 2];
  const float dx = ray_d[r * 3 + 0], dy = ray_d[r * 3 + 1], dz = ray_d[r * 3 + 2];
  for (int i = lane; i < T; i += 64) {
    const float zz = zsort[i];
    const size_t o = (size_t)r * T + i;
    z_out[o] = zz;
    pts[o * 3 + 0] = zz * dx + ox;
    pts[o * 3 + 1] = zz * dy + oy;
    pts[o * 3 + 2] = zz * dz + oz;
  }
}

__global__ void k_raw2outputs(const float* __restrict__ raw, const float* __restrict__ z_vals,
                              const float* __restrict__ vis, int R, int S, int white,
                              float* __restrict__ rgb, float* __restrict__ depth,
                              float* __restrict__ weights, float* __restrict__ alpha_out,
                              float* __restrict__ ray_mask) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  float T = 1.0f, c0 = 0.f, c1 = 0.f, c2 = 0.f, d = 0.f, wsum = 0.f, nvis = 0.f;
  for (int s = 0; s < S; ++s) {
    const size_t o = (size_t)r * S + s;
    const float4 rw = *reinterpret_cast<const float4*>(raw + o * 4);
    const float alpha = 1.0f - expf(-rw.w);
    const float wgt = alpha * T;
    T = T * (1.0f - alpha + 1e-10f);
    c0 += wgt * rw.x;
    c1 += wgt * rw.y;
    c2 += wgt * rw.z;
    d += wgt * z_vals[o];
    wsum += wgt;
    nvis += vis[o];
    weights[o] = wgt;
    alpha_out[o] = alpha;
  }
  if (white) {
    const float bg = 1.0f - wsum;
    c0 += bg;
    c1 += bg;
    c2 += bg;
  }
  rgb[r * 3 + 0] = c0;
  rgb[r * 3 + 1] = c1;
  rgb[r * 3 + 2] = c2;
  depth[r] = d;
  ray_mask[r] = nvis > 8.0f ? 1.0f : 0.0f;
}

extern "C" int gpn_sample_along_ray(const float* ray_o, const float* ray_d, float near_depth,
                                    float far_depth, int R, int S, int inv_uniform,
                                    const float* t_rand, float* pts, float* z_vals,
                                    gpn_stream_t stream) {
  if (R == 0) return GPN_OK;  // empty batch (pointers may be null)
  GPN_REQUIRE(ray_o && ray_d && pts && z_vals, "gpn_sample_along_ray: null pointer");
  GPN_REQUIRE(R >= 0 && S >= 2, "gpn_sample_along_ray: need R>=0, S>=2 (got R=%d S=%d)", R, S);
  GPN_REQUIRE(near_depth > 0 && far_depth > near_depth,
              "gpn_sample_along_ray: need 0 < near < far (got %g, %g)", near_depth, far_depth);
  if (R == 0) return GPN_OK;
  const long long n = (long long)R * S;
  GpnStageTimer timer_k_sample_along_ray_(GPN_ST_SAMPLE, (hipStream_t)stream);
  hipLaunchKernelGGL(k_sample_along_ray, dim3(gpn_ceil_div(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, ray_o, ray_d, near_depth, far_depth, R, S, inv_uniform,
                     t_rand, pts, z_vals);
  GPN_LAUNCH_CHECK("k_sample_along_ray");
  return GPN_OK;
}

extern "C" int gpn_sample_fine(const float* ray_o, const float* ray_d, const float* weights,
                               const float* z_vals, const float* u, int u_row_stride, int R, int S,
                               int N_imp, int inv_uniform, float* pts, float* z_out,
                               gpn_stream_t stream) {
  if (R == 0) return GPN_OK;  // empty batch (pointers may be null)
  GPN_REQUIRE(ray_o && ray_d && weights && z_vals && u && pts && z_out,
              "gpn_sample_fine: null pointer");
  GPN_REQUIRE(R >= 0 && S >= 3 && N_imp >= 1 && S + N_imp <= 256,
              "gpn_sample_fine: need S>=3, N_imp>=1, S+N_imp<=256 (got S=%d N_imp=%d)", S, N_imp);
  GPN_REQUIRE(u_row_stride == 0 || u_row_stride >= N_imp, "gpn_sample_fine: bad u_row_stride");
  if (R == 0) return GPN_OK;
  const int T = S + N_imp;
  const size_t lds = sizeof(float) * ((S - 1) + (S - 2) + (S - 1) + 2 * T);
  GpnStageTimer timer_k_sample_fine_(GPN_ST_SAMPLE, (hipStream_t)stream);
  hipLaunchKernelGGL(k_sample_fine, dim3(R), dim3(64), lds, (hipStream_t)stream, ray_o, ray_d,
                     weights, z_vals, u, u_row_stride, R, S, N_imp, inv_uniform, pts, z_out);
  GPN_LAUNCH_CHECK("k_sample_fine");
  return GPN_OK;
}

extern "C" int gpn_sample_pdf(const float* bins, const float* weights, const float* u,
                              int u_row_stride, int R, int M, int N_samples, float* samples,
                              gpn_stream_t stream) {
  if (R == 0) return GPN_OK;  // empty batch (pointers may be null)
  GPN_REQUIRE(bins && weights && u && samples, "gpn_sample_pdf: null pointer");
  GPN_REQUIRE(R >= 0 && M >= 1 && M <= 4096 && N_samples >= 1, "gpn_sample_pdf: bad shape");
  GPN_REQUIRE(u_row_stride == 0 || u_row_stride >= N_samples, "gpn_sample_pdf: bad u_row_stride");
  if (R == 0) return GPN_OK;
  const size_t lds = sizeof(float) * (3 * M + 2);
  GpnStageTimer timer_k_sample_pdf_(GPN_ST_SAMPLE, (hipStream_t)stream);
  hipLaunchKernelGGL(k_sample_pdf, dim3(R), dim3(64), lds, (hipStream_t)stream, bins, weights, u,
                     u_row_stride, M, N_samples, samples);
  GPN_LAUNCH_CHECK("k_sample_pdf");
  return GPN_OK;
}

extern "C" int gpn_raw2outputs(const float* raw, const float* z_vals, const float* vis_mask, int R,
                               int S, int white_bkgd, float* rgb, float* depth, float* weights,
                               float* alpha, float* ray_mask, gpn_stream_t stream) {
  if (R == 0) return GPN_OK;  // empty batch (pointers may be null)
  GPN_REQUIRE(raw && z_vals && vis_mask && rgb && depth && weights && alpha && ray_mask,
              "gpn_raw2outputs: null pointer");
  GPN_REQUIRE(R >= 0 && S >= 1, "gpn_raw2outputs: bad shape");
  if (R == 0) return GPN_OK;
  GpnStageTimer timer_k_raw2outputs_(GPN_ST_SAMPLE, (hipStream_t)stream);
  hipLaunchKernelGGL(k_raw2outputs, dim3(gpn_ceil_div(R, 128)), dim3(128), 0, (hipStream_t)stream,
                     raw, z_vals, vis_mask, R, S, white_bkgd, rgb, depth, weights, alpha, ray_mask);
  GPN_LAUNCH_CHECK("k_raw2outputs");
  return GPN_OK;
}

// ------------------------------------------------------------------------------------ ray generation
// RaySamplerSingleImage.get_rays_single_image (reference sample_ray.py:108-132): pixel (u, v) of the
// strided grid -> d = M . [u, v, 1] with M = c2w[:3,:3] . K[:3,:3]^-1 (3x3, computed by the caller
// with the reference's own torch ops), o = c2w[:3,3].  torch's CPU bmm evaluates each output as the
// k-ascending chain  M[r][0]*u, fma(M[r][1], v, .), fma(M[r][2], 1, .)  (verified bit for bit on
// 76,800 rays), which is what the kernel does; optional `inds` picks pixels of the strided grid
// (random_sample, :172-214) instead of enumerating all of them.
struct GpnRayGenArgs {
  float m[9];
  float o[3];
};

__global__ void k_rays_single_image(GpnRayGenArgs a, int Ws, int stride, long long n,
                                    const long long* __restrict__ inds, float* __restrict__ rays_o,
                                    float* __restrict__ rays_d) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long p = inds ? inds[i] : i;
  const float u = (float)((int)(p % Ws) * stride), v = (float)((int)(p / Ws) * stride);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    float acc = a.m[3 * r] * u;
    acc = __fmaf_rn(a.m[3 * r + 1], v, acc);
    acc = __fmaf_rn(a.m[3 * r + 2], 1.0f, acc);
    rays_d[i * 3 + r] = acc;
    rays_o[i * 3 + r] = a.o[r];
  }
}

extern "C" int gpn_rays_single_image(const float* M, const float* origin, int H, int W, int stride,
                                     const long long* inds, long long n_inds, float* rays_o,
                                     float* rays_d, gpn_stream_t stream) {
  GPN_REQUIRE(M && origin, "gpn_rays_single_image: null matrix / origin (host pointers)");
  GPN_REQUIRE(H >= 1 && W >= 1 && stride >= 1, "gpn_rays_single_image: bad image size %dx%d stride %d",
              H, W, stride);
  const int Hs = (H + stride - 1) / stride, Ws = (W + stride - 1) / stride;
  const long long n = inds ? n_inds : (long long)Hs * Ws;
  GPN_REQUIRE(n >= 0, "gpn_rays_single_image: negative count");
  if (n == 0) return GPN_OK;
  GPN_REQUIRE(rays_o && rays_d, "gpn_rays_single_image: null output");
  GpnRayGenArgs a;
  for (int i = 0; i < 9; ++i) a.m[i] = M[i];
  for (int i = 0; i < 3; ++i) a.o[i] = origin[i];
  GpnStageTimer timer(GPN_ST_SAMPLE, (hipStream_t)stream);
  hipLaunchKernelGGL(k_rays_single_image, dim3(gpn_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     a, Ws, stride, n, inds, rays_o, rays_d);
  GPN_LAUNCH_CHECK("k_rays_single_image");
  return GPN_OK;
}

// ------------------------------------------------------------------------------------ semantic labels
// Label epilogue of the semantic branch (reference eval_gpnerf.py:201-206 + loss.py:293-300): the
// caller permutes the head's logits [1,C,H,W] to NHWC and takes argmax over the classes on the host
// path; here one pass over the NCHW logits writes the int64 label image directly (first maximal
// index, as torch.argmax; a NaN counts as maximal).  Pixel-major threads: every class plane is read
// with fully coalesced loads.
__global__ void k_argmax_labels(const float* __restrict__ logits, int C, long long HW,
                                long long* __restrict__ labels, unsigned char* __restrict__ labels_u8) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  float best = logits[p];
  int arg = 0;
  for (int c = 1; c < C; ++c) {
    const float v = logits[(long long)c * HW + p];
    if (v > best || (v != v && best == best)) best = v, arg = c;
  }
  if (labels) labels[p] = arg;
  if (labels_u8) labels_u8[p] = (unsigned char)arg;
}

extern "C" int gpn_argmax_labels(const float* logits, int C, long long HW, long long* labels,
                                 unsigned char* labels_u8, gpn_stream_t stream) {
  GPN_REQUIRE(C >= 1 && C <= 256, "gpn_argmax_labels: class count must be in [1,256] (got %d)", C);
  GPN_REQUIRE(HW >= 0, "gpn_argmax_labels: negative pixel count");
  if (HW == 0) return GPN_OK;
  GPN_REQUIRE(logits && (labels || labels_u8), "gpn_argmax_labels: null pointer");
  hipLaunchKernelGGL(k_argmax_labels, dim3(gpn_ceil_div(HW, 256)), dim3(256), 0, (hipStream_t)stream,
                     logits, C, HW, labels, labels_u8);
  GPN_LAUNCH_CHECK("k_argmax_labels");
  return GPN_OK;
}
