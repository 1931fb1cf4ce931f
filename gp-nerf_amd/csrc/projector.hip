// Fused projector (north_star subsystem 1): world->pixel projection of every ray sample into V
// source views, in-front / in-bounds mask, bilinear gather of RGB + 32-ch features (+ optionally the
// 512-ch deep-semantic map) from NHWC maps, and the ray-direction-difference feature.
// Replaces reference gpnerf/projection.py:31-139 (Projector.compute and helpers).
// HBM-bound: NHWC maps make every tap a contiguous 128 B (features) / 2 KB (deep semantics) read;
// outputs are written once, fully coalesced.  Built with -ffp-contract=off.
#include "common.h"
#include "proj_common.h"
#include <climits>

// ---------------------------------------------------------------- NCHW -> NHWC repack
__global__ void k_repack_nhwc(const float* __restrict__ src, float* __restrict__ dst, int C,
                              long long HW) {
  __shared__ float tile[32][33];
  const int v = blockIdx.z;
  const long long p0 = (long long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const float* s = src + (size_t)v * C * HW;
  float* d = dst + (size_t)v * C * HW;
  const int tx = threadIdx.x, ty = threadIdx.y;  // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k;
    const long long p = p0 + tx;
    tile[k][tx] = (c < C && p < HW) ? s[(size_t)c * HW + p] : 0.0f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const long long p = p0 + k;
    const int c = c0 + tx;
    if (c < C && p < HW) d[(size_t)p * C + c] = tile[tx][k];
  }
}

// ---------------------------------------------------------------- RGB + feature gather, mask, ray_diff
// A workgroup of PAIRS threads works on PAIRS (point, view) pairs at a time, in two phases:
//   1. one LANE per pair does the per-pair arithmetic once -- projection, in-front / in-bounds mask,
//      the tap positions and bilinear weights on both maps (8 correctly rounded divisions), ray_diff
//      (9 divisions, 3 square roots) -- writes mask / ray_diff / the optional geometry outputs
//      (consecutive lanes, consecutive addresses) and leaves the taps in LDS;
//   2. 8 lanes per pair gather: lane q owns feature channels 4q..4q+3 (one float4 per tap), lanes 0-2
//      one RGB channel each; rows go through LDS and leave as one contiguous, coalesced span.
// (With 8 lanes per pair doing phase 1 as well -- the first form of this kernel -- every wave ran the
// whole per-pair arithmetic for 8 pairs only: it was instruction-bound at 2.7 TB/s.)
#ifndef GPN_PROJ_PAIRS
#define GPN_PROJ_PAIRS 256  // measured per step: 256 -> 0.36 ms, 128 -> 0.42 ms, 64 -> 0.43 ms
#endif
constexpr int PAIRS = GPN_PROJ_PAIRS;  // pairs per pass = threads per workgroup

__global__ void __launch_bounds__(PAIRS)
k_project_feat(gpn_project_args_t a, long long n_pairs) {
  __shared__ float sP[GPN_MAX_VIEWS * 12];
  __shared__ float sPos[GPN_MAX_VIEWS * 3];
  __shared__ __attribute__((aligned(16))) float sOut[PAIRS * 36];
  // taps of the pairs: [0] feature map, [1] RGB image; info = valid bits (vx0 vx1 vy0 vy1) | view << 4
  __shared__ int sX0[2][PAIRS], sY0[2][PAIRS], sInfo[2][PAIRS];
  __shared__ float sW[2][4][PAIRS];
  const int tid = threadIdx.x;
  for (int i = tid; i < a.V * 12; i += PAIRS) sP[i] = a.proj[(i / 12) * 16 + (i % 12)];
  for (int i = tid; i < a.V * 3; i += PAIRS) sPos[i] = a.src_pos[i];
  __syncthreads();
  // 35 channels; rows of the output are LD floats apart (36 on the fused path: 16-byte aligned rows,
  // channels OC.. are written as zeros)
  const int OC = 3 + a.feat_c, LD = a.rgb_feat_ld > 0 ? a.rgb_feat_ld : OC;
  const bool gather = a.rgb_feat != nullptr;  // false: geometry only (pix / mask / ray_diff)
  const bool small = n_pairs < 0x7fffffffLL;  // 32-bit index arithmetic (a 64-bit division is ~100 instructions)

  for (long long base = (long long)blockIdx.x * PAIRS; base < n_pairs;
       base += (long long)gridDim.x * PAIRS) {
    {  // ---- phase 1: pair base + tid
      const long long pair = base + tid;
      if (pair < n_pairs) {
        long long n;
        int v;
        if (small) {
          const unsigned pn = (unsigned)pair / (unsigned)a.V;
          n = pn, v = (int)((unsigned)pair - pn * (unsigned)a.V);
        } else {
          n = pair / a.V, v = (int)(pair - n * a.V);
        }
        const float x = a.xyz[n * 3 + 0], y = a.xyz[n * 3 + 1], z = a.xyz[n * 3 + 2];
        const GpnPix pp = gpn_project(sP + v * 12, x, y, z);
        const GpnTaps tr = gpn_taps(pp.px, pp.py, a.img_h, a.img_w, a.rgb_h, a.rgb_w);
        if (gather) {
          const GpnTaps tf = gpn_taps(pp.px, pp.py, a.img_h, a.img_w, a.feat_h, a.feat_w);
          sX0[0][tid] = tf.x0, sY0[0][tid] = tf.y0;
          sInfo[0][tid] = (tf.vx0 ? 1 : 0) | (tf.vx1 ? 2 : 0) | (tf.vy0 ? 4 : 0) | (tf.vy1 ? 8 : 0) | (v << 4);
          sW[0][0][tid] = tf.nw, sW[0][1][tid] = tf.ne, sW[0][2][tid] = tf.sw, sW[0][3][tid] = tf.se;
          sX0[1][tid] = tr.x0, sY0[1][tid] = tr.y0;
          sInfo[1][tid] = (tr.vx0 ? 1 : 0) | (tr.vx1 ? 2 : 0) | (tr.vy0 ? 4 : 0) | (tr.vy1 ? 8 : 0) | (v << 4);
          sW[1][0][tid] = tr.nw, sW[1][1][tid] = tr.ne, sW[1][2][tid] = tr.sw, sW[1][3][tid] = tr.se;
        }
        if (a.taps) {
          a.taps[((size_t)v * a.N + n) * 2 + 0] = tr.x0;
          a.taps[((size_t)v * a.N + n) * 2 + 1] = tr.y0;
        }
        // mask (projection.py:133-138)
        const bool m = pp.front && gpn_inbound(pp.px, pp.py, a.img_h, a.img_w);
        if (a.mask) a.mask[pair] = m ? 1.0f : 0.0f;
        if (a.front) a.front[(size_t)v * a.N + n] = pp.front ? 1 : 0;
        if (a.pix) {
          a.pix[((size_t)v * a.N + n) * 2 + 0] = pp.px;
          a.pix[((size_t)v * a.N + n) * 2 + 1] = pp.py;
        }
        if (a.ray_diff)  // ray_diff (projection.py:65-82)
          *reinterpret_cast<float4*>(a.ray_diff + pair * 4) =
              gpn_ray_diff(x, y, z, a.qry_pos[0], a.qry_pos[1], a.qry_pos[2], sPos[v * 3 + 0], sPos[v * 3 + 1],
                           sPos[v * 3 + 2]);
      }
    }
    if (!gather) continue;  // uniform across the block
    __syncthreads();
    // ---- phase 2: 8 lanes per pair
    const int q = tid & 7;
    for (int sub = 0; sub < 8; ++sub) {  // PAIRS / 8 pairs per pass
      const int slot = sub * (PAIRS / 8) + (tid >> 3);
      if (base + slot < n_pairs) {
        GpnTaps t;
        {
          const int info = sInfo[0][slot];
          const int v = info >> 4;
          t.x0 = sX0[0][slot], t.y0 = sY0[0][slot];
          t.vx0 = info & 1, t.vx1 = info & 2, t.vy0 = info & 4, t.vy1 = info & 8;
          t.nw = sW[0][0][slot], t.ne = sW[0][1][slot], t.sw = sW[0][2][slot], t.se = sW[0][3][slot];
          // 32-channel feature map, NHWC: one float4 per lane and tap
          const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
          const float* fm = a.feat + (size_t)v * a.feat_h * a.feat_w * a.feat_c + q * 4;
          const size_t rs = (size_t)a.feat_w * a.feat_c;
          const size_t o00 = (size_t)t.y0 * rs + (size_t)t.x0 * a.feat_c;
          float4 vnw = zero4, vne = zero4, vsw = zero4, vse = zero4;
          if (t.vy0 && t.vx0) vnw = *reinterpret_cast<const float4*>(fm + o00);
          if (t.vy0 && t.vx1) vne = *reinterpret_cast<const float4*>(fm + o00 + a.feat_c);
          if (t.vy1 && t.vx0) vsw = *reinterpret_cast<const float4*>(fm + o00 + rs);
          if (t.vy1 && t.vx1) vse = *reinterpret_cast<const float4*>(fm + o00 + rs + a.feat_c);
          float* o = sOut + slot * LD + 3 + q * 4;
          o[0] = gpn_bilerp(vnw.x, vne.x, vsw.x, vse.x, t);
          o[1] = gpn_bilerp(vnw.y, vne.y, vsw.y, vse.y, t);
          o[2] = gpn_bilerp(vnw.z, vne.z, vsw.z, vse.z, t);
          o[3] = gpn_bilerp(vnw.w, vne.w, vsw.w, vse.w, t);
        }
        if (q < 3) {  // RGB image [V,H,W,3], channel q
          const int info = sInfo[1][slot];
          const int v = info >> 4;
          t.x0 = sX0[1][slot], t.y0 = sY0[1][slot];
          t.vx0 = info & 1, t.vx1 = info & 2, t.vy0 = info & 4, t.vy1 = info & 8;
          t.nw = sW[1][0][slot], t.ne = sW[1][1][slot], t.sw = sW[1][2][slot], t.se = sW[1][3][slot];
          const float* im = a.rgb + (size_t)v * a.rgb_h * a.rgb_w * 3 + q;
          const size_t rs = (size_t)a.rgb_w * 3;
          const size_t o00 = (size_t)t.y0 * rs + (size_t)t.x0 * 3;
          float vnw = 0.f, vne = 0.f, vsw = 0.f, vse = 0.f;
          if (t.vy0 && t.vx0) vnw = im[o00];
          if (t.vy0 && t.vx1) vne = im[o00 + 3];
          if (t.vy1 && t.vx0) vsw = im[o00 + rs];
          if (t.vy1 && t.vx1) vse = im[o00 + rs + 3];
          sOut[slot * LD + q] = gpn_bilerp(vnw, vne, vsw, vse, t);
        } else if (q == 3) {
          for (int c = OC; c < LD; ++c) sOut[slot * LD + c] = 0.0f;
        }
      }
    }
    __syncthreads();
    // coalesced copy-out of this block's contiguous [pairs, LD] span
    const long long left = n_pairs - base;
    const int cnt = (int)(left < PAIRS ? left : PAIRS) * LD;
    float* dst = a.rgb_feat + base * LD;
    if (((base * LD) & 3) == 0) {
      for (int i = tid * 4; i + 3 < cnt; i += PAIRS * 4)
        *reinterpret_cast<float4*>(dst + i) = *reinterpret_cast<const float4*>(sOut + i);
      for (int i = (cnt & ~3) + tid; i < cnt; i += PAIRS) dst[i] = sOut[i];
    } else {
      for (int i = tid; i < cnt; i += PAIRS) dst[i] = sOut[i];
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- 512-channel deep-semantic gather
// One wave per (point, view) pair; each lane owns 2 x float4 of the 512 channels: every tap is two
// fully coalesced 1 KiB wave loads, the output row two 1 KiB wave stores (non-temporal: 5.4 GB at the
// north-star shape, written once -- they must not evict the texels the next pairs re-read).
// Order of the pairs.  The four taps of a pair are 8 KB of map; neighbouring RAYS at the same sample
// index and view project half a texel apart and share most of them, and the samples of the rays of one image
// row run along (nearly) one epipolar line of a source view: everything a block of a few hundred consecutive rays
// reads from ONE view over ALL its samples is two or three texel rows (~2 MB), while the chunk as a whole
// touches each texel about ten times, spread over far-apart (sample, view) slabs.  Every XCD has its own 4 MB
// L2 and workgroups are dealt to the XCDs round-robin (blockIdx % 8), so with the samples-per-ray hint the
// work is cut into UNITS = (block of PD_RB rays, view); XCD x walks a contiguous run of units (ray blocks of
// one view back to back: consecutive image rows share texel rows), all of its workgroups together inside
// one unit, sample-major with the rays innermost (the waves of a workgroup take consecutive rays of one
// sample at the same time: shared taps meet in the L1).  A unit's texels are then fetched from HBM once.
// Measured HBM fetch per call at the north-star shape (0.16 GB of distinct texels, 1.57 GB of maps): pair-major
// order 6.9 GB (r02a), slabs of one (sample, view) across the chip 3.1 GB (r02), units per XCD: profiles/r03_*.
// Without the hint (S = 1), or with fewer units than XCDs, the old slab order is used.
#define PD_RUN 16
#ifndef PD_RB
#define PD_RB 256  // rays per unit (with PD_URUN 8: 2 items per wave and unit at 32 waves per CU)
#endif
#ifndef PD_STORE
#define PD_STORE 0  // 0: nontemporal; 1: sc1, 2: sc0 sc1 (both measured 1.93 vs 1.40 ms: write-through in 16-byte pieces)
#endif
#ifndef PD_VIEW_MAJOR
#define PD_VIEW_MAJOR 0
#endif
#ifndef PD_URUN
#define PD_URUN 8  // consecutive rays a wave takes per item inside a unit (sliding texel window in registers)
#endif
__global__ void __launch_bounds__(256)
k_project_deep(gpn_project_args_t a, long long n_pairs, int by_units) {
  __shared__ float sP[GPN_MAX_VIEWS * 12];
  const int tid = threadIdx.x;
  for (int i = tid; i < a.V * 12; i += 256) sP[i] = a.proj[(i / 12) * 16 + (i % 12)];
  __syncthreads();
  const int lane = tid & 63;
  const int D = a.deep_c;  // 512
  const size_t rs = (size_t)a.deep_w * D;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int S = a.samples_per_ray > 0 && a.N % a.samples_per_ray == 0 ? a.samples_per_ray : 1;
  const long long R = a.N / S;                      // rays (S == 1: every point is its own "ray")
  // one (point, view) pair: project, four taps x 2 KB, blend, one 2 KB row out
  auto do_pair = [&](long long n, int v) {
    const float* dm = a.deep + (size_t)v * a.deep_h * rs;
    const long long pair = n * a.V + v;
    const GpnPix pp = gpn_project(sP + v * 12, a.xyz[n * 3 + 0], a.xyz[n * 3 + 1], a.xyz[n * 3 + 2]);
    const GpnTaps t = gpn_taps(pp.px, pp.py, a.img_h, a.img_w, a.deep_h, a.deep_w);
    const size_t o00 = (size_t)t.y0 * rs + (size_t)t.x0 * D;
    float* out = a.deep_out + pair * D;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int c = half * 256 + lane * 4;
      float4 vnw = zero4, vne = zero4, vsw = zero4, vse = zero4;
      if (t.vy0 && t.vx0) vnw = *reinterpret_cast<const float4*>(dm + o00 + c);
      if (t.vy0 && t.vx1) vne = *reinterpret_cast<const float4*>(dm + o00 + D + c);
      if (t.vy1 && t.vx0) vsw = *reinterpret_cast<const float4*>(dm + o00 + rs + c);
      if (t.vy1 && t.vx1) vse = *reinterpret_cast<const float4*>(dm + o00 + rs + D + c);
      typedef float vf4 __attribute__((ext_vector_type(4)));
      vf4 o;
      o.x = gpn_bilerp(vnw.x, vne.x, vsw.x, vse.x, t);
      o.y = gpn_bilerp(vnw.y, vne.y, vsw.y, vse.y, t);
      o.z = gpn_bilerp(vnw.z, vne.z, vsw.z, vse.z, t);
      o.w = gpn_bilerp(vnw.w, vne.w, vsw.w, vse.w, t);
      // the 2 KB output row is written once and never read here: an `sc1` store leaves the XCD's L2 without
      // keeping its line there (plain and `nt` stores keep it, MI355X_MICROARCH.md: the 5.4 GB output stream
      // then evicts the texels the next pairs re-read)
#if PD_STORE == 1
      // (s_nop: a VALU write of the data registers right behind a > 64-bit store is a hazard the compiler
      // pads for its own stores only -- without it 0.6 % of the values came out wrong)
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(out + c), "v"(o) : "memory");
#elif PD_STORE == 2
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(out + c), "v"(o) : "memory");
#else
      __builtin_nontemporal_store(o, reinterpret_cast<vf4*>(out + c));
#endif
    }
  };
  if (by_units) {  // (gridDim.x is a multiple of 8 here)
    const int xcd = blockIdx.x & 7;
    const long long wx = (long long)(blockIdx.x >> 3) * 4 + (tid >> 6), nwx = (long long)(gridDim.x >> 3) * 4;
    const long long nrb = (R + PD_RB - 1) / PD_RB, U = nrb * a.V;
    typedef float vf4 __attribute__((ext_vector_type(4)));
    for (long long u = xcd * U / 8; u < (xcd + 1) * U / 8; ++u) {
#if PD_VIEW_MAJOR
      const int v = (int)(u / nrb);
      const long long r0 = (u - (long long)v * nrb) * PD_RB;
#else
      // ray-block major: an XCD walks the views of ONE ray block (the views differ in how many of their pairs
      // are inside the map, i.e. in cost: view-major runs left the XCDs unbalanced, 1.90 vs 1.54 ms)
      const long long rbk = u / a.V;
      const int v = (int)(u - rbk * a.V);
      const long long r0 = rbk * PD_RB;
#endif
      const long long nr = R - r0 < PD_RB ? R - r0 : PD_RB;
      const long long runs = (nr + PD_URUN - 1) / PD_URUN;
      const float* dm = a.deep + (size_t)v * a.deep_h * rs;
      for (long long it = wx; it < runs * S; it += nwx) {
        const long long s = it / runs, run = it - s * runs;
        const long long re = (run + 1) * PD_URUN < nr ? (run + 1) * PD_URUN : nr;
        // A wave walks PD_URUN consecutive rays of one (sample, view) and keeps the 2 x 2 texels of the last
        // pair in registers (two 16-byte pieces per lane and texel; a texel outside the map is kept as zeros,
        // grid_sample's padding).  Neighbouring pixels project half a texel apart: the window of the next ray
        // is the same one (nothing to load) or one column further (two texels to load instead of four) --
        // same values in the same blend, bit for bit, with a third of the tap requests.
        int cx = INT_MIN, cy = INT_MIN;  // north-west corner of the cached window (wave-uniform)
        vf4 c00[2], c01[2], c10[2], c11[2];
        auto texel = [&](int x, int y, vf4 (&d)[2]) {
          if (x >= 0 && x < a.deep_w && y >= 0 && y < a.deep_h) {
            const float* p = dm + (size_t)y * rs + (size_t)x * D + lane * 4;
            d[0] = *reinterpret_cast<const vf4*>(p);
            d[1] = *reinterpret_cast<const vf4*>(p + 256);
          } else {
            d[0] = d[1] = (vf4){0.f, 0.f, 0.f, 0.f};
          }
        };
        for (long long r = run * PD_URUN; r < re; ++r) {
          const long long n = (r0 + r) * S + s;
          const GpnPix pp = gpn_project(sP + v * 12, a.xyz[n * 3 + 0], a.xyz[n * 3 + 1], a.xyz[n * 3 + 2]);
          const GpnTaps t = gpn_taps(pp.px, pp.py, a.img_h, a.img_w, a.deep_h, a.deep_w);
          // (the point is wave-uniform, so are these branches)
          const int x0 = __builtin_amdgcn_readfirstlane(t.x0), y0 = __builtin_amdgcn_readfirstlane(t.y0);
          if (y0 == cy && x0 == cx) {
          } else if (y0 == cy && x0 == cx + 1 && cx != INT_MIN) {
            c00[0] = c01[0], c00[1] = c01[1], c10[0] = c11[0], c10[1] = c11[1];
            texel(x0 + 1, y0, c01);
            texel(x0 + 1, y0 + 1, c11);
          } else {
            texel(x0, y0, c00);
            texel(x0 + 1, y0, c01);
            texel(x0, y0 + 1, c10);
            texel(x0 + 1, y0 + 1, c11);
          }
          cx = x0, cy = y0;
          float* out = a.deep_out + (n * a.V + v) * D + lane * 4;
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            vf4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = gpn_bilerp(c00[half][e], c01[half][e], c10[half][e], c11[half][e], t);
#if PD_STORE == 3
            *reinterpret_cast<vf4*>(out + half * 256) = o;
#else
            __builtin_nontemporal_store(o, reinterpret_cast<vf4*>(out + half * 256));
#endif
          }
        }
      }
    }
    return;
  }
  const long long wave = (long long)blockIdx.x * 4 + (tid >> 6);
  const long long nwaves = (long long)gridDim.x * 4;
  const long long runs = (R + PD_RUN - 1) / PD_RUN;  // runs of rays per (sample, view)
  const long long items = runs * S * a.V;
  for (long long it = wave; it < items; it += nwaves) {
    const long long sv = it / runs, run = it - sv * runs;
    const int s = (int)(sv / a.V), v = (int)(sv - (long long)s * a.V);
    const long long r_end = (run + 1) * PD_RUN < R ? (run + 1) * PD_RUN : R;
    for (long long r = run * PD_RUN; r < r_end; ++r) do_pair(r * S + s, v);
  }
}

extern "C" int gpn_repack_nhwc(const float* src, float* dst, int V, int C, int H, int W,
                               gpn_stream_t stream) {
  GPN_REQUIRE(src && dst, "gpn_repack_nhwc: null pointer");
  GPN_REQUIRE(V >= 1 && C >= 1 && H >= 1 && W >= 1 && V <= 65535, "gpn_repack_nhwc: bad shape");
  const long long HW = (long long)H * W;
  dim3 grid(gpn_ceil_div(HW, 32), gpn_ceil_div(C, 32), V);
  GPN_REQUIRE(grid.y <= 65535, "gpn_repack_nhwc: too many channels");
  hipLaunchKernelGGL(k_repack_nhwc, grid, dim3(32, 8), 0, (hipStream_t)stream, src, dst, C, HW);
  GPN_LAUNCH_CHECK("k_repack_nhwc");
  return GPN_OK;
}

extern "C" int gpn_project_gather(const gpn_project_args_t* a, gpn_stream_t stream) {
  GPN_REQUIRE(a, "gpn_project_gather: null args");
  if (a->N == 0) return GPN_OK;  // empty batch: nothing to validate or launch (pointers may be null)
  GPN_REQUIRE(a->xyz && a->proj && a->src_pos, "gpn_project_gather: null pointer");
  GPN_REQUIRE(a->N >= 0 && a->V >= 1 && a->V <= GPN_MAX_VIEWS,
              "gpn_project_gather: need 1 <= V <= %d (got %d)", GPN_MAX_VIEWS, a->V);
  GPN_REQUIRE(a->rgb_h >= 1 && a->rgb_w >= 1, "gpn_project_gather: bad rgb map size");
  if (a->rgb_feat) {
    GPN_REQUIRE(a->rgb && a->feat, "gpn_project_gather: rgb_feat requested without maps");
    GPN_REQUIRE(a->feat_c == 32, "gpn_project_gather: feature maps must have 32 channels (got %d)",
                a->feat_c);
    GPN_REQUIRE(a->feat_h >= 1 && a->feat_w >= 1, "gpn_project_gather: bad feature map size");
    GPN_REQUIRE(a->rgb_feat_ld == 0 || (a->rgb_feat_ld >= 3 + a->feat_c && a->rgb_feat_ld <= 36),
                "gpn_project_gather: rgb_feat_ld must be 0 or in [%d,36] (got %d)", 3 + a->feat_c,
                a->rgb_feat_ld);
  }
  GPN_REQUIRE(a->img_h > 1.0f && a->img_w > 1.0f, "gpn_project_gather: bad image size");
  const bool want_deep = a->deep_out != nullptr;
  if (want_deep) {
    GPN_REQUIRE(a->deep, "gpn_project_gather: deep_out requested without a deep map");
    GPN_REQUIRE(a->deep_c == GPN_SEM && a->deep_h >= 1 && a->deep_w >= 1,
                "gpn_project_gather: deep-semantic maps must have %d channels (got %d)", GPN_SEM,
                a->deep_c);
  }
  const long long n_pairs = (long long)a->N * a->V;
  const int cus = gpn_num_cus();
  {
    const long long blocks = (n_pairs + PAIRS - 1) / PAIRS;
    const int grid = (int)(blocks < (long long)cus * 16 ? blocks : (long long)cus * 16);
    GpnStageTimer timer_k_project_feat_(GPN_ST_PROJECT, (hipStream_t)stream);
    hipLaunchKernelGGL(k_project_feat, dim3(grid), dim3(PAIRS), 0, (hipStream_t)stream, *a, n_pairs);
    GPN_LAUNCH_CHECK("k_project_feat");
  }
  if (want_deep) {
    const long long blocks = (n_pairs / PD_RUN + 4) / 4;
    int grid = (int)(blocks < (long long)cus * 8 ? blocks : (long long)cus * 8);
    // (ray block, view) units per XCD: needs the samples-per-ray hint, at least two units per XCD and enough
    // pairs to fill the chip (the grid is then a multiple of the 8 XCDs)
    const int S_ = a->samples_per_ray > 0 && a->N % a->samples_per_ray == 0 ? a->samples_per_ray : 1;
    const long long units = (((long long)a->N / S_ + PD_RB - 1) / PD_RB) * a->V;
    const int by_units = S_ > 1 && units >= 16 && blocks >= (long long)cus * 8;
    if (by_units) grid = (cus * 8) & ~7;
    GpnStageTimer timer_k_project_deep_(GPN_ST_PROJECT_DEEP, (hipStream_t)stream);
    hipLaunchKernelGGL(k_project_deep, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a, n_pairs, by_units);
    GPN_LAUNCH_CHECK("k_project_deep");
  }
  return GPN_OK;
}
