// GNT forward (north_star subsystem 2) on fp32 MFMA: reference gpnerf/transformer_network.py.
// One kernel per stage; every stage keeps its activations in registers in the transposed "T-tile"
// layout of gnt_mfma.h (16 points per wave-tile), weights staged once per workgroup into LDS,
// workgroups persistent over tiles.  Stages:
//   k_posenc      Embedder x2                                   (:6-37, :297-306)
//   k_rgbfeat     rgbfeat_fc + max over views                   (:309-311)
//   k_view_attn   attn_norm + Attention2D + residual            (:74-88, :107-112)
//   k_ff          x + FeedForward(LayerNorm(x))                 (:40-51, :114-117, :210-213)
//   k_qpos        q_fcs[i] on cat(q, posenc, viewenc)           (:322-324)
//   k_ray_attn    attn_norm + Attention('qk') + residual        (:149-178, :203-208)
//   k_sem         deep-semantic aggregation, fused over layers  (:90, :180-184, :335-336)
//   k_heads_rgb / k_sem_adaptor   output heads                  (:332-338)
#include "common.h"
#include "gnt_mfma.h"
#include "proj_common.h"
#include <mutex>
#include <type_traits>

#define C64 64
// threads per workgroup of the persistent per-point kernels (one 16-point tile per wave)
#ifndef RF_TPB
#define RF_TPB 512
#endif
#ifndef VA_TPB
#define VA_TPB 512
#endif
#ifndef FF_TPB
#define FF_TPB 512
#endif
#ifndef QP_TPB
#define QP_TPB 512
#endif
#ifndef FF_NTL
#define FF_NTL 2  // point tiles per wave of the split-mode feed-forward kernel
#endif
#define LDW 68  // padded leading dimension of a 64-wide weight matrix in LDS
#define LOG2E_F 1.44269504088896341f
#define LN2_F 0.693147180559945309f
#define VM_HOMOG 0
#define VM_MASKED 1
#define VM_ONLINE 2
#define VA_SLOT_BYTES (4096 + 256)  // one (tile, view) pair: 16 feature rows + 64 ray_diff floats
#define VA_DMA_OPS 5
// cache policy of the feature-row DMA: read once per launch -> non-temporal (measured -5 %)
#ifndef RAY_FF_MAXNT
#define RAY_FF_MAXNT 4  // largest NT (16-sample tiles per ray) whose ray kernel takes the feed-forward phase
#endif
#ifndef GPN_FUSE_VIEW_QP
#define GPN_FUSE_VIEW_QP 1  // ... and k_qpos of the even layers as its third phase
#endif
#ifndef GPN_FUSE_RAY_FF
#define GPN_FUSE_RAY_FF 1  // gnt_forward_impl: k_ff of the ray stage as a second phase of k_ray1 / k_ray_attn
#endif
#ifndef GPN_FUSE_VIEW_FF
#define GPN_FUSE_VIEW_FF 1  // gnt_forward_impl: k_ff of the view stage as a second phase of k_view_attn
#endif
#ifndef VA_TILED_DMA
#define VA_TILED_DMA 1
#endif
#ifndef VA_DMA_NT
#define VA_DMA_NT "nt"
#endif

extern __shared__ __attribute__((aligned(16))) float smem[];

// Development aid (tools/kbench.py, -DGPN_STAMP builds only): wall-clock stamps (10 ns units) taken by
// thread 0 of workgroup 0 at marked places of a kernel, read back with gpn_debug_stamps.
#ifdef GPN_STAMP
__device__ unsigned long long g_stamps[64];
#define STAMP(i) do { if (threadIdx.x == 0) { \
    if (blockIdx.x == 0) { g_stamps[i] = wall_clock64(); g_stamps[32 + (i)] = __builtin_readcyclecounter(); } \
    if (blockIdx.x == gridDim.x - 1) g_stamps[16 + (i)] = wall_clock64(); \
    if (blockIdx.x == gridDim.x / 2 + 3) g_stamps[48 + (i)] = wall_clock64(); } } while (0)
// per-wave start / end stamps of one kernel (no atomics: one slot per wave), up to 4096 waves
__device__ unsigned long long g_wave_t[2][4096];
#define WAVE_T(which) do { if ((threadIdx.x & 63) == 0) { const unsigned w_ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); \
    if (w_ < 4096) g_wave_t[which][w_] = wall_clock64(); } } while (0)
extern "C" int gpn_debug_wave_times(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_t), sizeof(g_wave_t)) == hipSuccess ? 0 : 1;
}
// earliest / latest stamp over every wave of the grid (slots 60 / 61)
#define STAMP_MIN() do { if ((threadIdx.x & 63) == 0) atomicMin(&g_stamps[60], (unsigned long long)wall_clock64()); } while (0)
#define STAMP_MAX() do { if ((threadIdx.x & 63) == 0) { const unsigned long long t_ = wall_clock64(); \
    atomicMax(&g_stamps[61], t_); atomicMax(&g_stamps[52 + (blockIdx.x & 7)], t_); } } while (0)
extern "C" int gpn_debug_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(g_stamps)) == hipSuccess ? 0 : 1;
}
extern "C" int gpn_debug_stamps_reset(void) {
  unsigned long long z[64] = {0};
  z[60] = ~0ull;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) == hipSuccess ? 0 : 1;
}
// shader-cycle stamps inside one wave's second tile iteration of k_ff32 (slot k = region boundary k)
__device__ unsigned long long g_ffc[4][32];
#define FFC(k) do { if (ffc_on) { const unsigned long long t_ = __builtin_readcyclecounter(); \
    if (lane == 0) g_ffc[wave & 3][k] = t_; } __builtin_amdgcn_sched_barrier(0); } while (0)
extern "C" int gpn_debug_ffc(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ffc), sizeof(g_ffc)) == hipSuccess ? 0 : 1;
}
// k_ray_attn: cycles a wave of workgroup 3 spends in each phase, summed over its iterations
// [wave][0 = projections, 1 = barrier A, 2 = attention + out_fc + stores, 3 = barrier B, 4 = abar reduction, 5 = iterations]
__device__ unsigned long long g_ra_phase[16][8];
#define RA_T0() unsigned long long ra_t_ = __builtin_readcyclecounter(); unsigned long long ra_acc_[6] = {0, 0, 0, 0, 0, 0}
#define RA_MARK(k) do { const unsigned long long n_ = __builtin_readcyclecounter(); ra_acc_[k] += n_ - ra_t_; ra_t_ = n_; } while (0)
#define RA_END() do { if (blockIdx.x == 3 && lane == 0 && wave < 16) { for (int k_ = 0; k_ < 6; ++k_) g_ra_phase[wave][k_] = ra_acc_[k_]; } } while (0)
extern "C" int gpn_debug_ra_phase(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ra_phase), sizeof(g_ra_phase)) == hipSuccess ? 0 : 1;
}
#define VA_T0() unsigned va_t_ = (unsigned)__builtin_readcyclecounter(); unsigned va_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define VA_MARK(k) do { const unsigned n_ = (unsigned)__builtin_readcyclecounter(); va_acc_[k] += n_ - va_t_; va_t_ = n_; } while (0)
#define VA_COUNT(k) do { va_acc_[k] += 1; } while (0)
#define VA_END() do { if (blockIdx.x == 3 && lane == 0 && wave < 16) { for (int k_ = 0; k_ < 8; ++k_) g_ra_phase[wave][k_] = va_acc_[k_]; } } while (0)
#else
#define VA_T0() do { } while (0)
#define VA_MARK(k) do { } while (0)
#define VA_COUNT(k) do { } while (0)
#define VA_END() do { } while (0)
#define RA_T0() do { } while (0)
#define RA_MARK(k) do { } while (0)
#define RA_END() do { } while (0)
#define FFC(k) do { } while (0)
#define STAMP(i) do { } while (0)
#define STAMP_MIN() do { } while (0)
#define STAMP_MAX() do { } while (0)
#define WAVE_T(which) do { } while (0)
#endif



static int set_lds_limit(const void* fn, size_t bytes) {
  if (bytes <= 64 * 1024) return GPN_OK;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess)
    return gpn_set_error(GPN_E_LAUNCH, "hipFuncSetAttribute(%zu B LDS): %s", bytes,
                         hipGetErrorString(e));
  return GPN_OK;
}

static int persistent_grid(long long work_items, int blocks_per_cu = 1) {
  const long long cus = (long long)gpn_num_cus() * blocks_per_cu;
  return (int)(work_items < cus ? (work_items < 1 ? 1 : work_items) : cus);
}
// workgroups per CU of k_rgbfeat (124 VGPRs, 28 KB of LDS: two fit): four waves per SIMD in flight
// take 9 % off that memory-bound kernel; the same setting made k_qpos 5 % slower (measured, A/B)
#ifndef GPN_MEM_BPC
#define GPN_MEM_BPC 2
#endif

// sin and cos of one fp32 argument.  |x| <= 1e5 (the encodings reach 2^9 |coordinate|: ~5e3 in a room):
// four-term Cody-Waite reduction by pi/2 on fused multiply-adds (the terms sum to pi/2 to 2^-70; the
// first has 8 significant bits, so n * term is exact for n < 2^16) and the fdlibm single-precision
// kernels on [-pi/4, pi/4]: <= 9e-8 absolute error against the exact value over the whole range
// (2 M random arguments per decade, numpy emulation), the same as libm's correctly reduced sinf / cosf
// to within one ulp -- at ~30 vector instructions instead of the ~150 of the full-range routine, which
// made k_posenc the one VALU-bound elementwise kernel.  Larger arguments take libm's sincosf.
__device__ __forceinline__ void sincos_cw(float x, float& s, float& c) {
  if (__builtin_expect(!(fabsf(x) <= 1.0e5f), 0)) {
    sincosf(x, &s, &c);
    return;
  }
  const float n = rintf(x * 0.636619772367581343f);
  float r = fmaf(n, -1.5703125f, x);
  r = fmaf(n, -0.00048351287841796875f, r);
  r = fmaf(n, -3.13855707645416259765625e-07f, r);
  r = fmaf(n, -6.077100628276710381e-11f, r);
  const float r2 = r * r;
  float ps = fmaf(r2, 2.7183114939898219064e-06f, -1.98393348360966317347e-04f);
  ps = fmaf(r2, ps, 8.3333293858894631756e-03f);
  ps = fmaf(r2, ps, -1.66666666416265235595e-01f);
  const float sp = fmaf(r * r2, ps, r);
  float pc = fmaf(r2, 2.43904487962774090654e-05f, -1.38867637746099294692e-03f);
  pc = fmaf(r2, pc, 4.16666233237390631894e-02f);
  pc = fmaf(r2, pc, -4.99999997251031003120e-01f);
  const float cp = fmaf(r2, pc, 1.0f);
  const int q = (int)n;
  const float a = (q & 1) ? cp : sp, b = (q & 1) ? sp : cp;
  s = (q & 2) ? -a : a;
  c = ((q + 1) & 2) ? -b : b;
}

// ------------------------------------------------------------------------------------ posenc
// The 126 + 2 columns that q_fcs concatenates to q are stored in two pieces (same column order):
//   emb_pt  [N, 64]  = posenc(pts) (63) | first view-encoding value (the normalised direction's x)
//   emb_ray [R, 64]  = the other 62 view-encoding values | 0 0
// The view encoding depends on the ray only: k_qpos reads it from the (cached) per-ray row instead
// of 256 more bytes per point.
__global__ void k_posenc(const float* __restrict__ pts, const float* __restrict__ ray_d, int S,
                         long long N, long long R, float* __restrict__ emb_pt,
                         float* __restrict__ emb_ray) {
  // 32 lanes per 64-column row: lane p < 30 owns the (component p % 3, frequency p / 3) pair and
  // writes its sine AND its cosine from one range reduction (a lane per column ran the reduction
  // twice as often); lane 30 the identity columns, lane 31 what is left of the row.
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (N + R) * 32) return;
  const bool ray_part = idx >= N * 32;
  const long long row = (ray_part ? idx - N * 32 : idx) >> 5;
  const int p = (int)(idx & 31);
  float* out = (ray_part ? emb_ray : emb_pt) + (row << 6);
  float sx = 0.f, sy = 0.f, sz = 0.f;
  if (!ray_part && p != 31) {
    sx = pts[row * 3 + 0], sy = pts[row * 3 + 1], sz = pts[row * 3 + 2];
  } else {
    // (32-bit division: N fits an int in the C-ABI)
    const long long r = ray_part ? row : (long long)((unsigned)row / (unsigned)S);
    const float dx = ray_d[r * 3 + 0], dy = ray_d[r * 3 + 1], dz = ray_d[r * 3 + 2];
    const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
    sx = dx / nrm, sy = dy / nrm, sz = dz / nrm;
  }
  // column of encoding entry k (0..62: x y z, then per frequency sin xyz, cos xyz): the point row holds
  // posenc entries 0..62 and view entry 0 in column 63; the ray row holds view entries 1..62 and 0 0
  const int shift = ray_part ? -1 : 0;
  if (p < 30) {
    const int f = p / 3, comp = p - 3 * f;
    const float arg = (comp == 0 ? sx : (comp == 1 ? sy : sz)) * (float)(1 << f);
    float sv, cv;
    sincos_cw(arg, sv, cv);
    out[3 + 6 * f + comp + shift] = sv;
    out[6 + 6 * f + comp + shift] = cv;
  } else if (p == 30) {
    if (!ray_part) out[0] = sx;
    out[1 + shift] = sy;
    out[2 + shift] = sz;
  } else {
    if (ray_part) out[62] = 0.0f, out[63] = 0.0f;
    else out[63] = sx;  // first entry of the view encoding: the normalised direction's x
  }
}

// ------------------------------------------------------------------------------------ rgbfeat
// LDS: W0 [64][52] (in_ch zero-padded to 48), W2 [64][68], b0 [64], b2 [64]
#define RF_LD0 52
template <int SPL>
__global__ void __launch_bounds__(RF_TPB)
k_rgbfeat(gpn_linear_t fc0, gpn_linear_t fc2, const float* __restrict__ rgb_feat, long long N, int V,
          int in_ch, int in_ld, const float* __restrict__ mask, float* __restrict__ feat,
          float* __restrict__ q0) {
  float* W0 = smem;
  float* W2 = W0 + wimg_floats(SPL, 64, 3, RF_LD0);
  float* b0 = W2 + wimg_floats(SPL, 64, 4, LDW);
  float* b2 = b0 + 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  {  // two-phase staging (gnt_mfma.h): every global load first, then the LDS images
    WStage<SPL, 64, 3, RF_TPB> s0;
    WStage<SPL, 64, 4, RF_TPB> s2;
    VecStage<64, RF_TPB> v0, v2;
    s0.load(fc0.w, 64, in_ch, tid), s2.load(fc2.w, 64, 64, tid);
    v0.load(fc0.b, 64, tid), v2.load(fc2.b, 64, tid);
    s0.store(W0, RF_LD0, tid), s2.store(W2, LDW, tid);
    v0.store(b0, tid), v2.store(b2, tid);
  }
  __syncthreads();
  const int j = lane & 15, g = lane >> 4;
  const long long ntiles = (N + 15) >> 4;
  const long long tstride = (long long)gridDim.x * (RF_TPB / 64);
  // input row of (point, view), zero-padded to 48 channels.  Rows are in_ld floats apart: the
  // reference's dense layout (35) is read element-wise (rows are not 16-byte aligned: 12 requests per
  // lane); rows padded to a multiple of 4 floats (the fused path: 36) are read as 16-byte units
  const bool vec = (in_ld & 3) == 0;  // (base alignment checked on the host)
  auto load_row = [&](long long n_, int v_, f4* x_) {
    const float* row = rgb_feat + ((size_t)n_ * V + v_) * in_ld;
    if (vec) {
#pragma unroll
      for (int T = 0; T < 3; ++T) {
        const int c = 16 * T + 4 * g;
        x_[T] = c < in_ld ? ld4(row + c) : f4_zero();
#pragma unroll
        for (int r = 0; r < 4; ++r) x_[T][r] = c + r < in_ch ? x_[T][r] : 0.0f;
      }
    } else {
#pragma unroll
      for (int T = 0; T < 3; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * T + 4 * g + r;
          x_[T][r] = c < in_ch ? row[c] : 0.0f;
        }
    }
  };
  // mask given: the feature row of a masked pair is not stored unless its point has no visible view
  // at all (gpn_view_attn reads the rows of visible pairs, and every row of such a point)
  auto vis_of = [&](long long n_) {
    unsigned b = 0;
    if (mask != nullptr) {
      for (int v_ = 0; v_ < V; ++v_) b |= (mask[(size_t)n_ * V + v_] != 0.0f ? 1u : 0u) << (v_ & 31);
      if (b == 0) b = 0xffffffffu;
    } else {
      b = 0xffffffffu;
    }
    return b;
  };
  long long tile = (long long)blockIdx.x * (RF_TPB / 64) + wave;
  if (tile >= ntiles) return;
  f4 xpre[3];  // the next (tile, view) row is loaded one iteration ahead
  unsigned wr_next;
  {
    const long long n0 = tile * 16 + j;
    load_row(n0 < N ? n0 : N - 1, 0, xpre);
    wr_next = vis_of(n0 < N ? n0 : N - 1);
  }
  for (; tile < ntiles; tile += tstride) {
    const long long n = tile * 16 + j;
    const bool valid = n < N;
    const long long nc = valid ? n : N - 1;
    const long long nn = (tile + tstride) * 16 + j, nnc = nn < N ? nn : N - 1;
    const unsigned wr = valid ? wr_next : 0u;  // views whose rows this lane's point stores
    if (tile + tstride < ntiles) wr_next = vis_of(nnc);
    f4 qmax[4];
#pragma unroll
    for (int M = 0; M < 4; ++M) qmax[M] = f4_splat(-INFINITY);
    for (int v = 0; v < V; ++v) {
      f4 x[3];
#pragma unroll
      for (int T = 0; T < 3; ++T) x[T] = xpre[T];
      if (v + 1 < V) load_row(nc, v + 1, xpre);
      else load_row(nnc, 0, xpre);
      f4 h[4], y[4];
#pragma unroll
      for (int M = 0; M < 4; ++M) h[M] = ld4(b0 + 16 * M + 4 * g);
      lin<SPL, 3, 4>(W0, RF_LD0, x, h, lane);
#pragma unroll
      for (int M = 0; M < 4; ++M) {
        h[M] = f4_relu_t<(SPL != 0)>(h[M]);
        y[M] = ld4(b2 + 16 * M + 4 * g);
      }
      lin<SPL, 4, 4>(W2, LDW, h, y, lane);
      if constexpr (SPL != 0) {
        // split encoding of a feature row (gnt_mfma.h: feat_row_*): the operand split is done once
        // here instead of once per transformer layer in k_view_attn
        if ((wr >> (v & 31)) & 1u) store_feat_split(feat + ((size_t)nc * V + v) * C64, y, g);
      } else {
        float* out = feat + ((size_t)nc * V + v) * C64 + 4 * g;
#pragma unroll
        for (int M = 0; M < 4; ++M)
          if ((wr >> (v & 31)) & 1u) st4(out + 16 * M, y[M]);
      }
#pragma unroll
      for (int M = 0; M < 4; ++M) qmax[M] = f4_max(qmax[M], y[M]);
    }
    if (valid) {
#pragma unroll
      for (int M = 0; M < 4; ++M) st4(q0 + (size_t)n * C64 + 16 * M + 4 * g, qmax[M]);
    }
  }
}

// ------------------------------------------------------------------------------------ fused projector + rgbfeat
// The fast path of render_rays: projection.py:31-139 (RGB + 32-channel gather, masks, ray_diff) and
// transformer_network.py:309-311 (rgbfeat_fc, max over views) in ONE kernel -- the [N,V,35] tensor between
// them (330 MB written and read back per 64-sample pass) never exists, and the pairs whose gathered row is
// exactly zero (every bilinear tap outside both maps: ~half of the pairs of an indoor rig) cost neither a
// gather nor an MLP evaluation: their feature row is the constant c0 = fc2(relu(fc0.bias)) + bias.
// Same T-tile scheme as k_rgbfeat (16 points per wave-tile).  Per tile:
//   pre-pass  lane (j, g) projects point j into views g, g+4, ...: mask bits of every view (written out),
//             vis = the point's visibility word, wr = the views whose feature rows gpn_view_attn will read
//             (the visible ones; every view when the point has no visible view at all)
//   per group of 4 views
//     phase 1 lane (j, g) does the per-pair arithmetic of (point j, view 4k+g) once: tap positions and
//             bilinear weights on both maps -> a per-wave LDS table; ray_diff for the pairs in wr
//     phase 2 per view: 4 lanes per point gather their channels straight from the NHWC maps (feature
//             channels first, RGB last: fc0's columns are permuted to match when its image is staged),
//             bilinear blend with the projector's fma chain (same bits as k_project_feat), MLP on MFMA,
//             feature row stored if in wr, running maximum for q0 (over ALL views, as in the reference)
// A point's results never depend on its tile mates, so chunking / rank count change no bits.
#ifndef PR_TPB
#define PR_TPB 256  // 146 VGPRs, ~45 KB of LDS: three workgroups (3 waves per SIMD) per CU
#endif
#ifndef PR_BPC
#define PR_BPC 3
#endif
struct PrTab {  // per wave: 4 views x 16 points
  int fx[64], fy[64], rx[64], ry[64];  // clamped tap columns / rows, (x1 << 16) | x0
  float fw[4][64], rw[4][64];          // nw ne sw se; 0 for a tap outside the map
};

// Visibility masks alone (projection.py:31-55, :99-104): the pass gpn_vis_order needs BEFORE the fused
// kernel when that kernel is to write its rows in k_view_attn's tile order.  Same two functions as the
// pre-pass of k_proj_rgbfeat: same bits.
__global__ void __launch_bounds__(256)
k_proj_mask(gpn_project_args_t a) {
  __shared__ float sP[GPN_MAX_VIEWS * 12];
  const int V = a.V;
  for (int i = threadIdx.x; i < V * 12; i += 256) sP[i] = a.proj[(i / 12) * 16 + (i % 12)];
  __syncthreads();
  for (long long n = (long long)blockIdx.x * 256 + threadIdx.x; n < a.N; n += (long long)gridDim.x * 256) {
    const float px = a.xyz[n * 3 + 0], py = a.xyz[n * 3 + 1], pz = a.xyz[n * 3 + 2];
    for (int v = 0; v < V; ++v) {
      const GpnPix pp = gpn_project(sP + v * 12, px, py, pz);
      a.mask[(size_t)n * V + v] = (pp.front && gpn_inbound(pp.px, pp.py, a.img_h, a.img_w)) ? 1.0f : 0.0f;
    }
  }
}

// perm != nullptr: tile t of this kernel = slots 16 t .. 16 t + 15 of gpn_vis_order (the tiles of
// k_view_attn), the masks are already in a.mask (k_proj_mask) and are not written again.
// tiled != 0: the feature row of (slot 16 t + j, view v) goes to feat[((t V + v) 16 + j) 64] -- the 16 rows
// of a (tile, view) pair are one contiguous 4 KB block, read back by k_view_attn as whole sequential
// lines -- and, with rdp != nullptr, its ray_diff to the tile-major image of k_rd_pack.  Slots past N
// repeat the last slot's point (rows written, q0 not).
template <int SPL>
__global__ void __launch_bounds__(PR_TPB)
k_proj_rgbfeat(gpn_project_args_t a, gpn_linear_t fc0, gpn_linear_t fc2, float* __restrict__ feat,
               float* __restrict__ q0, const int* __restrict__ perm, float* __restrict__ rdp, int tiled) {
  float* W0 = smem;
  float* W2 = W0 + wimg_floats(SPL, 64, 3, RF_LD0);
  float* b0 = W2 + wimg_floats(SPL, 64, 4, LDW);
  float* b2 = b0 + 64;
  float* sP = b2 + 64;                      // [V][12]
  float* sPos = sP + GPN_MAX_VIEWS * 12;    // [V][3]
  PrTab* tabs = reinterpret_cast<PrTab*>(sPos + GPN_MAX_VIEWS * 3);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  PrTab& tab = tabs[wave];
  const int V = a.V;
  {
    // fc0's image with the columns in gather order: [feature 0..31 | r g b | 0 ...]  (reference order: rgb first)
    const int in_ch = 3 + a.feat_c;
    if constexpr (SPL != 0) {
      bf8* dst = reinterpret_cast<bf8*>(W0);
      for (int idx = tid; idx < 64 * 8; idx += PR_TPB) {  // 64 rows x (2 K pairs x 4 g)
        const int row = idx >> 3, u = idx & 7, P = u >> 2, g = u & 3;
        f4 x0 = f4_zero(), x1 = f4_zero();
        const float* r = fc0.w + (size_t)row * in_ch;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c0 = 32 * P + 4 * g + e, c1 = c0 + 16;  // gather-order columns
          x0[e] = c0 < 32 ? r[3 + c0] : (c0 < 35 ? r[c0 - 32] : 0.0f);
          x1[e] = c1 < 32 ? r[3 + c1] : (c1 < 35 ? r[c1 - 32] : 0.0f);
        }
        const XS sp = split8(x0, x1);
        const int blk = (((row >> 4) * 2 + P) * 2) * 64 + 16 * g + (row & 15);
        dst[blk] = sp.hi;
        dst[blk + 64] = sp.mid;
      }
    } else {
      for (int idx = tid; idx < 64 * 48; idx += PR_TPB) {
        const int row = idx / 48, c = idx - row * 48;
        const float* r = fc0.w + (size_t)row * in_ch;
        W0[row * RF_LD0 + c] = c < 32 ? r[3 + c] : (c < 35 ? r[c - 32] : 0.0f);
      }
    }
    WStage<SPL, 64, 4, PR_TPB> s2;
    VecStage<64, PR_TPB> v0, v2;
    s2.load(fc2.w, 64, 64, tid);
    v0.load(fc0.b, 64, tid), v2.load(fc2.b, 64, tid);
    s2.store(W2, LDW, tid);
    v0.store(b0, tid), v2.store(b2, tid);
    for (int i = tid; i < V * 12; i += PR_TPB) sP[i] = a.proj[(i / 12) * 16 + (i % 12)];
    for (int i = tid; i < V * 3; i += PR_TPB) sPos[i] = a.src_pos[i];
  }
  __syncthreads();
  const int j = lane & 15, g = lane >> 4;
  const long long N = a.N;
  const long long ntiles = (N + 15) >> 4;
  const long long tstride = (long long)gridDim.x * (PR_TPB / 64);
  const unsigned all_views = V >= 32 ? 0xffffffffu : ((1u << V) - 1u);

  // one MLP evaluation: x (3 T-tiles, gather order) -> y (4 T-tiles)
  auto mlp = [&](const f4* x, f4* y) {
    f4 h[4];
#pragma unroll
    for (int M = 0; M < 4; ++M) h[M] = ld4(b0 + 16 * M + 4 * g);
    lin<SPL, 3, 4>(W0, RF_LD0, x, h, lane);
#pragma unroll
    for (int M = 0; M < 4; ++M) {
      h[M] = f4_relu_t<(SPL != 0)>(h[M]);
      y[M] = ld4(b2 + 16 * M + 4 * g);
    }
    lin<SPL, 4, 4>(W2, LDW, h, y, lane);
  };
  auto store_row = [&](long long n, long long tile_, int v, const f4* y) {
    float* row = feat + (tiled ? (((size_t)tile_ * V + v) * 16 + (lane & 15)) : ((size_t)n * V + v)) * C64;
    if constexpr (SPL != 0) {
      store_feat_split(row, y, g);
    } else {
#pragma unroll
      for (int M = 0; M < 4; ++M) st4(row + 16 * M + 4 * g, y[M]);
    }
  };
  // the feature row of an all-zero input, by the same instruction sequence as any other row
  f4 c0[4];
  {
    const f4 xz[3] = {f4_zero(), f4_zero(), f4_zero()};
    mlp(xz, c0);
  }
  const int fw_ = a.feat_w, fh_ = a.feat_h, rw_ = a.rgb_w, rh_ = a.rgb_h;

  for (long long tile = (long long)blockIdx.x * (PR_TPB / 64) + wave; tile < ntiles; tile += tstride) {
    const long long slot = tile * 16 + j;
    const bool valid = slot < N;
    const long long sc = valid ? slot : N - 1;
    const long long n = perm != nullptr ? perm[sc] : sc;  // the point of this lane
    const bool real = valid || tiled != 0;                // tiled: padding slots carry the last point's rows
    const float px = a.xyz[n * 3 + 0], py = a.xyz[n * 3 + 1], pz = a.xyz[n * 3 + 2];
    // ---- pre-pass: masks of every view
    unsigned vis = 0;
    for (int v = g; v < V; v += 4) {
      const GpnPix pp = gpn_project(sP + v * 12, px, py, pz);
      const bool m = pp.front && gpn_inbound(pp.px, pp.py, a.img_h, a.img_w);
      if (valid && perm == nullptr) a.mask[(size_t)n * V + v] = m ? 1.0f : 0.0f;
      vis |= (m ? 1u : 0u) << v;
    }
    vis |= __shfl_xor(vis, 16);
    vis |= __shfl_xor(vis, 32);
    const unsigned wr = real ? (vis != 0 ? vis : all_views) : 0u;  // rows gpn_view_attn reads
    f4 qmax[4];
#pragma unroll
    for (int M = 0; M < 4; ++M) qmax[M] = f4_splat(-INFINITY);

    for (int vb = 0; vb < V; vb += 4) {
      // ---- phase 1: lane (j, g) <-> pair (point j, view vb + g)
      bool zero_row = true;
      {
        const int v = vb + g;
        if (v < V) {
          const GpnPix pp = gpn_project(sP + v * 12, px, py, pz);
          const GpnTaps tf = gpn_taps(pp.px, pp.py, a.img_h, a.img_w, fh_, fw_);
          const GpnTaps tr = gpn_taps(pp.px, pp.py, a.img_h, a.img_w, rh_, rw_);
          const bool any_f = (tf.vx0 || tf.vx1) && (tf.vy0 || tf.vy1);
          const bool any_r = (tr.vx0 || tr.vx1) && (tr.vy0 || tr.vy1);
          zero_row = !real || !(any_f || any_r);
          auto cl = [](int c, int hi) { return min(max(c, 0), hi); };
          tab.fx[lane] = cl(tf.x0, fw_ - 1) | (cl(tf.x0 + 1, fw_ - 1) << 16);
          tab.fy[lane] = cl(tf.y0, fh_ - 1) | (cl(tf.y0 + 1, fh_ - 1) << 16);
          tab.rx[lane] = cl(tr.x0, rw_ - 1) | (cl(tr.x0 + 1, rw_ - 1) << 16);
          tab.ry[lane] = cl(tr.y0, rh_ - 1) | (cl(tr.y0 + 1, rh_ - 1) << 16);
          tab.fw[0][lane] = (tf.vy0 && tf.vx0) ? tf.nw : 0.0f, tab.fw[1][lane] = (tf.vy0 && tf.vx1) ? tf.ne : 0.0f;
          tab.fw[2][lane] = (tf.vy1 && tf.vx0) ? tf.sw : 0.0f, tab.fw[3][lane] = (tf.vy1 && tf.vx1) ? tf.se : 0.0f;
          tab.rw[0][lane] = (tr.vy0 && tr.vx0) ? tr.nw : 0.0f, tab.rw[1][lane] = (tr.vy0 && tr.vx1) ? tr.ne : 0.0f;
          tab.rw[2][lane] = (tr.vy1 && tr.vx0) ? tr.sw : 0.0f, tab.rw[3][lane] = (tr.vy1 && tr.vx1) ? tr.se : 0.0f;
          if ((a.ray_diff != nullptr || rdp != nullptr) && ((wr >> v) & 1u)) {  // dead pairs are never read by gpn_view_attn: their 9 divisions are skipped
            const float4 r4 = gpn_ray_diff(px, py, pz, a.qry_pos[0], a.qry_pos[1], a.qry_pos[2], sPos[v * 3 + 0],
                                           sPos[v * 3 + 1], sPos[v * 3 + 2]);
            if (a.ray_diff != nullptr && valid) *reinterpret_cast<float4*>(a.ray_diff + ((size_t)n * V + v) * 4) = r4;
            if (rdp != nullptr) {
              float* o = rdp + ((size_t)tile * V + v) * 64 + j;
              o[0] = r4.x, o[16] = r4.y, o[32] = r4.z, o[48] = r4.w;
            }
          }
        }
      }
      const unsigned long long zb = __ballot(zero_row);  // bit 16 g + j
      // ---- phase 2: the group's views one after the other
      for (int g2 = 0; g2 < 4 && vb + g2 < V; ++g2) {
        const int v = vb + g2;
        const unsigned zmask = (unsigned)(zb >> (16 * g2)) & 0xffffu;  // wave-uniform
        f4 y[4];
        if (zmask == 0xffffu) {  // every row of this (tile, view) is zero: no gather, no MLP
#pragma unroll
          for (int M = 0; M < 4; ++M) y[M] = c0[M];
        } else {
          const int e = 16 * g2 + j;
          const int fx = tab.fx[e], fy = tab.fy[e];
          const f4 w = (f4){tab.fw[0][e], tab.fw[1][e], tab.fw[2][e], tab.fw[3][e]};
          const float* fm = a.feat + (size_t)v * fh_ * fw_ * 32 + 4 * g;
          const int x0 = fx & 0xffff, x1 = (int)((unsigned)fx >> 16), y0 = (fy & 0xffff) * fw_, y1 = (int)((unsigned)fy >> 16) * fw_;
          const int o00 = (y0 + x0) * 32, o01 = (y0 + x1) * 32, o10 = (y1 + x0) * 32, o11 = (y1 + x1) * 32;
          f4 x[3];
#pragma unroll
          for (int T = 0; T < 2; ++T) {
            const f4 tnw = ld4(fm + o00 + 16 * T), tne = ld4(fm + o01 + 16 * T);
            const f4 tsw = ld4(fm + o10 + 16 * T), tse = ld4(fm + o11 + 16 * T);
#pragma unroll
            for (int c = 0; c < 4; ++c) x[T][c] = gpn_bilerp4(tnw[c], tne[c], tsw[c], tse[c], w.x, w.y, w.z, w.w);
          }
          x[2] = f4_zero();
          if (g == 0) {  // RGB: channels 32..34 of the gather order
            const int rx = tab.rx[e], ry = tab.ry[e];
            const f4 u = (f4){tab.rw[0][e], tab.rw[1][e], tab.rw[2][e], tab.rw[3][e]};
            const float* im = a.rgb + (size_t)v * rh_ * rw_ * 3;
            const int X0 = rx & 0xffff, X1 = (int)((unsigned)rx >> 16), Y0 = (ry & 0xffff) * rw_, Y1 = (int)((unsigned)ry >> 16) * rw_;
            const int p00 = (Y0 + X0) * 3, p01 = (Y0 + X1) * 3, p10 = (Y1 + X0) * 3, p11 = (Y1 + X1) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c)
              x[2][c] = gpn_bilerp4(im[p00 + c], im[p01 + c], im[p10 + c], im[p11 + c], u.x, u.y, u.z, u.w);
          }
          mlp(x, y);
        }
        if ((wr >> v) & 1u) store_row(n, tile, v, y);
#pragma unroll
        for (int M = 0; M < 4; ++M) qmax[M] = f4_max(qmax[M], y[M]);
      }
    }
    if (valid) {
#pragma unroll
      for (int M = 0; M < 4; ++M) st4(q0 + (size_t)n * C64 + 16 * M + 4 * g, qmax[M]);
    }
  }
}

// ------------------------------------------------------------------------------------ visibility order
// Counting sort of the points by their set of visible views (bit v = mask[n][v] != 0; hashed to 12
// bits when V > 12).  The order inside a bin is arbitrary: every point is computed independently of
// its tile mates, so results do not depend on it.
#define VIS_BINS 4096
#define VIS_CHUNK 2048  // points per block

__device__ __forceinline__ unsigned vis_bits(const float* __restrict__ mrow, int V) {
  unsigned key = 0;
  for (int v = 0; v < V; ++v) key |= (mrow[v] != 0.0f ? 1u : 0u) << v;
  return key;
}
__device__ __forceinline__ unsigned vis_bin_of(unsigned key, int V) {
  return V <= 12 ? key : ((key * 2654435761u) >> 20);
}
__device__ __forceinline__ unsigned vis_bin(const float* __restrict__ mrow, int V) {
  return vis_bin_of(vis_bits(mrow, V), V);
}

__global__ void __launch_bounds__(256)
k_vis_hist(const float* __restrict__ mask, int N, int V, int* __restrict__ hist) {
  __shared__ int lh[VIS_BINS];
  for (int b = threadIdx.x; b < VIS_BINS; b += 256) lh[b] = 0;
  __syncthreads();
  const int n0 = blockIdx.x * VIS_CHUNK;
  for (int i = threadIdx.x; i < VIS_CHUNK && n0 + i < N; i += 256)
    atomicAdd(&lh[vis_bin(mask + (size_t)(n0 + i) * V, V)], 1);
  __syncthreads();
  for (int b = threadIdx.x; b < VIS_BINS; b += 256)
    if (lh[b]) atomicAdd(&hist[b], lh[b]);
}

// Start offset of every bin.  Bins are laid out by decreasing cost class (number of views the tile
// has to compute: popcount of the pattern, V for the all-masked pattern), so that the static
// round-robin tile assignment of k_view_attn gives every wave the same mix of heavy and light tiles.
__global__ void __launch_bounds__(1024)
k_vis_scan(const int* __restrict__ hist, int* __restrict__ cursor, int V) {
  // exclusive scan of the 4096 bin counts in cost-class order (classes V .. 0, bins ascending inside a
  // class).  Per class: scan inside each wave with shuffles, wave totals through LDS -- two barriers per
  // class (a 1024-wide Hillis-Steele scan per class took 22 of them: 21 us per launch).
  __shared__ int wtot[16];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;  // 4 bins per thread
  int h[4], cls[4];
  for (int k = 0; k < 4; ++k) {
    const int b = 4 * t + k;
    h[k] = hist[b];
    cls[k] = V <= 12 ? (b == 0 ? V : __popc((unsigned)b)) : 0;
  }
  int running = 0;  // every thread keeps the same running offset
  for (int c = (V <= 12 ? V : 0); c >= 0; --c) {
    int sum = 0;
    for (int k = 0; k < 4; ++k) sum += cls[k] == c ? h[k] : 0;
    int incl = sum;  // inclusive scan over the wave
    for (int off = 1; off < 64; off <<= 1) {
      const int up = __shfl_up(incl, off);
      if (lane >= off) incl += up;
    }
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    int before = 0, total = 0;
    for (int i = 0; i < 16; ++i) {
      const int v = wtot[i];
      before += i < w ? v : 0;
      total += v;
    }
    int base = running + before + incl - sum;
    for (int k = 0; k < 4; ++k)
      if (cls[k] == c) cursor[4 * t + k] = base, base += h[k];
    running += total;
    __syncthreads();  // wtot is rewritten by the next class
  }
}

__global__ void __launch_bounds__(256)
k_vis_scatter(const float* __restrict__ mask, int N, int V, int* __restrict__ cursor,
              int* __restrict__ perm) {  // perm [2N]: point of every slot, then its visibility bits
  __shared__ int lh[VIS_BINS];
  for (int b = threadIdx.x; b < VIS_BINS; b += 256) lh[b] = 0;
  __syncthreads();
  const int n0 = blockIdx.x * VIS_CHUNK;
  int rank[VIS_CHUNK / 256];
  unsigned bin[VIS_CHUNK / 256], bits[VIS_CHUNK / 256];
#pragma unroll
  for (int k = 0; k < VIS_CHUNK / 256; ++k) {
    const int i = threadIdx.x + 256 * k;
    rank[k] = -1;
    if (n0 + i < N) {
      bits[k] = vis_bits(mask + (size_t)(n0 + i) * V, V);
      bin[k] = vis_bin_of(bits[k], V);
      rank[k] = atomicAdd(&lh[bin[k]], 1);
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < VIS_BINS; b += 256)
    if (lh[b]) lh[b] = atomicAdd(&cursor[b], lh[b]);  // this block's first slot in bin b
  __syncthreads();
#pragma unroll
  for (int k = 0; k < VIS_CHUNK / 256; ++k)
    if (rank[k] >= 0) {
      const int slot = lh[bin[k]] + rank[k];
      perm[slot] = n0 + threadIdx.x + 256 * k;
      perm[N + slot] = (int)bits[k];
    }
}

// ------------------------------------------------------------------------------------ view attention
// LDS image (floats):
//   Wq, Wk, Wv, Wo [64][68] | pos2 [64][20] | attn0 [16][68] | attn2 [64][20] | pos0 [16][4]
//   | b_pos0[16] b_pos2[64] b_attn0[16] b_attn2[64] b_out[64] gamma[64] beta[64]
#define VA_LD8 20
#define VA_LDAF 100  // fp32 image of the folded [16][96] matrix (A0 Wk | A0 P2 | 0): 96 columns + 4 pad
struct ViewLds {
  float *Wq, *Wk, *Wv, *Wo, *P2, *A0, *A2, *P0, *bP0, *bP2, *bA0, *bA2, *bO, *gam, *bet;
  float *AF, *c0;  // folded mode: (A0 Wk | A0 P2) image and the constant A0 b_pos2 + b_attn0
};
template <int SPL>
__device__ __forceinline__ ViewLds view_lds(float* s) {
  static_assert(16 * VA_LDAF <= 64 * LDW, "the fp32 folded image must fit the unused Wk slot");
  ViewLds L;
  L.Wq = s;
  L.Wk = L.Wq + wimg_floats(SPL, 64, 4, LDW);
  L.Wv = L.Wk + wimg_floats(SPL, 64, 4, LDW);
  L.Wo = L.Wv + wimg_floats(SPL, 64, 4, LDW);
  L.P2 = L.Wo + wimg_floats(SPL, 64, 4, LDW);
  L.A0 = L.P2 + wimg_floats(SPL, 64, 1, VA_LD8);
  L.A2 = L.A0 + wimg_floats(SPL, 16, 4, LDW);
  L.P0 = L.A2 + wimg_floats(SPL, 64, 1, VA_LD8);
  L.bP0 = L.P0 + 64;
  L.bP2 = L.bP0 + 16;
  L.bA0 = L.bP2 + 64;
  L.bA2 = L.bA0 + 16;
  L.bO = L.bA2 + 64;
  L.gam = L.bO + 64;
  L.bet = L.gam + 64;
  L.c0 = L.bet + 64;
  // split mode: its own slot behind the vectors; fp32 mode: the Wk slot (k itself is never formed when folded)
  L.AF = SPL ? L.c0 + 16 : L.Wk;
  return L;
}
constexpr int view_lds_floats(bool spl) {
  return 4 * wimg_floats(spl, 64, 4, LDW) + 2 * wimg_floats(spl, 64, 1, VA_LD8) +
         wimg_floats(spl, 16, 4, LDW) + 64 + 16 + 64 + 16 + 64 * 4 + 16 +
         (spl ? wimg_floats(true, 16, 6, 0) : 0);
}

// Folded weight products of one Attention2D layer (gpn_view_fold in the header), fp64 accumulation.
#define FOLD_VK 0
#define FOLD_AF (64 * 64)
#define FOLD_C0 (64 * 64 + 16 * 96)
__global__ void __launch_bounds__(256)
k_view_fold(gpn_view_layer_t Lw, float* __restrict__ out) {
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < GPN_VIEW_FOLD_FLOATS; idx += gridDim.x * 256) {
    double acc = 0.0;
    if (idx < FOLD_AF) {  // (Wv Wk)[i][j]
      const int i = idx >> 6, jj = idx & 63;
      for (int c = 0; c < 64; ++c) acc += (double)Lw.v.w[i * 64 + c] * (double)Lw.k.w[c * 64 + jj];
    } else if (idx < FOLD_C0) {
      const int r = (idx - FOLD_AF) / 96, c = (idx - FOLD_AF) % 96;
      if (r < 8 && c < 64) {         // (A0 Wk)[r][c]
        for (int m = 0; m < 64; ++m) acc += (double)Lw.attn0.w[r * 64 + m] * (double)Lw.k.w[m * 64 + c];
      } else if (r < 8 && c < 72) {  // (A0 P2)[r][c-64]
        for (int m = 0; m < 64; ++m) acc += (double)Lw.attn0.w[r * 64 + m] * (double)Lw.pos2.w[m * 8 + (c - 64)];
      }
    } else {
      const int r = idx - FOLD_C0;
      if (r < 8) {
        acc = Lw.attn0.b ? (double)Lw.attn0.b[r] : 0.0;
        if (Lw.pos2.b)
          for (int m = 0; m < 64; ++m) acc += (double)Lw.attn0.w[r * 64 + m] * (double)Lw.pos2.b[m];
      }
    }
    out[idx] = (float)acc;
  }
}

// ray_diff in the order k_view_attn consumes it: out[((tile * V + v) * 4 + c) * 16 + j] = component c of
// the pair (point of slot 16 tile + j, view v); slots past N repeat the last point.  Source: the
// [N,V,4] array of the caller (rd != nullptr), else compute_angle itself from the points and the camera
// centres (proj_common.h: the very function the projector kernels call, same bits).
__global__ void __launch_bounds__(256)
k_rd_pack(const float* __restrict__ rd, const float* __restrict__ xyz, float qx, float qy, float qz,
          const float* __restrict__ src_pos, const int* __restrict__ perm, long long N, int V,
          float* __restrict__ out) {
  const long long total = ((N + 15) >> 4) * V * 16;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int j = (int)(idx & 15);
    const long long tv = idx >> 4;  // tile * V + v
    const long long tile = tv / V;
    const int v = (int)(tv - tile * V);
    const long long slot = tile * 16 + j, sc = slot < N ? slot : N - 1;
    const long long n = perm != nullptr ? perm[sc] : sc;
    float4 r;
    if (rd != nullptr) {
      r = *reinterpret_cast<const float4*>(rd + ((size_t)n * V + v) * 4);
    } else {
      r = gpn_ray_diff(xyz[n * 3 + 0], xyz[n * 3 + 1], xyz[n * 3 + 2], qx, qy, qz,
                       src_pos[v * 3 + 0], src_pos[v * 3 + 1], src_pos[v * 3 + 2]);
    }
    float* o = out + (size_t)tv * 64 + j;
    o[0] = r.x, o[16] = r.y, o[32] = r.z, o[48] = r.w;
  }
}

// ------------------------------------------------------------------------------------ feed-forward
// LDS: W1 [256][68] | W2 [64][260] | b1 [256] | b2 [64] | gamma [64] | beta [64]
#define FF_LD2 260
constexpr int ff_lds_floats(bool spl) {
  return wimg_floats(spl, 256, 4, LDW) + wimg_floats(spl, 64, 16, FF_LD2) + 256 + 64 * 3;
}

// The feed-forward block as device code shared by k_ff and by the second phase of k_view_attn (which
// runs it on the tiles the workgroup has just written, after re-staging LDS with the FF weights).
// Tiles of a wave: first + t * inner + i * outer (t < NTL tiles per iteration i); slot -> point through
// `perm` when given (the visibility order of the view attention), else identity.
struct FfLds {
  float *W1, *W2, *b1, *b2, *gam, *bet;
};
template <int SPL>
__device__ __forceinline__ FfLds ff_lds() {
  FfLds L;
  L.W1 = smem;
  L.W2 = L.W1 + wimg_floats(SPL, 256, 4, LDW);
  L.b1 = L.W2 + wimg_floats(SPL, 64, 16, FF_LD2);
  L.b2 = L.b1 + 256;
  L.gam = L.b2 + 64;
  L.bet = L.gam + 64;
  return L;
}
// two-phase staging (gnt_mfma.h): every global load first (load), then the LDS images (store); as a later
// phase of an attention kernel the loads are issued BEFORE the barrier that frees LDS
template <int SPL, int TPB>
struct FfStager {
  WStage<SPL, 256, 4, TPB> s1;
  WStage<SPL, 64, 16, TPB> s2;
  VecStage<256, TPB> v1;
  VecStage<64, TPB> v2, vg, vb;
  __device__ __forceinline__ void load(const float* __restrict__ norm_w, const float* __restrict__ norm_b,
                                       const gpn_linear_t& fc1, const gpn_linear_t& fc2, int tid) {
    s1.load(fc1.w, 256, 64, tid), s2.load(fc2.w, 64, 256, tid);
    v1.load(fc1.b, 256, tid), v2.load(fc2.b, 64, tid), vg.load(norm_w, 64, tid), vb.load(norm_b, 64, tid);
  }
  __device__ __forceinline__ void store(const FfLds& L, int tid) const {
    s1.store(L.W1, LDW, tid), s2.store(L.W2, FF_LD2, tid);
    STAMP(9);
    v1.store(L.b1, tid), v2.store(L.b2, tid), vg.store(L.gam, tid), vb.store(L.bet, tid);
  }
};
template <int SPL, int TPB>
__device__ __forceinline__ void ff_stage(const FfLds& L, const float* __restrict__ norm_w,
                                         const float* __restrict__ norm_b, const gpn_linear_t& fc1,
                                         const gpn_linear_t& fc2, int tid) {
  FfStager<SPL, TPB> st;
  st.load(norm_w, norm_b, fc1, fc2, tid);
  st.store(L, tid);
}
// A wave's tiles are numbered u = 0 .. nu-1; `pt(u, j, n, ok)` gives the point (row of x) of lane j of
// tile u and whether it is a real point (else: any valid row, never stored).
// split mode: the input rows of the wave's tiles u0 .. u0+NTL-1 (the first ones are requested before /
// while the weights are staged)
template <class PT>
__device__ __forceinline__ void ff_prefetch(f4 (&xpre)[FF_NTL][4], const float* x_in, long long u0,
                                            int lane, PT&& pt) {
#pragma unroll
  for (int t = 0; t < FF_NTL; ++t) {
    long long n;
    bool ok;
    pt(u0 + t, lane & 15, n, ok);
#pragma unroll
    for (int T = 0; T < 4; ++T) xpre[t][T] = ld4(x_in + (size_t)n * C64 + 16 * T + 4 * (lane >> 4));
  }
}
// REV: walk the tiles from the last to the first -- as a later phase of an attention kernel the rows a wave
// wrote LAST are the ones still in L2 (a workgroup writes 200-260 KB per phase, its share of the L2 is 128 KB:
// re-reading in writing order would miss on every row)
template <int SPL, bool REV, class PT>
__device__ __forceinline__ void ff_run(const FfLds& L, [[maybe_unused]] f4 (&xpre)[FF_NTL][4],
                                       const float* x_in, float* x_out, long long nu, int lane,
                                       PT&& pt) {
  const int j = lane & 15, g = lane >> 4;
  if constexpr (SPL != 0) {
    // FF_NTL point tiles per wave iteration; the hidden layer is produced and consumed in four 64-wide
    // chunks, so the 256 hidden values of a point never exist at once
    constexpr int NTL = FF_NTL;
    const bf8* w1p = reinterpret_cast<const bf8*>(L.W1) + lane;
    const bf8* w2p = reinterpret_cast<const bf8*>(L.W2) + lane;
    // the input tiles of the next iteration are loaded one iteration ahead (the first ones: ff_prefetch)
    const long long nur = (nu + NTL - 1) / NTL * NTL;  // whole iterations (`pt` marks tiles >= nu as not real)
    for (long long uf = 0; uf < nur; uf += NTL) {
      const long long u = REV ? nur - NTL - uf : uf;
      long long n[NTL];
      bool ok[NTL];
      XS xs[NTL][2];
      f4 y[NTL][4], x[NTL][4];
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        pt(u + t, j, n[t], ok[t]);
#pragma unroll
        for (int T = 0; T < 4; ++T) x[t][T] = xpre[t][T];
      }
      ff_prefetch(xpre, x_in, REV ? (u >= NTL ? u - NTL : 0) : u + NTL, lane, pt);
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        f4 xn[4];
        layernorm_T(x[t], xn, L.gam, L.bet, 1e-6f, lane);
        xs[t][0] = split8(xn[0], xn[1]);
        xs[t][1] = split8(xn[2], xn[3]);
#pragma unroll
        for (int M = 0; M < 4; ++M) y[t][M] = ld4(L.b2 + 16 * M + 4 * g);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        f4 h[NTL][4];
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
          for (int M = 0; M < 4; ++M) h[t][M] = ld4(L.b1 + 64 * c + 16 * M + 4 * g);
        lin_S2<2, 2, 4, NTL, bf_terms(SPL)>(w1p + (4 * c * 2) * 128, xs, h);  // rows 64c.., all 64 inputs
        XS hs[NTL][2];
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
          hs[t][0] = split8(f4_relu_t<true>(h[t][0]), f4_relu_t<true>(h[t][1]));
          hs[t][1] = split8(f4_relu_t<true>(h[t][2]), f4_relu_t<true>(h[t][3]));
        }
        lin_S2<8, 2, 4, NTL, bf_terms(SPL)>(w2p + (2 * c) * 128, hs, y);      // all 64 rows, hidden 64c..64c+63
      }
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        if (ok[t]) {
#pragma unroll
          for (int M = 0; M < 4; ++M) {
            FF_ST(x_out + (size_t)n[t] * C64 + 16 * M + 4 * g, y[t][M] + x[t][M]);
          }
        }
      }
    }
  } else {
    for (long long uf = 0; uf < nu; ++uf) {
      const long long u = REV ? nu - 1 - uf : uf;
      long long n;
      bool valid;
      pt(u, j, n, valid);
      f4 x[4], xn[4], h[16], y[4];
#pragma unroll
      for (int T = 0; T < 4; ++T) x[T] = ld4(x_in + (size_t)n * C64 + 16 * T + 4 * g);
      layernorm_T(x, xn, L.gam, L.bet, 1e-6f, lane);
#pragma unroll
      for (int M = 0; M < 16; ++M) h[M] = ld4(L.b1 + 16 * M + 4 * g);
      lin_T<4, 16>(L.W1, LDW, xn, h, lane);
#pragma unroll
      for (int M = 0; M < 16; ++M) h[M] = f4_relu(h[M]);
#pragma unroll
      for (int M = 0; M < 4; ++M) y[M] = ld4(L.b2 + 16 * M + 4 * g);
      lin_T<16, 4>(L.W2, FF_LD2, h, y, lane);
      if (valid) {
#pragma unroll
        for (int M = 0; M < 4; ++M) st4(x_out + (size_t)n * C64 + 16 * M + 4 * g, y[M] + x[M]);
      }
    }
  }
}
// tile u of a wave = point tile `first + (u % NTL) * inner + (u / NTL) * outer` of a [N, 64] array, slot -> point
// through `perm` when given (the visibility order of the view attention), else identity
template <int NTL>  // tiles per iteration: FF_NTL in split mode, 1 in fp32 mode
struct FfTilePt {
  const int* perm;
  long long N, first, inner, outer;
  __device__ __forceinline__ void operator()(long long u, int j, long long& n, bool& ok) const {
    const long long tile = first + (u % NTL) * inner + (u / NTL) * outer;
    const long long slot = tile * 16 + j, sc = slot < N ? slot : N - 1;
    ok = slot < N;
    n = perm != nullptr ? (long long)perm[sc] : sc;
  }
};
// tiles a wave walks: u < nu  <=>  first + (u / NTL) * outer < ntiles  (a group is cut at its first tile)
__device__ __forceinline__ long long ff_tile_count(long long ntiles, long long first, long long outer) {
  return first < ntiles ? ((ntiles - first + outer - 1) / outer) * FF_NTL : 0;
}

// ------------------------------------------------------------------------------------ q positional MLP
// input = [q (64) | emb_pt (64) | emb_ray of the point's ray (64)] = 192 channels (k_posenc); LDS: W0 [64][196] | W2 [64][68] | b0 | b2
#define QP_LD0 196
constexpr int qp_lds_floats(bool spl) {
  return wimg_floats(spl, 64, 12, QP_LD0) + wimg_floats(spl, 64, 4, LDW) + 128;
}

struct QpLds {
  float *W0, *W2, *b0, *b2;
};
template <int SPL>
__device__ __forceinline__ QpLds qp_lds() {
  QpLds L;
  L.W0 = smem;
  L.W2 = L.W0 + wimg_floats(SPL, 64, 12, QP_LD0);
  L.b0 = L.W2 + wimg_floats(SPL, 64, 4, LDW);
  L.b2 = L.b0 + 64;
  return L;
}
template <int SPL, int TPB>
struct QpStager {  // (see FfStager)
  WStage<SPL, 64, 12, TPB> s0;
  WStage<SPL, 64, 4, TPB> s2;
  VecStage<64, TPB> v0, v2;
  __device__ __forceinline__ void load(const gpn_linear_t& fc0, const gpn_linear_t& fc2, int tid) {
    s0.load(fc0.w, 64, 190, tid), s2.load(fc2.w, 64, 64, tid);
    v0.load(fc0.b, 64, tid), v2.load(fc2.b, 64, tid);
  }
  __device__ __forceinline__ void store(const QpLds& L, int tid) const {
    s0.store(L.W0, QP_LD0, tid), s2.store(L.W2, LDW, tid);
    v0.store(L.b0, tid), v2.store(L.b2, tid);
  }
};
template <int SPL, int TPB>
__device__ __forceinline__ void qp_stage(const QpLds& L, const gpn_linear_t& fc0, const gpn_linear_t& fc2, int tid) {
  QpStager<SPL, TPB> st;
  st.load(fc0, fc2, tid);
  st.store(L, tid);
}
// the MLP on a wave's tiles u = 0 .. nu-1 (`pt` as in ff_run); q_in and q_out may be the same array
template <int SPL, class PT>
__device__ __forceinline__ void qp_run(const QpLds& L, const float* q_in, const float* __restrict__ emb_pt,
                                       const float* __restrict__ emb_ray, int S, float* q_out, long long nu, int lane,
                                       PT&& pt) {
  const int j = lane & 15, g = lane >> 4;
  for (long long u = 0; u < nu; ++u) {
    // (prefetching the next tile's 48 input registers was measured 6 % slower: the copies cost more
    // than the exposed latency)
    long long n;
    bool valid;
    pt(u, j, n, valid);
    f4 x[12], h[4], y[4];
#pragma unroll
    for (int T = 0; T < 4; ++T) x[T] = ld4(q_in + (size_t)n * C64 + 16 * T + 4 * g);
#pragma unroll
    for (int T = 0; T < 4; ++T) x[4 + T] = ld4(emb_pt + (size_t)n * GPN_EMB_HALF + 16 * T + 4 * g);
    const int ray = (int)n / S;  // N fits an int (C-ABI)
#pragma unroll
    for (int T = 0; T < 4; ++T) x[8 + T] = ld4(emb_ray + (size_t)ray * GPN_EMB_HALF + 16 * T + 4 * g);
#pragma unroll
    for (int M = 0; M < 4; ++M) h[M] = ld4(L.b0 + 16 * M + 4 * g);
    lin<SPL, 12, 4>(L.W0, QP_LD0, x, h, lane);
#pragma unroll
    for (int M = 0; M < 4; ++M) {
      h[M] = f4_relu_t<(SPL != 0)>(h[M]);
      y[M] = ld4(L.b2 + 16 * M + 4 * g);
    }
    lin<SPL, 4, 4>(L.W2, LDW, h, y, lane);
    if (valid) {
#pragma unroll
      for (int M = 0; M < 4; ++M) st4(q_out + (size_t)n * C64 + 16 * M + 4 * g, y[M]);
    }
  }
}

// second phase of an attention kernel (ff_out != nullptr): the layer's feed-forward block on the tiles this
// workgroup has just written, after re-staging LDS with the FF weights
template <int SPL, int TPB, class PT>
__device__ __forceinline__ void ff_phase(const float* __restrict__ norm_w, const float* __restrict__ norm_b,
                                         const gpn_linear_t& fc1, const gpn_linear_t& fc2, const float* x, float* ff_out,
                                         long long nu, int tid, PT&& pt) {
  const int lane = tid & 63;
  const FfLds F = ff_lds<SPL>();
  f4 xpre[FF_NTL][4];
  // this wave's own stores of x are in program order before these loads
  if constexpr (SPL != 0)  // ff_run<REV> starts at the last iteration
    ff_prefetch(xpre, x, nu > 0 ? (nu + FF_NTL - 1) / FF_NTL * FF_NTL - FF_NTL : 0, lane, pt);
  FfStager<SPL, TPB> st;
  st.load(norm_w, norm_b, fc1, fc2, tid);  // in flight across the barrier
  __syncthreads();  // every wave is done with the attention weights / keys / values in LDS
  st.store(F, tid);
  __syncthreads();
  ff_run<SPL, true>(F, xpre, x, ff_out, nu, lane, pt);
}

// third phase of k_view_attn on even layers (qp_out != nullptr): q = q_fcs[l](cat(q, emb_pt, emb_ray)) on the
// same tiles, LDS re-staged once more
template <int SPL, int TPB, class PT>
__device__ __forceinline__ void qp_phase(const gpn_linear_t& fc0, const gpn_linear_t& fc2, const float* q,
                                         const float* __restrict__ emb_pt, const float* __restrict__ emb_ray, int S,
                                         float* qp_out, long long nu, int tid, PT&& pt) {
  const QpLds Q = qp_lds<SPL>();
  QpStager<SPL, TPB> st;
  st.load(fc0, fc2, tid);  // in flight across the barrier
  __syncthreads();  // every wave is done with the feed-forward weights
  st.store(Q, tid);
  __syncthreads();
  qp_run<SPL>(Q, q, emb_pt, emb_ray, S, qp_out, nu, tid & 63, pt);
}

template <int SPL, bool FOLD, bool TILED = false>
__global__ void __launch_bounds__(VA_TPB)
k_view_attn(gpn_view_layer_t Lw, const float* q_in, const float* __restrict__ feat,
            const float* __restrict__ ray_diff, const float* __restrict__ mask,
            const int* __restrict__ perm, long long N, int V, float* q_out,
            float* __restrict__ hidden, int hidden_ld, float* __restrict__ stats, int stats_ld,
            int layout, float* ff_out, gpn_linear_t qfc0, gpn_linear_t qfc2, const float* __restrict__ emb_pt,
            const float* __restrict__ emb_ray, int S, float* qp_out) {
  // ff_out != nullptr: the layer's feed-forward block runs as a second phase of this launch, on the tiles
  // this workgroup has just written (q_out -> ff_out; LDS re-staged with the FF weights in between): one
  // launch, one ramp and one tail instead of two, and the rows come back from L2.
  // layout & 1: `ray_diff` is the tile-major image written by k_rd_pack ([tile][view][component][16
  // points of the tile, in slot order]): the 64 floats of a (tile, view) pair are ONE contiguous 256-byte
  // run (two whole lines) instead of 16 sixteen-byte entries in 16 different 64-byte sectors.
  // layout & 2: `feat` is tile-major too ([tile][view][16 slots][64], written by the fused projector):
  // the 16 rows of a pair are one sequential 4 KB block instead of 16 random rows.
  const int rd_packed = layout & 1, feat_tiled = layout & 2;
  const ViewLds L = view_lds<SPL>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform (tile loop, LDS-DMA base)
  STAMP(0);
  // Everything the first tile needs from global memory is requested BEFORE / WHILE the weights are
  // staged: slot -> point lookups first, then the weight loads, then the first two ring fills and
  // the q rows; their latencies run under the LDS stores and the barrier instead of after it.
  const int j = lane & 15, g = lane >> 4;
  const long long ntiles = (N + 15) >> 4;
  const long long tstride = (long long)gridDim.x * (VA_TPB / 64);
  const unsigned all_views = V >= 32 ? 0xffffffffu : ((1u << V) - 1u);
  const int* vbits = perm != nullptr ? perm + N : nullptr;  // gpn_vis_order: bit masks in slot order

  // The tile loop is software-pipelined over tiles: while a tile computes, the point index and
  // visibility bits (top of the tile), the q rows (before the view loop) and the first live view
  // (in the last view iteration) of the wave's NEXT tile are already being fetched, so a tile
  // starts without a chain of dependent global loads.
  // slot -> point: identity, or the visibility-sorted order of gpn_vis_order (points with the same
  // set of visible views share tiles, which makes the dead-view skip below effective).
  auto point_of = [&](long long tile_, int& n_, unsigned& vis_) {
    const long long slot = tile_ * 16 + j, sc = slot < N ? slot : N - 1;
    if (perm != nullptr) {
      n_ = perm[sc];
      vis_ = (unsigned)vbits[sc];
    } else {
      n_ = (int)sc;
      unsigned b = 0;  // lane (j, g) reads views g, g+4, ... of point j; all loads independent
#pragma unroll
      for (int k = 0; k < GPN_MAX_VIEWS / 4; ++k) {
        const int v = 4 * k + g, vc = v < V ? v : V - 1;
        const float m = mask[(size_t)sc * V + vc];
        b |= (v < V && m != 0.0f ? 1u : 0u) << (v & 31);
      }
      vis_ = or_over_g(b);
    }
  };
  // live views of a tile (wave-uniform): visible for some point, or needed by a point that has no
  // visible view at all (its softmax is uniform over all V views)
  auto live_of = [&](unsigned vis_) {
    const unsigned m = or_over_row(vis_ != 0 ? vis_ : all_views);
    return (unsigned)__builtin_amdgcn_readfirstlane(m);
  };

  // ---- per-wave ring of two LDS slots filled by LDS-DMA (global_load_lds): the feature row and the
  // ray_diff entry of a (tile, view) pair are requested TWO view iterations before they are used,
  // without holding registers meanwhile (at two waves per SIMD one pair in flight per wave does not
  // cover the HBM latency: the kernel was measured stalled ~70 % of its cycles).  A lane reads back
  // exactly the bytes it requested (slot + k KiB + 16 lane), so no barrier is involved; the only
  // ordering is this wave's own s_waitcnt vmcnt before the ds_read (MI355X_MICROARCH.md item 7).
  // Every view iteration issues exactly VA_DMA_OPS DMA instructions, so the wait count is static.
  const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane(wave);
  char* ring = reinterpret_cast<char*>(smem + view_lds_floats(SPL)) + wave_u * (2 * VA_SLOT_BYTES);
  const unsigned ring_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ring;  // LDS byte address
  // issue the DMA of pair (point nn of lane j, view vv) into slot `sl`.  Every DMA instruction
  // fetches WHOLE 128-byte lines (8 lanes x 16 B per line, 8 points per instruction): two
  // instructions that each take half of every line of a row leave the second one waiting in the
  // L1 for the first one's fill (measured: TCP_PENDING_STALL 62 % of the kernel's cycles).
  // Instruction k = 2*h + half covers line h of the rows of points 8*half .. 8*half+7; lane
  // (jj = lane>>3, r = lane&7) fetches piece c = (r - s) & 7 of point 8*half + jj, s = (jj>>1) +
  // 4*half: the rotation makes the read-back below bank-conflict free.
  const int jj8 = lane >> 3, r8 = lane & 7;
  const int cA = (r8 - (jj8 >> 1)) & 7, cB = (r8 - (jj8 >> 1) - 4) & 7;
  // TILED (layout == 3, the fused path): the 16 rows and the ray_diff run of a pair sit at a wave-uniform
  // base, so the addresses are a scalar base (scalar unit: free beside the MFMAs) plus per-lane offsets
  // that never change -- no vector address arithmetic and no cross-lane reads per pair (an fp32 MFMA
  // kernel pays every vector instruction in full)
  const unsigned offA = (unsigned)(jj8 * 256 + cA * 16), offB = (unsigned)((8 + jj8) * 256 + cB * 16);
  const unsigned offR = (unsigned)lane * 4u;
  auto dma_pair = [&](int nn, long long tile_, int vv, unsigned sl) {
    if constexpr (TILED) {
      const unsigned tv = (unsigned)__builtin_amdgcn_readfirstlane((int)(tile_ * V + vv));
      const char* sf = reinterpret_cast<const char*>(feat) + (size_t)tv * (16 * C64 * 4);
      const char* sr = reinterpret_cast<const char*>(ray_diff) + (size_t)tv * 256;
      const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane(ring_lds + sl * VA_SLOT_BYTES);
      const unsigned m0 = base, m1 = base + 1024, m2 = base + 2048 - 128, m3 = base + 3072 - 128, m4 = base + 4096;
      unsigned keep;
      asm volatile(
          "s_waitcnt lgkmcnt(0)\n\t"  // the slot's previous contents have been read into registers
          "s_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4 " VA_DMA_NT "\n\t"
          "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4 " VA_DMA_NT "\n\t"
          "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4 offset:128 " VA_DMA_NT "\n\t"
          "s_mov_b32 m0, %9\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4 offset:128 " VA_DMA_NT "\n\t"
          "s_mov_b32 m0, %10\n\ts_nop 0\n\tglobal_load_lds_dword %3, %5\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(offA), "v"(offB), "v"(offR), "s"(sf), "s"(sr), "s"(m0), "s"(m1), "s"(m2), "s"(m3), "s"(m4)
          : "memory");
      return;
    }
    const int nA = __shfl(nn, jj8), nB = __shfl(nn, 8 + jj8);
    const size_t rowA = feat_tiled ? ((size_t)tile_ * V + vv) * 16 + jj8 : (size_t)nA * V + vv;
    const size_t rowB = feat_tiled ? ((size_t)tile_ * V + vv) * 16 + 8 + jj8 : (size_t)nB * V + vv;
    const char* fa = reinterpret_cast<const char*>(feat) + rowA * (C64 * 4) + cA * 16;
    const char* fb = reinterpret_cast<const char*>(feat) + rowB * (C64 * 4) + cB * 16;
    const char* ra = reinterpret_cast<const char*>(ray_diff) +
                     (rd_packed ? ((size_t)tile_ * V + vv) * 256 + lane * 4 : ((size_t)nn * V + vv) * 16 + g * 4);
    // instruction offsets apply to the global AND the LDS address: M0 is compensated per piece
    const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane(ring_lds + sl * VA_SLOT_BYTES);
    const unsigned m0 = base, m1 = base + 1024, m2 = base + 2048 - 128, m3 = base + 3072 - 128, m4 = base + 4096;
    unsigned keep;
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n\t"  // the slot's previous contents have been read into registers
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off " VA_DMA_NT "\n\t"
        "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off " VA_DMA_NT "\n\t"
        "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:128 " VA_DMA_NT "\n\t"
        "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off offset:128 " VA_DMA_NT "\n\t"
        "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dword %3, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(fa), "v"(fb), "v"(ra), "s"(m0), "s"(m1), "s"(m2), "s"(m3), "s"(m4)
        : "memory");
  };
  // read a slot back (after the covering wait): unit (line h, piece c) of this lane's point j
  // landed at piece k = 2h + (j>>3), lane (j&7)*8 + ((c + s_j) & 7), s_j = ((j&7)>>1) + 4*(j>>3)
  const int sj = ((j & 7) >> 1) + 4 * (j >> 3);
  int rb_off[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    // split mode: f[2P + hm] = unit (line P, piece 4 hm + g); fp32 mode: f[T] = (line T>>1, piece 4(T&1) + g)
    const int h = k >> 1, c = 4 * (k & 1) + g;
    rb_off[k] = (2 * h + (j >> 3)) * 1024 + ((j & 7) * 8 + ((c + sj) & 7)) * 16;
  }
  auto read_slot = [&](unsigned sl, f4* f, float& rd_) {
    const char* sp = ring + sl * VA_SLOT_BYTES;
#pragma unroll
    for (int k = 0; k < 4; ++k) f[k] = *reinterpret_cast<const f4*>(sp + rb_off[k]);
    rd_ = *reinterpret_cast<const float*>(sp + 4096 + lane * 4);
  };

  long long tile = (long long)blockIdx.x * (VA_TPB / 64) + wave;
  const bool has_tile = tile < ntiles;  // (a wave without a tile still takes part in the staging)
  // points and visibility of the current tile and of the next two; the third one ahead is in flight
  int n = 0, n1, n2, n3;  // N fits an int (C-ABI)
  unsigned vis = 0, vis1, vis2, vis3;
  if (has_tile) point_of(tile, n, vis);
  n1 = n2 = n3 = n;
  vis1 = vis2 = vis3 = vis;
  if (tile + tstride < ntiles) point_of(tile + tstride, n1, vis1);
  if (tile + 2 * tstride < ntiles) point_of(tile + 2 * tstride, n2, vis2);
  unsigned live = live_of(vis);
  unsigned live1 = tile + tstride < ntiles ? live_of(vis1) : 0u;
  unsigned live2 = tile + 2 * tstride < ntiles ? live_of(vis2) : 0u;

  // the pair after next in this wave's sequence of (tile, view) pairs, given the views of the
  // current tile that remain after the current one (all wave-uniform): tile index 0 / 1 / 2 relative
  // to the current tile, or -1 past the end
  auto pair_plus2 = [&](unsigned rest, int& t2, int& v2) {
    const unsigned rest1 = rest & (rest - 1);
    t2 = -1, v2 = 0;
    if (rest1) {                       // two or more views remain in this tile
      t2 = 0, v2 = __builtin_ctz(rest1);
    } else if (rest) {                 // one remains: then the first view of the next tile
      if (live1) t2 = 1, v2 = __builtin_ctz(live1);
    } else if (live1) {                // none remains: the next tile's second view, or the first of the one after
      const unsigned l1r = live1 & (live1 - 1);
      if (l1r) t2 = 1, v2 = __builtin_ctz(l1r);
      else if (live2) t2 = 2, v2 = __builtin_ctz(live2);
    }
  };
  unsigned par = 0;  // slot of the current pair
  // (re)start the ring at the first view of the current tile: pair 0 -> slot 0, pair 1 -> slot 1
  auto ring_start = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing may still land in the slots
    par = 0;
    const int v0 = __builtin_ctz(live);
    dma_pair(n, tile, v0, 0);
    const unsigned rest = live & (live - 1);
    if (rest) dma_pair(n, tile, __builtin_ctz(rest), 1);
    else if (live1) dma_pair(n1, tile + tstride, __builtin_ctz(live1), 1);
    else dma_pair(n, tile, v0, 1);  // no further pair: keep the op count
  };
  f4 xq[4];
  {  // two-phase staging (gnt_mfma.h): every global load first, then the LDS images
    WStage<SPL, 64, 4, VA_TPB> sq, sk, sv, so;
    WStage<SPL, 16, 6, VA_TPB> saf;
    WStage<SPL, 64, 1, VA_TPB> sp2, sa2;
    WStage<SPL, 16, 4, VA_TPB> sa0;
    RowStage<16, 1, VA_TPB> sp0;
    VecStage<16, VA_TPB> vc0, vbp0, vba0;
    VecStage<64, VA_TPB> vbp2, vba2, vbo, vga, vbe;
    sq.load(Lw.q.w, 64, 64, tid);
    if constexpr (FOLD) {  // Wv slot holds Wv Wk; k itself is never formed
      sv.load(Lw.folded + FOLD_VK, 64, 64, tid);
      saf.load(Lw.folded + FOLD_AF, 16, 96, tid);
      vc0.load(Lw.folded + FOLD_C0, 16, tid);
    } else {
      sk.load(Lw.k.w, 64, 64, tid);
      sv.load(Lw.v.w, 64, 64, tid);
    }
    so.load(Lw.out.w, 64, 64, tid);
    sp2.load(Lw.pos2.w, 64, 8, tid);
    sa0.load(Lw.attn0.w, 8, 64, tid);
    sa2.load(Lw.attn2.w, 64, 8, tid);
    sp0.load(Lw.pos0.w, 8, 4, tid);
    vbp0.load(Lw.pos0.b, 8, tid), vbp2.load(Lw.pos2.b, 64, tid), vba0.load(Lw.attn0.b, 8, tid);
    vba2.load(Lw.attn2.b, 64, tid), vbo.load(Lw.out.b, 64, tid);
    vga.load(Lw.attn_norm_w, 64, tid), vbe.load(Lw.attn_norm_b, 64, tid);
    if (has_tile) {
      ring_start();
#pragma unroll
      for (int T = 0; T < 4; ++T) xq[T] = ld4(q_in + (size_t)n * C64 + 16 * T + 4 * g);
    }
    sq.store(L.Wq, LDW, tid);
    if constexpr (FOLD) {
      sv.store(L.Wv, LDW, tid);
      saf.store(L.AF, VA_LDAF, tid);
      vc0.store(L.c0, tid);
    } else {
      sk.store(L.Wk, LDW, tid);
      sv.store(L.Wv, LDW, tid);
    }
    so.store(L.Wo, LDW, tid);
    sp2.store(L.P2, VA_LD8, tid);
    sa0.store(L.A0, LDW, tid);
    sa2.store(L.A2, VA_LD8, tid);
    sp0.store(L.P0, 4, tid);
    vbp0.store(L.bP0, tid), vbp2.store(L.bP2, tid), vba0.store(L.bA0, tid), vba2.store(L.bA2, tid);
    STAMP(1);
    vbo.store(L.bO, tid), vga.store(L.gam, tid), vbe.store(L.bet, tid);
  }
  __syncthreads();

  STAMP(2);
  if (!has_tile && ff_out == nullptr) return;
  auto attention_phase = [&]() {
  const float p0w = L.P0[j * 4 + g];  // A operand of the 4 -> 8 layer: W[i = j][k = g]

  unsigned long long redo_lanes = 0;  // lanes (points) of the current tile to redo in the online form
  // a point = the 4 lanes (j, g = 0..3): all of them redo if any of their channels overflowed
  auto ovf_point = [&](bool o) {
    return or_over_g(o ? 1u : 0u) != 0;
  };
  f2 l2e = (f2){LOG2E_F, LOG2E_F};
  asm volatile("" : "+v"(l2e));  // keep log2(e) in registers: a literal operand blocks v_pk_fma_f32
  VA_T0();
  for (;;) {
    VA_MARK(5);
    VA_COUNT(7);
    const bool valid = tile * 16 + j < N;
    const bool has_next = tile + tstride < ntiles;  // wave-uniform
    // points of the tile three ahead (consumed when it has become the tile two ahead)
    const bool online = redo_lanes != 0;            // wave-uniform: second pass over this tile
    if (!online) {
      n3 = n2, vis3 = vis2;
      if (tile + 3 * tstride < ntiles) point_of(tile + 3 * tstride, n3, vis3);
    }

    f4 qp[4];
    {
      f4 qn[4];
      layernorm_T(xq, qn, L.gam, L.bet, 1e-6f, lane);
#pragma unroll
      for (int M = 0; M < 4; ++M) qp[M] = f4_zero();
      lin<SPL, 4, 4>(L.Wq, LDW, qn, qp, lane);
    }
    f4 aq = f4_zero();  // folded mode: A0 b_pos2 + b_attn0 - A0 q  (the per-point part of attn_fc[0])
    if constexpr (FOLD) {
      f4 t = f4_zero();
      lin<SPL, 4, 1>(L.A0, LDW, qp, &t, lane);
      aq = ld4(L.c0 + 4 * g) - t;
    }
    // next tile's q rows (the residual input of THIS tile is re-read at the end instead of being
    // held in 16 registers through the view loop)
    f4 xres[4];  // FOLD: the residual input stays in registers (one L1 request per 64 B saved)
    if constexpr (FOLD) {  // (the other variants have no registers to spare here: they fetch it after the loop)
#pragma unroll
      for (int T = 0; T < 4; ++T) xres[T] = xq[T];
#pragma unroll
      for (int T = 0; T < 4; ++T) xq[T] = ld4(q_in + (size_t)n1 * C64 + 16 * T + 4 * g);
    }

    f4 mrun[4], lrun[4], acc[4];
    // A masked view contributes exp(-1e9 - max) == 0 exactly as soon as the point has one visible
    // view, so a (tile, view) whose 16 points are all masked-but-visible-elsewhere is skipped whole.
    // Softmax over the views, per channel (:84-87).  It is shift invariant, so instead of a running
    // maximum (compare / select / rescale per value) the exponentials are taken relative to the
    // logits of the point's FIRST VISIBLE view: per value one fma, one v_exp_f32, one add, one fma.
    // The result of a point does not depend on its tile mates (masked views add exact zeros), so
    // any point order and any chunking give the same bits.  Three instantiations of the view loop:
    //   HOMOG   every point of the tile sees exactly the tile's live views (the common case after
    //           gpn_vis_order): no mask handling at all
    //   MASKED  mixed tile: masked_fill(-1e9) and a per-lane reference
    //   ONLINE  running-maximum form, only for points whose logits spread over more than the fp32
    //           exponent range (> ~87 between two views): their sum overflows in the forms above,
    //           which is detected after the loop; the tile is then run a second time in this form
    //           and only those points are stored from it.
    STAMP(3);
    const unsigned live0 = live;
    const int vfirst = __builtin_ctz(live0);
    // lanes whose results this pass stores (a redo pass: the overflowed points only)
    const bool lane_ok = valid && (!online || ((redo_lanes >> lane) & 1ull) != 0);
    const bool homog = __all(vis == live0);
    const int vref = vis != 0 ? __builtin_ctz(vis) : vfirst;  // MASKED: this point's reference view
#pragma unroll
    for (int M = 0; M < 4; ++M) {
      mrun[M] = online ? f4_splat(-INFINITY) : f4_zero();  // fast forms: -log2(e) * reference logit
      lrun[M] = f4_zero();
      acc[M] = f4_zero();
    }

    auto view_loop = [&](auto mode_tag) {
      constexpr int MODE = decltype(mode_tag)::value;
      while (live) {
        if (MODE == VM_HOMOG) { VA_MARK(4); VA_COUNT(6); }
        const int v = __builtin_ctz(live);
        live &= live - 1;
        // This pair's slot was filled by a DMA issued two iterations ago.  Younger vector-memory
        // operations of this wave in the HOMOG form: the hidden-layer stores of the two iterations
        // since (1 each, always issued: every lane is visible and lane 0 of a tile is a valid point)
        // and the 5 DMA pieces of the next pair; the other forms do not count, they wait for everything.
        if (MODE == VM_HOMOG && hidden != nullptr) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VA_DMA_OPS + 2) : "memory");
        else if (MODE == VM_HOMOG) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VA_DMA_OPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        f4 fcur[4];
        float rd;
        read_slot(par, fcur, rd);
        if (MODE == VM_HOMOG) VA_MARK(1);
        {  // refill the slot with the pair after next (or, past the end, any valid pair: fixed op count)
          int t2, v2;
          pair_plus2(live, t2, v2);
          const int nn = t2 == 1 ? n1 : (t2 == 2 ? n2 : n);
          dma_pair(nn, tile + (t2 > 0 ? t2 : 0) * tstride, t2 < 0 ? v : v2, par);
          par ^= 1u;
        }
        if (MODE == VM_HOMOG) VA_MARK(2);
        f4 vp[4], a[4];  // vp = v + pos (the value that is attended over)
#pragma unroll
        for (int M = 0; M < 4; ++M) {
          vp[M] = ld4(L.bP2 + 16 * M + 4 * g);
          a[M] = ld4(L.bA2 + 16 * M + 4 * g);
        }
        // pos hidden layer: 4 -> 8 (ReLU), one fp32 MFMA
        f4 ph = f4_relu_t<(SPL != 0)>(mfma16(p0w, rd, ld4(L.bP0 + 4 * g)));
        f4 ah;  // post-ReLU hidden layer of attn_fc: rows 4g..4g+3 of the (zero-padded) 16
        if constexpr (FOLD && SPL != 0) {
          // v + pos = (Wv Wk) f + P2 h + b ; attn hidden = (A0 Wk) f + (A0 P2) h + aq : every product
          // reads the split of f and h made once here
          XS xs[3];
          xs[0] = feat_pair(fcur[0], fcur[1]);
          xs[1] = feat_pair(fcur[2], fcur[3]);
          xs[2] = split8(ph, f4_zero());
          lin_Sx<4, 4, true, bf_terms(SPL)>(L.Wv, xs, vp, lane);
          lin_Sx<1, 4, true, bf_terms(SPL)>(L.P2, xs + 2, vp, lane);
          ah = aq;
          lin_Sx<6, 1, true, bf_terms(SPL)>(L.AF, xs, &ah, lane);
          ah = f4_relu_t<(SPL != 0)>(ah);
          lin<SPL, 1, 4>(L.A2, VA_LD8, &ah, a, lane);
        } else if constexpr (FOLD) {
          // the same two chains on fp32 MFMAs: 117 instead of 177 MFMAs per (tile, view) -- k is never formed
          const f4 xin[5] = {fcur[0], fcur[1], fcur[2], fcur[3], ph};  // K = 64 feature channels + 16 (8) hidden
          lin_T<4, 4>(L.Wv, LDW, fcur, vp, lane);
          lin_T<1, 4>(L.P2, VA_LD8, &ph, vp, lane);
          ah = aq;
          lin_T<5, 1>(L.AF, VA_LDAF, xin, &ah, lane);
          ah = f4_relu_t<(SPL != 0)>(ah);
          lin_T<1, 4>(L.A2, VA_LD8, &ah, a, lane);
        } else {
          f4 k[4];
#pragma unroll
          for (int M = 0; M < 4; ++M) k[M] = f4_zero();
          if constexpr (SPL != 0) {
            const XS xf[2] = {feat_pair(fcur[0], fcur[1]), feat_pair(fcur[2], fcur[3])};
            lin_Sx<4, 4, true, bf_terms(SPL)>(L.Wk, xf, k, lane);
          } else {
            lin_T<4, 4>(L.Wk, LDW, fcur, k, lane);   // k = k_fc(feat)
          }
          f4 pos[4];
#pragma unroll
          for (int M = 0; M < 4; ++M) pos[M] = vp[M];
          lin<SPL, 1, 4>(L.P2, VA_LD8, &ph, pos, lane);  // pos = pos_fc(ray_diff): 4 -> 8 -> 64
          // attn = attn_fc(k - q + pos): 64 -> 8 (ReLU) -> 64
          f4 ain[4];
#pragma unroll
          for (int M = 0; M < 4; ++M) ain[M] = k[M] - qp[M] + pos[M];
#pragma unroll
          for (int M = 0; M < 4; ++M) vp[M] = f4_zero();
          lin<SPL, 4, 4>(L.Wv, LDW, k, vp, lane);     // v = v_fc(k)   (reference quirk, :77)
#pragma unroll
          for (int M = 0; M < 4; ++M) vp[M] = vp[M] + pos[M];
          ah = ld4(L.bA0 + 4 * g);
          lin<SPL, 4, 1>(L.A0, LDW, ain, &ah, lane);
          ah = f4_relu_t<(SPL != 0)>(ah);
          lin<SPL, 1, 4>(L.A2, VA_LD8, &ah, a, lane);
        }
        if (MODE == VM_HOMOG) VA_MARK(3);
        const bool vis_v = MODE == VM_HOMOG || ((vis >> v) & 1u) != 0;
        // side output for the semantic pass: the 8 hidden values (lanes g < 2), from which
        // gpn_sem_aggregate rebuilds the 64 logits; masked pairs are never read back
        if (hidden != nullptr && lane_ok && vis_v && g < 2)
          st4(hidden + ((size_t)n * V + v) * hidden_ld + 4 * g, ah);
        if constexpr (MODE != VM_HOMOG) {
          if (!vis_v) {
#pragma unroll
            for (int M = 0; M < 4; ++M) a[M] = f4_splat(-1e9f);  // masked_fill (:83)
            // its weight is exactly 0 once the point has a visible view; the feature row of such a
            // pair may never have been stored (gpn_rgbfeat with a mask): keep 0 * garbage out of the sum
            if (vis != 0) {
#pragma unroll
              for (int M = 0; M < 4; ++M) vp[M] = f4_zero();
            }
          }
        }
        if constexpr (MODE == VM_HOMOG) {
          if (v == vfirst) {
#pragma unroll
            for (int M = 0; M < 4; ++M) mrun[M] = a[M] * f4_splat(-LOG2E_F);
          }
        } else if constexpr (MODE == VM_MASKED) {
          if (v == vref) {
#pragma unroll
            for (int M = 0; M < 4; ++M) mrun[M] = a[M] * f4_splat(-LOG2E_F);
          }
        }
        if constexpr (MODE != VM_ONLINE) {
#pragma unroll
          for (int M = 0; M < 4; ++M) {
            f4 d, p;
            d.lo = a[M].lo * l2e + mrun[M].lo;
            d.hi = a[M].hi * l2e + mrun[M].hi;
#pragma unroll
            for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(d[r]);
            lrun[M] = lrun[M] + p;
            acc[M] = acc[M] + vp[M] * p;
          }
        } else {
          // one of the two rescale factors is always exp(0) = 1, so a single exponential of
          // -|a - max| serves both
#pragma unroll
          for (int M = 0; M < 4; ++M)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float d = a[M][r] - mrun[M][r];
              const float e = __expf(-fabsf(d));
              const bool up = d > 0.0f;  // new maximum
              const float alpha = up ? e : 1.0f, p = up ? 1.0f : e;
              lrun[M][r] = lrun[M][r] * alpha + p;
              acc[M][r] = acc[M][r] * alpha + vp[M][r] * p;
              mrun[M][r] = up ? a[M][r] : mrun[M][r];
            }
        }
      }
    };
    STAMP(4);
    VA_MARK(0);
    bool store_ok = valid;
    unsigned long long redo_next = 0;
    if (online) {
      view_loop(std::integral_constant<int, VM_ONLINE>{});
      store_ok = lane_ok;
    } else {
      if (homog) view_loop(std::integral_constant<int, VM_HOMOG>{});
      else view_loop(std::integral_constant<int, VM_MASKED>{});
      float lm = 0.f;
#pragma unroll
      for (int M = 0; M < 4; ++M) lm = max3_nc(lm, max_nc(lrun[M].x, lrun[M].y), max_nc(lrun[M].z, lrun[M].w));
      const bool ovf = !(lm < 1e30f);  // some channel's sum overflowed (or is NaN)
      redo_next = __ballot(ovf_point(ovf));
      store_ok = valid && ((redo_next >> lane) & 1ull) == 0;
    }
    f4 xr[4], xs[4], y[4], rl[4];
#pragma unroll
    for (int T = 0; T < 4; ++T) {
      if constexpr (FOLD) xr[T] = xres[T];
      else xr[T] = ld4(q_in + (size_t)n * C64 + 16 * T + 4 * g);  // residual input
    }
    if constexpr (!FOLD) {
#pragma unroll
      for (int T = 0; T < 4; ++T) xq[T] = ld4(q_in + (size_t)n1 * C64 + 16 * T + 4 * g);
    }
#pragma unroll
    for (int M = 0; M < 4; ++M) {
#pragma unroll
      for (int r = 0; r < 4; ++r) rl[M][r] = __builtin_amdgcn_rcpf(lrun[M][r]);  // v_rcp_f32, 1 ulp
      xs[M] = acc[M] * rl[M];
      y[M] = ld4(L.bO + 16 * M + 4 * g);
    }
    lin<SPL, 4, 4>(L.Wo, LDW, xs, y, lane);
    if (store_ok) {
#pragma unroll
      for (int M = 0; M < 4; ++M) st4(q_out + (size_t)n * C64 + 16 * M + 4 * g, y[M] + xr[M]);
      if (stats != nullptr) {  // log-sum-exp over the views: softmax weight = exp(logit - lse)
#pragma unroll
        for (int M = 0; M < 4; ++M) {
          f4 lse;  // the sum is >= 1 in both forms: bare v_log_f32 (log2), no denormal handling
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float l2 = __builtin_amdgcn_logf(lrun[M][r]);
            lse[r] = online ? mrun[M][r] + l2 * LN2_F : (l2 - mrun[M][r]) * LN2_F;
          }
          st4(stats + (size_t)n * stats_ld + 16 * M + 4 * g, lse);
        }
      }
    }
    if (redo_next != 0) {
      // run this tile again in the online form for the overflowed points.  Their rows of q_in are
      // untouched (nothing of theirs has been stored); the other lanes recompute from whatever is
      // there now and are not stored.
      redo_lanes = redo_next;
      live = live0;
#pragma unroll
      for (int T = 0; T < 4; ++T) xq[T] = ld4(q_in + (size_t)n * C64 + 16 * T + 4 * g);
      ring_start();
      continue;
    }
    redo_lanes = 0;
    if (!has_next) break;
    tile += tstride;
    n = n1, vis = vis1, live = live1;
    n1 = n2, vis1 = vis2, live1 = live2;
    n2 = n3, vis2 = vis3;
    live2 = tile + 2 * tstride < ntiles ? live_of(vis2) : 0u;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last (unused) ring fills land before the wave ends
  VA_MARK(5);
  VA_END();
  };
  if (has_tile) attention_phase();
  STAMP(5);
  if (ff_out != nullptr) {  // (wave-uniform: a kernel argument)
    const long long tile0 = (long long)blockIdx.x * (VA_TPB / 64) + wave;
    const FfTilePt<SPL ? FF_NTL : 1> pt = {perm, N, tile0, tstride, SPL ? tstride * FF_NTL : tstride};
    const long long nu = SPL ? ff_tile_count(ntiles, tile0, tstride * FF_NTL)
                             : (tile0 < ntiles ? (ntiles - tile0 + tstride - 1) / tstride : 0);
    ff_phase<SPL, VA_TPB>(Lw.ff_norm_w, Lw.ff_norm_b, Lw.ff1, Lw.ff2, q_out, ff_out, nu, tid, pt);
    if (qp_out != nullptr)  // (only after the feed-forward phase; wave-uniform)
      qp_phase<SPL, VA_TPB>(qfc0, qfc2, ff_out, emb_pt, emb_ray, S, qp_out, nu, tid, pt);
  }
}

// ------------------------------------------------------------------------------------ feed-forward (kernel)
template <int SPL>
__global__ void __launch_bounds__(FF_TPB)
k_ff(const float* __restrict__ norm_w, const float* __restrict__ norm_b, gpn_linear_t fc1,
     gpn_linear_t fc2, const float* x_in, long long N, float* x_out) {
  const FfLds L = ff_lds<SPL>();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  STAMP(8);
  WAVE_T(0);
  const long long w0 = (long long)blockIdx.x * (FF_TPB / 64) + wave, stride = (long long)gridDim.x * (FF_TPB / 64);
  const long long ntiles = (N + 15) >> 4;
  // split mode: a wave takes FF_NTL consecutive tiles per iteration; the first input tiles are requested
  // before the weights (their HBM latency then runs under the staging instead of after the barrier)
  const FfTilePt<SPL ? FF_NTL : 1> pt = {nullptr, N, SPL ? w0 * FF_NTL : w0, 1, SPL ? stride * FF_NTL : stride};
  const long long nu = SPL ? ff_tile_count(ntiles, w0 * FF_NTL, stride * FF_NTL)
                           : (w0 < ntiles ? (ntiles - w0 + stride - 1) / stride : 0);
  f4 xpre[FF_NTL][4];
  if constexpr (SPL != 0) ff_prefetch(xpre, x_in, 0, lane, pt);
  ff_stage<SPL, FF_TPB>(L, norm_w, norm_b, fc1, fc2, tid);
  __syncthreads();
  STAMP(10);
  ff_run<SPL, false>(L, xpre, x_in, x_out, nu, lane, pt);
  STAMP(11);
  WAVE_T(1);
}

// Split-mode feed-forward on 32x32x16 MFMAs (gnt_mfma.h, "32x32x16 form"): one 32-point tile per wave
// iteration, 192 MFMAs of 32 cycles per tile -- the pipe time of the 16x16x32 kernel above at half the
// MFMA issue slots and 40 % fewer vector instructions.  The residual rides in the accumulator (y starts
// as x + b2), the hidden layer is produced and consumed in eight 32-wide chunks.  Same LDS budget:
// W1, W2 images 64 KB each.
// NOT the default (-DGPN_FF32=1 selects it): parity-green, but measured equal to k_ff<true> in a
// back-to-back microbench (52-57 us vs 51-55 us at N = 262144) and 0.7 % slower on the whole step.
// Ablations (tools/ffbench.py, results in DESIGN.md section 6): dropping the whole ReLU + operand
// split and every in-loop LDS weight read only takes 57 -> 50 us; the slope is 45 us per 262144
// points where the MFMA pipe time is 22 us, in both tilings.
#ifndef GPN_FF32
#define GPN_FF32 0
#endif

__device__ __forceinline__ f4 v16_get(const f16v& v, int b) {
  return (f4){v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3]};
}
__device__ __forceinline__ void v16_set(f16v& v, int b, f4 t) {
  v[4 * b] = t.x, v[4 * b + 1] = t.y, v[4 * b + 2] = t.z, v[4 * b + 3] = t.w;
}

// weight fragments of the first layer, hidden channels 32c .. 32c+31: [K-step s][hi, mid]
__device__ __forceinline__ void ff32_w1(const bf8* w1p, int c, bf8 (&w)[8]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    w[2 * s] = w1p[((c * 4 + s) * 2) * 64];
    w[2 * s + 1] = w1p[((c * 4 + s) * 2 + 1) * 64];
  }
}
// first layer, hidden channels 32c .. 32c+31 of the wave's 32 points: h = b1 + W1[32c.., :] . xn
__device__ __forceinline__ void ff32_fc1(const float* b1, int c, int gg, const bf8 (&w)[8], const XS (&xs)[4], f16v& h) {
#pragma unroll
  for (int b = 0; b < 4; ++b) v16_set(h, b, ld4(b1 + 32 * c + 8 * b + 4 * gg));
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    h = mfma_bf32(w[2 * s + 1], xs[s].hi, h);
    h = mfma_bf32(w[2 * s], xs[s].mid, h);
    h = mfma_bf32(w[2 * s], xs[s].hi, h);
  }
}

// one 32-point input tile (rows clamped to the last point)
__device__ __forceinline__ void ff32_load(const float* __restrict__ x_in, long long np, long long N, int gg, f4 (&x)[2][4]) {
  const long long npc = np < N ? np : N - 1;
#pragma unroll
  for (int M = 0; M < 2; ++M)
#pragma unroll
    for (int b = 0; b < 4; ++b) x[M][b] = ld4(x_in + (size_t)npc * C64 + 32 * M + 8 * b + 4 * gg);
}
// LayerNorm over the 64 channels of a point (32 registers of this lane + the partner lane gg ^ 1),
// split into the B fragments of the first layer
__device__ __forceinline__ void ff32_ln(const f4 (&x)[2][4], const float* gam, const float* bet, int gg, XS (&xs)[4]) {
  float sm = 0.f;
#pragma unroll
  for (int M = 0; M < 2; ++M)
#pragma unroll
    for (int b = 0; b < 4; ++b) sm += (x[M][b].x + x[M][b].y) + (x[M][b].z + x[M][b].w);
  sm += __shfl_xor(sm, 32);
  const float mean = sm * (1.0f / 64.0f);
  float sq = 0.f;
#pragma unroll
  for (int M = 0; M < 2; ++M)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const f4 d = x[M][b] - f4_splat(mean);
      sq += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
    }
  sq += __shfl_xor(sq, 32);
  const float rstd = __builtin_amdgcn_rsqf(sq * (1.0f / 64.0f) + 1e-6f);
#pragma unroll
  for (int M = 0; M < 2; ++M) {
    f4 xn[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int c = 32 * M + 8 * b + 4 * gg;
      xn[b] = (x[M][b] - f4_splat(mean)) * f4_splat(rstd) * ld4(gam + c) + ld4(bet + c);
    }
    xs[2 * M] = split8(xn[0], xn[1]);
    xs[2 * M + 1] = split8(xn[2], xn[3]);
  }
}

__global__ void __launch_bounds__(FF_TPB)
k_ff32(const float* __restrict__ norm_w, const float* __restrict__ norm_b, gpn_linear_t fc1,
       gpn_linear_t fc2, const float* __restrict__ x_in, long long N, float* __restrict__ x_out) {
  float* W1 = smem;
  float* W2 = W1 + 256 * 64;
  float* b1 = W2 + 64 * 256;
  float* b2 = b1 + 256;
  float* gam = b2 + 64;
  float* bet = gam + 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  {  // two-phase staging: every global load first, then the LDS images
    SplitStage32<256, 4, FF_TPB> s1;
    SplitStage32<64, 16, FF_TPB> s2;
    VecStage<256, FF_TPB> v1;
    VecStage<64, FF_TPB> v2, vg, vb;
    s1.load(fc1.w, tid), s2.load(fc2.w, tid);
    v1.load(fc1.b, 256, tid), v2.load(fc2.b, 64, tid), vg.load(norm_w, 64, tid), vb.load(norm_b, 64, tid);
    s1.store(W1, tid), s2.store(W2, tid);
    v1.store(b1, tid), v2.store(b2, tid), vg.store(gam, tid), vb.store(bet, tid);
  }
  __syncthreads();
  const int i = lane & 31, gg = lane >> 5;
  const bf8* w1p = reinterpret_cast<const bf8*>(W1) + lane;
  const bf8* w2p = reinterpret_cast<const bf8*>(W2) + lane;
  const long long ntiles = (N + 31) >> 5;
  const long long stride = (long long)gridDim.x * (FF_TPB / 64);
  long long tile = (long long)blockIdx.x * (FF_TPB / 64) + wave;
  // lane (i, gg) holds channels 32M + 8b + 4gg + r of point 32 tile + i in x[M][b][r]
  f4 xpre[2][4];
  ff32_load(x_in, tile * 32 + i, N, gg, xpre);
  // Pipelined across tiles: while the second-layer MFMAs of a tile's last chunks issue, the wave
  // normalises and splits the NEXT tile's input (its registers are free once the last first-layer
  // chunk has issued) and starts that tile's first hidden chunk -- no MFMA-free stretch per tile.
  XS xs[4];  // K-step 2M + h of the first layer
  bf8 w1f[8], w2f[8];
  f16v hn;
  ff32_ln(xpre, gam, bet, gg, xs);
  ff32_w1(w1p, 0, w1f);
  ff32_fc1(b1, 0, gg, w1f, xs, hn);
  ff32_w1(w1p, 1, w1f);
  __builtin_amdgcn_sched_barrier(0);
  [[maybe_unused]] int ffc_it = 0;
  for (; tile < ntiles; tile += stride) {
    const long long n = tile * 32 + i;
    [[maybe_unused]] const bool ffc_on = blockIdx.x == 7 && wave < 4 && ffc_it == 1;
    ++ffc_it;
    FFC(0);
    f16v y[2];  // the residual rides in the accumulator
#pragma unroll
    for (int M = 0; M < 2; ++M)
#pragma unroll
      for (int b = 0; b < 4; ++b) v16_set(y[M], b, xpre[M][b] + ld4(b2 + 32 * M + 8 * b + 4 * gg));
    ff32_load(x_in, (tile + stride) * 32 + i, N, gg, xpre);  // the next tile's input, used from chunk 7 on
    __builtin_amdgcn_sched_barrier(0);
    FFC(1);
    // eight 32-wide hidden chunks, two regions per chunk separated by sched_barriers.  Region A:
    // first-layer MFMAs of chunk c + 1 next to the ReLU + operand split of chunk c; region B:
    // second-layer MFMAs of chunk c.  Every region starts with its 8 weight fragments already in
    // registers (read from LDS during the previous region) -- left to itself hipcc reads each
    // fragment right before its MFMA and waits out the LDS latency 8 times a region.
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const f16v h = hn;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int My = 0; My < 2; ++My) {
          w2f[(hh * 2 + My) * 2] = w2p[((My * 16 + 2 * c + hh) * 2) * 64];
          w2f[(hh * 2 + My) * 2 + 1] = w2p[((My * 16 + 2 * c + hh) * 2 + 1) * 64];
        }
      if (c + 1 < 8) ff32_fc1(b1, c + 1, gg, w1f, xs, hn);
      XS hs[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
        hs[hh] = split8(f4_relu_i(v16_get(h, 2 * hh)), f4_relu_i(v16_get(h, 2 * hh + 1)));
      if (c == 7) ff32_ln(xpre, gam, bet, gg, xs);  // xs is dead since chunk 7's first layer issued
      __builtin_amdgcn_sched_barrier(0);
      FFC(2 + 2 * c);
      if (c + 2 < 8) ff32_w1(w1p, c + 2, w1f);
      else if (c == 6) ff32_w1(w1p, 0, w1f);
      if (c == 7) ff32_fc1(b1, 0, gg, w1f, xs, hn);  // next tile, chunk 0
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {  // K-step 2c + hh of the second layer = 16 hidden channels
#pragma unroll
        for (int My = 0; My < 2; ++My) {
          const bf8 wh = w2f[(hh * 2 + My) * 2], wm = w2f[(hh * 2 + My) * 2 + 1];
          y[My] = mfma_bf32(wm, hs[hh].hi, y[My]);
          y[My] = mfma_bf32(wh, hs[hh].mid, y[My]);
          y[My] = mfma_bf32(wh, hs[hh].hi, y[My]);
        }
      }
      if (c == 7) ff32_w1(w1p, 1, w1f);
      __builtin_amdgcn_sched_barrier(0);
      FFC(3 + 2 * c);
    }
    if (n < N) {
#pragma unroll
      for (int M = 0; M < 2; ++M)
#pragma unroll
        for (int b = 0; b < 4; ++b) st4(x_out + (size_t)n * C64 + 32 * M + 8 * b + 4 * gg, v16_get(y[M], b));
    }
  }
}

// ------------------------------------------------------------------------------------ q positional MLP (kernel)
template <int SPL>
__global__ void __launch_bounds__(QP_TPB)
k_qpos(gpn_linear_t fc0, gpn_linear_t fc2, const float* q_in,
       const float* __restrict__ emb_pt, const float* __restrict__ emb_ray, long long N, int S,
       float* q_out) {
  const QpLds L = qp_lds<SPL>();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  qp_stage<SPL, QP_TPB>(L, fc0, fc2, tid);
  __syncthreads();
  const long long ntiles = (N + 15) >> 4;
  const long long tstride = (long long)gridDim.x * (QP_TPB / 64), t0 = (long long)blockIdx.x * (QP_TPB / 64) + wave;
  const FfTilePt<1> pt = {nullptr, N, t0, 0, tstride};
  qp_run<SPL>(L, q_in, emb_pt, emb_ray, S, q_out, t0 < ntiles ? (ntiles - t0 + tstride - 1) / tstride : 0, lane, pt);
}

// Feed-forward phase of the ray kernels: tile u of a wave = samples 16 t .. 16 t + 15 of ray
// `ray0 + (u / TPR) * rstride` with t = t0 + u % TPR (TPR tiles of a ray per wave: NT in k_ray1, 1 in k_ray_attn)
template <int TPR>
struct FfRayPt {
  int R, S, ray0, rstride, t0;
  __device__ __forceinline__ void operator()(long long u, int j, long long& n, bool& ok) const {
    const int ui = (int)u, ray = ray0 + (ui / TPR) * rstride, s = 16 * (t0 + ui % TPR) + j;
    ok = ray < R && s < S;
    n = (long long)(ray < R ? ray : R - 1) * S + (s < S ? s : S - 1);
  }
};
// ------------------------------------------------------------------------------------ ray attention
#ifndef GPN_RAY1
#define GPN_RAY1 1  // S <= 48: k_ray1 (one wave per ray, keys / values in registers); 0: k_ray_attn for every S
#endif
// One wave per 16-sample tile of a ray; RPW rays per workgroup share the weights.
// LDS: Wq Wk Wv Wo [64][68] | bO gamma beta [64] | Ks [RPW][S][68] | Vt [RPW][64][S+4]
//      | part [RPW][NT][4][S]   (per-wave column sums of A, reduced in fixed order)
template <int NT>
struct RayCfg {
  // rays per workgroup (3 rays for NT == 3 fits LDS but measured 5 % slower: uneven iteration count)
  static constexpr int RPW = NT <= 2 ? 4 : (NT <= 4 ? 2 : 1);
  static constexpr int SP = NT * 16;  // samples per ray, padded to whole 16-sample tiles
  static constexpr int LDV = SP + 4;
  static constexpr int THREADS = RPW * NT * 64;
  static constexpr int LDS_FLOATS =
      4 * 64 * LDW + 3 * 64 + RPW * SP * LDW + RPW * 64 * LDV + RPW * NT * 4 * SP;
  static_assert(LDS_FLOATS * 4 <= 160 * 1024, "ray attention LDS image exceeds 160 KiB");
  static_assert(THREADS <= 1024, "too many waves per workgroup");
};

// Split (bf16x3) mode keeps the same LDS budget with 16-byte units [hi(4) | mid(4)], fragment-major
// (64 consecutive units = the A fragment of one MFMA, read at [block + lane]: conflict-free):
//   Ks: unit (key 16kt+j, head h, g) = channels 16h+4g..+3 of the key     at (4kt + h)*64 + 16g + j
//   Vt: unit (channel 16h+j, key tile kt, g) = keys 16kt+4g..+3           at (NT*h + kt)*64 + 16g + j
// so that one unit is the A operand of both MFMAs of a product: against B = [x_hi | x_hi] it yields
// a_hi*x_hi + a_mid*x_hi, against B = [x_mid | 0] it yields a_hi*x_mid.
template <int NT, int SPL>
__global__ void __launch_bounds__(RayCfg<NT>::THREADS)
k_ray_attn(gpn_ray_layer_t Lw, const float* x_in, int R, int S,
           float* x_out, float* __restrict__ abar, int abar_ld,
           float* __restrict__ weights, float* ff_out) {
  // S (<= SP) real samples per ray; keys >= S are masked out of the softmax, queries >= S are
  // computed on clamped data and never stored.
  using Cfg = RayCfg<NT>;
  constexpr int SP = Cfg::SP, RPW = Cfg::RPW, LDV = Cfg::LDV, NTHR = Cfg::THREADS;
  float* Wq = smem;
  float* Wk = Wq + 64 * LDW;
  float* Wv = Wk + 64 * LDW;
  float* Wo = Wv + 64 * LDW;
  float* bO = Wo + 64 * LDW;
  float* gam = bO + 64;
  float* bet = gam + 64;
  float* Ks = bet + 64;
  float* Vt = Ks + RPW * SP * LDW;
  float* part = Vt + RPW * 64 * LDV;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int slot = wave / NT, t = wave - slot * NT;
  float* Ks_s = Ks + slot * SP * LDW;
  float* Vt_s = Vt + slot * 64 * LDV;
  bf8* Ku_s = reinterpret_cast<bf8*>(Ks_s);
  bf8* Vu_s = reinterpret_cast<bf8*>(Vt_s);
  const int sq = 16 * t + j;          // this lane's query / sample index
  const int jrev = (int)(__builtin_bitreverse32((unsigned)j) >> 28);
  const bool qvalid = sq < S;
  const int sqc = qvalid ? sq : S - 1;
  const float qsel = qvalid ? 1.0f : 0.0f;

  // next iteration's input tile, loaded one iteration ahead; the first one is requested before the
  // weights, so that its HBM latency runs under the staging
  f4 xpre[4];
  {
    const int r0 = blockIdx.x * RPW + slot;
    const size_t row0 = ((size_t)(r0 < R ? r0 : R - 1) * S + sqc) * C64;
#pragma unroll
    for (int T = 0; T < 4; ++T) xpre[T] = ld4(x_in + row0 + 16 * T + 4 * g);
  }
  {  // two-phase staging (gnt_mfma.h): every global load first, then the LDS images
    WStage<SPL, 64, 4, NTHR> sq, sk, sv, so;
    VecStage<64, NTHR> vo, vg, vb;
    sq.load(Lw.q.w, 64, 64, tid), sk.load(Lw.k.w, 64, 64, tid), sv.load(Lw.v.w, 64, 64, tid);
    so.load(Lw.out.w, 64, 64, tid);
    vo.load(Lw.out.b, 64, tid), vg.load(Lw.attn_norm_w, 64, tid), vb.load(Lw.attn_norm_b, 64, tid);
    sq.store(Wq, LDW, tid), sk.store(Wk, LDW, tid), sv.store(Wv, LDW, tid), so.store(Wo, LDW, tid);
    vo.store(bO, tid), vg.store(gam, tid), vb.store(bet, tid);
  }
  __syncthreads();
  RA_T0();
  for (int grp = blockIdx.x; grp * RPW < R; grp += gridDim.x) {
    RA_MARK(4);
    const int ray = grp * RPW + slot;
    const bool valid = ray < R;
    f4 x[4], xn[4], qT[4];
#pragma unroll
    for (int T = 0; T < 4; ++T) x[T] = xpre[T];
    {
      const int rn = (grp + (int)gridDim.x) * RPW + slot;
      const size_t rown = ((size_t)(rn < R ? rn : R - 1) * S + sqc) * C64;
#pragma unroll
      for (int T = 0; T < 4; ++T) xpre[T] = ld4(x_in + rown + 16 * T + 4 * g);
    }
    layernorm_T(x, xn, gam, bet, 1e-6f, lane);
#pragma unroll
    for (int M = 0; M < 4; ++M) qT[M] = f4_zero();
    if constexpr (SPL != 0) {
      XS xs[2] = {split8(xn[0], xn[1]), split8(xn[2], xn[3])};
      lin_Sx<4, 4, true, bf_terms(SPL)>(Wq, xs, qT, lane);
      f4 kT[4], vN[4];
#pragma unroll
      for (int M = 0; M < 4; ++M) kT[M] = f4_zero(), vN[M] = f4_zero();
      lin_Sx<4, 4, true, bf_terms(SPL)>(Wk, xs, kT, lane);
#pragma unroll
      for (int M = 0; M < 4; ++M) Ku_s[(4 * t + M) * 64 + lane] = split4(kT[M]);
      // V in normal orientation: lane (j, g) holds V[key 16t + 4g + reg][channel 16M + j]
      lin_Sx<4, 4, false, bf_terms(SPL)>(Wv, xs, vN, lane);
#pragma unroll
      for (int M = 0; M < 4; ++M) Vu_s[(NT * M + t) * 64 + lane] = split4(vN[M]);
    } else {
      lin_T<4, 4>(Wq, LDW, xn, qT, lane);
      {
        f4 kT[4];
#pragma unroll
        for (int M = 0; M < 4; ++M) kT[M] = f4_zero();
        lin_T<4, 4>(Wk, LDW, xn, kT, lane);
#pragma unroll
        for (int M = 0; M < 4; ++M) st4(Ks_s + (16 * t + j) * LDW + 16 * M + 4 * g, kT[M]);
      }
      {
        f4 vT[4];
#pragma unroll
        for (int M = 0; M < 4; ++M) vT[M] = f4_zero();
        lin_T<4, 4>(Wv, LDW, xn, vT, lane);
#pragma unroll
        for (int M = 0; M < 4; ++M)
#pragma unroll
          for (int r = 0; r < 4; ++r) Vt_s[(16 * M + 4 * g + r) * LDV + 16 * t + j] = vT[M][r];
      }
    }
    RA_MARK(0);
    __syncthreads();
    RA_MARK(1);

    // scores / sqrt(16): scale q once instead of every score; log2(e) rides along so that the softmax is a
    // bare v_exp_f32 (2^x) in both modes
    {
      const float qscale = 0.25f * 1.44269504088896341f;
#pragma unroll
      for (int M = 0; M < 4; ++M) qT[M] = qT[M] * f4_splat(qscale);
    }
    const bool all_q = __all(qvalid);  // no padding query in this tile (S multiple of 16, or not the last tile)
    f4 oT[4], wacc[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) wacc[kt] = f4_zero();
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      // S^T tile: rows = keys, columns (lanes) = this wave's 16 queries; k-dim = the head's 16 channels
      f4 sT[NT];
      auto scores = [&]() {
        if constexpr (SPL != 0) {
          const bf8 qs = split4(qT[h]);
          bf8 qb1, qb2;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            qb1[e] = qs[e], qb1[4 + e] = SPL == 2 ? (__bf16)0.0f : qs[e];  // plain bf16: k_hi * q_hi only
            qb2[e] = qs[4 + e], qb2[4 + e] = (__bf16)0.0f;
          }
          bf8 ku[NT];
#pragma unroll
          for (int kt = 0; kt < NT; ++kt) ku[kt] = Ku_s[(4 * kt + h) * 64 + lane];
#pragma unroll
          for (int kt = 0; kt < NT; ++kt) sT[kt] = SPL == 2 ? f4_zero() : mfma_bf(ku[kt], qb2, f4_zero());
#pragma unroll
          for (int kt = 0; kt < NT; ++kt) sT[kt] = mfma_bf(ku[kt], qb1, sT[kt]);
        } else {
          f4 kf[NT];
#pragma unroll
          for (int kt = 0; kt < NT; ++kt) {
            sT[kt] = f4_zero();
            kf[kt] = ld4(Ks_s + (16 * kt + j) * LDW + 16 * h + 4 * g);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) sT[kt] = mfma16(kf[kt][r], qT[h][r], sT[kt]);
        }
        // only the last key tile can hold padding keys
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 * (NT - 1) + 4 * g + r >= S) sT[NT - 1][r] = -INFINITY;
      };
      // softmax over keys (registers x the 4 lane groups), scores / sqrt(16); returns the denominator
      auto softmax_stable = [&]() -> float {
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) mx = max3_nc(mx, max_nc(sT[kt].x, sT[kt].y), max_nc(sT[kt].z, sT[kt].w));
        mx = max_over_g(mx);
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            sT[kt][r] = __builtin_amdgcn_exp2f(sT[kt][r] - mx);
          sum += (sT[kt].x + sT[kt].y) + (sT[kt].z + sT[kt].w);
        }
        return sum_over_g(sum);
      };
      scores();
      float tot;
      {
        // Softmax is shift invariant and the scores of LayerNorm'ed rows are small: take the bare
        // exponentials (no running maximum: 16 NT compare / subtract pairs and two cross-lane maxima
        // fewer per head) and fall back to the shifted form when a denominator leaves the safe range.
        // One wave = queries of ONE ray, so a ray's result does not depend on its workgroup mates.
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) sT[kt][r] = __builtin_amdgcn_exp2f(sT[kt][r]);
          sum += (sT[kt].x + sT[kt].y) + (sT[kt].z + sT[kt].w);
        }
        tot = sum_over_g(sum);
        if (__builtin_expect(__any(!(tot > 1e-30f && tot < 1e30f)), 0)) {
          scores();
          tot = softmax_stable();
        }
      }
      const float inv = __builtin_amdgcn_rcpf(tot);
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) sT[kt] = sT[kt] * f4_splat(inv);
      // column sums over this wave's queries (for abar) and the query-0 row (for the weights).
      // Padding queries (last tile only) are zeroed in place: their outputs are never stored.
      if (!all_q) {
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) sT[kt] = sT[kt] * f4_splat(qsel);
      }
#pragma unroll
      for (int kt = 0; kt < NT; ++kt)
        if (t == 0) wacc[kt] = wacc[kt] + sT[kt];
#pragma unroll
      for (int grp = 0; grp < (NT + 3) / 4; ++grp) {  // 16 (key tile, register) values per pass, gnt_mfma.h
        float cv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) cv[e] = 4 * grp + (e >> 2) < NT ? sT[(4 * grp + (e >> 2)) % NT][e & 3] : 0.0f;
        const float c = row_sums16(cv, j);
        const int kt = 4 * grp + (jrev >> 2);  // lane j holds the total of value bitrev4(j)
        if (kt < NT) part[((slot * NT + t) * 4 + h) * SP + 16 * kt + 4 * g + (jrev & 3)] = c;
      }
      // O^T[h] = V_h^T . P^T : A = V (channels of head h on rows), B = P^T (keys in registers)
      f4 o0 = f4_zero(), o1 = f4_zero();
      if constexpr (SPL != 0) {
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
          const bf8 vu = Vu_s[(NT * h + kt) * 64 + lane];
          const bf8 ps = split4(sT[kt]);
          bf8 pb1, pb2;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            pb1[e] = ps[e], pb1[4 + e] = SPL == 2 ? (__bf16)0.0f : ps[e];  // plain bf16: v_hi * p_hi only
            pb2[e] = ps[4 + e], pb2[4 + e] = (__bf16)0.0f;
          }
          if constexpr (SPL == 2) {
            if ((kt & 1) == 0) o0 = mfma_bf(vu, pb1, o0);
            else o1 = mfma_bf(vu, pb1, o1);
          } else {
            if ((kt & 1) == 0) o0 = mfma_bf(vu, pb1, mfma_bf(vu, pb2, o0));
            else o1 = mfma_bf(vu, pb1, mfma_bf(vu, pb2, o1));
          }
        }
      } else {
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
          const f4 vf = ld4(Vt_s + (16 * h + j) * LDV + 16 * kt + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if ((kt & 1) == 0) o0 = mfma16(vf[r], sT[kt][r], o0);
            else o1 = mfma16(vf[r], sT[kt][r], o1);
          }
        }
      }
      oT[h] = o0 + o1;
    }
    f4 y[4];
#pragma unroll
    for (int M = 0; M < 4; ++M) y[M] = ld4(bO + 16 * M + 4 * g);
    lin<SPL, 4, 4>(Wo, LDW, oT, y, lane);
    if (valid) {
      if (qvalid) {
        const size_t orow = ((size_t)ray * S + sq) * C64;
#pragma unroll
        for (int M = 0; M < 4; ++M) st4(x_out + orow + 16 * M + 4 * g, y[M] + x[M]);
      }
      if (weights != nullptr && t == 0 && j == 0) {  // head-mean of attention row 0 (:216)
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = 16 * kt + 4 * g + r;
            if (key < S) weights[(size_t)ray * S + key] = wacc[kt][r] * 0.25f;
          }
      }
    }
    RA_MARK(2);
    __syncthreads();
    RA_MARK(3);
#ifdef GPN_STAMP
    ra_acc_[5] += 1;
#endif
    if (abar != nullptr) {
      for (int idx = tid; idx < RPW * 4 * SP; idx += NTHR) {
        const int sl = idx / (4 * SP), rem = idx - sl * 4 * SP;
        const int h = rem / SP, key = rem - h * SP;
        const int rr = grp * RPW + sl;
        float c = 0.f;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) c += part[(sl * NT + tt) * 4 * SP + rem];
        if (rr < R && key < S) abar[((size_t)rr * S + key) * abar_ld + h] = c / (float)S;
      }
    }
    // no third barrier: the next iteration writes Ks / Vt only (every wave is past the attention
    // phase here) and its own barrier separates these reads of `part` from the next writes to it
  }
  RA_MARK(4);
  RA_END();
  if constexpr (NT <= RAY_FF_MAXNT)
  if (ff_out != nullptr) {  // (wave-uniform: a kernel argument)
    const int ngrp = (R + RPW - 1) / RPW;
    const long long nu = (int)blockIdx.x < ngrp ? (ngrp - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    const FfRayPt<1> pt = {R, S, (int)blockIdx.x * RPW + slot, (int)gridDim.x * RPW, t};
    ff_phase<SPL, NTHR>(Lw.ff_norm_w, Lw.ff_norm_b, Lw.ff1, Lw.ff2, x_out, ff_out, nu, tid, pt);
  }
}

// ---- k_ray1: ONE wave per ray (instantiated for NT <= 4, used for NT <= 3, i.e. S <= 48).  The keys and values of a ray never leave the
// wave that made them: the T-tile a k projection produces IS the A operand of the score MFMAs of that
// key tile (lane (j, g) holds K[key j][channel 16h + 4g + r]) and a v projection in normal orientation
// (lin_N) IS the A operand of the P.V MFMAs, so K and V stay in 2 x 16 NT registers instead of going
// through LDS, and no barrier is left in the kernel: the waves of a workgroup share nothing but the
// weight images.  Same arithmetic in the same order as k_ray_attn (bit-identical results).
#ifndef RAY1_WAVES
#define RAY1_WAVES 8
#endif
#ifndef RAY1_MINW
#define RAY1_MINW 2
#endif
#ifndef RAY1_MAXNT
#define RAY1_MAXNT 3
#endif
// weight images | bO gamma beta | per wave: the attention row of query 0 of every head (for `weights`)
constexpr int ray1_lds_floats(bool spl) { return 4 * wimg_floats(spl, 64, 4, LDW) + 3 * 64 + RAY1_WAVES * 4 * 64; }

template <int NT, int SPL>
__global__ void __launch_bounds__(RAY1_WAVES * 64, RAY1_MINW)
k_ray1(gpn_ray_layer_t Lw, const float* x_in, int R, int S, float* x_out,
       float* __restrict__ abar, int abar_ld, float* __restrict__ weights, float* ff_out) {
  constexpr int NTHR = RAY1_WAVES * 64;
  float* Wq = smem;
  float* Wk = Wq + wimg_floats(SPL, 64, 4, LDW);
  float* Wv = Wk + wimg_floats(SPL, 64, 4, LDW);
  float* Wo = Wv + wimg_floats(SPL, 64, 4, LDW);
  float* bO = Wo + wimg_floats(SPL, 64, 4, LDW);
  float* gam = bO + 64;
  float* bet = gam + 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* wrow = bet + 64 + wave * 256;  // [4 heads][64 keys]: query 0's attention row (lanes j == 0 only)
  const int j = lane & 15, g = lane >> 4;
  const int jrev = (int)(__builtin_bitreverse32((unsigned)j) >> 28);
  {  // two-phase staging (gnt_mfma.h): every global load first, then the LDS images
    WStage<SPL, 64, 4, NTHR> sq, sk, sv, so;
    VecStage<64, NTHR> vo, vg, vb;
    sq.load(Lw.q.w, 64, 64, tid), sk.load(Lw.k.w, 64, 64, tid), sv.load(Lw.v.w, 64, 64, tid);
    so.load(Lw.out.w, 64, 64, tid);
    vo.load(Lw.out.b, 64, tid), vg.load(Lw.attn_norm_w, 64, tid), vb.load(Lw.attn_norm_b, 64, tid);
    sq.store(Wq, LDW, tid), sk.store(Wk, LDW, tid), sv.store(Wv, LDW, tid), so.store(Wo, LDW, tid);
    vo.store(bO, tid), vg.store(gam, tid), vb.store(bet, tid);
  }
  __syncthreads();
  const float qscale = 0.25f * 1.44269504088896341f;  // 1/sqrt(16) and log2(e): the softmax is a bare 2^x
  for (int ray = blockIdx.x * RAY1_WAVES + wave; ray < R; ray += (int)gridDim.x * RAY1_WAVES) {
    const size_t rbase = (size_t)ray * S;
    // ---- keys and values of the whole ray, tile by tile, into registers
    typedef typename std::conditional<(SPL != 0), bf8, f4>::type kv_t;
    kv_t kk[NT][4], vv[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int sq_ = 16 * t + j, sc = sq_ < S ? sq_ : S - 1;
      f4 x[4], xn[4];
#pragma unroll
      for (int T = 0; T < 4; ++T) x[T] = ld4(x_in + (rbase + sc) * C64 + 16 * T + 4 * g);
      layernorm_T(x, xn, gam, bet, 1e-6f, lane);
      f4 kT[4], vN[4];
#pragma unroll
      for (int M = 0; M < 4; ++M) kT[M] = f4_zero(), vN[M] = f4_zero();
      if constexpr (SPL != 0) {
        XS xs[2] = {split8(xn[0], xn[1]), split8(xn[2], xn[3])};
        lin_Sx<4, 4, true, bf_terms(SPL)>(Wk, xs, kT, lane);
        lin_Sx<4, 4, false, bf_terms(SPL)>(Wv, xs, vN, lane);
#pragma unroll
        for (int M = 0; M < 4; ++M) kk[t][M] = split4(kT[M]), vv[t][M] = split4(vN[M]);
      } else {
        lin_T<4, 4>(Wk, LDW, xn, kT, lane);
        lin_N<4, 4>(Wv, LDW, xn, vN, lane);
#pragma unroll
        for (int M = 0; M < 4; ++M) kk[t][M] = kT[M], vv[t][M] = vN[M];
      }
    }
    // ---- queries, tile by tile
    float csum[4][(NT + 3) / 4];  // column sums of the attention tiles over all queries (abar), per head
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
      for (int q = 0; q < (NT + 3) / 4; ++q) csum[h][q] = 0.0f;
#pragma unroll 1
    for (int t = 0; t < NT; ++t) {
      const int sq_ = 16 * t + j;
      const bool qvalid = sq_ < S;
      const int sc = qvalid ? sq_ : S - 1;
      const float qsel = qvalid ? 1.0f : 0.0f;
      f4 x[4], xn[4], qT[4];
#pragma unroll
      for (int T = 0; T < 4; ++T) x[T] = ld4(x_in + (rbase + sc) * C64 + 16 * T + 4 * g);
      layernorm_T(x, xn, gam, bet, 1e-6f, lane);
#pragma unroll
      for (int M = 0; M < 4; ++M) qT[M] = f4_zero();
      if constexpr (SPL != 0) {
        XS xs[2] = {split8(xn[0], xn[1]), split8(xn[2], xn[3])};
        lin_Sx<4, 4, true, bf_terms(SPL)>(Wq, xs, qT, lane);
      } else {
        lin_T<4, 4>(Wq, LDW, xn, qT, lane);
      }
#pragma unroll
      for (int M = 0; M < 4; ++M) qT[M] = qT[M] * f4_splat(qscale);
      const bool all_q = __all(qvalid);
      f4 oT[4];
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        f4 sT[NT];
        auto scores = [&]() {
          if constexpr (SPL != 0) {
            const bf8 qs = split4(qT[h]);
            bf8 qb1, qb2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              qb1[e] = qs[e], qb1[4 + e] = SPL == 2 ? (__bf16)0.0f : qs[e];  // plain bf16: k_hi * q_hi only
              qb2[e] = qs[4 + e], qb2[4 + e] = (__bf16)0.0f;
            }
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) sT[kt] = SPL == 2 ? f4_zero() : mfma_bf(kk[kt][h], qb2, f4_zero());
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) sT[kt] = mfma_bf(kk[kt][h], qb1, sT[kt]);
          } else {
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) sT[kt] = f4_zero();
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int kt = 0; kt < NT; ++kt) sT[kt] = mfma16(kk[kt][h][r], qT[h][r], sT[kt]);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (16 * (NT - 1) + 4 * g + r >= S) sT[NT - 1][r] = -INFINITY;
        };
        scores();
        float tot;
        {
          float sum = 0.f;
#pragma unroll
          for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sT[kt][r] = __builtin_amdgcn_exp2f(sT[kt][r]);
            sum += (sT[kt].x + sT[kt].y) + (sT[kt].z + sT[kt].w);
          }
          tot = sum_over_g(sum);
          if (__builtin_expect(__any(!(tot > 1e-30f && tot < 1e30f)), 0)) {  // shifted form (see k_ray_attn)
            scores();
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) mx = max3_nc(mx, max_nc(sT[kt].x, sT[kt].y), max_nc(sT[kt].z, sT[kt].w));
            mx = max_over_g(mx);
            sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
              for (int r = 0; r < 4; ++r) sT[kt][r] = __builtin_amdgcn_exp2f(sT[kt][r] - mx);
              sum += (sT[kt].x + sT[kt].y) + (sT[kt].z + sT[kt].w);
            }
            tot = sum_over_g(sum);
          }
        }
        const float inv = __builtin_amdgcn_rcpf(tot);
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) sT[kt] = sT[kt] * f4_splat(inv);
        if (!all_q) {
#pragma unroll
          for (int kt = 0; kt < NT; ++kt) sT[kt] = sT[kt] * f4_splat(qsel);
        }
        if (weights != nullptr && t == 0 && j == 0) {  // (parked in LDS: 16 registers less through the head loop)
#pragma unroll
          for (int kt = 0; kt < NT; ++kt) st4(wrow + h * 64 + 16 * kt + 4 * g, sT[kt]);
        }
        if (abar != nullptr) {
#pragma unroll
          for (int grp = 0; grp < (NT + 3) / 4; ++grp) {
            float cv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) cv[e] = 4 * grp + (e >> 2) < NT ? sT[(4 * grp + (e >> 2)) % NT][e & 3] : 0.0f;
            csum[h][grp] += row_sums16(cv, j);
          }
        }
        f4 o0 = f4_zero(), o1 = f4_zero();
        if constexpr (SPL != 0) {
#pragma unroll
          for (int kt = 0; kt < NT; ++kt) {
            const bf8 ps = split4(sT[kt]);
            bf8 pb1, pb2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              pb1[e] = ps[e], pb1[4 + e] = SPL == 2 ? (__bf16)0.0f : ps[e];  // plain bf16: v_hi * p_hi only
              pb2[e] = ps[4 + e], pb2[4 + e] = (__bf16)0.0f;
            }
            if constexpr (SPL == 2) {
              if ((kt & 1) == 0) o0 = mfma_bf(vv[kt][h], pb1, o0);
              else o1 = mfma_bf(vv[kt][h], pb1, o1);
            } else {
              if ((kt & 1) == 0) o0 = mfma_bf(vv[kt][h], pb1, mfma_bf(vv[kt][h], pb2, o0));
              else o1 = mfma_bf(vv[kt][h], pb1, mfma_bf(vv[kt][h], pb2, o1));
            }
          }
        } else {
#pragma unroll
          for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if ((kt & 1) == 0) o0 = mfma16(vv[kt][h][r], sT[kt][r], o0);
              else o1 = mfma16(vv[kt][h][r], sT[kt][r], o1);
            }
          }
        }
        oT[h] = o0 + o1;
      }
      // the residual input is read AGAIN here (L2-hot) instead of being held through the attention: 16
      // registers that the NT = 4 instantiation does not have
      f4 y[4], xr[4];
#pragma unroll
      for (int T = 0; T < 4; ++T) xr[T] = ld4(x_in + (rbase + sc) * C64 + 16 * T + 4 * g);
#pragma unroll
      for (int M = 0; M < 4; ++M) y[M] = ld4(bO + 16 * M + 4 * g);
      lin<SPL, 4, 4>(Wo, LDW, oT, y, lane);
      if (qvalid) {
#pragma unroll
        for (int M = 0; M < 4; ++M) st4(x_out + (rbase + sq_) * C64 + 16 * M + 4 * g, y[M] + xr[M]);
      }
      if (weights != nullptr && t == 0 && j == 0) {  // head-mean of attention row 0 (:216)
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
          const float* wr = wrow + 16 * kt + 4 * g;
          const f4 w4 = ((ld4(wr) + ld4(wr + 64)) + ld4(wr + 128)) + ld4(wr + 192);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = 16 * kt + 4 * g + r;
            if (key < S) weights[rbase + key] = w4[r] * 0.25f;
          }
        }
      }
    }
    if (abar != nullptr) {  // lane j holds the column total of value bitrev4(j) of its 16-value group
#pragma unroll
      for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int grp = 0; grp < (NT + 3) / 4; ++grp) {
          const int kt = 4 * grp + (jrev >> 2), key = 16 * kt + 4 * g + (jrev & 3);
          if (kt < NT && key < S) abar[(rbase + key) * abar_ld + h] = csum[h][grp] / (float)S;
        }
    }
  }
  if (ff_out != nullptr) {  // (wave-uniform: a kernel argument)
    const int ray0 = blockIdx.x * RAY1_WAVES + wave, rstride = (int)gridDim.x * RAY1_WAVES;
    const long long nu = ray0 < R ? (long long)((R - ray0 + rstride - 1) / rstride) * NT : 0;
    const FfRayPt<NT> pt = {R, S, ray0, rstride, 0};
    ff_phase<SPL, NTHR>(Lw.ff_norm_w, Lw.ff_norm_b, Lw.ff1, Lw.ff2, x_out, ff_out, nu, tid, pt);
  }
}

// ------------------------------------------------------------------------------------ semantic aggregation
// One wave per ray, 8 adjacent rays per workgroup, lane = one 8-channel group of the 512 deep-semantic
// channels (so the per-group attention weights are computed exactly once).  The 8 waves walk the
// samples together (one barrier per sample): neighbouring rays project to neighbouring texels, so
// their bilinear taps hit the same lines while those are still in L1/L2.  Per visible (sample, view)
// the attention weights of every layer are rebuilt from the stored logits + softmax stats, summed
// into W3d / W2d and applied to the sampled (or dense) deep-semantic row.  Masked views have weight
// exactly 0 and are skipped; a sample with no visible view has the uniform weight 1/V in every layer.
#define SEM_RAYS 8
#ifndef SEM_WAVES_PER_SIMD
#define SEM_WAVES_PER_SIMD 4
#endif
__device__ __forceinline__ float rl_f(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ unsigned rl_u(unsigned v, int l) {
  return (unsigned)__builtin_amdgcn_readlane((int)v, l);
}

template <int DEPTH>
__global__ void __launch_bounds__(SEM_RAYS * 64, SEM_WAVES_PER_SIMD)
k_sem(gpn_sem_args_t a, int strip_groups) {
  constexpr int LP = (DEPTH + 1) / 2;  // layers are processed in pairs (v_pk_fma_f32)
  __shared__ float sP[GPN_MAX_VIEWS * 12];
  __shared__ __attribute__((aligned(16))) float sH[SEM_RAYS][LP * 16];
  __shared__ __attribute__((aligned(16))) float sW[LP * 4 * 64 * 4];
  const int tid = threadIdx.x, lane = tid & 63;
  // the wave index (hence the ray and every per-sample address) is wave-uniform: keep it in SGPRs
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2): give each XCD a
  // contiguous run of ray groups.
  int blk = blockIdx.x;
  if (strip_groups > 0) {
    // Rays are consecutive pixels in row-major order, G = 8 x strip_groups groups of SEM_RAYS rays per image
    // row, the chunk starting g0 groups into its first row (gpn_sem_args_t.ray_row_*).  The groups are put in
    // the order (column strip, row, column inside the strip) and that sequence is cut into 8 equal runs, one
    // per XCD (workgroups b and b + 8 share an XCD): an XCD works on a 2-D patch of the image -- the rows of a
    // strip share texel rows -- instead of a 1-D run of rays, with the same number of workgroups as before.
    const int groups = (a.R + SEM_RAYS - 1) / SEM_RAYS, G = strip_groups * 8;
    const int g0 = a.ray_row_offset / SEM_RAYS, g1 = g0 + groups;  // the chunk in "virtual" group numbers
    const int nb = gridDim.x, per = nb >> 3;
    int p = blk < per * 8 ? (blk & 7) * per + (blk >> 3) : blk;  // position in the sequence
    // virtual groups below n that lie in strip k: whole rows + the part of the last row
    auto below = [&](int n, int k) { return (n / G) * strip_groups + min(max(n % G - k * strip_groups, 0), strip_groups); };
    int k = 0;
    for (; k < 7; ++k) {
      const int nk = below(g1, k) - below(g0, k);
      if (p < nk) break;
      p -= nk;
    }
    // the p-th group of strip k: its first row may start inside the strip (or past it)
    const int row0 = g0 / G, first = strip_groups - min(max(g0 % G - k * strip_groups, 0), strip_groups);
    int row, col;
    if (p < first) {
      row = row0, col = (k + 1) * strip_groups - first + p;
    } else {
      p -= first;
      row = row0 + 1 + p / strip_groups, col = k * strip_groups + p % strip_groups;
    }
    blk = row * G + col - g0;
  } else {
    const int nb = gridDim.x, per = nb >> 3;
    if (per > 0 && blk < per * 8) blk = (blk & 7) * per + (blk >> 3);
  }
  const int S = a.S, V = a.V;
  if (a.deep_map != nullptr) {
    for (int i = tid; i < V * 12; i += SEM_RAYS * 64) sP[i] = a.proj[(i / 12) * 16 + (i % 12)];
  }
  for (int i = tid; i < SEM_RAYS * LP * 16; i += SEM_RAYS * 64) (&sH[0][0])[i] = 0.0f;
  __syncthreads();
  // (Banding the workgroups of an XCD into 2-D pixel patches was measured: no gain over 1-D runs.)
  const int r = blk * SEM_RAYS + wave;
  const bool rvalid = r < a.R;
  const int rc = rvalid ? r : a.R - 1;
  // lane == channel group (8 channels) for the attention weights; the 512-channel rows themselves
  // are read and written as two fully coalesced 1 KB halves: lane holds channels 4*lane..+3 and
  // 256 + 4*lane..+3, whose groups are lane/2 and 32 + lane/2 (weights fetched from those lanes)
  const int c4 = lane * 4, head = lane >> 4;
  f4 acc2a = f4_zero(), acc2b = f4_zero();
  const size_t rs = (size_t)a.deep_w * GPN_SEM;
  const unsigned urs = (unsigned)rs;  // one map is < 2^31 floats: 32-bit element offsets inside a view
  const float inv_v = 1.0f / (float)V;
  const unsigned all_views = V >= 32 ? 0xffffffffu : ((1u << V) - 1u);
  const bool dense = a.deep_dense != nullptr;
  // attn_fc[2] of every layer: this lane's row (its channel group) of the [64,8] weight and its
  // bias, scaled by log2(e) so that the softmax weight is a bare v_exp_f32; layers l = 2p, 2p+1
  // sit in the two halves of an f2 (one v_pk_fma_f32 per hidden value and layer pair)
  // (weights live in LDS, one 16-byte unit per lane, layer pair and pair of hidden values:
  // (w[2p][k], w[2p+1][k], w[2p][k+1], w[2p+1][k+1]); 72 registers fewer = twice the waves per SIMD)
  f2 a2b[LP];
#pragma unroll
  for (int l = 0; l < 2 * LP; ++l) {
    f4 w0 = f4_zero(), w1 = f4_zero();
    float b = 0.0f;
    if (l < DEPTH) {
      w0 = ld4(a.attn2_w[l] + lane * 8), w1 = ld4(a.attn2_w[l] + lane * 8 + 4);
      b = a.attn2_b[l] ? a.attn2_b[l][lane] : 0.0f;
    }
    if (wave == 0) {  // one wave fills the image (every wave would write the same values)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        sW[(((l >> 1) * 4 + (k >> 1)) * 64 + lane) * 4 + (k & 1) * 2 + (l & 1)] = w0[k] * LOG2E_F;
        sW[(((l >> 1) * 4 + 2 + (k >> 1)) * 64 + lane) * 4 + (k & 1) * 2 + (l & 1)] = w1[k] * LOG2E_F;
      }
    }
    a2b[l >> 1][l & 1] = b;
  }
  __syncthreads();
  // hidden value i = 8 l + k of a pair is parked at LDS slot ((l>>1)*8 + k)*2 + (l&1): the two
  // layers of a pair side by side, read back 4 floats (2 hidden values x 2 layers) at a time
  const int hslot = lane < DEPTH * 8 ? ((lane >> 4) * 8 + (lane & 7)) * 2 + ((lane >> 3) & 1) : 0;
  // Bilinear taps of a sample in every source view, one view per lane (the arithmetic of the
  // projector, proj_common.h): element offsets inside the view's map (clamped to valid texels) and
  // the four weights, zero for taps outside the map (grid_sample's zero padding).
  // The table is made for a BATCH of SB = 64 / V consecutive samples at once, lane = (sample of the batch, view):
  // the projection arithmetic runs once per batch with SB * V of the 64 lanes busy instead of once per sample
  // with V of them (it was 148 of the ~310 instructions a sample costs outside its view loop).
  struct TapTab {
    unsigned o00, o01, o10, o11;
    float wnw, wne, wsw, wse;
  };
  const int SB = min(64 / V, 8), jl = lane / V, vl = lane - jl * V;  // (V <= GPN_MAX_VIEWS = 32: SB >= 2)
  const bool tl = jl < SB;                                         // lanes that carry a (sample, view) of the batch
  auto batch_row = [&](int s0_) { return (size_t)rc * S + min(s0_ + jl, S - 1); };  // this lane's sample (clamped)
  unsigned long long hit64 = ~0ull;  // views with a tap inside their map, SB fields of V bits (wave-uniform)
  auto tap_table = [&](float px, float py, float pz) {
    TapTab t_ = {0, 0, 0, 0, 0.f, 0.f, 0.f, 0.f};
    bool hit_ = false;
    if (!dense && tl) {
      const GpnPix pp = gpn_project(sP + vl * 12, px, py, pz);
      const GpnTaps t = gpn_taps(pp.px, pp.py, a.img_h, a.img_w, a.deep_h, a.deep_w);
      const int x0 = min(max(t.x0, 0), a.deep_w - 1), x1 = min(max(t.x0 + 1, 0), a.deep_w - 1);
      const int y0 = min(max(t.y0, 0), a.deep_h - 1), y1 = min(max(t.y0 + 1, 0), a.deep_h - 1);
      t_.o00 = (unsigned)y0 * urs + (unsigned)x0 * GPN_SEM, t_.o01 = (unsigned)y0 * urs + (unsigned)x1 * GPN_SEM;
      t_.o10 = (unsigned)y1 * urs + (unsigned)x0 * GPN_SEM, t_.o11 = (unsigned)y1 * urs + (unsigned)x1 * GPN_SEM;
      t_.wnw = (t.vy0 && t.vx0) ? t.nw : 0.0f, t_.wne = (t.vy0 && t.vx1) ? t.ne : 0.0f;
      t_.wsw = (t.vy1 && t.vx0) ? t.sw : 0.0f, t_.wse = (t.vy1 && t.vx1) ? t.se : 0.0f;
      hit_ = (t.vy0 || t.vy1) && (t.vx0 || t.vx1);
    }
    if (!dense) hit64 = __ballot(hit_);
    return t_;
  };
  // the 8 x 16 bytes a lane needs of one (sample, view): four taps x this lane's 8 channels; requested right
  // before the pair's weight rebuild, which runs while they are in flight (requesting them a pair earlier costs
  // 32 registers, i.e. the fourth wave per SIMD, or the L1 sharing of the lock-stepped rays: DESIGN.md 7, 7b)
  f4 tp[8];
  auto fetch_taps = [&](const TapTab& t_, int v_, int l_) {  // l_: the lane that holds (sample, view v_)
    const unsigned s00 = rl_u(t_.o00, l_), s01 = rl_u(t_.o01, l_), s10 = rl_u(t_.o10, l_), s11 = rl_u(t_.o11, l_);
    const float* dm = a.deep_map + (size_t)v_ * a.deep_h * rs + (unsigned)c4;
    tp[0] = ld4(dm + s00), tp[1] = ld4(dm + s00 + 256), tp[2] = ld4(dm + s01), tp[3] = ld4(dm + s01 + 256);
    tp[4] = ld4(dm + s10), tp[5] = ld4(dm + s10 + 256), tp[6] = ld4(dm + s11), tp[7] = ld4(dm + s11 + 256);
  };
  // mask and sample positions of a batch are loaded one batch ahead (lane = (sample of the batch, view))
  float mv_next = tl ? a.mask[batch_row(0) * V + vl] : 0.0f;
  float nx = 0.f, ny = 0.f, nz = 0.f;
  if (!dense && tl) {
    const float* xp = a.xyz + batch_row(0) * 3;
    nx = xp[0], ny = xp[1], nz = xp[2];
  }
  TapTab tab = {0, 0, 0, 0, 0.f, 0.f, 0.f, 0.f};
  unsigned long long vis64 = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) tp[k] = f4_zero();
  int jb = 0, lb = 0;  // sample inside the batch, first lane of its views
  for (int s = 0; s < S; ++s) {
    const size_t n = (size_t)rc * S + s;
    if (jb == 0) {  // new batch: its table from the positions loaded a batch ago, then the next batch's loads
      vis64 = __ballot(mv_next != 0.0f);
      tab = tap_table(nx, ny, nz);
      if (s + SB < S) {
        mv_next = tl ? a.mask[batch_row(s + SB) * V + vl] : 0.0f;
        if (!dense && tl) {
          const float* xp = a.xyz + batch_row(s + SB) * 3;
          nx = xp[0], ny = xp[1], nz = xp[2];
        }
      }
    }
    const unsigned vis = (unsigned)(vis64 >> lb) & all_views;
    const unsigned hit = (unsigned)(hit64 >> lb) & all_views;
    const int lane0 = lb;  // lane lane0 + v holds view v of this sample
    if (++jb == SB) jb = 0, lb = 0; else lb += V;
    f2 bl[LP], ab[LP];
    const float* abp = a.abar + n * (DEPTH * 4);
    const float* stp = a.stats + n * (DEPTH * C64);
#pragma unroll
    for (int l = 0; l < 2 * LP; ++l) {
      float abv = 0.0f, lse = 0.0f;
      if (l < DEPTH) {
        abv = abp[4 * l + head];
        lse = __builtin_nontemporal_load(stp + (C64 * l + lane));
      }
      ab[l >> 1][l & 1] = abv;
      bl[l >> 1][l & 1] = lse;
    }
    // an all-masked sample takes the uniform weight over EVERY view, but a view whose four taps all
    // lie outside its map contributes exact zeros (grid_sample zero padding): skipped, same bits
    unsigned live = vis != 0 ? vis : hit;
    // hidden layer of attn_fc for all DEPTH layers of a (sample, view): 8*DEPTH floats, one per lane,
    // fetched one view ahead; every lane needs all of them, so they go through an LDS slot of the
    // wave and come back as broadcast reads (same address in all lanes: conflict free)
    float hvn = 0.f;
    if (vis != 0 && lane < DEPTH * 8)
      hvn = __builtin_nontemporal_load(a.hidden + (n * V + __builtin_ctz(live)) * (DEPTH * 8) + lane);
    // bias - log-sum-exp, in log2 units: the constant part of every logit of this sample
#pragma unroll
    for (int p = 0; p < LP; ++p) bl[p] = (a2b[p] - bl[p]) * (f2){LOG2E_F, LOG2E_F};
    float w3u = 0.f, w2u = 0.f;  // uniform weights of an all-masked sample
#pragma unroll
    for (int l = 0; l < DEPTH; ++l) w3u += inv_v, w2u += ab[l >> 1][l & 1] * inv_v;
    f4 acc3a = f4_zero(), acc3b = f4_zero();
    while (live) {
      const int v = __builtin_ctz(live);
      live &= live - 1;
      if (vis != 0) {
        if (lane < DEPTH * 8) sH[wave][hslot] = hvn;
        if (live && lane < DEPTH * 8)
          hvn = __builtin_nontemporal_load(a.hidden + (n * V + __builtin_ctz(live)) * (DEPTH * 8) + lane);
      }
      f4 d0, d1;
      if (dense) {
        const float* dd = a.deep_dense + (n * V + v) * GPN_SEM;
        d0 = ld4(dd + (unsigned)c4);
        d1 = ld4(dd + (unsigned)(c4 + 256));
      } else {
        // this view's tap weights: wave-uniform (scalar registers); out-of-map taps carry weight 0
        const int tlane = lane0 + v;
        const float tnw_w = rl_f(tab.wnw, tlane), tne_w = rl_f(tab.wne, tlane), tsw_w = rl_f(tab.wsw, tlane), tse_w = rl_f(tab.wse, tlane);
        fetch_taps(tab, v, tlane);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          d0[c] = gpn_bilerp4(tp[0][c], tp[2][c], tp[4][c], tp[6][c], tnw_w, tne_w, tsw_w, tse_w);
          d1[c] = gpn_bilerp4(tp[1][c], tp[3][c], tp[5][c], tp[7][c], tnw_w, tne_w, tsw_w, tse_w);
        }
      }
      float w3 = w3u, w2 = w2u;
      if (vis != 0) {
        // logits of this lane's channel group in all layers: attn_fc[2] row . hidden + bias (:66-70),
        // two layers per packed fma; softmax weight = 2^(logit - lse) in log2 units
        const float* hp = sH[wave];
        f2 w3p = (f2){0.f, 0.f}, w2p = (f2){0.f, 0.f};
#pragma unroll
        for (int p = 0; p < LP; ++p) {
          f2 lg = bl[p];
#pragma unroll
          for (int k = 0; k < 8; k += 2) {
            const f4 h = ld4(hp + (p * 8 + k) * 2);  // (k, 2p) (k, 2p+1) (k+1, 2p) (k+1, 2p+1)
            const f4 w = ld4(sW + ((p * 4 + (k >> 1)) * 64 + lane) * 4);
            lg = w.lo * h.lo + lg;
            lg = w.hi * h.hi + lg;
          }
          f2 at;
          at.x = __builtin_amdgcn_exp2f(lg.x);
          at.y = (2 * p + 1 < DEPTH) ? __builtin_amdgcn_exp2f(lg.y) : 0.0f;
          w3p = w3p + at;
          w2p = ab[p] * at + w2p;
        }
        w3 = w3p.x + w3p.y, w2 = w2p.x + w2p.y;
      }
      // weights of the two channel groups this lane's channels belong to
      const float w3A = __shfl(w3, lane >> 1), w3B = __shfl(w3, 32 + (lane >> 1));
      const float w2A = __shfl(w2, lane >> 1), w2B = __shfl(w2, 32 + (lane >> 1));
      acc3a = acc3a + d0 * f4_splat(w3A);
      acc3b = acc3b + d1 * f4_splat(w3B);
      acc2a = acc2a + d0 * f4_splat(w2A);
      acc2b = acc2b + d1 * f4_splat(w2B);
    }
    if (a.sem3d != nullptr && rvalid) {
      __builtin_nontemporal_store(acc3a, reinterpret_cast<f4*>(a.sem3d + n * GPN_SEM + c4));
      __builtin_nontemporal_store(acc3b, reinterpret_cast<f4*>(a.sem3d + n * GPN_SEM + c4 + 256));
    }
#ifndef SEM_NO_SYNC
    __syncthreads();  // keep the 8 rays of the group on the same sample (cache locality only)
#endif
  }
  if (rvalid) {
    st4(a.pre + (size_t)r * GPN_SEM + c4, acc2a);
    st4(a.pre + (size_t)r * GPN_SEM + c4 + 256, acc2b);
  }
}

// ------------------------------------------------------------------------------------ heads
// One wave per ray: final LayerNorm (eps 1e-5), mean over samples, rgb_fc, cat with weights, depth.
__global__ void __launch_bounds__(256)
k_heads_rgb(const float* __restrict__ norm_w, const float* __restrict__ norm_b, gpn_linear_t rgb_fc,
            const float* __restrict__ q, const float* __restrict__ weights,
            const float* __restrict__ z_vals, int R, int S, float* __restrict__ out,
            float* __restrict__ depth) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  // One wave per ray, four samples at a time: the 16 lanes of a DPP row own one sample's 64 channels
  // (lane (g, j): sample s0 + g, channels 4j..4j+3, one 16-byte load); the per-sample LayerNorm
  // statistics are xor-butterflies inside the row (every lane gets the total), independent across
  // the unrolled iterations.
  const int j = lane & 15, g = lane >> 4;
  const f4 ga = ld4(norm_w + 4 * j), be = ld4(norm_b + 4 * j);
  auto row_all = [](float v) {  // sum over the 16 lanes of the row, result in every lane
    v += dpp_f<0xB1>(v);   // ^1
    v += dpp_f<0x4E>(v);   // ^2
    v += dpp_f<0x141>(v);  // row_half_mirror: the other quad of the half
    v += dpp_f<0x140>(v);  // row_mirror: the other half
    return v;
  };
  f4 hsum = f4_zero();
  for (int s0 = 0; s0 < S; s0 += 16) {  // 4 x 4 samples per trip: four independent loads / butterflies
    f4 x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int s = s0 + 4 * u + g;
      x[u] = ld4(q + ((size_t)r * S + (s < S ? s : S - 1)) * C64 + 4 * j);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool ok = s0 + 4 * u + g < S;
      const float m = row_all((x[u].x + x[u].y) + (x[u].z + x[u].w)) * (1.0f / 64.0f);
      const f4 d = x[u] - f4_splat(m);
      const float vv = row_all((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w)) * (1.0f / 64.0f);
      const float rstd = ok ? 1.0f / sqrtf(vv + 1e-5f) : 0.0f;
      const f4 t = d * f4_splat(rstd) * ga + be;
      if (ok) hsum = hsum + t;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) hsum[e] = sum_over_g(hsum[e]);  // the four sample groups
  const f4 hmean = hsum / f4_splat((float)S);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const f4 w = ld4(rgb_fc.w + c * C64 + 4 * j);
    const float p = row_all((w.x * hmean.x + w.y * hmean.y) + (w.z * hmean.z + w.w * hmean.w));
    if (lane == 0) out[(size_t)r * (3 + S) + c] = p + (rgb_fc.b ? rgb_fc.b[c] : 0.f);
  }
  float dsum = 0.f;
  for (int s = lane; s < S; s += 64) {
    const float w = weights[(size_t)r * S + s];
    out[(size_t)r * (3 + S) + 3 + s] = w;
    if (z_vals != nullptr) dsum += w * z_vals[(size_t)r * S + s];
  }
  if (depth != nullptr && z_vals != nullptr) {
    for (int o = 32; o > 0; o >>= 1) dsum += __shfl_xor(dsum, o);
    if (lane == 0) depth[r] = dsum;
  }
}

// sem_feat = ReLU(W pre + b): [R,512] x [512,512]^T on fp32 MFMA.  A workgroup takes 64 rays x 128 output
// columns: wave (rt, ch) = 16 rays x 64 columns, the 128 x 512 weight slab goes through LDS in eight
// double-buffered 64-wide K chunks loaded by the whole workgroup (the first form read its weight rows
// straight from L2, one dependent round trip per 16 K: 44 us per launch at R = 4096, latency-bound).
#define SA_LDW 68
#define SA_LDS_FLOATS (2 * 128 * SA_LDW)
__global__ void __launch_bounds__(512)
k_sem_adaptor(gpn_linear_t fc, const float* __restrict__ pre, int R, float* __restrict__ sem_feat) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int rt = wave & 3, ch = wave >> 2;
  const int c0 = (blockIdx.x & 3) * 128;             // first output column of the workgroup
  const int rr = (blockIdx.x >> 2) * 64 + rt * 16 + j;
  const bool valid = rr < R;
  const int rc = valid ? rr : R - 1;
  // slab chunk kc: rows c0 .. c0+127 of W, columns 64 kc .. 64 kc + 63; 2048 float4, four per thread
  f4 wreg[4];
  auto slab_load = [&](int kc) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = tid + it * 512, row = idx >> 4, c4 = idx & 15;
      wreg[it] = ld4(fc.w + (size_t)(c0 + row) * GPN_SEM + 64 * kc + 4 * c4);
    }
  };
  auto slab_store = [&](int buf) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = tid + it * 512, row = idx >> 4, c4 = idx & 15;
      st4(smem + buf * (128 * SA_LDW) + row * SA_LDW + 4 * c4, wreg[it]);
    }
  };
  slab_load(0);
  f4 x[32];
#pragma unroll
  for (int T = 0; T < 32; ++T) x[T] = ld4(pre + (size_t)rc * GPN_SEM + 16 * T + 4 * g);
  f4 y[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) y[m] = fc.b ? ld4(fc.b + c0 + 64 * ch + 16 * m + 4 * g) : f4_zero();
  slab_store(0);
  __syncthreads();
#pragma unroll
  for (int kc = 0; kc < 8; ++kc) {
    if (kc + 1 < 8) slab_load(kc + 1);  // in flight under this chunk's MFMAs
    lin_T<4, 4>(smem + (kc & 1) * (128 * SA_LDW) + 64 * ch * SA_LDW, SA_LDW, x + 4 * kc, y, lane);
    if (kc + 1 < 8) {
      slab_store((kc + 1) & 1);  // (that buffer was last read in chunk kc - 1: every wave is past the barrier since)
      __syncthreads();
    }
  }
  if (valid) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
      st4(sem_feat + (size_t)rr * GPN_SEM + c0 + 64 * ch + 16 * m + 4 * g, f4_relu(y[m]));
  }
}

// ==================================================================================== C-ABI
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// f(std::integral_constant<int, GPN_PREC_*>) for the arithmetic mode `precision` (validated by GPN_REQUIRE_PREC):
// every GNT kernel is a template over the mode
template <class F>
static int by_precision(int precision, F&& f) {
  switch (precision) {
    case GPN_PREC_BF16X3: return f(std::integral_constant<int, GPN_PREC_BF16X3>{});
    case GPN_PREC_BF16: return f(std::integral_constant<int, GPN_PREC_BF16>{});
    default: return f(std::integral_constant<int, GPN_PREC_FP32>{});
  }
}
#define REQ_LIN(l, name) \
  GPN_REQUIRE((l).w && aligned16((l).w), "%s: weight pointer null or not 16-byte aligned", name)

extern "C" int gpn_posenc(const float* pts, const float* ray_d, int R, int S, float* emb_pt,
                          float* emb_ray, gpn_stream_t stream) {
  if (R == 0) return GPN_OK;  // empty batch (pointers may be null)
  GPN_REQUIRE(pts && ray_d && emb_pt && emb_ray, "gpn_posenc: null pointer");
  GPN_REQUIRE(R >= 0 && S >= 1, "gpn_posenc: bad shape");
  const long long N = (long long)R * S;
  if (N == 0) return GPN_OK;
  GpnStageTimer timer_k_posenc_(GPN_ST_POSENC, (hipStream_t)stream);
  hipLaunchKernelGGL(k_posenc, dim3(gpn_ceil_div((N + R) * 32, 256)), dim3(256), 0,
                     (hipStream_t)stream, pts, ray_d, S, N, (long long)R, emb_pt, emb_ray);
  GPN_LAUNCH_CHECK("k_posenc");
  return GPN_OK;
}

extern "C" int gpn_rgbfeat(const gpn_linear_t* fc0, const gpn_linear_t* fc2, const float* rgb_feat,
                           int N, int V, int in_ch, int in_ld, const float* mask, float* feat,
                           float* q0, int precision, gpn_stream_t stream) {
  GPN_REQUIRE_PREC(precision, "gpn_rgbfeat");
  if (N == 0) return GPN_OK;  // empty batch (pointers may be null)
  GPN_REQUIRE(fc0 && fc2 && rgb_feat && feat && q0, "gpn_rgbfeat: null pointer");
  REQ_LIN(*fc2, "gpn_rgbfeat.fc2");
  GPN_REQUIRE(fc0->w, "gpn_rgbfeat: null weight");
  GPN_REQUIRE(N >= 0 && V >= 1 && V <= GPN_MAX_VIEWS && in_ch >= 1 && in_ch <= 48,
              "gpn_rgbfeat: need 1<=V<=%d, in_ch<=48 (got V=%d in_ch=%d)", GPN_MAX_VIEWS, V, in_ch);
  GPN_REQUIRE(aligned16(feat) && aligned16(q0), "gpn_rgbfeat: outputs must be 16-byte aligned");
  if (in_ld == 0) in_ld = in_ch;
  GPN_REQUIRE(in_ld >= in_ch && in_ld <= 48, "gpn_rgbfeat: in_ld must be 0 or in [in_ch,48] (got %d)", in_ld);
  GPN_REQUIRE((in_ld & 3) != 0 || aligned16(rgb_feat),
              "gpn_rgbfeat: rgb_feat with rows of a multiple of 4 floats must be 16-byte aligned");
  GPN_REQUIRE_INT_RANGE(N, 1, V, "gpn_rgbfeat");
  const bool spl = precision != GPN_PREC_FP32;
  const size_t lds = sizeof(float) * (wimg_floats(spl, 64, 3, RF_LD0) + wimg_floats(spl, 64, 4, LDW) + 128);
  const int grid = persistent_grid(((long long)N + (RF_TPB / 4) - 1) / (RF_TPB / 4), GPN_MEM_BPC);
  GpnStageTimer timer_k_rgbfeat_(GPN_ST_RGBFEAT, (hipStream_t)stream);
  by_precision(precision, [&](auto m) {
    hipLaunchKernelGGL(k_rgbfeat<decltype(m)::value>, dim3(grid), dim3(RF_TPB), lds, (hipStream_t)stream, *fc0, *fc2,
                       rgb_feat, (long long)N, V, in_ch, in_ld, mask, feat, q0);
    return 0;
  });
  GPN_LAUNCH_CHECK("k_rgbfeat");
  return GPN_OK;
}

static size_t proj_rgbfeat_lds(bool spl) {
  return sizeof(float) * (wimg_floats(spl, 64, 3, RF_LD0) + wimg_floats(spl, 64, 4, LDW) + 128 +
                          GPN_MAX_VIEWS * 15) + (PR_TPB / 64) * sizeof(PrTab);
}

static int proj_rgbfeat_launch(const gpn_project_args_t* a, const gpn_linear_t* fc0,
                               const gpn_linear_t* fc2, float* feat, float* q0, const int32_t* perm, float* rdp,
                               int tiled, int precision, gpn_stream_t stream) {
  GPN_REQUIRE_PREC(precision, "gpn_project_rgbfeat");
  GPN_REQUIRE(a, "gpn_project_rgbfeat: null args");
  if (a->N == 0) return GPN_OK;  // empty batch
  GPN_REQUIRE(fc0 && fc2 && feat && q0, "gpn_project_rgbfeat: null pointer");
  GPN_REQUIRE(a->xyz && a->proj && a->src_pos && a->rgb && a->feat && a->mask,
              "gpn_project_rgbfeat: xyz, proj, src_pos, rgb, feat and mask are required");
  GPN_REQUIRE(!a->rgb_feat && !a->deep_out && !a->pix && !a->taps && !a->front,
              "gpn_project_rgbfeat: rgb_feat / deep_out / pix / taps / front are outputs of gpn_project_gather only");
  GPN_REQUIRE(a->N > 0 && a->V >= 1 && a->V <= GPN_MAX_VIEWS, "gpn_project_rgbfeat: need 1 <= V <= %d (got %d)",
              GPN_MAX_VIEWS, a->V);
  GPN_REQUIRE_INT_RANGE(a->N, 1, a->V, "gpn_project_rgbfeat");
  GPN_REQUIRE(a->feat_c == 32, "gpn_project_rgbfeat: feature maps must have 32 channels (got %d)", a->feat_c);
  GPN_REQUIRE(a->feat_h >= 1 && a->feat_w >= 1 && a->rgb_h >= 1 && a->rgb_w >= 1 && a->feat_w <= 32767 &&
                  a->feat_h <= 32767 && a->rgb_w <= 32767 && a->rgb_h <= 32767 &&
                  (long long)a->feat_h * a->feat_w * 32 < 2147483647LL &&
                  (long long)a->rgb_h * a->rgb_w * 3 < 2147483647LL,
              "gpn_project_rgbfeat: unsupported map size (features %dx%d, rgb %dx%d)", a->feat_h, a->feat_w,
              a->rgb_h, a->rgb_w);
  GPN_REQUIRE(a->img_h > 1.0f && a->img_w > 1.0f, "gpn_project_rgbfeat: bad image size");
  GPN_REQUIRE(fc0->w, "gpn_project_rgbfeat: null weight");
  REQ_LIN(*fc2, "gpn_project_rgbfeat.fc2");
  GPN_REQUIRE(aligned16(feat) && aligned16(q0) && aligned16(a->feat) && (!a->ray_diff || aligned16(a->ray_diff)),
              "gpn_project_rgbfeat: feat, q0, the feature maps and ray_diff must be 16-byte aligned");
  const bool spl = precision != GPN_PREC_FP32;
  const size_t lds = proj_rgbfeat_lds(spl);
  const int grid = persistent_grid(((long long)a->N + (PR_TPB / 4) - 1) / (PR_TPB / 4), PR_BPC);
  int rc = by_precision(precision, [&](auto m) {
    constexpr int M = decltype(m)::value;
    int rc_ = set_lds_limit(reinterpret_cast<const void*>(k_proj_rgbfeat<M>), lds);
    if (rc_) return rc_;
    GpnStageTimer timer_(GPN_ST_PROJ_RGBFEAT, (hipStream_t)stream);
    hipLaunchKernelGGL(k_proj_rgbfeat<M>, dim3(grid), dim3(PR_TPB), lds, (hipStream_t)stream, *a, *fc0, *fc2,
                       feat, q0, perm, rdp, tiled);
    return 0;
  });
  if (rc) return rc;
  GPN_LAUNCH_CHECK("k_proj_rgbfeat");
  return GPN_OK;
}

extern "C" int gpn_project_rgbfeat(const gpn_project_args_t* a, const gpn_linear_t* fc0,
                                   const gpn_linear_t* fc2, float* feat, float* q0, int precision,
                                   gpn_stream_t stream) {
  return proj_rgbfeat_launch(a, fc0, fc2, feat, q0, nullptr, nullptr, 0, precision, stream);
}

// masks of every (point, view) pair: the pre-pass of the tile-ordered fused projector
static int proj_mask_launch(const gpn_project_args_t* a, gpn_stream_t stream) {
  GPN_REQUIRE(a && a->xyz && a->proj && a->mask, "k_proj_mask: null pointer");
  GPN_REQUIRE(a->N > 0 && a->V >= 1 && a->V <= GPN_MAX_VIEWS, "k_proj_mask: bad shape");
  const int grid = (int)std::min<long long>(((long long)a->N + 255) / 256, 65535);
  GpnStageTimer timer_(GPN_ST_PROJ_RGBFEAT, (hipStream_t)stream);
  hipLaunchKernelGGL(k_proj_mask, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
  GPN_LAUNCH_CHECK("k_proj_mask");
  return GPN_OK;
}

extern "C" size_t gpn_vis_order_scratch_bytes(void) { return 2 * VIS_BINS * sizeof(int); }

extern "C" int gpn_vis_order(const float* mask, int N, int V, int32_t* perm, void* scratch,
                             gpn_stream_t stream) {
  if (N == 0) return GPN_OK;
  GPN_REQUIRE(mask && perm && scratch, "gpn_vis_order: null pointer");
  GPN_REQUIRE(N > 0 && V >= 1 && V <= GPN_MAX_VIEWS, "gpn_vis_order: bad shape");
  GPN_REQUIRE_INT_RANGE(N, 1, V, "gpn_vis_order");
  int* hist = static_cast<int*>(scratch);
  int* cursor = hist + VIS_BINS;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(hist, 0, VIS_BINS * sizeof(int), st) != hipSuccess)
    return gpn_set_error(GPN_E_LAUNCH, "gpn_vis_order: hipMemsetAsync failed");
  const int blocks = gpn_ceil_div(N, VIS_CHUNK);
  GpnStageTimer timer(GPN_ST_VIS_ORDER, st);
  hipLaunchKernelGGL(k_vis_hist, dim3(blocks), dim3(256), 0, st, mask, N, V, hist);
  hipLaunchKernelGGL(k_vis_scan, dim3(1), dim3(1024), 0, st, hist, cursor, V);
  hipLaunchKernelGGL(k_vis_scatter, dim3(blocks), dim3(256), 0, st, mask, N, V, cursor, perm);
  GPN_LAUNCH_CHECK("k_vis_order");
  return GPN_OK;
}

extern "C" int gpn_view_fold(const gpn_view_layer_t* L, float* out, gpn_stream_t stream) {
  GPN_REQUIRE(L && out, "gpn_view_fold: null pointer");
  GPN_REQUIRE(L->k.w && L->v.w && L->pos2.w && L->attn0.w, "gpn_view_fold: null weight");
  hipLaunchKernelGGL(k_view_fold, dim3(gpn_ceil_div(GPN_VIEW_FOLD_FLOATS, 256)), dim3(256), 0,
                     (hipStream_t)stream, *L, out);
  GPN_LAUNCH_CHECK("k_view_fold");
  return GPN_OK;
}

static int view_attn_launch(const gpn_view_layer_t* L, const float* q_in, const float* feat,
                            const float* ray_diff, int layout, const float* mask, const int32_t* perm, int N,
                            int V, float* q_out, float* hidden, int hidden_ld, float* stats,
                            int stats_ld, int precision, gpn_stream_t stream, float* ff_out = nullptr,
                            const gpn_linear_t* qfc0 = nullptr, const gpn_linear_t* qfc2 = nullptr,
                            const float* emb_pt = nullptr, const float* emb_ray = nullptr, int S = 1,
                            float* qp_out = nullptr) {
  GPN_REQUIRE_PREC(precision, "gpn_view_attn");
  if (N == 0) return GPN_OK;  // empty batch (pointers may be null)
  GPN_REQUIRE(L && q_in && feat && ray_diff && mask && q_out, "gpn_view_attn: null pointer");
  REQ_LIN(L->q, "gpn_view_attn.q");
  REQ_LIN(L->k, "gpn_view_attn.k");
  REQ_LIN(L->v, "gpn_view_attn.v");
  REQ_LIN(L->out, "gpn_view_attn.out");
  GPN_REQUIRE(L->pos0.w && L->pos2.w && L->attn0.w && L->attn2.w && L->attn_norm_w &&
                  L->attn_norm_b, "gpn_view_attn: null weight");
  GPN_REQUIRE(N >= 0 && V >= 1 && V <= GPN_MAX_VIEWS, "gpn_view_attn: need 1<=V<=%d (got %d)",
              GPN_MAX_VIEWS, V);
  GPN_REQUIRE(aligned16(q_in) && aligned16(feat) && aligned16(q_out) && aligned16(hidden) &&
                  aligned16(stats), "gpn_view_attn: tensors must be 16-byte aligned");
  GPN_REQUIRE(!hidden || (hidden_ld >= 8 && hidden_ld % 4 == 0),
              "gpn_view_attn: hidden_ld must be a multiple of 4 and >= 8 (got %d)", hidden_ld);
  GPN_REQUIRE(!stats || (stats_ld >= C64 && stats_ld % 4 == 0),
              "gpn_view_attn: stats_ld must be a multiple of 4 and >= %d (got %d)", C64, stats_ld);
  GPN_REQUIRE_INT_RANGE(N, 1, V, "gpn_view_attn");
  const bool spl = precision != GPN_PREC_FP32;
  const bool fold = L->folded != nullptr;
  // the launch's dynamic LDS must hold the largest image of its phases: attention weights + DMA ring, the
  // feed-forward weights, the q-MLP weights (the later phases re-stage from the base of the same allocation)
  constexpr size_t LDS_CAP = 160 * 1024;
  constexpr size_t va_lds_f = sizeof(float) * view_lds_floats(false) + (VA_TPB / 64) * 2 * VA_SLOT_BYTES;
  constexpr size_t va_lds_s = sizeof(float) * view_lds_floats(true) + (VA_TPB / 64) * 2 * VA_SLOT_BYTES;
  static_assert(va_lds_f <= LDS_CAP && va_lds_s <= LDS_CAP, "k_view_attn: attention image + ring exceed 160 KB");
  static_assert(sizeof(float) * ff_lds_floats(false) <= LDS_CAP && sizeof(float) * ff_lds_floats(true) <= LDS_CAP,
                "k_view_attn: feed-forward phase image exceeds 160 KB");
  static_assert(sizeof(float) * qp_lds_floats(false) <= LDS_CAP && sizeof(float) * qp_lds_floats(true) <= LDS_CAP,
                "k_view_attn: q-MLP phase image exceeds 160 KB");
  size_t lds = spl ? va_lds_s : va_lds_f;
  if (ff_out != nullptr) lds = std::max(lds, sizeof(float) * (size_t)ff_lds_floats(spl));
  if (qp_out != nullptr) lds = std::max(lds, sizeof(float) * (size_t)qp_lds_floats(spl));
  const bool tiled = VA_TILED_DMA && fold && layout == 3;  // the fused path: scalar-base DMA addressing
  GPN_REQUIRE(ff_out == nullptr || (L->ff1.w && L->ff2.w && L->ff_norm_w && L->ff_norm_b),
              "gpn_view_attn: fused feed-forward without its weights");
  GPN_REQUIRE(qp_out == nullptr || (ff_out && qfc0 && qfc2 && qfc0->w && qfc2->w && emb_pt && emb_ray && S >= 1),
              "gpn_view_attn: fused q MLP needs the feed-forward phase, its weights and the encodings");
  const gpn_linear_t q0_ = qp_out ? *qfc0 : gpn_linear_t{}, q2_ = qp_out ? *qfc2 : gpn_linear_t{};
  const int grid = persistent_grid(((long long)N + (VA_TPB / 4) - 1) / (VA_TPB / 4));
  const dim3 g3(grid), b3(VA_TPB);
  hipStream_t st = (hipStream_t)stream;
  auto go = [&](auto kernel) {
    int rc_ = set_lds_limit(reinterpret_cast<const void*>(kernel), lds);
    if (rc_) return rc_;
    GpnStageTimer timer_k_view_attn_(GPN_ST_VIEW_ATTN, st);
    hipLaunchKernelGGL(kernel, g3, b3, lds, st, *L, q_in, feat, ray_diff, mask, perm, (long long)N, V, q_out, hidden,
                       hidden_ld, stats, stats_ld, layout, ff_out, q0_, q2_, emb_pt, emb_ray, S, qp_out);
    return 0;
  };
  int rc = by_precision(precision, [&](auto m) {
    constexpr int M = decltype(m)::value;
    if (tiled) return go(k_view_attn<M, true, true>);
    if (fold) return go(k_view_attn<M, true>);
    return go(k_view_attn<M, false>);
  });
  if (rc) return rc;
  GPN_LAUNCH_CHECK("k_view_attn");
  return GPN_OK;
}

extern "C" int gpn_view_attn(const gpn_view_layer_t* L, const float* q_in, const float* feat,
                             const float* ray_diff, const float* mask, const int32_t* perm, int N,
                             int V, float* q_out, float* hidden, int hidden_ld, float* stats,
                             int stats_ld, int precision, gpn_stream_t stream) {
  return view_attn_launch(L, q_in, feat, ray_diff, 0, mask, perm, N, V, q_out, hidden, hidden_ld, stats, stats_ld,
                          precision, stream);
}

// ray_diff re-laid for k_view_attn (k_rd_pack); `rd` given: gather, else computed from the geometry
static int rd_pack_launch(const float* rd, const float* xyz, const float* qry_pos, const float* src_pos,
                          const int32_t* perm, int N, int V, float* out, gpn_stream_t stream) {
  GPN_REQUIRE(out && (rd || (xyz && qry_pos && src_pos)), "k_rd_pack: null pointer");
  const float qx = qry_pos ? qry_pos[0] : 0.f, qy = qry_pos ? qry_pos[1] : 0.f, qz = qry_pos ? qry_pos[2] : 0.f;
  GPN_REQUIRE(aligned16(rd) && aligned16(out), "k_rd_pack: ray_diff must be 16-byte aligned");
  const long long total = (((long long)N + 15) >> 4) * V * 16;
  const int grid = (int)std::min<long long>((total + 255) / 256, 65535);
  GpnStageTimer timer_(GPN_ST_VIS_ORDER, (hipStream_t)stream);
  hipLaunchKernelGGL(k_rd_pack, dim3(grid), dim3(256), 0, (hipStream_t)stream, rd, xyz, qx, qy, qz, src_pos, perm,
                     (long long)N, V, out);
  GPN_LAUNCH_CHECK("k_rd_pack");
  return GPN_OK;
}

extern "C" int gpn_ff(const float* norm_w, const float* norm_b, const gpn_linear_t* fc1,
                      const gpn_linear_t* fc2, const float* x_in, int N, float* x_out,
                      int precision, gpn_stream_t stream) {
  GPN_REQUIRE_PREC(precision, "gpn_ff");
  if (N == 0) return GPN_OK;  // empty batch (pointers may be null)
  GPN_REQUIRE(norm_w && norm_b && fc1 && fc2 && x_in && x_out, "gpn_ff: null pointer");
  REQ_LIN(*fc1, "gpn_ff.fc1");
  REQ_LIN(*fc2, "gpn_ff.fc2");
  GPN_REQUIRE(aligned16(x_in) && aligned16(x_out), "gpn_ff: tensors must be 16-byte aligned");
  GPN_REQUIRE(N >= 0, "gpn_ff: bad N");
  const bool spl = precision != GPN_PREC_FP32;
  const size_t lds = sizeof(float) * ff_lds_floats(spl);
  const int grid = persistent_grid(((long long)N + (FF_TPB / 4) - 1) / (FF_TPB / 4));
  int rc = 0;
  if (precision == GPN_PREC_BF16X3 && GPN_FF32) {
    rc = set_lds_limit(reinterpret_cast<const void*>(k_ff32), lds);
    if (rc) return rc;
    GpnStageTimer timer_k_ff_(GPN_ST_FF, (hipStream_t)stream);
    const int grid32 = persistent_grid(((long long)N + (FF_TPB / 2) - 1) / (FF_TPB / 2));
    hipLaunchKernelGGL(k_ff32, dim3(grid32), dim3(FF_TPB), lds, (hipStream_t)stream, norm_w, norm_b, *fc1,
                       *fc2, x_in, (long long)N, x_out);
  } else {
    rc = by_precision(precision, [&](auto m) {
      constexpr int M = decltype(m)::value;
      int rc_ = set_lds_limit(reinterpret_cast<const void*>(k_ff<M>), lds);
      if (rc_) return rc_;
      GpnStageTimer timer_k_ff_(GPN_ST_FF, (hipStream_t)stream);
      hipLaunchKernelGGL(k_ff<M>, dim3(grid), dim3(FF_TPB), lds, (hipStream_t)stream, norm_w, norm_b, *fc1,
                         *fc2, x_in, (long long)N, x_out);
      return 0;
    });
    if (rc) return rc;
  }
  GPN_LAUNCH_CHECK("k_ff");
  return GPN_OK;
}

extern "C" int gpn_qpos(const gpn_linear_t* fc0, const gpn_linear_t* fc2, const float* q_in,
                        const float* emb_pt, const float* emb_ray, int N, int S, float* q_out,
                        int precision, gpn_stream_t stream) {
  GPN_REQUIRE_PREC(precision, "gpn_qpos");
  if (N == 0) return GPN_OK;  // empty batch (pointers may be null)
  GPN_REQUIRE(fc0 && fc2 && q_in && emb_pt && emb_ray && q_out, "gpn_qpos: null pointer");
  GPN_REQUIRE(fc0->w, "gpn_qpos: null weight");
  REQ_LIN(*fc2, "gpn_qpos.fc2");
  GPN_REQUIRE(aligned16(q_in) && aligned16(emb_pt) && aligned16(emb_ray) && aligned16(q_out),
              "gpn_qpos: tensors must be 16-byte aligned");
  GPN_REQUIRE(N >= 0 && S >= 1 && N % S == 0, "gpn_qpos: N must be a multiple of S >= 1 (got N=%d S=%d)", N, S);
  const bool spl = precision != GPN_PREC_FP32;
  const size_t lds = sizeof(float) * qp_lds_floats(spl);
  const int grid = persistent_grid(((long long)N + (QP_TPB / 4) - 1) / (QP_TPB / 4));
  int rc = by_precision(precision, [&](auto m) {
    constexpr int M = decltype(m)::value;
    int rc_ = set_lds_limit(reinterpret_cast<const void*>(k_qpos<M>), lds);
    if (rc_) return rc_;
    GpnStageTimer timer_k_qpos_(GPN_ST_QPOS, (hipStream_t)stream);
    hipLaunchKernelGGL(k_qpos<M>, dim3(grid), dim3(QP_TPB), lds, (hipStream_t)stream, *fc0, *fc2, q_in, emb_pt,
                       emb_ray, (long long)N, S, q_out);
    return 0;
  });
  if (rc) return rc;
  GPN_LAUNCH_CHECK("k_qpos");
  return GPN_OK;
}

template <int NT, int SPL>
static int launch_ray_attn_p(const gpn_ray_layer_t* L, const float* x_in, int R, int S, float* x_out,
                             float* abar, int abar_ld, float* weights, hipStream_t stream, float* ff_out_req) {
  // S > 64: k_ray_attn<5..8> has no registers to spare (the extra phase costs spills: measured on the
  // 96 + 32-sample stress shape, 154 K vs 152 K rays/s in fp32): the feed-forward block stays a launch of
  // its own there
  float* ff_out = NT <= RAY_FF_MAXNT ? ff_out_req : nullptr;
  if (ff_out_req != nullptr && ff_out == nullptr) {
    int rc0 = launch_ray_attn_p<NT, SPL>(L, x_in, R, S, x_out, abar, abar_ld, weights, stream, nullptr);
    if (rc0) return rc0;
    return gpn_ff(L->ff_norm_w, L->ff_norm_b, &L->ff1, &L->ff2, x_out, R * S, ff_out_req,
                  SPL, (gpn_stream_t)stream);
  }
  // (ff_out: the feed-forward phase re-stages LDS with its own weights: the launch needs the larger image)
  const size_t lds_ff = ff_out ? sizeof(float) * ff_lds_floats(SPL != 0) : 0;
#if GPN_RAY1
  // one wave per ray, keys / values in registers (k_ray1).  Per launch at R = 4096 (rocprofv3): S = 48 fp32
  // 111 -> 89 us, bf16x3 59 -> 52.5 us; S = 64 (the 128 key / value registers leave the compiler no room)
  // fp32 124.5 -> 125, bf16x3 68.5 -> 72 us: k_ray_attn stays the kernel of S > 48
  if constexpr (NT <= RAY1_MAXNT) {
    const size_t lds1 = std::max(sizeof(float) * ray1_lds_floats(SPL != 0), lds_ff);
    int rc1 = set_lds_limit(reinterpret_cast<const void*>(k_ray1<NT, SPL>), lds1);
    if (rc1) return rc1;
    const int grid1 = persistent_grid(((long long)R + RAY1_WAVES - 1) / RAY1_WAVES);
    GpnStageTimer timer_k_ray1_(GPN_ST_RAY_ATTN, stream);
    hipLaunchKernelGGL((k_ray1<NT, SPL>), dim3(grid1), dim3(RAY1_WAVES * 64), lds1, stream, *L, x_in, R, S, x_out,
                       abar, abar_ld, weights, ff_out);
    GPN_LAUNCH_CHECK("k_ray1");
    return GPN_OK;
  }
#endif
  using Cfg = RayCfg<NT>;
  const size_t lds = std::max(sizeof(float) * Cfg::LDS_FLOATS, lds_ff);
  int rc = set_lds_limit(reinterpret_cast<const void*>(k_ray_attn<NT, SPL>), lds);
  if (rc) return rc;
  const int grid = persistent_grid(((long long)R + Cfg::RPW - 1) / Cfg::RPW);
  GpnStageTimer timer_k_ray_attn_(GPN_ST_RAY_ATTN, (hipStream_t)stream);
  hipLaunchKernelGGL((k_ray_attn<NT, SPL>), dim3(grid), dim3(Cfg::THREADS), lds, stream, *L, x_in, R, S,
                     x_out, abar, abar_ld, weights, ff_out);
  GPN_LAUNCH_CHECK("k_ray_attn");
  return GPN_OK;
}

template <int NT>
static int launch_ray_attn(const gpn_ray_layer_t* L, const float* x_in, int R, int S, float* x_out,
                           float* abar, int abar_ld, float* weights, int precision, hipStream_t stream,
                           float* ff_out) {
  return by_precision(precision, [&](auto m) {
    return launch_ray_attn_p<NT, decltype(m)::value>(L, x_in, R, S, x_out, abar, abar_ld, weights, stream, ff_out);
  });
}

static int ray_attn_launch(const gpn_ray_layer_t* L, const float* x_in, int R, int S, float* x_out,
                           float* abar, int abar_ld, float* weights, int precision,
                           gpn_stream_t stream, float* ff_out) {
  GPN_REQUIRE_PREC(precision, "gpn_ray_attn");
  if (R == 0) return GPN_OK;  // empty batch (pointers may be null)
  GPN_REQUIRE(L && x_in && x_out, "gpn_ray_attn: null pointer");
  REQ_LIN(L->q, "gpn_ray_attn.q");
  REQ_LIN(L->k, "gpn_ray_attn.k");
  REQ_LIN(L->v, "gpn_ray_attn.v");
  REQ_LIN(L->out, "gpn_ray_attn.out");
  GPN_REQUIRE(L->attn_norm_w && L->attn_norm_b, "gpn_ray_attn: null weight");
  GPN_REQUIRE(R >= 0 && S >= 1 && S <= GPN_MAX_SAMPLES,
              "gpn_ray_attn: samples per ray must be in [1,%d] (got %d)", GPN_MAX_SAMPLES, S);
  GPN_REQUIRE(aligned16(x_in) && aligned16(x_out), "gpn_ray_attn: tensors must be 16-byte aligned");
  GPN_REQUIRE(!abar || abar_ld >= 4, "gpn_ray_attn: abar_ld must be >= 4 (got %d)", abar_ld);
  GPN_REQUIRE(ff_out == nullptr || (L->ff1.w && L->ff2.w && L->ff_norm_w && L->ff_norm_b),
              "gpn_ray_attn: fused feed-forward without its weights");
  GPN_REQUIRE_INT_RANGE(R, S, 1, "gpn_ray_attn");
  hipStream_t st = (hipStream_t)stream;
  switch ((S + 15) / 16) {
    case 1: return launch_ray_attn<1>(L, x_in, R, S, x_out, abar, abar_ld, weights, precision, st, ff_out);
    case 2: return launch_ray_attn<2>(L, x_in, R, S, x_out, abar, abar_ld, weights, precision, st, ff_out);
    case 3: return launch_ray_attn<3>(L, x_in, R, S, x_out, abar, abar_ld, weights, precision, st, ff_out);
    case 4: return launch_ray_attn<4>(L, x_in, R, S, x_out, abar, abar_ld, weights, precision, st, ff_out);
    case 5: return launch_ray_attn<5>(L, x_in, R, S, x_out, abar, abar_ld, weights, precision, st, ff_out);
    case 6: return launch_ray_attn<6>(L, x_in, R, S, x_out, abar, abar_ld, weights, precision, st, ff_out);
    case 7: return launch_ray_attn<7>(L, x_in, R, S, x_out, abar, abar_ld, weights, precision, st, ff_out);
    default: return launch_ray_attn<8>(L, x_in, R, S, x_out, abar, abar_ld, weights, precision, st, ff_out);
  }
}

extern "C" int gpn_ray_attn(const gpn_ray_layer_t* L, const float* x_in, int R, int S, float* x_out,
                            float* abar, int abar_ld, float* weights, int precision,
                            gpn_stream_t stream) {
  return ray_attn_launch(L, x_in, R, S, x_out, abar, abar_ld, weights, precision, stream, nullptr);
}

extern "C" int gpn_sem_aggregate(const gpn_sem_args_t* a, gpn_stream_t stream) {
  GPN_REQUIRE(a, "gpn_sem_aggregate: null args");
  if (a->R == 0) return GPN_OK;  // empty batch
  GPN_REQUIRE(a->R >= 0 && a->S >= 1 && a->V >= 1 && a->V <= GPN_MAX_VIEWS && a->depth >= 1 &&
                  a->depth <= GPN_MAX_DEPTH, "gpn_sem_aggregate: bad shape");
  GPN_REQUIRE_INT_RANGE(a->R, a->S, a->V, "gpn_sem_aggregate");
  GPN_REQUIRE(a->mask && a->pre, "gpn_sem_aggregate: null pointer");
  GPN_REQUIRE(a->hidden && a->stats && a->abar, "gpn_sem_aggregate: null hidden / stats / abar");
  for (int l = 0; l < a->depth; ++l)
    GPN_REQUIRE(a->attn2_w[l], "gpn_sem_aggregate: attn2_w[%d] is null", l);
  GPN_REQUIRE((a->deep_dense != nullptr) != (a->deep_map != nullptr),
              "gpn_sem_aggregate: exactly one of deep_dense / deep_map must be given");
  if (a->deep_map) {
    GPN_REQUIRE(a->xyz && a->proj && a->deep_h >= 1 && a->deep_w >= 1 && a->img_h > 1 && a->img_w > 1,
                "gpn_sem_aggregate: map mode needs xyz, proj and sizes");
    GPN_REQUIRE(aligned16(a->deep_map), "gpn_sem_aggregate: deep_map must be 16-byte aligned");
  } else {
    GPN_REQUIRE(aligned16(a->deep_dense), "gpn_sem_aggregate: deep_dense must be 16-byte aligned");
  }
  GPN_REQUIRE(aligned16(a->pre) && aligned16(a->sem3d), "gpn_sem_aggregate: outputs must be aligned");
  if (a->R == 0) return GPN_OK;
  GpnStageTimer timer_k_sem_(GPN_ST_SEM, (hipStream_t)stream);
  const int groups = gpn_ceil_div(a->R, SEM_RAYS);
  // 2-D strips per XCD (gpn_sem_args_t.ray_row_*): whole groups per strip, whole workgroups per XCD, the chunk
  // at least two rows tall (GPN_SEM_NO_STRIPS=1: developer switch for A/B measurements)
  int strip_groups = 0;
  const int ngrid = groups;
  if (a->ray_row_width >= 8 * SEM_RAYS && a->ray_row_width % (8 * SEM_RAYS) == 0 && a->ray_row_offset >= 0 &&
      a->ray_row_offset < a->ray_row_width && a->ray_row_offset % SEM_RAYS == 0 &&
      a->R >= 2 * a->ray_row_width && a->deep_map != nullptr && !getenv("GPN_SEM_NO_STRIPS"))
    strip_groups = a->ray_row_width / (8 * SEM_RAYS);
  const dim3 grid(ngrid), block(SEM_RAYS * 64);
  hipStream_t st = (hipStream_t)stream;
  switch (a->depth) {
    case 1: hipLaunchKernelGGL(k_sem<1>, grid, block, 0, st, *a, strip_groups); break;
    case 2: hipLaunchKernelGGL(k_sem<2>, grid, block, 0, st, *a, strip_groups); break;
    case 3: hipLaunchKernelGGL(k_sem<3>, grid, block, 0, st, *a, strip_groups); break;
    case 4: hipLaunchKernelGGL(k_sem<4>, grid, block, 0, st, *a, strip_groups); break;
    case 5: hipLaunchKernelGGL(k_sem<5>, grid, block, 0, st, *a, strip_groups); break;
    case 6: hipLaunchKernelGGL(k_sem<6>, grid, block, 0, st, *a, strip_groups); break;
    case 7: hipLaunchKernelGGL(k_sem<7>, grid, block, 0, st, *a, strip_groups); break;
    default: hipLaunchKernelGGL(k_sem<8>, grid, block, 0, st, *a, strip_groups); break;
  }
  GPN_LAUNCH_CHECK("k_sem");
  return GPN_OK;
}

extern "C" int gpn_heads(const gpn_gnt_weights_t* W, const float* q, const float* weights,
                         const float* z_vals, const float* pre, int R, int S, float* out,
                         float* depth, float* sem_feat, gpn_stream_t stream) {
  if (R == 0) return GPN_OK;  // empty batch (pointers may be null)
  GPN_REQUIRE(W && q && weights && out, "gpn_heads: null pointer");
  GPN_REQUIRE(W->norm_w && W->norm_b && W->rgb_fc.w, "gpn_heads: null weight");
  GPN_REQUIRE(R >= 0 && S >= 1, "gpn_heads: bad shape");
  GPN_REQUIRE(aligned16(q) && aligned16(W->norm_w) && aligned16(W->norm_b) && aligned16(W->rgb_fc.w),
              "gpn_heads: q and the norm / rgb_fc weights must be 16-byte aligned");
  if (R == 0) return GPN_OK;
  {
    GpnStageTimer timer_k_heads_rgb_(GPN_ST_HEADS, (hipStream_t)stream);
    hipLaunchKernelGGL(k_heads_rgb, dim3(gpn_ceil_div(R, 4)), dim3(256), 0, (hipStream_t)stream,
                       W->norm_w, W->norm_b, W->rgb_fc, q, weights, z_vals, R, S, out, depth);
    GPN_LAUNCH_CHECK("k_heads_rgb");
  }
  if (pre != nullptr) {
    GPN_REQUIRE(sem_feat, "gpn_heads: pre given without sem_feat");
    REQ_LIN(W->sem_adaptor, "gpn_heads.sem_adaptor");
    GPN_REQUIRE(aligned16(pre) && aligned16(sem_feat), "gpn_heads: sem tensors must be aligned");
    int rc_a = set_lds_limit(reinterpret_cast<const void*>(k_sem_adaptor), sizeof(float) * SA_LDS_FLOATS);
    if (rc_a) return rc_a;
    GpnStageTimer timer_k_sem_adaptor_(GPN_ST_HEADS, (hipStream_t)stream);
    hipLaunchKernelGGL(k_sem_adaptor, dim3(4 * gpn_ceil_div(R, 64)), dim3(512), sizeof(float) * SA_LDS_FLOATS, (hipStream_t)stream,
                       W->sem_adaptor, pre, R, sem_feat);
    GPN_LAUNCH_CHECK("k_sem_adaptor");
  }
  return GPN_OK;
}

// ------------------------------------------------------------------------------------ whole forward
static size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct GntWs {
  size_t emb, emb_ray, feat, q, hidden, stats, abar, weights, pre, perm, vis_scratch, rdiff, mask, total;
};
static GntWs gnt_ws_layout(int R, int S, int V, int depth) {
  const size_t N = (size_t)R * S, f = sizeof(float);
  GntWs w;
  size_t o = 0;
  w.emb = o, o += align_up(N * GPN_EMB_HALF * f);
  w.emb_ray = o, o += align_up((size_t)R * GPN_EMB_HALF * f);
  w.feat = o, o += align_up(((N + 15) / 16 * 16) * V * C64 * f);  // (whole tiles: the fused path writes tile-major)
  w.q = o, o += align_up(N * C64 * f);
  w.hidden = o, o += align_up(N * V * depth * 8 * f);    // [N,V,depth,8]
  w.stats = o, o += align_up(N * depth * C64 * f);        // [N,depth,64]
  w.abar = o, o += align_up(N * depth * 4 * f);           // [R,S,depth,4]
  w.weights = o, o += align_up(N * f);
  w.pre = o, o += align_up((size_t)R * GPN_SEM * f);
  w.perm = o, o += align_up(2 * N * sizeof(int));  // points in slot order | their visibility bits
  w.vis_scratch = o, o += align_up(2 * VIS_BINS * sizeof(int));
  w.rdiff = o, o += align_up(((N + 15) / 16 * 16) * V * 4 * f);  // ray_diff in k_view_attn's tile order (k_rd_pack)
  w.mask = o, o += align_up(N * V * f);
  w.total = o;
  return w;
}

extern "C" size_t gpn_gnt_workspace_bytes(int R, int S, int V, int depth) {
  if (R < 0 || S < 1 || V < 1 || depth < 1) return 0;
  return gnt_ws_layout(R, S, V, depth).total;
}

// proj == nullptr: gpn_gnt_forward (rgb_feat / ray_diff / mask given).  proj != nullptr: gpn_render_pass --
// the fused projector + rgbfeat kernel produces feat, q0, ray_diff and mask from the source maps.
static int gnt_forward_impl(const gpn_gnt_weights_t* W, const gpn_project_args_t* proj, const float* rgb_feat,
                            int rgb_feat_ld, const float* ray_diff, const float* mask, const float* pts,
                            const float* ray_d, const float* z_vals, int R, int S, int V,
                            gpn_sem_args_t* sem, float* out, float* depth, float* sem_feat,
                            void* workspace, size_t workspace_bytes, gpn_stream_t stream) {
  if (R == 0) return GPN_OK;  // empty batch (pointers may be null)
  GPN_REQUIRE(W && (proj || (rgb_feat && ray_diff && mask)) && pts && ray_d && out && workspace,
              "gpn_gnt_forward: null pointer");
  const int D = W->trans_depth;
  GPN_REQUIRE(D >= 1 && D <= GPN_MAX_DEPTH, "gpn_gnt_forward: trans_depth must be in [1,%d]",
              GPN_MAX_DEPTH);
  GPN_REQUIRE(R >= 0 && S >= 1 && S <= GPN_MAX_SAMPLES && V >= 1 && V <= GPN_MAX_VIEWS,
              "gpn_gnt_forward: unsupported shape R=%d S=%d V=%d", R, S, V);
  GPN_REQUIRE_INT_RANGE(R, S, V, "gpn_gnt_forward");
  const int prec = W->precision;
  GPN_REQUIRE_PREC(prec, "gpn_gnt_forward (gpn_gnt_weights_t.precision)");
  const GntWs L = gnt_ws_layout(R, S, V, D);
  GPN_REQUIRE(workspace_bytes >= L.total, "gpn_gnt_forward: workspace too small (%zu < %zu)",
              workspace_bytes, L.total);
  GPN_REQUIRE(aligned16(workspace), "gpn_gnt_forward: workspace must be 16-byte aligned");
  if (sem) GPN_REQUIRE(sem_feat, "gpn_gnt_forward: sem given without sem_feat output");
  if (R == 0) return GPN_OK;
  char* ws = static_cast<char*>(workspace);
  const size_t N = (size_t)R * S;
  float* emb = reinterpret_cast<float*>(ws + L.emb);
  float* emb_ray = reinterpret_cast<float*>(ws + L.emb_ray);
  float* feat = reinterpret_cast<float*>(ws + L.feat);
  float* q = reinterpret_cast<float*>(ws + L.q);
  float* wts = reinterpret_cast<float*>(ws + L.weights);
  float* pre = reinterpret_cast<float*>(ws + L.pre);
  int32_t* perm = reinterpret_cast<int32_t*>(ws + L.perm);
  int rc;
  gpn_project_args_t pa_;
  if (proj) {
    gpn_project_args_t& pa = pa_;
    pa = *proj;
    pa.xyz = pts, pa.N = (int)N;
    // (pa.ray_diff stays null unless the caller wants the [N,V,4] array: k_rd_pack computes its own)
    if (!pa.mask) pa.mask = reinterpret_cast<float*>(ws + L.mask);
    GPN_REQUIRE(pa.V == V && pa.feat_c == W->in_feat_ch, "gpn_render_pass: proj.V / proj.feat_c do not match");
    // masks -> visibility order -> the fused kernel walks the points in that order and writes the
    // feature rows and ray_diff of every (tile, view) pair as the contiguous blocks k_view_attn reads
    if ((rc = proj_mask_launch(&pa, stream))) return rc;
    mask = pa.mask;
  }
  if ((rc = gpn_vis_order(mask, (int)N, V, perm, ws + L.vis_scratch, stream))) return rc;
  float* rdp = reinterpret_cast<float*>(ws + L.rdiff);
  if (proj) {
    if ((rc = proj_rgbfeat_launch(&pa_, &W->rgbfeat0, &W->rgbfeat2, feat, q, perm, rdp, 1, prec, stream))) return rc;
  } else if ((rc = rd_pack_launch(ray_diff, nullptr, nullptr, nullptr, perm, (int)N, V, rdp, stream))) {
    return rc;
  }
  const int va_layout = proj ? 3 : 1;
  if ((rc = gpn_posenc(pts, ray_d, R, S, emb, emb_ray, stream))) return rc;
  // (mask given: the feature rows of masked pairs, which gpn_view_attn never uses, are not stored)
  if (!proj &&
      (rc = gpn_rgbfeat(&W->rgbfeat0, &W->rgbfeat2, rgb_feat, (int)N, V, 3 + W->in_feat_ch, rgb_feat_ld,
                        mask, feat, q, prec, stream)))
    return rc;
  const bool fuse_qp = GPN_FUSE_VIEW_FF && GPN_FUSE_VIEW_QP;  // the q MLP as a third phase of the view launch
  for (int l = 0; l < D; ++l) {
    // side outputs for the semantic aggregation, interleaved over the layers
    float* hd = sem ? reinterpret_cast<float*>(ws + L.hidden) + l * 8 : nullptr;
    float* st = sem ? reinterpret_cast<float*>(ws + L.stats) + l * C64 : nullptr;
    float* ab = sem ? reinterpret_cast<float*>(ws + L.abar) + l * 4 : nullptr;
    const gpn_view_layer_t* vl = &W->view[l];
    const gpn_ray_layer_t* rl = &W->ray[l];
    // the view stage's feed-forward block runs as the second phase of the attention launch
    if ((rc = view_attn_launch(vl, q, feat, rdp, va_layout, mask, perm, (int)N, V, q, hd, D * 8, st, D * C64, prec, stream,
                               GPN_FUSE_VIEW_FF ? q : nullptr, &W->qfc0[l], &W->qfc2[l], emb, emb_ray, S,
                               fuse_qp && l % 2 == 0 ? q : nullptr)))
      return rc;
    if (!GPN_FUSE_VIEW_FF &&
        (rc = gpn_ff(vl->ff_norm_w, vl->ff_norm_b, &vl->ff1, &vl->ff2, q, (int)N, q, prec, stream)))
      return rc;
    if (l % 2 == 0 && !fuse_qp)
      if ((rc = gpn_qpos(&W->qfc0[l], &W->qfc2[l], q, emb, emb_ray, (int)N, S, q, prec, stream))) return rc;
    if ((rc = ray_attn_launch(rl, q, R, S, q, ab, D * 4, l == D - 1 ? wts : nullptr, prec, stream,
                              GPN_FUSE_RAY_FF ? q : nullptr)))
      return rc;
    if (!GPN_FUSE_RAY_FF &&
        (rc = gpn_ff(rl->ff_norm_w, rl->ff_norm_b, &rl->ff1, &rl->ff2, q, (int)N, q, prec, stream)))
      return rc;
  }
  if (sem) {
    sem->R = R, sem->S = S, sem->V = V, sem->depth = D;
    sem->hidden = reinterpret_cast<float*>(ws + L.hidden);
    for (int l = 0; l < D; ++l) sem->attn2_w[l] = W->view[l].attn2.w, sem->attn2_b[l] = W->view[l].attn2.b;
    sem->stats = reinterpret_cast<float*>(ws + L.stats);
    sem->abar = reinterpret_cast<float*>(ws + L.abar);
    sem->mask = mask;
    sem->pre = pre;
    if ((rc = gpn_sem_aggregate(sem, stream))) return rc;
  }
  return gpn_heads(W, q, wts, z_vals, sem ? pre : nullptr, R, S, out, depth, sem_feat, stream);
}

extern "C" int gpn_gnt_forward(const gpn_gnt_weights_t* W, const float* rgb_feat, int rgb_feat_ld,
                               const float* ray_diff, const float* mask, const float* pts,
                               const float* ray_d, const float* z_vals, int R, int S, int V,
                               gpn_sem_args_t* sem, float* out, float* depth, float* sem_feat,
                               void* workspace, size_t workspace_bytes, gpn_stream_t stream) {
  return gnt_forward_impl(W, nullptr, rgb_feat, rgb_feat_ld, ray_diff, mask, pts, ray_d, z_vals, R, S, V, sem,
                          out, depth, sem_feat, workspace, workspace_bytes, stream);
}

extern "C" int gpn_render_pass(const gpn_gnt_weights_t* W, const gpn_project_args_t* proj, const float* pts,
                               const float* ray_d, const float* z_vals, int R, int S, gpn_sem_args_t* sem,
                               float* out, float* depth, float* sem_feat, void* workspace,
                               size_t workspace_bytes, gpn_stream_t stream) {
  if (R == 0) return GPN_OK;
  GPN_REQUIRE(proj, "gpn_render_pass: null projector arguments");
  return gnt_forward_impl(W, proj, nullptr, 0, nullptr, nullptr, pts, ray_d, z_vals, R, S, proj->V, sem, out,
                          depth, sem_feat, workspace, workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------ allocation-free forward
// gpn_gnt_forward on a workspace that lives in the library's own device image instead of being
// allocated by the caller (header: gpn_gnt_forward_static).  `scratch` is a host-side stand-in for the
// array: its device address differs per GPU and is looked up once per device with
// hipGetSymbolAddress; every use below goes through that pointer.
static __device__ __align__(256) unsigned char scratch[GPN_STATIC_WS_BYTES];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
void* g_scratch_ptr[kMaxDevices] = {};  // per device: that GPU's copy of `scratch`
}  // namespace

// device address of `scratch` on the current device (first call on a device: one symbol lookup)
static int static_scratch(void** out) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices)
    return gpn_set_error(GPN_E_LAUNCH, "gpn_gnt_forward_static: no current HIP device (or index >= %d)",
                         kMaxDevices);
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (g_scratch_ptr[dev] == nullptr) {
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || p == nullptr)
      return gpn_set_error(GPN_E_LAUNCH, "gpn_gnt_forward_static: hipGetSymbolAddress(scratch) on device %d: %s",
                           dev, hipGetErrorString(e));
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return GPN_OK;
}

extern "C" int gpn_gnt_forward_static(const gpn_gnt_weights_t* W, const float* rgb_feat, int rgb_feat_ld,
                                      const float* ray_diff, const float* mask, const float* pts,
                                      const float* ray_d, const float* z_vals, int R, int S, int V,
                                      gpn_sem_args_t* sem, float* out, float* depth, float* sem_feat,
                                      gpn_stream_t stream) {
  if (R == 0) return GPN_OK;
  GPN_REQUIRE(W, "gpn_gnt_forward_static: null pointer");
  GPN_REQUIRE(R >= 0 && S >= 1 && S <= GPN_MAX_SAMPLES && V >= 1 && V <= GPN_MAX_VIEWS &&
                  W->trans_depth >= 1 && W->trans_depth <= GPN_MAX_DEPTH,
              "gpn_gnt_forward_static: unsupported shape R=%d S=%d V=%d depth=%d", R, S, V, W->trans_depth);
  const size_t need = gnt_ws_layout(R, S, V, W->trans_depth).total;
  GPN_REQUIRE(need <= (size_t)GPN_STATIC_WS_BYTES,
              "gpn_gnt_forward_static: R=%d S=%d V=%d needs a %zu-byte workspace, the static one holds %zu "
              "(GPN_STATIC_WS_BYTES): pass fewer rays per call, or use gpn_gnt_forward with a caller workspace",
              R, S, V, need, (size_t)GPN_STATIC_WS_BYTES);
  void* ws = nullptr;
  int rc = static_scratch(&ws);
  if (rc) return rc;
  // every region of the workspace is written before it is read within one forward (the one counter
  // that must start at zero, the visibility histogram, is cleared by gpn_vis_order on `stream`), so the
  // previous call's contents are harmless; two calls in flight on one GPU would share it and race
  return gpn_gnt_forward(W, rgb_feat, rgb_feat_ld, ray_diff, mask, pts, ray_d, z_vals, R, S, V, sem, out,
                         depth, sem_feat, ws, need, stream);
}
