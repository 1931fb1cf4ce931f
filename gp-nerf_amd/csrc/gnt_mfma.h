// Register-resident fp32 MFMA building blocks for the GNT kernels (gfx950, wave64).
//
// Everything is built on v_mfma_f32_16x16x4_f32 (exact fp32: a k-ordered fmaf chain, 32-cycle issue):
//   D[i][j] += sum_{kk<4} A[i][kk] * B[kk][j]
//   lane l supplies A[i = l&15][kk = l>>4] and B[kk = l>>4][j = l&15] (one float each) and holds
//   D[i = 4*(l>>4) + reg][j = l&15], reg = 0..3.
//
// "T-tile" convention (activations live transposed: channels on the accumulator rows, points on the
// lanes): a tile covers 16 channels x 16 points; lane (j = l&15, g = l>>4) holds, in one f4,
//   X[point j][channel 16*T + 4*g + r],  r = 0..3.
// That is exactly the C/D layout, so the OUTPUT of one linear layer is directly the B operand of the
// next (k-step r of tile T pairs weight column 16T+4g+r with register r): chains of per-point
// linear layers never leave the register file, and per-channel elementwise work (bias, ReLU,
// softmax over views, max over views) is lane-local.  LayerNorm over the 64 channels of a point
// needs the 16 registers of a lane plus two cross-lane adds (lanes j, j+16, j+32, j+48).
//
// Weights are read from LDS as the A operand: lane (i, g) needs W[16M+i][16T+4g .. +3] = one
// ds_read_b128 per 4 MFMAs; rows are padded by 4 floats so that the 16 rows of a tile hit distinct
// 16-byte slots.
#pragma once
#include <hip/hip_runtime.h>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f4 mfma16(float a, float b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f4 f4_zero() { return (f4){0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ f4 f4_splat(float v) { return (f4){v, v, v, v}; }
// max without the canonicalising pre-pass fmaxf gets (operands are never NaN here)
__device__ __forceinline__ float max_nc(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, __builtin_inff()); }
__device__ __forceinline__ float max3_nc(float a, float b, float c) { return max_nc(max_nc(a, b), c); }
// max(x, 0): fmaxf / fmed3 make hipcc put a canonicalising v_max_f32 x, x in front of the real one
// (256 of the 921 vector instructions of a split-mode k_ff iteration were those pairs); relu1 keeps
// that exact-fp32 form (NaN-propagating like the reference), relu_i below is the one-instruction form.
__device__ __forceinline__ float relu1(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, __builtin_inff()); }
__device__ __forceinline__ f4 f4_relu(f4 v) { return (f4){relu1(v.x), relu1(v.y), relu1(v.z), relu1(v.w)}; }
// max(x, 0) on the bit pattern: a negative float is a negative int, so one v_max_i32 does it -- no
// canonicalising pre-pass, and a compiler-visible instruction: hipcc inserts the wait states a read of
// an MFMA result needs (an earlier inline-asm v_max_f32 form did not get them: wrong values when the
// ReLU directly followed the MFMA that produced its input).
__device__ __forceinline__ float relu_i(float x) {
  const int v = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, v > 0 ? v : 0);
}
__device__ __forceinline__ f4 f4_relu_i(f4 v) { return (f4){relu_i(v.x), relu_i(v.y), relu_i(v.z), relu_i(v.w)}; }
// VALU_CONSUMER = the result feeds the bf16x3 operand split: the single-instruction integer form
template <bool VALU_CONSUMER>
__device__ __forceinline__ f4 f4_relu_t(f4 v) {
  if constexpr (VALU_CONSUMER) return f4_relu_i(v);
  else return f4_relu(v);
}
__device__ __forceinline__ f4 f4_max(f4 a, f4 b) {
  return (f4){max_nc(a.x, b.x), max_nc(a.y, b.y), max_nc(a.z, b.z), max_nc(a.w, b.w)};
}
__device__ __forceinline__ f4 ld4(const float* p) { return *reinterpret_cast<const f4*>(p); }
__device__ __forceinline__ void st4(float* p, f4 v) { *reinterpret_cast<f4*>(p) = v; }
#ifdef GPN_NO_NT
__device__ __forceinline__ void st4_nt(float* p, f4 v) { *reinterpret_cast<f4*>(p) = v; }
#else
__device__ __forceinline__ void st4_nt(float* p, f4 v) { __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p)); }
#endif

// write-through store (sc0 sc1: the line does not stay dirty in this XCD's L2, so the release at the
// end of the kernel has nothing of it left to write back)
__device__ __forceinline__ void st4_wt(float* p, f4 v) {
  // (s_nop: the store-data hazard -- a VALU write of `v`'s registers right behind a > 64-bit store -- is padded
  // by the compiler for its own stores only)
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
#if defined(GPN_FF_ST) && GPN_FF_ST == 1
#define FF_ST st4_nt
#elif defined(GPN_FF_ST) && GPN_FF_ST == 2
#define FF_ST st4_wt
#else
#define FF_ST st4
#endif

// y[M] += W[16M.., 16T..] . x[T]   (transposed orientation: x, y are T-tiles; W row-major in LDS with
// leading dimension ldw floats).  y must be initialised (bias or zero) by the caller.
template <int KT, int MT>
__device__ __forceinline__ void lin_T(const float* W, int ldw, const f4* x, f4* y, int lane) {
  const float* wp = W + (lane & 15) * ldw + 4 * (lane >> 4);
  constexpr int MC = MT < 4 ? MT : 4;  // 4 independent accumulators per k-step hide the MFMA latency
  static_assert(MT % MC == 0, "MT must be a multiple of the accumulator chunk");
#pragma unroll
  for (int M0 = 0; M0 < MT; M0 += MC) {
    // software pipeline: the weight fragments of step T+1 are read while step T's MFMAs issue; the
    // sched_barrier keeps hipcc from hoisting every ds_read of the unrolled chain to the top
    // (which spills past 256 VGPRs).
    f4 w[MC];
#pragma unroll
    for (int m = 0; m < MC; ++m) w[m] = ld4(wp + (16 * (M0 + m)) * ldw);
#pragma unroll
    for (int T = 0; T < KT; ++T) {
      f4 wn[MC];
      if (T + 1 < KT) {
#pragma unroll
        for (int m = 0; m < MC; ++m) wn[m] = ld4(wp + (16 * (M0 + m)) * ldw + 16 * (T + 1));
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int m = 0; m < MC; ++m) y[M0 + m] = mfma16(w[m][r], x[T][r], y[M0 + m]);
      __builtin_amdgcn_sched_barrier(0);
      if (T + 1 < KT) {
#pragma unroll
        for (int m = 0; m < MC; ++m) w[m] = wn[m];
      }
    }
  }
}

// Same product with the activation tile as the A operand: the result comes out in NORMAL orientation
// (lane (j, g) holds Y[point 4g + reg][channel 16N + j]).  Used where the next product sums over points.
template <int KT, int NT>
__device__ __forceinline__ void lin_N(const float* W, int ldw, const f4* x, f4* y, int lane) {
  const float* wp = W + (lane & 15) * ldw + 4 * (lane >> 4);
#pragma unroll
  for (int T = 0; T < KT; ++T) {
    f4 w[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) w[n] = ld4(wp + (16 * n) * ldw + 16 * T);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int n = 0; n < NT; ++n) y[n] = mfma16(x[T][r], w[n][r], y[n]);
  }
}

// Sum over the 4 lanes (j, g = 0..3) that share a point.
#ifndef GPN_SWAP_SUM
#define GPN_SWAP_SUM 1
#endif
// v and its copy with the 16-lane rows (0,1) and (2,3) / the wave halves exchanged, on the vector unit
// (v_permlane16_swap / v_permlane32_swap of gfx950) instead of a ds_bpermute round trip through LDS.
// Inline asm: the builtin with two equal inputs is folded into a + a by hipcc (wrong).
__device__ __forceinline__ void swap16(float v, float& a, float& b) {
  a = v, b = v;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap32(float v, float& a, float& b) {
  a = v, b = v;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
// bitwise OR over the four lanes (j, g = 0..3) of a point / over the 16 lanes of a DPP row
__device__ __forceinline__ unsigned or_over_g(unsigned v) {
  float a, b;
  swap16(__builtin_bit_cast(float, v), a, b);
  v = __builtin_bit_cast(unsigned, a) | __builtin_bit_cast(unsigned, b);
  swap32(__builtin_bit_cast(float, v), a, b);
  return __builtin_bit_cast(unsigned, a) | __builtin_bit_cast(unsigned, b);
}
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ unsigned or_over_row(unsigned v) {
  v |= dpp_u<0xB1>(v);   // ^1
  v |= dpp_u<0x4E>(v);   // ^2
  v |= dpp_u<0x141>(v);  // row_half_mirror: the other quad of the half
  v |= dpp_u<0x140>(v);  // row_mirror: the other half
  return v;
}
__device__ __forceinline__ float sum_over_g(float v) {
#if GPN_SWAP_SUM
  float a, b;
  swap16(v, a, b);
  v = a + b;  // rows 0 / 1: v0 + v1, rows 2 / 3: v2 + v3 (the same two addends as v + shfl_xor(v, 16))
  swap32(v, a, b);
  return a + b;
#else
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
#endif
}
__device__ __forceinline__ float max_over_g(float v) {
#if GPN_SWAP_SUM
  float a, b;
  swap16(v, a, b);
  v = max_nc(a, b);
  swap32(v, a, b);
  return max_nc(a, b);
#else
  v = max_nc(v, __shfl_xor(v, 16));
  v = max_nc(v, __shfl_xor(v, 32));
  return v;
#endif
}

// Sum over the 16 lanes of a DPP row (lanes with equal l>>4); the total lands in lane 15 of the row.
// Four v_add_f32 with row_shr DPP modifiers instead of four ds_bpermute round trips.
__device__ __forceinline__ float row_sum16(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
  return v;
}

// Sums over the 16 lanes of a DPP row of SIXTEEN values at once (recursive halving): at every stage a
// lane keeps the half of its values its own lane bit selects and adds the partner's partial sums of
// the same half (partners j^15, j^7, j^3, j^1: row_mirror, row_half_mirror, two quad_perms), so
// 8 + 4 + 2 + 1 = 15 DPP adds replace the 64 of sixteen row_sum16 calls, and every lane ends up with
// one total -- lane j with that of v[bitrev4(j)] -- which a single full-wave store can write.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_sums16(const float (&v)[16], int j) {
  const bool b3 = j & 8, b2 = j & 4, b1 = j & 2, b0 = j & 1;
  float a[8], b[4], c[2];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (b3 ? v[2 * i + 1] : v[2 * i]) + dpp_f<0x140>(b3 ? v[2 * i] : v[2 * i + 1]);
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = (b2 ? a[2 * i + 1] : a[2 * i]) + dpp_f<0x141>(b2 ? a[2 * i] : a[2 * i + 1]);
#pragma unroll
  for (int i = 0; i < 2; ++i) c[i] = (b1 ? b[2 * i + 1] : b[2 * i]) + dpp_f<0x1B>(b1 ? b[2 * i] : b[2 * i + 1]);
  return (b0 ? c[1] : c[0]) + dpp_f<0xB1>(b0 ? c[0] : c[1]);
}

// LayerNorm over the 64 channels of each point, T-tiles in and out.  gamma/beta are [64] (LDS or
// global), eps as in the reference (1e-6 inside the transformers, 1e-5 for the final norm).
__device__ __forceinline__ void layernorm_T(const f4* x, f4* y, const float* gamma,
                                            const float* beta, float eps, int lane) {
  float s = 0.f;
#pragma unroll
  for (int T = 0; T < 4; ++T) s += (x[T].x + x[T].y) + (x[T].z + x[T].w);
  const float mean = sum_over_g(s) * (1.0f / 64.0f);
  float q = 0.f;
#pragma unroll
  for (int T = 0; T < 4; ++T) {
    const f4 d = x[T] - f4_splat(mean);
    q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
  }
  const float var = sum_over_g(q) * (1.0f / 64.0f);
  const float rstd = __builtin_amdgcn_rsqf(var + eps);  // v_rsq_f32, 1 ulp (a correctly rounded sqrt + divide costs ~25 instructions)
  const int g = lane >> 4;
#pragma unroll
  for (int T = 0; T < 4; ++T) {
    const f4 ga = ld4(gamma + 16 * T + 4 * g), be = ld4(beta + 16 * T + 4 * g);
    y[T] = (x[T] - f4_splat(mean)) * f4_splat(rstd) * ga + be;
  }
}

// Cooperative global -> LDS copy of a row-major [rows, cols] matrix into a padded image
// (leading dimension ldw); cols % 4 == 0 and 16-byte aligned source rows.
__device__ __forceinline__ void stage_matrix(float* dst, int ldw, const float* __restrict__ src,
                                             int rows, int cols, int tid, int nthreads) {
  // 8 independent 16-byte loads in flight per thread before the first LDS store: a one-load-per-
  // iteration loop serialises ~17 L2 round trips per launch (measured: ~15 % of k_ff's time).
  constexpr int U = 8;
  const int c4n = cols >> 2, total = rows * c4n;
  for (int base = tid; base < total; base += nthreads * U) {
    f4 tmp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * nthreads;
      if (idx < total) {
        const int r = idx / c4n, c4 = idx - r * c4n;
        tmp[u] = ld4(src + (size_t)r * cols + 4 * c4);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * nthreads;
      if (idx < total) {
        const int r = idx / c4n, c4 = idx - r * c4n;
        st4(dst + r * ldw + 4 * c4, tmp[u]);
      }
    }
  }
}

// Same, element-wise, zero-padding to [prow, pcol] (for the 8-wide hidden layers and odd widths).
__device__ __forceinline__ void stage_padded(float* dst, int ldw, int prow, int pcol,
                                             const float* __restrict__ src, int rows, int cols,
                                             int tid, int nthreads) {
  for (int idx = tid; idx < prow * pcol; idx += nthreads) {
    const int r = idx / pcol, c = idx - r * pcol;
    dst[r * ldw + c] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.0f;
  }
}

__device__ __forceinline__ void stage_vector(float* dst, int padded, const float* __restrict__ src,
                                             int n, int tid, int nthreads) {
  for (int i = tid; i < padded; i += nthreads) dst[i] = (src != nullptr && i < n) ? src[i] : 0.0f;
}

// ====================================================================================== bf16x3 ("split") mode
// fp32 operands are split into two bf16 pieces, x = hi + mid (+ 2^-17 |x| residue), and a product is
// evaluated as  w_mid*x_hi + w_hi*x_mid + w_hi*x_hi  on v_mfma_f32_16x16x32_bf16 with fp32
// accumulation: 3 MFMAs of 16 cycles replace 8 fp32 MFMAs of 32 cycles (5.3x the rate) at a relative
// error of ~2^-16 per product (measured end to end on GNT: 6e-6 of the tensor scale; tolerance 1e-4).
//
// K-step of 32 = a PAIR of T-tiles (2P, 2P+1).  Lane (j, g) supplies, as the 8 bf16 of its B
// fragment, its own registers of the two tiles: elements 0-3 = channels 32P+4g+r, elements 4-7 =
// channels 32P+16+4g+r.  The weight image in LDS is made of 16-byte units (the 8 matching weight
// columns of one output row, in that order), laid out FRAGMENT-MAJOR:
//   unit(row 16M+i, pair P, piece hm (0 = hi, 1 = mid), g)  at  ((M*KP + P)*2 + hm)*64 + 16*g + i
// so the A fragment of one MFMA is 64 consecutive units read at [block + lane]: each of
// ds_read_b128's four 16-lane groups then covers all 16 slots of the 256-byte bank row (a padded
// row-major image is 2-way conflicted on one slot per group: measured 45 % extra LDS cycles).
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
struct XS {
  bf8 hi, mid;
};

__device__ __forceinline__ f4 mfma_bf(bf8 a, bf8 b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ XS split8(f4 a, f4 b) {
  XS s;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    s.hi[i] = (__bf16)a[i];
    s.hi[4 + i] = (__bf16)b[i];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    s.mid[i] = (__bf16)(a[i] - (float)s.hi[i]);
    s.mid[4 + i] = (__bf16)(b[i] - (float)s.hi[4 + i]);
  }
  return s;
}

// Split encoding of a 64-channel feature row in global memory (the `feat` tensor in bf16x3 mode):
// 16 units of 16 bytes, unit ((P*2 + hm)*4 + g) = the hi (hm = 0) or mid (hm = 1) piece of the 8
// channels {32P+4g+r, 32P+16+4g+r : r < 4} -- exactly the B fragment lane (., g) feeds to K pair P.
// Same 256 bytes per row as fp32; the four lanes of a point write 64 contiguous bytes per store
// instruction (with g innermost; hm innermost made every store a scatter of 16-byte pieces:
// 19 M write requests per launch instead of 11 M).
__device__ __forceinline__ void store_feat_split(float* row, const f4* y, int g) {
  bf8* u = reinterpret_cast<bf8*>(row);
#pragma unroll
  for (int P = 0; P < 2; ++P) {
    const XS s = split8(y[2 * P], y[2 * P + 1]);
    u[(P * 2) * 4 + g] = s.hi;
    u[(P * 2 + 1) * 4 + g] = s.mid;
  }
}
// Row -> registers: fp32 mode, the four T-tiles; split mode, f[2P] = hi and f[2P+1] = mid of K pair P
// (raw 16-byte units held in f4 registers).  Both modes: f[k] = the 16 bytes at 64 k + 16 g.
template <int SP>
__device__ __forceinline__ void load_feat_row(const float* row, int g, f4* f) {
#pragma unroll
  for (int T = 0; T < 4; ++T) f[T] = ld4(row + 16 * T + 4 * g);
}
__device__ __forceinline__ XS feat_pair(f4 hi, f4 mid) {
  XS s;
  s.hi = __builtin_bit_cast(bf8, hi);
  s.mid = __builtin_bit_cast(bf8, mid);
  return s;
}

// floats of LDS taken by the weight image of a [prow x 16*kt] matrix
__host__ __device__ constexpr int wimg_floats(bool split, int prow, int kt, int ld) {
  return split ? ((prow + 15) / 16) * ((kt + 1) / 2) * 2 * 64 * 4 : prow * ld;
}

// [hi(a) | mid(a)] of four fp32 values as one 16-byte unit
__device__ __forceinline__ bf8 split4(f4 a) {
  bf8 u;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    u[i] = (__bf16)a[i];
    u[4 + i] = (__bf16)(a[i] - (float)u[i]);
  }
  return u;
}

// y[M] += W . x in split mode on pre-split activations xs[(KT+1)/2].  TRANSPOSED = true keeps the
// T-tile orientation of lin_T (weights are the A operand); false swaps the operands, so the result
// comes out in normal orientation: lane (j, g) holds Y[point 4g + reg][channel 16M + j].
// TERMS = 3: the three-term split product (GPN_PREC_BF16X3); TERMS = 1: w_hi * x_hi only (GPN_PREC_BF16, plain
// bf16 operands with fp32 accumulation: the mid pieces of the images / splits are then dead and never computed).
template <int KT, int MT, bool TRANSPOSED = true, int TERMS = 3>
__device__ __forceinline__ void lin_Sx(const float* Wf, const XS* xs, f4* y, int lane) {
  static_assert(TERMS == 1 || TERMS == 3, "TERMS: 1 (plain bf16) or 3 (bf16x3)");
  constexpr int KP = (KT + 1) / 2;
  const bf8* wp = reinterpret_cast<const bf8*>(Wf) + lane;
  constexpr int MC = MT < 4 ? MT : 4;
  static_assert(MT % MC == 0, "MT must be a multiple of the accumulator chunk");
#pragma unroll
  for (int M0 = 0; M0 < MT; M0 += MC) {
    bf8 wh[MC], wm[MC];
#pragma unroll
    for (int m = 0; m < MC; ++m) {
      wh[m] = wp[((M0 + m) * KP) * 128];
      if constexpr (TERMS == 3) wm[m] = wp[((M0 + m) * KP) * 128 + 64];
    }
#pragma unroll
    for (int P = 0; P < KP; ++P) {
      bf8 nh[MC], nm[MC];
      if (P + 1 < KP) {
#pragma unroll
        for (int m = 0; m < MC; ++m) {
          nh[m] = wp[((M0 + m) * KP + P + 1) * 128];
          if constexpr (TERMS == 3) nm[m] = wp[((M0 + m) * KP + P + 1) * 128 + 64];
        }
      }
      // small terms first, so that they are not rounded away against the large one
      if constexpr (TRANSPOSED) {
        if constexpr (TERMS == 3) {
#pragma unroll
          for (int m = 0; m < MC; ++m) y[M0 + m] = mfma_bf(wm[m], xs[P].hi, y[M0 + m]);
#pragma unroll
          for (int m = 0; m < MC; ++m) y[M0 + m] = mfma_bf(wh[m], xs[P].mid, y[M0 + m]);
        }
#pragma unroll
        for (int m = 0; m < MC; ++m) y[M0 + m] = mfma_bf(wh[m], xs[P].hi, y[M0 + m]);
      } else {
        if constexpr (TERMS == 3) {
#pragma unroll
          for (int m = 0; m < MC; ++m) y[M0 + m] = mfma_bf(xs[P].hi, wm[m], y[M0 + m]);
#pragma unroll
          for (int m = 0; m < MC; ++m) y[M0 + m] = mfma_bf(xs[P].mid, wh[m], y[M0 + m]);
        }
#pragma unroll
        for (int m = 0; m < MC; ++m) y[M0 + m] = mfma_bf(xs[P].hi, wh[m], y[M0 + m]);
      }
#ifndef GPN_NO_SB
      __builtin_amdgcn_sched_barrier(0);
#endif
      if (P + 1 < KP) {
#pragma unroll
        for (int m = 0; m < MC; ++m) {
          wh[m] = nh[m];
          if constexpr (TERMS == 3) wm[m] = nm[m];
        }
      }
    }
  }
}

// Two activation tiles against one weight block: y[t][m] += W[rows 16(M0+m).., pairs P0..P0+NP) . xs[t]
// for t = 0, 1.  Every weight unit read from LDS feeds two MFMAs (half the LDS traffic per flop of
// lin_Sx) and the two tiles give the MFMA pipe two independent dependency chains.
// wp = image base + lane + (M0*KPT + P0)*128 units, KPT = K pairs of the whole image.
template <int KPT, int NP, int MT, int NTL = 2, int TERMS = 3>
__device__ __forceinline__ void lin_S2(const bf8* wp, const XS (&xs)[NTL][NP], f4 (&y)[NTL][MT]) {
  bf8 wh[MT], wm[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    wh[m] = wp[(m * KPT) * 128];
    if constexpr (TERMS == 3) wm[m] = wp[(m * KPT) * 128 + 64];
  }
#pragma unroll
  for (int P = 0; P < NP; ++P) {
    bf8 nh[MT], nm[MT];
    if (P + 1 < NP) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        nh[m] = wp[(m * KPT + P + 1) * 128];
        if constexpr (TERMS == 3) nm[m] = wp[(m * KPT + P + 1) * 128 + 64];
      }
    }
    if constexpr (TERMS == 3) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NTL; ++t) y[t][m] = mfma_bf(wm[m], xs[t][P].hi, y[t][m]);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NTL; ++t) y[t][m] = mfma_bf(wh[m], xs[t][P].mid, y[t][m]);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < NTL; ++t) y[t][m] = mfma_bf(wh[m], xs[t][P].hi, y[t][m]);
    __builtin_amdgcn_sched_barrier(0);
    if (P + 1 < NP) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        wh[m] = nh[m];
        if constexpr (TERMS == 3) wm[m] = nm[m];
      }
    }
  }
}

// y[M] += W . x in split mode; same contract as lin_T (x, y are fp32 T-tiles).
template <int KT, int MT, int TERMS = 3>
__device__ __forceinline__ void lin_S(const float* Wf, const f4* x, f4* y, int lane) {
  constexpr int KP = (KT + 1) / 2;
  XS xs[KP];
#pragma unroll
  for (int P = 0; P < KP; ++P) xs[P] = split8(x[2 * P], 2 * P + 1 < KT ? x[2 * P + 1] : f4_zero());
  lin_Sx<KT, MT, true, TERMS>(Wf, xs, y, lane);
}

// Arithmetic mode as a template argument of the kernels: SP = GPN_PREC_FP32 (0), GPN_PREC_BF16X3 (1) or
// GPN_PREC_BF16 (2).  Both bf16 modes share the operand formats and LDS images ("split" layout: hi and mid
// pieces side by side); the plain mode issues only the hi * hi MFMA of every product.
constexpr int bf_terms(int sp) { return sp == 2 ? 1 : 3; }
template <int SP, int KT, int MT>
__device__ __forceinline__ void lin(const float* W, int ldw, const f4* x, f4* y, int lane) {
  if constexpr (SP != 0) lin_S<KT, MT, bf_terms(SP)>(W, x, y, lane);
  else lin_T<KT, MT>(W, ldw, x, y, lane);
}

// Build the split image of a row-major [rows, cols] fp32 matrix, zero-padded to [prow, 16*kt].
__device__ __forceinline__ void stage_split(float* dstf, const float* __restrict__ src, int rows,
                                            int cols, int prow, int kt, int tid, int nthreads) {
  const int KP = (kt + 1) / 2, upr = KP * 4;
  bf8* dst = reinterpret_cast<bf8*>(dstf);
  const bool vec = (cols & 3) == 0;
  for (int idx = tid; idx < prow * upr; idx += nthreads) {
    const int row = idx / upr, u = idx - row * upr, P = u >> 2, g = u & 3;
    f4 a = f4_zero(), b = f4_zero();
    if (row < rows) {
      const int c0 = 32 * P + 4 * g, c1 = c0 + 16;
      const float* r = src + (size_t)row * cols;
      if (vec) {
        if (c0 < cols) a = ld4(r + c0);
        if (c1 < cols) b = ld4(r + c1);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (c0 + e < cols) a[e] = r[c0 + e];
          if (c1 + e < cols) b[e] = r[c1 + e];
        }
      }
    }
    const XS s = split8(a, b);
    const int blk = (((row >> 4) * KP + P) * 2) * 64 + 16 * g + (row & 15);
    dst[blk] = s.hi;
    dst[blk + 64] = s.mid;
  }
}

// Weight staging in either mode: [rows, cols] -> image of [prow, 16*kt] (fp32: leading dimension ldw).
template <int SP>
__device__ __forceinline__ void stage_w(float* dst, int ldw, const float* __restrict__ src, int rows,
                                        int cols, int prow, int kt, int tid, int nthreads) {
  if constexpr (SP != 0) {
    stage_split(dst, src, rows, cols, prow, kt, tid, nthreads);
  } else {
    if (rows == prow && cols == 16 * kt && (cols & 3) == 0) stage_matrix(dst, ldw, src, rows, cols, tid, nthreads);
    else stage_padded(dst, ldw, prow, 16 * kt, src, rows, cols, tid, nthreads);
  }
}

// ====================================================================================== 32x32x16 form
// v_mfma_f32_32x32x16_bf16: D[32x32] += A[32x16] . B[16x32]; lane (i = l&31, gg = l>>5) supplies
// A[row i][k = 8gg..8gg+7] and B[k = 8gg..8gg+7][col i] and holds D[row 8b + 4gg + r][col i] in
// register 4b + r (b, r = 0..3).  Same flops per pipe cycle as the 16x16x32 form, but an MFMA holds
// the SIMD's vector issue for 8 of its 32 cycles instead of 8 of its 16: half the issue slots per flop.
// "T32-tile": 32 channels x 32 points, lane (i, gg) holds channels 32M + 8b + 4gg + r of point i.
// Registers 8h..8h+7 of an output tile are then the B fragment of K-step 2M + h of the next layer
// (k slots 8gg + e <-> channels 32M + 16h + 4gg + {0..3, 8..11}); the weight image stores its A
// fragments in that column order, fragment-major:
//   unit(row 32Mo + i, K-step s, piece hm, gg) at ((Mo*KS + s)*2 + hm)*64 + 32gg + i
typedef float f16v __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f16v mfma_bf32(bf8 a, bf8 b, f16v c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

template <int PROW, int KS, int NTHR>
struct SplitStage32 {  // exact [PROW, 16*KS] row-major fp32 matrix -> bf16x3 image for the 32x32x16 form
  static constexpr int UPR = KS * 2, TOTAL = PROW * UPR, IT = (TOTAL + NTHR - 1) / NTHR;
  static_assert(PROW % 32 == 0, "rows must fill whole 32-row tiles");
  f4 a[IT], b[IT];
  __device__ __forceinline__ void load(const float* __restrict__ src, int tid) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = tid + it * NTHR;
      a[it] = f4_zero(), b[it] = f4_zero();
      if (idx < TOTAL) {
        const int row = idx / UPR, u = idx - row * UPR, s = u >> 1, gg = u & 1;
        const float* r = src + (size_t)row * (16 * KS) + 16 * s + 4 * gg;
        a[it] = ld4(r);
        b[it] = ld4(r + 8);
      }
    }
  }
  __device__ __forceinline__ void store(float* dstf, int tid) const {
    bf8* dst = reinterpret_cast<bf8*>(dstf);
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = tid + it * NTHR;
      if (idx < TOTAL) {
        const int row = idx / UPR, u = idx - row * UPR, s = u >> 1, gg = u & 1;
        const XS sp = split8(a[it], b[it]);
        const int blk = (((row >> 5) * KS + s) * 2) * 64 + 32 * gg + (row & 31);
        dst[blk] = sp.hi;
        dst[blk + 64] = sp.mid;
      }
    }
  }
};

// ====================================================================================== two-phase staging
// A kernel prologue that stages its matrices one after the other (load, wait, store to LDS, next
// matrix) pays one global-memory round trip per matrix: measured 11 us (k_ff, 6 pieces) to ~40 us
// (k_view_attn, 15 pieces) of fixed cost per launch.  These stagers split every piece into a load()
// that only issues global loads into registers and a store() that converts and writes the LDS
// image: a prologue calls every load() first, then every store(), and pays ONE round trip.
// Same images as stage_w / stage_padded / stage_vector above (PROW, KT as there; NTHR = block size).
template <int PROW, int PC4, int NTHR>
struct RowStage {  // fp32 row-major image [PROW][ldw], PC4 16-byte columns, zero-padded
  static constexpr int TOTAL = PROW * PC4, IT = (TOTAL + NTHR - 1) / NTHR;
  f4 a[IT];
  __device__ __forceinline__ void load(const float* __restrict__ src, int rows, int cols, int tid) {
    const bool vec = (cols & 3) == 0;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = tid + it * NTHR;
      a[it] = f4_zero();
      if (idx < TOTAL) {
        const int r = idx / PC4, c = 4 * (idx - r * PC4);
        if (r < rows) {
          const float* p = src + (size_t)r * cols + c;
          if (vec) {
            if (c < cols) a[it] = ld4(p);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (c + e < cols) a[it][e] = p[e];
          }
        }
      }
    }
  }
  __device__ __forceinline__ void store(float* dst, int ldw, int tid) const {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = tid + it * NTHR;
      if (idx < TOTAL) {
        const int r = idx / PC4, c = 4 * (idx - r * PC4);
        st4(dst + r * ldw + c, a[it]);
      }
    }
  }
};

template <int PROW, int KT, int NTHR>
struct SplitStage {  // bf16x3 fragment-major image (stage_split)
  static constexpr int KP = (KT + 1) / 2, UPR = KP * 4, TOTAL = PROW * UPR, IT = (TOTAL + NTHR - 1) / NTHR;
  f4 a[IT], b[IT];
  __device__ __forceinline__ void load(const float* __restrict__ src, int rows, int cols, int tid) {
    const bool vec = (cols & 3) == 0;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = tid + it * NTHR;
      a[it] = f4_zero(), b[it] = f4_zero();
      if (idx < TOTAL) {
        const int row = idx / UPR, u = idx - row * UPR, P = u >> 2, g = u & 3;
        if (row < rows) {
          const int c0 = 32 * P + 4 * g, c1 = c0 + 16;
          const float* r = src + (size_t)row * cols;
          if (vec) {
            if (c0 < cols) a[it] = ld4(r + c0);
            if (c1 < cols) b[it] = ld4(r + c1);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (c0 + e < cols) a[it][e] = r[c0 + e];
              if (c1 + e < cols) b[it][e] = r[c1 + e];
            }
          }
        }
      }
    }
  }
  __device__ __forceinline__ void store(float* dstf, int /*ldw*/, int tid) const {
    bf8* dst = reinterpret_cast<bf8*>(dstf);
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = tid + it * NTHR;
      if (idx < TOTAL) {
        const int row = idx / UPR, u = idx - row * UPR, P = u >> 2, g = u & 3;
        const XS s = split8(a[it], b[it]);
        const int blk = (((row >> 4) * KP + P) * 2) * 64 + 16 * g + (row & 15);
        dst[blk] = s.hi;
        dst[blk + 64] = s.mid;
      }
    }
  }
};

template <bool SPLIT, int PROW, int KT, int NTHR>
struct WStageSel {
  typedef RowStage<PROW, 4 * KT, NTHR> type;
};
template <int PROW, int KT, int NTHR>
struct WStageSel<true, PROW, KT, NTHR> {
  typedef SplitStage<PROW, KT, NTHR> type;
};
// image of a [rows, cols] matrix zero-padded to [PROW, 16*KT] in any mode (stage_w); SP = GPN_PREC_*
template <int SP, int PROW, int KT, int NTHR>
using WStage = typename WStageSel<(SP != 0), PROW, KT, NTHR>::type;

template <int PADDED, int NTHR>
struct VecStage {  // stage_vector
  static constexpr int IT = (PADDED + NTHR - 1) / NTHR;
  float v[IT];
  __device__ __forceinline__ void load(const float* __restrict__ src, int n, int tid) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int i = tid + it * NTHR;
      v[it] = (src != nullptr && i < n && i < PADDED) ? src[i] : 0.0f;
    }
  }
  __device__ __forceinline__ void store(float* dst, int tid) const {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int i = tid + it * NTHR;
      if (i < PADDED) dst[i] = v[it];
    }
  }
};
