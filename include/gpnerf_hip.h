/*
 * gpnerf_hip.h -- C-ABI of libgpnerf_hip.so: hand-written gfx950 (MI355X) HIP kernels for the
 * per-ray rendering hot path of GP-NeRF.
 *
 * The reference (lifuguan/GP-NeRF) is pure Python/PyTorch and has no FFI; the boundary it offers
 * is the set of Python functions listed in SURVEY.md section 8(b).  Each entry point below names the
 * reference function (file:line under the reference repo) whose arithmetic it replaces.  The
 * Python host (`gp-nerf_amd/`) binds these with ctypes and re-exposes the reference signatures.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 unless stated otherwise;
 *   - `stream` is a hipStream_t (0 = default stream); every call only enqueues work;
 *   - no entry point allocates, frees or synchronises; workspaces are passed in by the caller;
 *   - return value: 0 on success, negative GPN_E_* on error (gpn_last_error() has the text);
 *   - N = R*S points in ray-major order (point n = r*S + s), V source views; N and N*V must fit a
 *     32-bit int (checked: GPN_E_ARG otherwise);
 *   - stateless and re-entrant: no entry point reads or writes process-global state (ABI 2: the
 *     arithmetic mode is an argument / a field of gpn_gnt_weights_t, not a global switch; ABI 3 adds
 *     GPN_PREC_BF16).  Calls
 *     that share an output tensor or a workspace must of course be ordered by the caller (same
 *     stream, or events).  The measurement hooks at the end of this header are the one exception:
 *     one timing session per process at a time (serialised by a mutex).
 */
#ifndef GPNERF_HIP_H
#define GPNERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPN_ABI_VERSION 3
#define GPN_OK 0
#define GPN_E_ARG (-1)     /* invalid argument (shape / null pointer / unsupported size) */
#define GPN_E_LAUNCH (-2)  /* HIP reported an error at launch */

#define GPN_C 64         /* netwidth (reference config.py:108) */
#define GPN_SEM 512      /* deep-semantic channels (hard-coded in reference render_image.py:95) */
#define GPN_EMB 128      /* pos-enc(63) + view-enc(63) + 2 zero pad: the columns q_fcs adds to q */
#define GPN_EMB_HALF 64  /* ... stored as a per-point half and a per-ray half (gpn_posenc) */
#define GPN_MAX_DEPTH 8  /* trans_depth (reference configs/gpnerf_*.txt) */
#define GPN_MAX_VIEWS 32
#define GPN_MAX_SAMPLES 128

typedef void* gpn_stream_t;

/* Arithmetic mode of the GNT linear layers (no counterpart in the reference, which is plain fp32);
 * an argument of gpn_rgbfeat / gpn_view_attn / gpn_ff / gpn_qpos / gpn_ray_attn and the `precision`
 * field of the weight struct gpn_gnt_forward takes.  (ABI 1 had a process-global switch instead.)
 *   GPN_PREC_FP32   v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation (each linear layer is an
 *                   fmaf chain).  The ray softmax is exp2 with log2(e) folded into the query scale; the view
 *                   attention uses the folded weight products when gpn_view_layer_t.folded is set (below).
 *   GPN_PREC_BF16X3 every fp32 operand is split into two bf16 pieces (hi + mid) and a product is
 *                   w_mid*x_hi + w_hi*x_mid + w_hi*x_hi on v_mfma_f32_16x16x32_bf16 with fp32
 *                   accumulation: ~2^-16 relative error per product, 5.3x the MFMA rate.
 *                   Softmax, LayerNorm, residuals, the projector and the semantic pass stay fp32.
 *   GPN_PREC_BF16   plain bf16 operands (round to nearest even), ONE v_mfma_f32_16x16x32_bf16 per product,
 *                   fp32 accumulation; softmax, LayerNorm, residuals, biases, the projector and the semantic
 *                   pass stay fp32 (BASELINE.json configs[4]: "bf16 attention").  Operand formats and weight
 *                   images are those of GPN_PREC_BF16X3 (the residual pieces are simply not multiplied), so
 *                   `feat` tensors of the two modes are interchangeable.  ~2^-9 relative error per product:
 *                   NOT within the 1e-4 bar of the other two modes -- measured tolerances in DESIGN.md
 *                   section 3.1 and tests/test_gpu_parity.py (BF16_TOL). */
#define GPN_PREC_FP32 0
#define GPN_PREC_BF16X3 1
#define GPN_PREC_BF16 2

int gpn_abi_version(void);
const char* gpn_last_error(void); /* thread-local, valid until the next failing call */

/* ------------------------------------------------------------------ samplers / compositing */

/* sample_along_camera_ray, reference gpnerf/render_ray.py:58-102.
 * ray_o,ray_d [R,3]; t_rand [R,S] in [0,1) or NULL (det=True); outputs pts [R,S,3], z_vals [R,S]. */
int gpn_sample_along_ray(const float* ray_o, const float* ray_d, float near_depth, float far_depth,
                         int R, int S, int inv_uniform, const float* t_rand, float* pts,
                         float* z_vals, gpn_stream_t stream);

/* sample_pdf, reference gpnerf/render_ray.py:9-55.  bins [R,M+1], weights [R,M] (not modified;
 * 1e-5 is added internally), u as in gpn_sample_fine -> samples [R,N_samples]. */
int gpn_sample_pdf(const float* bins, const float* weights, const float* u, int u_row_stride, int R,
                   int M, int N_samples, float* samples, gpn_stream_t stream);

/* sample_fine_pts + sample_pdf, reference gpnerf/render_ray.py:162-191 and :9-55.
 * weights,z_vals [R,S] (not modified); u [R,N_imp] or, with u_row_stride==0, one row [N_imp]
 * shared by all rays (det=True: torch.linspace(0,1,N_imp)); outputs pts [R,S+N_imp,3],
 * z_out [R,S+N_imp] sorted ascending.  S+N_imp <= 256. */
int gpn_sample_fine(const float* ray_o, const float* ray_d, const float* weights,
                    const float* z_vals, const float* u, int u_row_stride, int R, int S, int N_imp,
                    int inv_uniform, float* pts, float* z_out, gpn_stream_t stream);

/* raw2outputs, reference gpnerf/render_ray.py:110-159 (model_type != 'gnt').
 * raw [R,S,4], z_vals [R,S], vis_mask [R,S] (float 0/1) -> rgb [R,3], depth [R], weights [R,S],
 * alpha [R,S], ray_mask [R] (float 0/1: more than 8 visible samples). */
int gpn_raw2outputs(const float* raw, const float* z_vals, const float* vis_mask, int R, int S,
                    int white_bkgd, float* rgb, float* depth, float* weights, float* alpha,
                    float* ray_mask, gpn_stream_t stream);

/* RaySamplerSingleImage.get_rays_single_image, reference gpnerf/sample_ray.py:108-132 (and the
 * pixel selection of random_sample, :172-214): rays of the strided pixel grid of an H x W image,
 * generated on the device instead of built on the host and copied.  M [9] = c2w[:3,:3] . K[:3,:3]^-1
 * (row-major) and origin [3] = c2w[:3,3] are HOST pointers (computed once per image with the
 * reference's own torch ops).  inds == NULL: all ceil(H/stride) x ceil(W/stride) pixels, row-major
 * (u = stride*col, v = stride*row); otherwise n_inds flat indices (int64, device) into that grid.
 * rays_o, rays_d [n,3].  Bit-exact with the reference's CPU result. */
int gpn_rays_single_image(const float* M, const float* origin, int H, int W, int stride,
                          const long long* inds, long long n_inds, float* rays_o, float* rays_d,
                          gpn_stream_t stream);

/* Label epilogue of the semantic branch, reference eval_gpnerf.py:201-206 (permute to NHWC) +
 * loss.py:293-300 (argmax over classes).  logits [C,HW] (the head's NCHW output of one image) ->
 * labels [HW] int64 and/or labels_u8 [HW] (either may be NULL): first maximal index, NaN counts as
 * maximal (torch.argmax).  C <= 256.  The 1-byte form is what a multi-GPU caller gathers when only
 * labels are needed (1 B/pixel instead of 2 KB/ray of features). */
int gpn_argmax_labels(const float* logits, int C, long long HW, long long* labels,
                      unsigned char* labels_u8, gpn_stream_t stream);

/* ------------------------------------------------------------------ projector (subsystem 1) */

/* NCHW -> NHWC repack of a source-view map stack (done once per image; the reference samples NCHW
 * maps with F.grid_sample, gpnerf/projection.py:118-129). src [V,C,H,W] -> dst [V,H,W,C]. */
int gpn_repack_nhwc(const float* src, float* dst, int V, int C, int H, int W, gpn_stream_t stream);

typedef struct {
  const float* xyz;      /* [N,3] world points */
  const float* proj;     /* [V,4,4] P_v = K_v . inv(c2w_v), computed on the host exactly as the
                            reference does (projection.py:41-44) so that pixels match bit for bit */
  const float* src_pos;  /* [V,3] source camera centres  (c2w[:3,3]) */
  float qry_pos[3];      /* query camera centre */
  float img_h, img_w;    /* train_cameras[0][:2]  (projection.py:109) */
  int N, V;
  const float* rgb;      /* [V,H,W,3]  NHWC (the API hands src_rgbs over in this layout) */
  int rgb_h, rgb_w;
  const float* feat;     /* [V,Hf,Wf,F] NHWC */
  int feat_h, feat_w, feat_c;
  const float* deep;     /* [V,Hs,Ws,D] NHWC or NULL */
  int deep_h, deep_w, deep_c;
  float* rgb_feat;       /* out [N,V,3+F], or NULL: geometry only, no gather */
  float* deep_out;       /* out [N,V,D] or NULL (skipped) */
  float* ray_diff;       /* out [N,V,4] or NULL */
  float* mask;           /* out [N,V] float 0/1 (in front AND in bounds) or NULL */
  float* pix;            /* out [V,N,2] clamped pixel coordinates, or NULL */
  int32_t* taps;         /* out [V,N,2] floor() of the RGB-map sampling coordinate (x0,y0), or NULL */
  uint8_t* front;        /* out [V,N] 1 if the point is in front of the camera, or NULL */
  int rgb_feat_ld;       /* floats between consecutive rgb_feat rows; 0 = 3+F (the reference's dense
                            layout).  36 gives 16-byte aligned rows (pad written as zeros): what
                            gpn_rgbfeat reads fastest; used by the fused path only */
  int samples_per_ray;   /* S when xyz is [R,S,3] (N = R*S), else 0.  Scheduling hint only (results do
                            not depend on it): the deep-semantic gather then walks neighbouring RAYS at
                            one sample index and view back to back -- their bilinear taps share texels,
                            which turns most 2 KB tap reads into cache hits */
} gpn_project_args_t;

/* Projector.compute = compute_projections + inbound + normalize + 3x grid_sample + compute_angle,
 * reference gpnerf/projection.py:31-139, as one fused pass. */
int gpn_project_gather(const gpn_project_args_t* a, gpn_stream_t stream);

/* ------------------------------------------------------------------ GNT (subsystem 2) */
/* In-place use: the per-point stage kernels (gpn_view_attn q_in / q_out, gpn_ff, gpn_qpos, gpn_ray_attn x_in /
 * x_out) may be given the SAME array as input and output -- every row is read by the lanes that write it
 * (a ray's rows by the workgroup that owns the ray, behind a barrier); gpn_gnt_forward runs the whole
 * layer stack on one [N,C] array this way.  Partially overlapping arrays are not supported. */

typedef struct {
  const float* w; /* [out,in] row-major (nn.Linear.weight) */
  const float* b; /* [out] or NULL */
} gpn_linear_t;

/* Projector.compute (without the deep-semantic gather) fused with rgbfeat_fc + max over views, reference
 * gpnerf/projection.py:31-139 + transformer_network.py:309-311: the fast path of render_rays.  Same
 * arguments as gpn_project_gather (a->mask required, a->ray_diff optional; rgb_feat / deep_out / pix / taps /
 * front must be NULL) and the outputs of gpn_rgbfeat: feat [N,V,C] in the layout of `precision`, q0 [N,C].
 * The [N,V,3+F] tensor is never formed; pairs whose gathered row is exactly zero (every bilinear tap
 * outside both maps) take the constant row fc2(relu(fc0.bias)) without a gather or an MLP evaluation.
 * mask is written for every pair; ray_diff and the feature row only for the pairs gpn_view_attn reads
 * (visible pairs, and every view of a point without any visible view).  mask bit-exact with
 * gpn_project_gather; feat / q0 agree with gpn_rgbfeat(gpn_project_gather(...)) to fp32 rounding (the
 * input channels are summed features-first).  Map sides <= 32767. */
int gpn_project_rgbfeat(const gpn_project_args_t* a, const gpn_linear_t* fc0, const gpn_linear_t* fc2,
                        float* feat, float* q0, int precision, gpn_stream_t stream);

typedef struct {                 /* Transformer2D + Attention2D, transformer_network.py:55-122 */
  const float *attn_norm_w, *attn_norm_b, *ff_norm_w, *ff_norm_b;
  gpn_linear_t ff1, ff2;         /* [4C,C], [C,4C] */
  gpn_linear_t q, k, v;          /* [C,C], no bias */
  gpn_linear_t pos0, pos2;       /* [C/8,4], [C,C/8] */
  gpn_linear_t attn0, attn2;     /* [C/8,C], [C,C/8] */
  gpn_linear_t out;              /* [C,C] */
  /* Optional (may be NULL): products of the weights above, made by gpn_view_fold.  When set, gpn_view_attn
   * uses them in EVERY arithmetic mode (k is never formed): in GPN_PREC_FP32 this re-associates the linear
   * chains v_fc(k_fc(f)) and attn_fc[0](k - q + pos) -- products formed in fp64 and rounded to fp32 once,
   * results within 2e-5 (relative to the tensor scale; measured ~1e-6) of the unfolded order and of
   * the oracle (tests/test_gpu_parity.py::test_gnt_stages runs both and compares them with each other).  With folded == NULL the fp32 mode follows the reference's operation order
   * literally (k_fc, then v_fc on its output; the sum k - q + pos formed before attn_fc[0]). */
  const float* folded;           /* [GPN_VIEW_FOLD_FLOATS] */
} gpn_view_layer_t;

typedef struct {                 /* Transformer + Attention('qk'), transformer_network.py:130-218 */
  const float *attn_norm_w, *attn_norm_b, *ff_norm_w, *ff_norm_b;
  gpn_linear_t ff1, ff2;
  gpn_linear_t q, k, v;          /* no bias */
  gpn_linear_t out;
} gpn_ray_layer_t;

typedef struct {                 /* GNT, transformer_network.py:221-294 (state-dict contract a10) */
  int trans_depth;               /* <= GPN_MAX_DEPTH */
  int in_feat_ch;                /* 32 -> rgb_feat has 3+32 channels */
  int precision;                 /* GPN_PREC_*: arithmetic mode gpn_gnt_forward runs this net in */
  int reserved_;                 /* keeps the pointers 8-byte aligned; set to 0 */
  gpn_linear_t rgbfeat0, rgbfeat2;
  gpn_view_layer_t view[GPN_MAX_DEPTH];  /* view_crosstrans.i */
  gpn_ray_layer_t ray[GPN_MAX_DEPTH];    /* view_selftrans.i  */
  gpn_linear_t qfc0[GPN_MAX_DEPTH];      /* q_fcs.i.0 for even i (odd i: unused) [C,C+126] */
  gpn_linear_t qfc2[GPN_MAX_DEPTH];      /* q_fcs.i.2 */
  const float *norm_w, *norm_b;          /* final LayerNorm, eps 1e-5 */
  gpn_linear_t rgb_fc;                   /* [3,C] */
  gpn_linear_t sem_adaptor;              /* [512,512] */
} gpn_gnt_weights_t;

/* Embedder x2, transformer_network.py:6-37 and :297-306.  The GPN_EMB columns
 * [posenc(pts) 63 | posenc(ray_d/|ray_d|) 63 | 0 0] that q_fcs concatenates to q (:322-323) are
 * stored in two halves, in that column order: emb_pt [N,64] = columns 0..63 of every point,
 * emb_ray [R,64] = columns 64..127, which depend on the ray only (the reference broadcasts the view
 * encoding over the samples, :304-306). */
int gpn_posenc(const float* pts, const float* ray_d, int R, int S, float* emb_pt, float* emb_ray,
               gpn_stream_t stream);

/* rgbfeat_fc + max over views, transformer_network.py:309-311.
 * rgb_feat [N,V,3+F] -> feat [N,V,C], q0 [N,C].
 * `feat` is the tensor gpn_view_attn reads.  In GPN_PREC_FP32 mode it is plain fp32.  In
 * GPN_PREC_BF16X3 mode every row of C = 64 channels is stored in the operand format of that mode (same
 * 256 bytes): 16 units of 8 bf16, unit ((P*2 + hm)*4 + g), P < 2, g < 4, holding the high (hm = 0)
 * or the residual (hm = 1) bf16 piece of channels {32P+4g+r, 32P+16+4g+r : r < 4}; value = hi + mid.
 * Both calls must run in the same mode.
 * in_ld: floats between consecutive rgb_feat rows (0 = in_ch, the reference's dense layout; a
 * multiple of 4 -- gpn_project_args_t.rgb_feat_ld = 36 -- needs a 16-byte aligned rgb_feat and is read
 * as 16-byte units).  mask [N,V] or NULL: with a mask the feature row of a masked pair is not stored
 * (its attention weight is exactly 0) unless the point has no visible view at all; q0 is the maximum
 * over ALL views either way, as in the reference. */
int gpn_rgbfeat(const gpn_linear_t* fc0, const gpn_linear_t* fc2, const float* rgb_feat, int N,
                int V, int in_ch, int in_ld, const float* mask, float* feat, float* q0,
                int precision, gpn_stream_t stream);

/* Attention part of Transformer2D (attn_norm, Attention2D, residual), transformer_network.py:74-88
 * and :107-112.  q_in [N,C], feat [N,V,C], ray_diff [N,V,4], mask [N,V] -> q_out [N,C];
 * perm [2N] (or NULL): processing order + visibility bits from gpn_vis_order.
 * hidden: the C/8 = 8 post-ReLU hidden values of attn_fc (transformer_network.py:66-70, the
 * 64 -> 8 -> 64 MLP in front of the softmax) per VISIBLE (point, view) pair (mask != 0; entries of
 * masked pairs are left unwritten: their softmax weight is exactly 0, or exactly 1/V when the
 * point has no visible view at all), the row of pair (n, v) starting at hidden[(n*V+v)*hidden_ld].
 * The C pre-softmax logits of the pair are attn_fc[2].weight . hidden + attn_fc[2].bias: storing
 * their 8-value pre-image instead of the 64 logits cuts this side channel (the largest stream of
 * the whole forward pass) eightfold; gpn_sem_aggregate rebuilds the logits from it.
 * stats: log-sum-exp over the views, C values per point, row n starting at stats[n*stats_ld], so
 * that the per-channel view softmax of line :84 is exp(logit - lse).  hidden_ld >= 8 and
 * stats_ld >= C (multiples of 4) let the caller interleave the layers (see gpn_sem_args_t);
 * hidden/stats may be NULL. */
int gpn_view_attn(const gpn_view_layer_t* L, const float* q_in, const float* feat,
                  const float* ray_diff, const float* mask, const int32_t* perm, int N, int V,
                  float* q_out, float* hidden, int hidden_ld, float* stats, int stats_ld,
                  int precision, gpn_stream_t stream);

/* No counterpart in the reference (weight preprocessing for gpn_view_attn, once per model): folds the
 * linear chains of Attention2D that have no nonlinearity in between,
 *   v = v_fc(k_fc(f))                          -> (Wv Wk) f
 *   attn_fc[0](k - q + pos_fc[2](h) + b_pos2)  -> (A0 Wk) f + (A0 P2) h + (A0 b_pos2 + b_attn0) - A0 q
 * so that per (point, view) the k product and two fp32->bf16 operand splits disappear.  Products are
 * accumulated in fp64 and rounded to fp32 once.  out: GPN_VIEW_FOLD_FLOATS floats =
 * [Wv Wk (C x C) | (A0 Wk | A0 P2 | 0) (16 x 96, rows >= C/8 zero) | A0 b_pos2 + b_attn0 (16)]. */
#define GPN_VIEW_FOLD_FLOATS (64 * 64 + 16 * 96 + 16)
int gpn_view_fold(const gpn_view_layer_t* L, float* out, gpn_stream_t stream);

/* No counterpart in the reference (scheduling aid for gpn_view_attn): perm [2N] = the points sorted by
 * their set of visible views (first N entries), so that 16-point tiles are homogeneous and whole
 * (tile, view) pairs can be skipped exactly, followed by the visibility bit mask (bit v = mask[n][v]
 * != 0) of the point in each slot (last N entries).  perm == NULL in gpn_view_attn means identity
 * order with the bits read from `mask`; results are bit-identical either way.
 * scratch: gpn_vis_order_scratch_bytes() bytes of device memory. */
size_t gpn_vis_order_scratch_bytes(void);
int gpn_vis_order(const float* mask, int N, int V, int32_t* perm, void* scratch, gpn_stream_t stream);

/* x + FeedForward(LayerNorm(x)), transformer_network.py:114-117 / :210-213 (eps 1e-6). x [N,C]. */
int gpn_ff(const float* norm_w, const float* norm_b, const gpn_linear_t* fc1,
           const gpn_linear_t* fc2, const float* x_in, int N, float* x_out, int precision,
           gpn_stream_t stream);

/* q_fcs[i] on cat(q, pos-enc, view-enc), transformer_network.py:322-324. q [N,C], emb_pt [N,64],
 * emb_ray [N/S,64] (gpn_posenc); point n belongs to ray n / S. */
int gpn_qpos(const gpn_linear_t* fc0, const gpn_linear_t* fc2, const float* q_in, const float* emb_pt,
             const float* emb_ray, int N, int S, float* q_out, int precision, gpn_stream_t stream);

/* Attention part of the ray Transformer, transformer_network.py:149-178 and :203-208.
 * x_in [R,S,C] -> x_out [R,S,C]; abar = mean over queries of A per head (may be NULL): the 4 head
 * values of (ray r, key s) are written to abar[(r*S+s)*abar_ld + 0..3], abar_ld >= 4;
 * weights [R,S] = head-mean of attention row 0 (transformer_network.py:216; may be NULL).
 * 1 <= S <= GPN_MAX_SAMPLES (any S; multiples of 16 waste no lanes). */
int gpn_ray_attn(const gpn_ray_layer_t* L, const float* x_in, int R, int S, float* x_out,
                 float* abar, int abar_ld, float* weights, int precision, gpn_stream_t stream);

typedef struct {
  int R, S, V, depth;
  /* per-layer side outputs of gpn_view_attn / gpn_ray_attn, interleaved over the layers so that
   * everything one (sample, view) needs is contiguous: */
  const float* hidden;                /* [N,V,depth,8]  (layer l: hidden + l*8,  ld = depth*8)  */
  const float* stats;                 /* [N,depth,C]    (layer l: stats + l*C,   ld = depth*C)  */
  const float* abar;                  /* [R,S,depth,4]  (layer l: abar + l*4,    ld = depth*4)  */
  const float* mask;                  /* [N,V] */
  /* deep-semantic source: either a materialised tensor ... */
  const float* deep_dense;            /* [N,V,512] or NULL */
  /* ... or the NHWC map stack, sampled on the fly with the projector's arithmetic */
  const float* deep_map;              /* [V,Hs,Ws,512] or NULL */
  int deep_h, deep_w;
  const float* xyz;                   /* [N,3]   (map mode) */
  const float* proj;                  /* [V,4,4] (map mode) */
  float img_h, img_w;
  float* sem3d;                       /* out [N,512] = sem_feat_3d, or NULL */
  float* pre;                         /* out [R,512] = mean_s sum_l ray_sem_l (adaptor input) */
  /* attn_fc[2] of every view layer (weight [C,8] row-major, bias [C]): logits = W . hidden + b */
  const float* attn2_w[GPN_MAX_DEPTH];
  const float* attn2_b[GPN_MAX_DEPTH];
  /* Scheduling hint only (results do not depend on it; 0 = unknown): the R rays are consecutive pixels of an
   * image in row-major order, ray_row_width rays per image row, ray 0 sitting at column ray_row_offset of its
   * row (render_single_image's chunks, reference render_image.py:39-47).  With it every XCD works on a 2-D
   * patch of the image -- rows of one column strip share texel rows of the deep-semantic maps -- instead of
   * a 1-D run of rays: measured HBM fetch of the pass 2.31 -> 1.38 GB at the north-star shape, same time.
   * Used when both are multiples of 64 / 8 and the chunk spans at least two rows. */
  int ray_row_width, ray_row_offset;
} gpn_sem_args_t;

/* Deep-semantic aggregation of Attention2D (:90), Attention (:180-184) and GNT.forward (:335-336)
 * in one pass, using linearity in deep_sem. */
int gpn_sem_aggregate(const gpn_sem_args_t* a, gpn_stream_t stream);

/* Output heads, transformer_network.py:332-338 + depth expectation render_ray.py:255-256.
 * q [R,S,C], weights [R,S], z_vals [R,S] or NULL, pre [R,512] or NULL ->
 * out [R,3+S] = cat(rgb, weights); depth [R] (if z_vals); sem_feat [R,512] (if pre). */
int gpn_heads(const gpn_gnt_weights_t* W, const float* q, const float* weights, const float* z_vals,
              const float* pre, int R, int S, float* out, float* depth, float* sem_feat,
              gpn_stream_t stream);

/* Whole GNT.forward (transformer_network.py:295-338) on a caller-provided workspace.
 * Inputs as gpn_rgbfeat / gpn_view_attn (rgb_feat_ld = its in_ld); `sem` carries the deep-semantic source and outputs
 * (its hidden/stats/abar/mask/R/S/V/depth/attn2_* fields are filled in by this call); sem may be NULL. */
size_t gpn_gnt_workspace_bytes(int R, int S, int V, int depth);
int gpn_gnt_forward(const gpn_gnt_weights_t* W, const float* rgb_feat, int rgb_feat_ld,
                    const float* ray_diff,
                    const float* mask, const float* pts, const float* ray_d, const float* z_vals,
                    int R, int S, int V, gpn_sem_args_t* sem, float* out, float* depth,
                    float* sem_feat, void* workspace, size_t workspace_bytes, gpn_stream_t stream);

/* One pass of render_rays' inner block, reference gpnerf/render_ray.py:239-256 (coarse) / :275-296
 * (fine): Projector.compute + GNT.forward + the depth expectation, with the projector fused into the
 * network's first stage (gpn_project_rgbfeat).  proj: the source views as for gpn_project_gather (its xyz /
 * N are taken from pts / R*S; ray_diff / mask may be NULL: they then live in the workspace; rgb_feat,
 * deep_out, pix, taps, front must be NULL).  Everything else as gpn_gnt_forward, same workspace size. */
int gpn_render_pass(const gpn_gnt_weights_t* W, const gpn_project_args_t* proj, const float* pts,
                    const float* ray_d, const float* z_vals, int R, int S, gpn_sem_args_t* sem,
                    float* out, float* depth, float* sem_feat, void* workspace, size_t workspace_bytes,
                    gpn_stream_t stream);

/* gpn_gnt_forward without a caller workspace and without any allocation.  It has NO counterpart in
 * the reference: it exists for callers that cannot allocate device memory around the call -- code
 * under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  The workspace is a
 * file-scope device array of GPN_STATIC_WS_BYTES inside the library's device image (every GPU the
 * library runs on holds its own copy until the process ends); its address on the current device is
 * looked up once per device, at this entry point's first call there (make that first call outside a
 * graph capture).  Same arguments, results (bit for bit) and error behaviour as gpn_gnt_forward;
 * additionally GPN_E_ARG when gpn_gnt_workspace_bytes(R, S, V, depth) > GPN_STATIC_WS_BYTES
 * (64 MiB: e.g. 128 rays x 64 samples x 10 views at depth 8 -- a serving-loop tile, not a 4096-ray
 * chunk).  Calls no hipMalloc / hipFree / allocating library routine.  All calls on one GPU share the
 * one array: calls that overlap on a GPU (two streams, two threads) race -- order them. */
#define GPN_STATIC_WS_BYTES (64u << 20)
int gpn_gnt_forward_static(const gpn_gnt_weights_t* W, const float* rgb_feat, int rgb_feat_ld,
                           const float* ray_diff, const float* mask, const float* pts,
                           const float* ray_d, const float* z_vals, int R, int S, int V,
                           gpn_sem_args_t* sem, float* out, float* depth, float* sem_feat,
                           gpn_stream_t stream);

/* ------------------------------------------------------------------ measurement hooks
 * No counterpart in the reference: bench.py uses these to time individual kernels with HIP events on
 * the stream they are launched on, inside the timed region.  gpn_timing_begin creates `max_records`
 * event pairs (outside any launch path); while enabled, every launch of a stage selected in
 * `stage_mask` (bit i = stage i) is bracketed by an event pair; gpn_timing_end waits for the
 * events, returns per-stage summed milliseconds and launch counts (arrays of GPN_NSTAGES) and
 * releases the pool.  One session per process at a time (a second gpn_timing_begin fails until the
 * first session has ended); the pool itself is guarded by a mutex, so launches from other threads
 * while a session is open are safe (they are recorded if their stage is selected).
 * A stage is a LAUNCH: inside gpn_gnt_forward / gpn_render_pass the feed-forward blocks and the q MLP run as
 * later phases of the attention launches and are counted under GPN_ST_VIEW_ATTN /
 * GPN_ST_RAY_ATTN; GPN_ST_FF / GPN_ST_QPOS see the stand-alone gpn_ff / gpn_qpos calls (and the
 * feed-forward launches of passes with more than 64 samples per ray). */
enum {
  GPN_ST_SAMPLE = 0, GPN_ST_PROJECT, GPN_ST_PROJECT_DEEP, GPN_ST_POSENC, GPN_ST_RGBFEAT,
  GPN_ST_VIEW_ATTN, GPN_ST_FF, GPN_ST_QPOS, GPN_ST_RAY_ATTN, GPN_ST_SEM, GPN_ST_HEADS,
  GPN_ST_VIS_ORDER, GPN_ST_PROJ_RGBFEAT, GPN_NSTAGES
};
int gpn_timing_begin(unsigned stage_mask, int max_records);
int gpn_timing_end(double* ms_per_stage, long long* launches_per_stage);

#ifdef __cplusplus
}
#endif
#endif /* GPNERF_HIP_H */
