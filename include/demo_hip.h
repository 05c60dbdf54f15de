/*
 * demo_hip.h -- C ABI of libdemo_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for
 * the DeMo PIFE -> HDM -> ATMoE hot path.
 *
 * The reference (maxingan2412/DeMo2) is pure Python/PyTorch and has no FFI of its own; each
 * entry point below replaces the PyTorch ops of the cited reference lines (paths relative to
 * the reference root).  The library never allocates or frees device memory, keeps no mutable
 * global state, launches asynchronously on the caller's stream and is re-entrant (forward runs
 * on the Python main thread, backward on PyTorch's autograd thread).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless said otherwise; sizes are element counts;
 *   - `stream` is a hipStream_t passed as void*;
 *   - dtype codes: DEMO_F32 = 0, DEMO_BF16 = 1 (DEMO_D8 = 2: the 8-bit code of a saved activation derivative, demo_gemm only).
 *     "T" below = the compute/activation dtype;
 *   - return value 0 = launched; negative = argument/launch error, text via demo_last_error().
 */
#ifndef DEMO_HIP_H
#define DEMO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DEMO_F32 0
#define DEMO_BF16 1
/* 8-bit fixed-point code of a saved activation DERIVATIVE (demo_gemm: pre_out_kind 2 writes it, dtPre DEMO_D8 reads it): one byte per
 * element, q = rint((a + 0.13) * 200), a = q * 0.005 - 0.13.  act'(x) lies in [-0.0996, 1.0996] for QuickGELU and in [-0.1290, 1.1290] for
 * erf-GELU, so q is in [0, 252]; absolute error <= 0.0025 everywhere (bf16's half-ulp is 0.0020 on [0.5, 1) and 0.0039 on [1, 2)); the
 * saturated values 0 and 1 are codes of their own (26, 226: decoded to fp32 rounding, 2e-9 and 1.0).  Halves the bytes of the tensor
 * the MLP backward re-reads. */
#define DEMO_D8 2

#define DEMO_ACT_NONE 0
#define DEMO_ACT_QUICKGELU 1 /* x*sigmoid(1.702x): modeling/clip/model.py:163-165, modeling/moe/AttnMOE.py:5-7 */
#define DEMO_ACT_GELU 2      /* erf GELU: modeling/backbones/vit_pytorch.py:126 (nn.GELU) */
#define DEMO_ACT_SAVED 3     /* only as act_grad_kind: the tensor act_grad_pre IS the derivative (a forward saved it: pre_out_kind 1) */

int demo_abi_version(void);
const char* demo_last_error(void); /* thread-local, host pointer */

/* ---- GEMM --------------------------------------------------------------------------------
 * C[z][m,n] = act( alpha * sum_k A[z][m,k] * B[z][n,k] + bias[z][n] ) (+ residual[z][m,n])
 * Replaces every nn.Linear / F.linear / `@` on the path (e.g. clip/model.py:172-178,225,484;
 * vit_pytorch.py:159-170,132-140; AttnMOE.py:27-31,53-58; make_model.py:52-58,134-138).
 * A[m,k] lives at A + m*a_rs + k*a_cs (elements), B[n,k] at B + n*b_rs + k*b_cs, so every
 * transposition needed by forward and backward is a choice of strides.  Two-level batching
 * z = z1*batch2 + z2 with independent strides serves the per-head / per-branch grouped
 * products of HDM and ATMoE.  bf16 x bf16 with K-contiguous operands takes the MFMA
 * (v_mfma_f32_16x16x32_bf16) LDS-tiled kernel; everything else a generic fp32-accumulate kernel. */
typedef struct demo_gemm_desc {
  const void* A;
  const void* B;
  void* C;
  const float* bias;      /* [N] fp32 or NULL */
  const void* residual;   /* dtype dtRes (fp32 by default), indexed like C (same ldc / batch strides), or NULL */
  int32_t M, N, K;
  int64_t a_rs, a_cs, b_rs, b_cs, c_rs;
  int32_t dtA, dtB, dtC;
  int32_t act;
  float alpha;
  int32_t batch1, batch2; /* >=1 */
  int64_t sA1, sA2, sB1, sB2, sC1, sC2, sBias1, sBias2;
  /* optional backward epilogue: C *= act'(pre[m,n]) where pre is indexed like C (dtype dtPre);
     used for d(fc1 pre-activation) = (dY W2) * act'(pre)  */
  const void* act_grad_pre;
  int32_t dtPre;
  int32_t act_grad_kind;  /* DEMO_ACT_* whose derivative is applied */
  int32_t accumulate;     /* 1: C += result (fp32 C only) */
  void* pre_out;          /* optional: value BEFORE `act` (after alpha, bias), indexed and typed like C; saved for backward */
  int32_t dtRes;          /* dtype of `residual`: DEMO_F32 (0, the default) or DEMO_BF16 -- the bf16 residual stream of the bf16 compute mode */
  int32_t pre_out_kind;   /* what pre_out receives: 0 = that pre-activation x, 1 = act'(x), 2 = act'(x) as DEMO_D8 codes (pre_out is then a
                             uint8 buffer indexed like C, one byte per element).  With 1 / 2 the backward product's epilogue is one
                             multiply (act_grad_kind DEMO_ACT_SAVED; dtPre = dtC resp. DEMO_D8) instead of re-evaluating the derivative
                             from x: the exponential and the reciprocal are computed once, in the forward, where sigmoid(1.702x) / erf
                             are needed anyway */
} demo_gemm_desc;
int demo_gemm(const demo_gemm_desc* d, void* stream);
/* what demo_gemm would pick: 0 = generic fp32-FMA kernel, 1 = bf16 MFMA 128x128 kernel, 2 = bf16 MFMA 256x256 ping-pong kernel */
int demo_gemm_path(const demo_gemm_desc* d);
int demo_gemm_force_tile(int tile); /* tuning hook: 128 / 256 forces that MFMA tile configuration, 0 = heuristic; returns 0 */
/* A 256x256 launch whose tile count leaves a small last round is issued as two launches: rows [0, r) on the 256x256 kernel
 * (whole rounds of the 256 CUs) and rows [r, M) on the 128x128 kernel.  demo_gemm_split_rows: the r demo_gemm would use
 * (0 = one launch); demo_gemm_set_split: largest leftover, in 256x256 tiles, that is split off (default 128, 0 = never). */
int demo_gemm_split_rows(const demo_gemm_desc* d);
/* Round 3: the dispatcher's whole plan for a descriptor -- rows[i] and kernel configuration cfg[i] (2566 = 256x256, 1282 = 128x128, 0 = the
 * heuristic's single launch) of each part, in row order; returns the number of parts (1 = one launch).  Besides main + remainder there is an
 * EQUAL-PARTS plan: full + 1 launches of the 256x256 kernel with at most one tile per CU each (a tile runs faster the fewer CUs are active:
 * the main loop is bound by the chip-wide L2 -> LDS feed), taken when K <= the bound of demo_gemm_set_split_equal (0 = never, the default:
 * measured -0.12 ms of a 35.7 ms training step at 1024, i.e. for the K = 768 products of N = 768, and slower on the long-K products). */
int demo_gemm_split_plan(const demo_gemm_desc* d, int32_t* rows, int32_t* cfg, int32_t max_parts);
int demo_gemm_set_split_equal(int max_k);
int demo_gemm_set_rem160(int on);      /* tuning hook: 1 (default) = a row-split remainder that fits one round of 160x128 tiles takes that 3-buffer configuration (cfg 1603), 0 = always 128x128 */
int demo_gemm_set_pp_min_fill(int pct); /* tuning hook: least fill (%) of the last round of 256 CUs at which short-K products still take the 256x256 kernel (default 70) */
int demo_gemm_set_pp_epilogue(int reg); /* tuning hook, 256x256 kernel: 1 (default) = epilogue straight from the accumulators, 0 = LDS-staged */
int demo_gemm_set_split(int max_left_tiles);
int demo_gemm_set_split_overlap(int on); /* row-split products: 1 = the remainder part runs on a library-owned side stream beside the main part (fork / join with events on the caller's stream; measured slower, an experiment), 0 (default) = both parts back to back on the caller's stream */
int demo_gemm_set_epi_stage(int on);  /* 256x256 kernel, bf16 outputs: 1 (default) = whole-row stores through an LDS image of the output tile, 0 = 16-row x 64-B stores straight from the accumulator layout */
int demo_gemm_set_colslice(int s);   /* tuning hook, 256x256 kernel: 2 / 4 / 8 = every XCD keeps to N / 256 / s weight panels for the whole launch; 0 = heuristic, 1 = off */
int demo_gemm_set_group_m(int g);     /* tuning hook, 256x256 kernel: row panels per group of the XCD-aware tile order (0 = heuristic) */
int demo_gemm_set_reverse_m(int min_k); /* tuning hook, 256x256 kernel: products with K >= min_k walk their row panels last to first (0 = never, the default) */
int demo_gemm_set_dephase(int units); /* tuning hook, 256x256 kernel: first-round workgroups start ((id >> 3) & 3) * units * 64 clocks late (default 0) */
int demo_gemm_store_policy(void); /* cache policy the MFMA NT kernels' output stores were BUILT with (-DDEMO_STORE_POL): 0 default, 1 nt, 2 sc1 (write-through) */
int demo_gemm_desc_size(void); /* sizeof(demo_gemm_desc): lets a foreign binding verify its struct layout */
int demo_gemm_generic_tile(int tile); /* tuning / test hook, generic fp32-FMA kernel: 32 / 64 = that output tile for every launch, 0 = heuristic (32 x 32 tiles when the 64 x 64 grid would not fill the chip) */

/* ---- elementwise / layout ----------------------------------------------------------------- */
/* fp32 -> bf16 shadow copy of the flat parameter buffer (one launch per optimizer step). */
int demo_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream);
/* DropPath / stochastic depth of the ImageNet ViT blocks in training (backbones/vit_pytorch.py:48-65,186-187: x + drop_path(f(x))).
 * demo_droppath_mask: scale[b] = floor(keep + u_b) / keep (0 or 1/keep per image), u_b from a counter hash of (seed [+ *seed_dev], b);
 * demo_rowscale_add: out[r,:] = x[r,:] + scale[r / L] * t[r,:] (x, out fp32, may alias; t in dtype dt) -- the residual add;
 * demo_rowscale_cast: out[r,:] = scale[r / L] * x[r,:] (fp32 -> dt) -- the branch's incoming gradient. */
int demo_droppath_mask(float* scale, int32_t n, float keep_prob, uint64_t seed, const uint64_t* seed_dev, void* stream);
int demo_rowscale_add(const float* x, const void* t, int32_t dt, const float* scale, float* out, int32_t Bn, int32_t L, int32_t W, void* stream);
int demo_rowscale_cast(const float* x, const float* scale, void* out, int32_t dt, int32_t Bn, int32_t L, int32_t W, void* stream);
int demo_cast_bf16_f32(const void* src, float* dst, int64_t n, void* stream);
/* 2-D transpose with dtype preserved: dst[c, r] = src[r, c]; used for weight^T shadows (backward). */
int demo_transpose(const void* src, void* dst, int32_t rows, int32_t cols, int32_t dt, void* stream);
/* the same for `batch` equally shaped matrices src + z*src_stride -> dst + z*dst_stride (strides in elements): one launch for
   the W^T shadows of all layers' Linear weights of one kind, which sit at a constant stride in the flat parameter buffer. */
int demo_transpose_batched(const void* src, void* dst, int32_t rows, int32_t cols, int32_t batch, int64_t src_stride,
                           int64_t dst_stride, int32_t dt, void* stream);
/* column sums: out[n] (+)= sum_m X[m, n]   (bias gradients) */
int demo_colsum(const void* X, float* out, int32_t M, int32_t N, int64_t ldx, int32_t dt, int32_t accumulate, void* stream);

/* ---- device-side argument errors ------------------------------------------------------------ */
/* Bits a kernel ORs into the caller's device error word (int32) when an index argument is out of range; see the entries that take
 * `err`.  The word is caller-owned (the library keeps no device state); demo2_amd polls it without a device sync (ops.DeviceErrors). */
enum { DEMO_ERR_CAM_RANGE = 1, DEMO_ERR_LABEL_RANGE = 2 };

/* ---- patch embedding ---------------------------------------------------------------------- */
/* im2col for the patch-embed conv (clip/model.py:432,447-449; vit_pytorch.py:303,321-323):
 * img fp32 (Bn,3,H,W) -> cols T (Bn*ny*nx, 3*p*p), column order (c,ky,kx) = conv weight order. */
int demo_im2col(const float* img, void* cols, int32_t Bn, int32_t H, int32_t W, int32_t patch, int32_t stride,
                int32_t dt, void* stream);
/* adjoint of demo_im2col is not needed (images carry no gradient). */

/* token assembly + optional pre-LayerNorm (clip/model.py:450-457; vit_pytorch.py:419-431):
 * x[b,0,:] = cls + pos[0] + cam_scale*cam_embed[cam[b]] ; x[b,1+i,:] = patch[b,i,:] + pos[1+i]
 *            (+ cam_scale*cam_embed[cam[b]] on every token when cam_all_tokens, the ViT SIE rule);
 * if ln_w != NULL: x = LayerNorm(x; ln_w, ln_b, eps).  x is the fp32 residual stream (Bn*(1+np), W).
 * pre (optional, fp32) receives the un-normalised sum for the backward pass.
 * n_cam = rows of cam_embed.  The reference's torch indexing raises for cam[b] outside [0, n_cam) (meta_arch.py:95-99); a kernel
 * cannot raise, so such an index is clamped into the table (no out-of-bounds read of the neighbouring parameters of the flat
 * buffer) and bit DEMO_ERR_CAM_RANGE is OR-ed into *err (device int32, optional): the host raises at its next poll of the word. */
int demo_embed_tokens(const void* patch, int32_t dt_patch, const float* cls, const float* pos,
                      const float* cam_embed, const int64_t* cam, int32_t n_cam, float cam_scale, int32_t cam_all_tokens,
                      const float* ln_w, const float* ln_b, float eps, float* x, float* pre, float* mean, float* rstd,
                      int32_t Bn, int32_t np, int32_t W, int32_t* err, void* stream);

/* ---- normalisation ------------------------------------------------------------------------ */
/* LayerNorm over the last dim of fp32 x (M, D) -> y (dtype dtY); optional mean/rstd (M) saved for backward.
 * (clip/model.py:154-160; nn.LayerNorm in vit_pytorch.py:178,182,401; make_model.py:52) */
int demo_layernorm_fwd(const float* x, const float* w, const float* b, float eps, void* y, int32_t dtY,
                       float* mean, float* rstd, int32_t M, int32_t D, void* stream);
/* dx (fp32, optionally accumulated into: the residual-stream gradient) from dy (dtype dtDy); dw/db are accumulated into
 * (atomics); dx_bf16 (optional) receives a bf16 copy of the final dx = the A operand of the next backward GEMM. */
int demo_layernorm_bwd(const void* dy, int32_t dtDy, const float* x, const float* w, const float* mean, const float* rstd,
                       float* dx, int32_t accumulate_dx, void* dx_bf16, float* dw, float* db, float* workspace,
                       int64_t workspace_elems, int32_t M, int32_t D, void* stream);
/* workspace (optional, fp32, >= demo_layernorm_bwd_workspace(M, D) elements): per-workgroup dw/db partial rows summed by a
 * second kernel instead of contended global atomics. */
int64_t demo_layernorm_bwd_workspace(int32_t M, int32_t D);
/* Residual add fused into the LayerNorm that follows it (x + drop-free branch output, clip/model.py:227-230): s = x (+ add, bf16),
 * rounded to x's dtype first; xsum_out (optional, x's dtype, may alias x) = s; y (bf16) = LN(s); mean/rstd (optional) of s. */
int demo_layernorm_add_fwd(const void* x, int32_t dtX, const void* add, int32_t dtAdd, void* xsum_out, const float* w, const float* b,
                           float eps, void* y, int32_t dtY, float* mean, float* rstd, int32_t M, int32_t D, void* stream);
/* The same backward with the residual-gradient stream kept in bf16 (x in dtype dtX): dx_bf16 = bf16(dres_bf16 (optional) + LayerNorm-backward(dy)); no
 * fp32 dx is read or written (380 instead of 608 MB per launch at 49536 x 768).  dw/db/workspace as above.  D <= 1024. */
int demo_layernorm_bwd_lp(const void* dy, int32_t dtDy, const void* x, int32_t dtX, const float* w, const float* mean, const float* rstd,
                          const void* dres_bf16, void* dx_bf16, float* dw, float* db, float* workspace, int64_t workspace_elems,
                          int32_t M, int32_t D, void* stream);

/* BatchNorm1d over x (Bn, F) fp32 (nn.BatchNorm1d at AttnMOE.py:30,53; make_model.py:136,176).
 * training!=0: batch statistics (biased var), running stats updated in place with `momentum`
 * (unbiased var), save_mean/save_invstd (F) written.  Otherwise running stats are used.
 * Grouped parameters (the 7*H expert BNs of ATMoE live in one call): feature f belongs to group
 * g = f / gsz; its parameters are at base + pblock(g)*pstride + f%gsz where
 * pblock(g) = (g % n_inner)*n_outer + g / n_inner (expert (h,j) block order); gsz = 0 -> plain.
 * gate (optional, (Bn, n_inner, n_outer) fp32): y *= gate[b, g % n_inner, g / n_inner]  (AttnMOE.py:46). */
int demo_batchnorm_fwd(const float* x, const float* w, const float* b, float* running_mean, float* running_var,
                       int32_t training, float momentum, float eps, void* y, int32_t dtY, float* save_mean,
                       float* save_invstd, int32_t Bn, int32_t F, int32_t gsz, int32_t n_inner, int32_t n_outer,
                       int64_t pstride, int64_t sstride, const float* gate, void* stream);

/* ---- ViT self-attention (L <= 144 tokens, head dim 64) ------------------------------------- */
/* qkv T (Bn*L, 3*D) packed [q | k | v], heads contiguous inside each third
 * (nn.MultiheadAttention packed in-proj, clip/model.py:172,225; vit_pytorch.py:159-167).
 * out T (Bn*L, D) = softmax(q k^T / sqrt(dh)) v per (image, head); lse (Bn*H*L) fp32 optional. */
int demo_attention_fwd(const void* qkv, void* out, float* lse, int32_t Bn, int32_t L, int32_t H, int32_t dh,
                       int32_t dt, void* stream);

/* ---- PIFE ----------------------------------------------------------------------------------- */
/* z[b] = cat[ tokens[b,0,:], mean_i tokens[b,1+i,:] ] -> fp32 (Bn, 2C)  (make_model.py:301-303). */
int demo_pife_pool_cat(const void* tokens, int32_t dt, float* z, int32_t Bn, int32_t L, int32_t C, void* stream);
/* tokens[b,0,:] = glob[b,:] : the reduced global replaces the cls row of the HDM key/value stream
 * (AttnMOE.py:155-165). */
int demo_set_token0(void* tokens, int32_t dt, const float* glob, int64_t glob_ld, int32_t Bn, int32_t L, int32_t C, void* stream);
/* out[b, 0:C] = tokens[b,0,:] as fp32 (row stride out_ld): the cls rows when GLOBAL_LOCAL is off (make_model.py:370). */
int demo_get_token0(const void* tokens, int32_t dt, float* out, int64_t out_ld, int32_t Bn, int32_t L, int32_t C, void* stream);

/* ---- HDM (AttnMOE.py:153-191), folded single-query form ------------------------------------- */
/* scores S (3*Bs*L, ldS) fp32, column br*Hh + h = score of branch br / head h for that token (all 7 branches are
 * scored for every modality; only the attended ones are read), come from demo_gemm against the folded keys.  This kernel: for sample b, branch br (0..6 = r,n,t,rn,rt,nt,rnt)
 * and head h: softmax over the concatenated streams of that branch, optional dropout (seed/offset
 * counter RNG, p_drop), pooled[b,br,h,:] = sum_l p_l * tokens_l  -> fp32 (Bs,7,Hh,C);
 * psum[b,br,h] = sum_l p_l (== 1 without dropout); probs (optional) saved for backward.
 * seed_dev (optional, device memory) is added to seed on the device: a training step captured into a hipGraph replays
 * with the launch arguments frozen, so whatever must change between steps has to live in device memory. */
int demo_hdm_softmax_pool(const float* S, int64_t ldS, const void* tokens, int32_t dt, float* pooled, float* psum, float* probs,
                          float* probs_post, int32_t Bs, int32_t L, int32_t C, int32_t Hh, float p_drop, uint64_t seed,
                          const uint64_t* seed_dev, void* stream);
/* Backward of the above.  probs = the forward's softmax BEFORE dropout (Bs,7,Hh,3L); the dropout mask is regenerated
 * from (seed, p_drop).  dS (3*Bs*L, ldS) fp32 is fully written (zeros in the columns of branches that do not attend a
 * row's modality); dtokens (3*Bs*L, C) fp32 is fully written (row 0 of every image = gradient of the PIFE global). */
int demo_hdm_softmax_pool_bwd(const float* dpooled, const float* dpsum, const float* probs, const void* tokens, int32_t dt,
                              const float* pooled, const float* psum, float* dS, int64_t ldS, float* dtokens, int32_t Bs,
                              int32_t L, int32_t C, int32_t Hh, float p_drop, uint64_t seed, const uint64_t* seed_dev,
                              void* stream);
/* value-projection bias under dropout, backward: dbias[br*stride + c] += sum_b dy*psum ; dpsum[b,br,h] += sum_c dy*bias */
int demo_hdm_vbias_bwd(const float* dy, const float* bias, int64_t bias_stride, const float* psum, float* dbias, float* dpsum,
                       int32_t Bs, int32_t C, int32_t Hh, void* stream);
/* y[b, br, h*dh + e] += bias[br][h*dh+e] * psum[b,br,h]   (value-projection bias under dropout) */
int demo_hdm_vbias(float* y, const float* bias, int64_t bias_stride, const float* psum, int32_t Bs, int32_t C, int32_t Hh,
                   void* stream);

/* ---- ATMoE gate (AttnMOE.py:61-68) ----------------------------------------------------------- */
/* gates[b,h,j] = softmax_j( q[b, h*d:(h+1)*d] . k[b, j, h*d:(h+1)*d] * d^-0.5 ) */
int demo_atmoe_gate_softmax(const float* q, const float* k, float* gates, int32_t Bs, int32_t C, int32_t H, void* stream);
int demo_atmoe_gate_softmax_bwd(const float* dgates, const float* gates, const float* q, const float* k, float* dq, float* dk,
                                int32_t Bs, int32_t C, int32_t H, void* stream);

/* ---- DGAF-V3 (modeling/dual_gated_fusion.py:558-745: attention pooling :661-679, IEG :681-719, MIG :721-730, blend :732-745),
 * the fusion head of the reference's `DeMo` class with MODEL.USE_DGAF (make_model.py:835-865) -------------------------------- */
/* Attention pooling in the folded single-query form.  S (3*Bs*L, ldS) fp32 from demo_gemm against the folded keys: column
 * s*nh + h of the rows of modality s = score of head h.  For sample b, modality s, head h: softmax over the L-1 PATCH tokens
 * (row 0 of an image, the cls token, is not part of DGAF's input), pooled[b,s,h,:] = sum_n p_n tokens_n -> fp32 (Bs,3,nh,C);
 * probs (optional, (Bs,3,nh,L-1)) kept for the backward. */
int demo_dgaf_attn_pool(const float* S, int64_t ldS, const void* tokens, int32_t dt, float* pooled, float* probs, int32_t Bs,
                        int32_t L, int32_t C, int32_t nh, void* stream);
/* Backward of the above: dS (3*Bs*L, ldS) and dtokens (3*Bs*L, C), both fp32, are fully written (zeros outside a modality's
 * own columns and on the cls rows). */
int demo_dgaf_attn_pool_bwd(const float* dpooled, const float* pooled, const float* probs, const void* tokens, int32_t dt,
                            float* dS, int64_t ldS, float* dtokens, int32_t Bs, int32_t L, int32_t C, int32_t nh, void* stream);
/* Both gates and the blend for one sample per workgroup.  h (Bs,3,C) pooled + LayerNormed modality features, E (Bs,3,C) =
 * entropy_proj(h), g1n (Bs,C) = LayerNorm(gate_net.0(cat h)) before the ReLU, W2 (3,C) / b2 (3) = gate_net.3, alpha_raw = _alpha.
 * fused (Bs,C) = a * sum_s softmax_s(mean(E_s) exp(-H(h_s)/tau)) h_s + (1-a) * sum_s sigmoid(W2_s . relu(g1n) + b2_s) h_s, a = sigmoid(_alpha).
 * relu_out (optional, (Bs,C)) = relu(g1n); aux (optional, (Bs,16)) = {ew[3], gate[3], H[3], z[3], sum|h|[3]} for the backward. */
int demo_dgaf_gate_fuse(const float* h, const float* E, const float* g1n, const float* W2, const float* b2, const float* alpha_raw,
                        float tau, float* fused, float* relu_out, float* aux, int32_t Bs, int32_t C, void* stream);
/* Backward: dh (Bs,3,C) = dres (optional, gradient reaching h directly) + blend + entropy paths; dE (Bs,3,C); dg1n (Bs,C);
 * dlogit (Bs,3) = gradient of gate_net.3's output before the sigmoid (its weight/bias gradients are the caller's GEMMs);
 * dalpha (Bs) = per-sample gradient of _alpha. */
int demo_dgaf_gate_fuse_bwd(const float* dfused, const float* dres, const float* h, const float* g1n, const float* W2,
                            const float* aux, const float* alpha_raw, float tau, float* dh, float* dE, float* dg1n, float* dlogit,
                            float* dalpha, int32_t Bs, int32_t C, void* stream);
/* out[b, s*C + c] = h[b,s,c] + enh[b,c]  (:739-745) and the enhancement's gradient denh[b,c] = sum_s dout[b, s*C + c] */
int demo_dgaf_combine(const float* h, const float* enh, float* out, int32_t Bs, int32_t C, void* stream);
int demo_dgaf_sum3(const float* dout, float* denh, int32_t Bs, int32_t C, void* stream);

/* ---- backward-only entry points ------------------------------------------------------------------------- */
/* Weight-gradient GEMM, both operands reduction-major (rows = tokens, exactly as the forward left them):
 * C[Mo,No] (+)= alpha * sum_r A[r,Mo] * B[r,No]; bf16 MFMA, split over the reduction with fp32 atomics.
 * dW of every nn.Linear in the trunk: A = dY (rows,N), B = X (rows,K) -> C = dW (N,K) inside the flat grad buffer. */
int demo_gemm_tn_bf16(const void* A, const void* B, float* C, int32_t Mo, int32_t No, int32_t Kr, int64_t lda, int64_t ldb,
                      int64_t ldc, float alpha, int32_t accumulate, float* colsum, float* workspace, int64_t workspace_elems,
                      void* stream);
/* floats of `workspace` with which the call above combines its reduction slices in two passes (plain partial tiles + a small
 * summing kernel) instead of fp32 atomics into C; 0 if the shape is not split.  workspace may be NULL (atomics). */
int64_t demo_gemm_tn_workspace(int32_t Mo, int32_t No, int32_t Kr);
/* colsum (optional, fp32 [Mo]): colsum[m] += sum_r A[r,m] -- the bias gradient of the same Linear, computed by the waves
 * that already hold the A fragments (no extra pass over dY). */
/* Round 3: up to four such products over the SAME Kr reduction rows as one launch and one summing launch -- the weight gradients of
 * one transformer block (solver step: engine/processor.py:99-104 consumes them all at once).  Each item keeps its own split of the
 * reduction; the grids run as one, longest slices first, so every product's tail, its burst of partial-tile stores and three launch
 * ramps disappear under the next product's main loop.  Mo, No multiples of 256 and Kr >= 2048 for every item; colsum for every item or
 * for none; workspace of demo_gemm_tn_group_workspace floats (caller-owned, like every buffer). */
typedef struct demo_tn_item {
  const void* A;      /* (Kr, Mo) bf16, leading dimension lda */
  const void* B;      /* (Kr, No) bf16, leading dimension ldb */
  float* C;           /* (Mo, No) fp32, leading dimension ldc */
  int32_t Mo, No;
  int64_t lda, ldb, ldc;
  float* colsum;      /* optional [Mo] */
} demo_tn_item;
int demo_gemm_tn_bf16_group(const demo_tn_item* items, int32_t n, int32_t Kr, float alpha, int32_t accumulate, float* workspace,
                            int64_t workspace_elems, void* stream);
int64_t demo_gemm_tn_group_workspace(const demo_tn_item* items, int32_t n, int32_t Kr);
/* Self-attention backward (same packing as demo_attention_fwd; lse from the forward): dqkv (Bn*L, 3D). */
int demo_layernorm_set_lp8(int on);    /* tuning hook: 1 (default) = demo_layernorm_bwd_lp takes the 8-column kernel when dy and x are both bf16 */
int demo_attention_set_variant(int v); /* tuning hook: 1 (default) = forward with K/V only in LDS (4 workgroups per CU), 0 = the round-1 form */
int demo_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int32_t Bn,
                       int32_t L, int32_t H, int32_t dh, int32_t dt, void* stream);
/* Adjoint of demo_embed_tokens w.r.t. its un-normalised sum `pre` (dpre fp32 (Bn*(1+np), W)):
 * dpatch (dtype dt_patch) = rows 1.., dcls += sum_b row 0, dpos[t] += sum_b row t, dcam[cam[b]] += cam_scale * rows.
 * cam[b] outside [0, n_cam) contributes nothing to dcam (and sets DEMO_ERR_CAM_RANGE in *err, as in the forward). */
int demo_embed_tokens_bwd(const float* dpre, const int64_t* cam, int32_t n_cam, float cam_scale, int32_t cam_all_tokens, void* dpatch,
                          int32_t dt_patch, float* dcls, float* dpos, float* dcam, int32_t Bn, int32_t np, int32_t W, int32_t* err,
                          void* stream);
/* y = act(pre), fp32, n elements (used where the product itself ran split-K and could not apply the activation) */
int demo_act_fwd(const float* pre, int32_t kind, float* y, int64_t n, void* stream);
/* dx = dy * act'(pre), fp32, n elements */
int demo_act_bwd(const float* dy, const float* pre, int32_t kind, float* dx, int64_t n, void* stream);
/* BatchNorm1d backward with batch statistics (same grouping / gate conventions as demo_batchnorm_fwd);
 * dw, db are accumulated into; dgate (Bn,n_inner,n_outer) is accumulated into with atomics (zero it first). */
int demo_batchnorm_bwd(const float* dy, const float* x, const float* w, const float* b, const float* save_mean,
                       const float* save_invstd, float* dx, int32_t accumulate_dx, float* dw, float* db, int32_t Bn, int32_t F,
                       int32_t gsz, int32_t n_inner, int32_t n_outer, int64_t pstride, const float* gate, float* dgate, void* stream);
/* Fused flat Adam step over the flat fp32 parameter buffer (solver/make_optimizer.py: torch.optim.Adam with L2 weight
 * decay; all parameter groups of the reference share lr and decay): updates p, m, v, optionally writes the bf16 weight
 * shadow and zeroes g.  step >= 1 is the 1-based update count used for bias correction.  hyper_dev (optional, device
 * memory, floats {lr, update count t}) overrides lr and step: the form a captured (hipGraph) training step uses -- the
 * launch arguments of a replay are frozen, so the count is advanced by a captured device op and lr by a stream-ordered fill. */
int demo_adam_step(float* p, float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int32_t step, int32_t zero_grad, const float* hyper_dev, void* stream);
/* The flat optimizer step in general form (solver/make_optimizer.py:4-28): kind 0 = torch.optim.Adam, 1 = AdamW (decoupled decay),
 * 2 = SGD with momentum beta1 (v may be NULL; bias corrections unused).  Parameter groups: block_class (optional, uint8 per 64-element
 * block of the flat buffer -- every parameter starts on such a boundary) indexes class_table (device floats {lr multiplier, weight
 * decay} per class): WEIGHT_DECAY_BIAS for "bias" keys, 2 x lr for classifiers under LARGE_FC_LR, and {0, 0} for parameters the
 * reference's optimizer never moves (requires_grad False, or no gradient in this configuration: torch skips grad-None parameters). */
int demo_optim_step(int32_t kind, float* p, float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int32_t step, int32_t zero_grad, const float* hyper_dev, const uint8_t* block_class,
                    const float* class_table, void* stream);
/* out[b, 0:C] += dtokens[b, 0, :]   (row-0 gradient of a token stream -> gradient of the global feature) */
int demo_token0_grad(const float* dtokens, float* out, int64_t out_ld, int32_t Bn, int32_t L, int32_t C, void* stream);
/* final trunk-output gradient (dtype dt): row 0 = dcls[b] (+ dtok row 0 if keep_row0); row l>=1 = dtok[b,l] + dmean[b]/(L-1);
 * dtok / dcls / dmean may each be NULL (treated as zero). */
int demo_tokens_grad(const float* dtok, const float* dcls, int64_t cls_ld, const float* dmean, int64_t mean_ld, int32_t keep_row0,
                     void* out, int32_t dt, int32_t Bn, int32_t L, int32_t C, void* stream);

/* ---- ATMoE at inference, fused (AttnMOE.py:50-68 gate, :24-47 experts, :85-121 MoM.forward) --------------------------------------
 * Eval mode only (every BatchNorm1d is the affine map of its running statistics): the whole module in TWO launches.
 * The gate's score q_h . k_jh = (Wq[h] x2) . (Wk[h] f_j) is evaluated as x2 . (G_h f_j) with G_h = Wq[h]^T Wk[h] (C x C), a product
 * of weights only: gfold = [G_0; ..; G_{H-1}] (H*C, C) is computed by the caller once per parameter version (two demo_gemm calls per
 * head) -- the q projection, the one step that needs the finished x2, no longer exists.
 * stage 1: independent LDS-staged tiles of 64 output columns x 32 rows: the gate's linear_re.0 as ceil(7C/256) partial sums (ws_part),
 * the folded keys t_j = G f_j (ws_tkey, (7B, H*C)), the 7*H experts y_jh = BN(QuickGELU(W_hj f_j[h] + b_hj)) (ws_y, (B, 7C));
 * stage 2, one workgroup per sample: x2 = BN(QuickGELU(sum of the partials + b1)), g_hj = softmax_j(x2 . t_j[h] / sqrt(d)),
 * moe[b, j*C + h*d + n] = g_hj * y_jh[n].  No atomics; every workspace element is written before it is read, nothing needs clearing.
 * feats (B,7,C) fp32; weights are pointers into the flat parameter / statistics buffers: ew / eb / ebn_w / ebn_b address expert
 * (h=0, j=0), expert (h, j) sits (7h + j) * expert_stride floats further (stat_stride for the running mean / var).
 * ws_part: demo_atmoe_eval_part_elems(B, C) floats; ws_tkey: 7*B*H*C; ws_y: B*7*C; gates (B,H,7) optional.
 * demo_atmoe_eval_supported: 1 if (C, H) fits the kernels (C <= 1024, C % H == 0, (C/H) % 4 == 0, 7 H <= 128). */
int demo_atmoe_eval_supported(int32_t C, int32_t H);
int64_t demo_atmoe_eval_part_elems(int32_t B, int32_t C);
int demo_atmoe_eval_fwd(const float* feats, const float* w1, const float* b1, const float* bn1_w, const float* bn1_b,
                        const float* bn1_mean, const float* bn1_var, const float* gfold, const float* ew, const float* eb,
                        const float* ebn_w, const float* ebn_b, const float* ebn_mean, const float* ebn_var, int64_t expert_stride,
                        int64_t stat_stride, float eps, float* ws_part, float* ws_tkey, float* ws_y, float* moe, float* gates,
                        int32_t B, int32_t C, int32_t H, void* stream);

/* ---- training losses: value and gradient in one launch -----------------------------------------
 * CrossEntropyLabelSmooth (layers/softmax_loss.py:23-34): *loss += weight * mean_b sum_k -t_bk log softmax(logits_b)_k with
 * t = (1-eps) onehot(target) + eps/K;  dlogits (optional, (B, ldd)) = weight * d/dlogits.  loss is a device scalar the
 * caller zeroes once per step.  target[b] outside [0, K) (torch's scatter_ raises there, softmax_loss.py:30-31): the sample
 * contributes a NaN to *loss, its dlogits row is NaN, nothing out of bounds is read, and DEMO_ERR_LABEL_RANGE is OR-ed into *err
 * (device int32, optional). */
int demo_xent_label_smooth(const float* logits, int64_t ld, const int64_t* target, int32_t B, int32_t K, float eps, float weight,
                           float* loss, float* dlogits, int64_t ldd, int32_t* err, void* stream);
/* TripletLoss() without margin (layers/triplet_loss.py:16-31,51-104,121-135): euclidean distances as the reference forms them
 * (sqrt(clamp(|a|^2+|b|^2-2ab, 1e-12))), batch-hard mining (hardest positive incl. self, hardest negative), SoftMarginLoss:
 * *loss += weight * mean_i softplus(d_ap - d_an);  dfeat (optional, (B, ldd), OVERWRITTEN) = weight * d/dfeat.
 * workspace: 2*B*B floats (Gram matrix F F^T and the sparse gradient weights Wt with dfeat = Wt F; both products run on
 * demo_gemm's fp32 kernel). */
int demo_triplet_soft_margin(const float* feat, int64_t ld, const int64_t* labels, int32_t B, int32_t D, float weight, float* loss,
                             float* dfeat, int64_t ldd, float* workspace, void* stream);
/* The same with the reference's other triplet form (layers/triplet_loss.py:114-117,131-134; MODEL.NO_MARGIN False): margin >= 0 selects
 * MarginRankingLoss(d_an, d_ap, 1) = mean_i max(0, d_ap - d_an + margin); margin < 0 is the soft-margin form above. */
int demo_triplet_loss(const float* feat, int64_t ld, const int64_t* labels, int32_t B, int32_t D, float weight, float margin, float* loss,
                      float* dfeat, int64_t ldd, float* workspace, void* stream);

/* ---- retrieval evaluator (utils/metrics.py) ---------------------------------------------------------
 * R1_mAP_eval.compute (:341-369): F.normalize -> euclidean_distance (:395-401) -> eval_func (:110-169); R1_mAP (:172-218) with
 * eval_func_msrv (:12-107) differs only in the removal key (scene id instead of camera id). */
int demo_l2_normalize_rows(float* x, int64_t ld, int32_t N, int32_t D, float eps, void* stream);      /* x /= max(||x||_2, eps) */
int demo_row_sqnorm(const float* x, int64_t ld, int32_t N, int32_t D, float* out, void* stream);       /* out[i] = sum_d x[i,d]^2 */
/* dist[i,j] = (a[i] + b[j]) + dist[i,j]: completes |q|^2 + |g|^2 - 2 q.g^T around a demo_gemm with alpha = -2 */
int demo_add_outer_sum(float* dist, int64_t ld, const float* a, const float* b, int32_t nq, int32_t ng, void* stream);
/* per query i: sort the gallery by dist[i,:] (ties by gallery index), drop items with g_pids == q_pids[i] AND g_keys ==
 * q_keys[i] (keys = camera ids: eval_func; scene ids: eval_func_msrv), then ap[i] = average precision over the kept items
 * (float64; -1 if the identity does not occur), first_hit[i] = 0-based kept rank of the first match (-1 likewise),
 * num_rel[i] (optional) = number of kept matches, order_out (optional, (nq, ng) int32) = the sorted gallery indices.
 * ng <= 16384 (one workgroup sorts a row in LDS). */
int demo_rank_eval(const float* dist, int64_t ld, const int64_t* q_pids, const int64_t* g_pids, const int64_t* q_keys,
                   const int64_t* g_keys, int32_t nq, int32_t ng, double* ap, int32_t* first_hit, int32_t* num_rel,
                   int32_t* order_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DEMO_HIP_H */
