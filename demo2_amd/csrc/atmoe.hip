// ATMoE at inference in TWO launches (reference: modeling/moe/AttnMOE.py:50-68 gate CrossAttention, :24-47 Expert / ExpertHead,
// :85-121 MoM.forward).  In eval mode every BatchNorm1d is a per-feature affine map of its running statistics, so nothing couples
// the samples and the whole module is a short dependency chain over the 7 HDM features f_1..f_7 (each (B, C)):
//
//   stage 1 -- everything that depends on the features only, one launch over independent 64-column tiles:
//       x2  = BN_g(QuickGELU(W_1 [f_1..f_7] + b_1))                                  (B, C)      gate's linear_re   (:52-58, :62)
//       k_j = W_k f_j                                                                (B, 7, C)   gate keys          (:64)
//       y_jh = BN_hj(QuickGELU(W_hj f_j[h*d:(h+1)*d] + b_hj))                        (B, 7C)     the 7*H experts    (:27-35, :43-47)
//   stage 2 -- per head h and block of samples:
//       q_h = W_q[h*d:(h+1)*d, :] x2 ;  g_hj = softmax_j(q_h . k_jh / sqrt(d)) ;  out[b, j*C + h*d + n] = g_hj y_jh[n]   (:63-68, :116-117)
//
// Both stages are small fp32 products whose cost is the weights they stream (11.3 MB at C = 512, H = 4) and, below B ~ 1k, launch
// latency: the training-time schedule issues 8 launches for the same arithmetic.  Weight tiles are staged through LDS (64 x 32
// fp32, padded rows: conflict-free for the 4-column register tiles), every thread owns a 2 x 4 (stage 1) or 1 x d/16 (stage 2) block
// of outputs.  Training keeps the multi-launch schedule (batch statistics couple the samples; activations are kept for backward).
#include "common.h"

struct demo_atmoe_eval_args {
  const float* feats;                                       // (B, 7, C)
  const float *w1, *b1, *bn1_w, *bn1_b, *bn1_m, *bn1_v;     // gate.linear_re.{0,2}
  const float *wq, *wk;                                     // gate.q_ / gate.k_ (C, C), no bias
  const float *ew, *eb, *ebn_w, *ebn_b;                     // expert (h = 0, j = 0): mlp.0.{weight (d,d), bias}, mlp.2.{weight, bias}; expert (h, j) at + (7h + j) * estride
  const float *ebn_m, *ebn_v;                               // running statistics of expert (0, 0); expert (h, j) at + (7h + j) * sstride
  int64_t estride, sstride;
  float *x2, *kg, *y;                                       // workspaces (B, C), (B, 7, C), (B, 7C)
  float* moe;                                               // (B, 7C) output, layout j*C + h*d + n
  float* gates;                                             // optional (B, H, 7)
  int B, C, H;
  float eps;
};

namespace {

constexpr int TN = 64, TB = 32, TK = 32;

__global__ __launch_bounds__(256) void atmoe_eval_stage1_kernel(const demo_atmoe_eval_args p) {
  __shared__ float Xs[TB][TK + 1];
  __shared__ float Ws[TN][TK + 1];
  const int C = p.C, d = C / p.H;
  const int ctC = (C + TN - 1) / TN, ctD = (d + TN - 1) / TN;
  const int nG = ctC, nK = 7 * ctC;
  int t = blockIdx.x;
  const int b0 = blockIdx.y * TB;
  const float *X, *W, *bias = nullptr, *bw = nullptr, *bb = nullptr, *bm = nullptr, *bv = nullptr;
  float* out;
  int K, ncol;                                               // reduction length, valid columns of this tile
  const int64_t x_rs = 7 * (int64_t)C;
  int64_t o_rs;
  if (t < nG) {                                              // gate: x2 = BN(QuickGELU(W1 [f_1..f_7] + b1))
    const int n0 = t * TN;
    X = p.feats; W = p.w1 + (int64_t)n0 * 7 * C; K = 7 * C; ncol = min(TN, C - n0);
    bias = p.b1 + n0; bw = p.bn1_w + n0; bb = p.bn1_b + n0; bm = p.bn1_m + n0; bv = p.bn1_v + n0;
    out = p.x2 + n0; o_rs = C;
  } else if (t < nG + nK) {                                  // gate keys: k_j = Wk f_j
    t -= nG;
    const int j = t / ctC, n0 = (t - j * ctC) * TN;
    X = p.feats + (int64_t)j * C; W = p.wk + (int64_t)n0 * C; K = C; ncol = min(TN, C - n0);
    out = p.kg + (int64_t)j * C + n0; o_rs = x_rs;
  } else {                                                   // expert (h, j): y = BN(QuickGELU(W f_j[h*d:(h+1)*d] + b))
    t -= nG + nK;
    const int e = t / ctD, n0 = (t - e * ctD) * TN, h = e / 7, j = e - 7 * h;
    X = p.feats + (int64_t)j * C + h * d; W = p.ew + e * p.estride + (int64_t)n0 * d; K = d; ncol = min(TN, d - n0);
    bias = p.eb + e * p.estride + n0; bw = p.ebn_w + e * p.estride + n0; bb = p.ebn_b + e * p.estride + n0;
    bm = p.ebn_m + e * p.sstride + n0; bv = p.ebn_v + e * p.sstride + n0;
    out = p.y + (int64_t)j * C + h * d + n0; o_rs = x_rs;
  }
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  for (int k0 = 0; k0 < K; k0 += TK) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {                            // weight tile: 64 rows x 32 k = 512 float4
      const int f = tid + i * 256, n = f >> 3, kq = f & 7;
      float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < ncol) w4 = *reinterpret_cast<const float4*>(W + (int64_t)n * K + k0 + kq * 4);
      Ws[n][kq * 4 + 0] = w4.x; Ws[n][kq * 4 + 1] = w4.y; Ws[n][kq * 4 + 2] = w4.z; Ws[n][kq * 4 + 3] = w4.w;
    }
    {                                                        // input tile: 32 samples x 32 k = 256 float4
      const int s = tid >> 3, kq = tid & 7;
      float4 x4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b0 + s < p.B) x4 = *reinterpret_cast<const float4*>(X + (int64_t)(b0 + s) * x_rs + k0 + kq * 4);
      Xs[s][kq * 4 + 0] = x4.x; Xs[s][kq * 4 + 1] = x4.y; Xs[s][kq * 4 + 2] = x4.z; Xs[s][kq * 4 + 3] = x4.w;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TK; ++kk) {
      const float a0 = Xs[ty * 2][kk], a1 = Xs[ty * 2 + 1][kk];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float w = Ws[tx * 4 + c][kk];
        acc[0][c] = fmaf(a0, w, acc[0][c]);
        acc[1][c] = fmaf(a1, w, acc[1][c]);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int n = tx * 4 + c;
    if (n >= ncol) continue;
    float scale = 1.f, shift = 0.f;
    const float bs = bias ? bias[n] : 0.f;
    if (bw) {                                                // eval-mode BatchNorm1d: (v - mean) / sqrt(var + eps) * w + b
      scale = bw[n] * rsqrtf(bv[n] + p.eps);
      shift = bb[n] - bm[n] * scale;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int b = b0 + ty * 2 + r;
      if (b >= p.B) continue;
      float v = acc[r][c] + bs;
      if (bw) v = fmaf(act_apply(v, DEMO_ACT_QUICKGELU), scale, shift);
      out[(int64_t)b * o_rs + n] = v;
    }
  }
}

// grid (H, ceil(B / 16)); thread (s = tid / 16, tx = tid % 16) owns q_h[s][tx + 16 i], i < d / 16
constexpr int SB = 16, DMAX = 128;
__global__ __launch_bounds__(256) void atmoe_eval_stage2_kernel(const demo_atmoe_eval_args p) {
  __shared__ float Xs[SB][TK + 1];
  __shared__ float Ws[DMAX][TK + 1];
  const int C = p.C, H = p.H, d = C / H, h = blockIdx.x, b0 = blockIdx.y * SB;
  const int tid = threadIdx.x, tx = tid & 15, s = tid >> 4, nI = d >> 4;
  const int b = b0 + s;
  const float* Wq = p.wq + (int64_t)h * d * C;
  float acc[DMAX / 16];
#pragma unroll
  for (int i = 0; i < DMAX / 16; ++i) acc[i] = 0.f;
  for (int k0 = 0; k0 < C; k0 += TK) {
    for (int f = tid; f < d * 8; f += 256) {                 // weight tile: d rows x 32 k
      const int n = f >> 3, kq = f & 7;
      const float4 w4 = *reinterpret_cast<const float4*>(Wq + (int64_t)n * C + k0 + kq * 4);
      Ws[n][kq * 4 + 0] = w4.x; Ws[n][kq * 4 + 1] = w4.y; Ws[n][kq * 4 + 2] = w4.z; Ws[n][kq * 4 + 3] = w4.w;
    }
    if (tid < SB * 8) {
      const int ss = tid >> 3, kq = tid & 7;
      float4 x4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b0 + ss < p.B) x4 = *reinterpret_cast<const float4*>(p.x2 + (int64_t)(b0 + ss) * C + k0 + kq * 4);
      Xs[ss][kq * 4 + 0] = x4.x; Xs[ss][kq * 4 + 1] = x4.y; Xs[ss][kq * 4 + 2] = x4.z; Xs[ss][kq * 4 + 3] = x4.w;
    }
    __syncthreads();
#pragma unroll 8
    for (int kk = 0; kk < TK; ++kk) {
      const float a = Xs[s][kk];
#pragma unroll
      for (int i = 0; i < DMAX / 16; ++i)
        if (i < nI) acc[i] = fmaf(a, Ws[tx + 16 * i][kk], acc[i]);
    }
    __syncthreads();
  }
  // scores against the 7 keys of this head, softmax over the 7, gate the experts' outputs
  const int64_t row = (int64_t)(b < p.B ? b : 0) * 7 * C;     // (clamped: the wave must stay whole for the shuffles below)
  float lg[7];
  const float sc = rsqrtf((float)d);
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < DMAX / 16; ++i)
      if (i < nI) v = fmaf(acc[i], p.kg[row + (int64_t)j * C + h * d + tx + 16 * i], v);
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    lg[j] = v * sc;
  }
  float m = lg[0];
#pragma unroll
  for (int j = 1; j < 7; ++j) m = fmaxf(m, lg[j]);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 7; ++j) { lg[j] = __expf(lg[j] - m); sum += lg[j]; }
  const float inv = 1.f / sum;
  if (b >= p.B) return;
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const float gj = lg[j] * inv;
    if (p.gates && tx == 0) p.gates[((int64_t)b * H + h) * 7 + j] = gj;
#pragma unroll
    for (int i = 0; i < DMAX / 16; ++i)
      if (i < nI) {
        const int64_t o = row + (int64_t)j * C + h * d + tx + 16 * i;
        p.moe[o] = p.y[o] * gj;
      }
  }
}

}  // namespace

extern "C" int demo_atmoe_eval_supported(int32_t C, int32_t H) {
  if (H < 1 || C % H) return 0;
  const int d = C / H;
  return (C % TK == 0 && d % TK == 0 && d % 16 == 0 && d <= DMAX) ? 1 : 0;
}

extern "C" int demo_atmoe_eval_fwd(const float* feats, const float* w1, const float* b1, const float* bn1_w, const float* bn1_b,
                                   const float* bn1_mean, const float* bn1_var, const float* wq, const float* wk, const float* ew,
                                   const float* eb, const float* ebn_w, const float* ebn_b, const float* ebn_mean, const float* ebn_var,
                                   int64_t expert_stride, int64_t stat_stride, float eps, float* ws_x2, float* ws_k, float* ws_y, float* moe,
                                   float* gates, int32_t B, int32_t C, int32_t H, void* stream) {
  DEMO_REQUIRE(B >= 1 && demo_atmoe_eval_supported(C, H), "demo_atmoe_eval_fwd: unsupported B=%d C=%d H=%d", B, C, H);
  DEMO_REQUIRE(feats && w1 && b1 && bn1_w && bn1_b && bn1_mean && bn1_var && wq && wk && ew && eb && ebn_w && ebn_b && ebn_mean && ebn_var &&
                   ws_x2 && ws_k && ws_y && moe, "demo_atmoe_eval_fwd: null argument");
  demo_atmoe_eval_args p;
  p.feats = feats; p.w1 = w1; p.b1 = b1; p.bn1_w = bn1_w; p.bn1_b = bn1_b; p.bn1_m = bn1_mean; p.bn1_v = bn1_var;
  p.wq = wq; p.wk = wk; p.ew = ew; p.eb = eb; p.ebn_w = ebn_w; p.ebn_b = ebn_b; p.ebn_m = ebn_mean; p.ebn_v = ebn_var;
  p.estride = expert_stride; p.sstride = stat_stride; p.x2 = ws_x2; p.kg = ws_k; p.y = ws_y; p.moe = moe; p.gates = gates;
  p.B = B; p.C = C; p.H = H; p.eps = eps;
  const int d = C / H, ctC = (C + TN - 1) / TN, ctD = (d + TN - 1) / TN;
  const dim3 g1((unsigned)(ctC + 7 * ctC + 7 * H * ctD), (unsigned)((B + TB - 1) / TB));
  hipLaunchKernelGGL(atmoe_eval_stage1_kernel, g1, dim3(256), 0, (hipStream_t)stream, p);
  DEMO_LAUNCH_CHECK("demo_atmoe_eval_fwd(stage 1)");
  const dim3 g2((unsigned)H, (unsigned)((B + SB - 1) / SB));
  hipLaunchKernelGGL(atmoe_eval_stage2_kernel, g2, dim3(256), 0, (hipStream_t)stream, p);
  DEMO_LAUNCH_CHECK("demo_atmoe_eval_fwd(stage 2)");
  return 0;
}
