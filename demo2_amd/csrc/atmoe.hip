// ATMoE at inference in TWO launches (reference: modeling/moe/AttnMOE.py:50-68 gate CrossAttention, :24-47 Expert / ExpertHead,
// :85-121 MoM.forward).  In eval mode every BatchNorm1d is a per-feature affine map of its running statistics, so nothing couples
// the samples and the whole module is a short dependency chain over the 7 HDM features f_1..f_7 (each (B, C)).
//
// The gate's score of head h for feature j is q_h . k_jh with q_h = W_q[h] x2 and k_jh = W_k[h] f_j (W_q[h], W_k[h]: the d rows of
// head h), i.e. x2^T (W_q[h]^T W_k[h]) f_j: with the weights-only product G_h = W_q[h]^T W_k[h] (C x C; the caller caches the stack
// G = [G_0; ..; G_{H-1}], (H C, C), per parameter version) the q projection -- the one step that needs the FINISHED x2 and made a
// third dependent stage, or a second stage of H workgroups walking 256 KB of W_q each (measured: 103 us at B = 1) -- disappears:
//
//   stage 1 -- everything that depends on the features only, one launch over independent tiles of 64 output columns x 32 rows:
//       x1[ks] = W_1[:, chunk ks] [f_1..f_7][chunk ks]         (7C/256, B, C)  gate's linear_re.0 as 7C/256 partial sums   (:52-58, :62)
//       t_j    = G f_j                                         (7B, H C)       folded keys                                  (:63-66)
//       y_jh   = BN_hj(QuickGELU(W_hj f_j[h*d:(h+1)*d] + b_hj)) (B, 7C)        the 7*H experts                               (:27-35, :43-47)
//   stage 2 -- one workgroup per sample:  x2 = BN_g(QuickGELU(sum_ks x1[ks] + b_1));  s_hj = x2 . t_j[h] / sqrt(d);
//       g_hj = softmax_j(s_hj);  out[b, j*C + h*d + n] = g_hj y_jh[n]                                                      (:63-68, :116-117)
//
// No atomics, no accumulator to clear: every workspace element is written by exactly one thread before it is read.  Both stages are
// small fp32 products whose cost is the weights they stream (7.3 MB gate + 4 MB G + 1.8 MB experts at C = 512, H = 4) and, below
// B ~ 1k, launch latency: the training-time schedule issues 8 launches for the same arithmetic.  Weight tiles are staged through LDS
// as they lie in memory (row = output column, 128 reduction elements + 4 pad: 132 floats, so that ds_read_b128 along the reduction
// is conflict-free when the 16 lanes of a read group own columns 16 apart); a thread owns 2 rows x 4 columns of outputs and
// consumes four reduction elements per LDS read.  Training keeps the multi-launch schedule (batch statistics couple the samples;
// activations are kept for backward).
#include "common.h"

struct atmoe_eval_args {
  const float* feats;                                       // (B, 7, C)
  const float *w1, *b1, *bn1_w, *bn1_b, *bn1_m, *bn1_v;     // gate.linear_re.{0,2}
  const float* gfold;                                       // (H C, C): rows h*C + c = (W_q[h]^T W_k[h])[c, :]
  const float *ew, *eb, *ebn_w, *ebn_b;                     // expert (h = 0, j = 0): mlp.0.{weight (d,d), bias}, mlp.2.{weight, bias}; expert (h, j) at + (7h + j) * estride
  const float *ebn_m, *ebn_v;                               // running statistics of expert (0, 0); expert (h, j) at + (7h + j) * sstride
  int64_t estride, sstride;
  float* part;                                              // workspace (ceil(7C/256), B, C): partial sums of the gate's first Linear
  float* tkey;                                              // workspace (7B, H C): folded keys
  float* y;                                                 // workspace (B, 7C): expert outputs
  float* moe;                                               // (B, 7C) output, layout j*C + h*d + n
  float* gates;                                             // optional (B, H, 7)
  int B, C, H;
  float eps;
};

namespace {

constexpr int TN = 64, TB = 32, TK = 128, LDK = TK + 4, KCHUNK = 256, CMAX = 1024;

// 1-D grid: [gate tiles: ctC x ksG x rtB] [folded-key tiles: ctHC x rtK] [expert tiles: 7H x ctD x rtB]
__global__ __launch_bounds__(256) void atmoe_eval_stage1_kernel(const atmoe_eval_args p) {
  __shared__ __attribute__((aligned(16))) float Ws[TN][LDK];
  __shared__ __attribute__((aligned(16))) float Xs[TB][LDK];
  const int C = p.C, d = C / p.H, HC = p.H * C;
  const int ctC = (C + TN - 1) / TN, ctD = (d + TN - 1) / TN, ctHC = (HC + TN - 1) / TN;
  const int ksG = (7 * C + KCHUNK - 1) / KCHUNK;
  const int rtB = (p.B + TB - 1) / TB, rtK = (7 * p.B + TB - 1) / TB;
  const int nG = ctC * ksG * rtB, nK = ctHC * rtK;
  int t = blockIdx.x;
  const float *X, *W, *bias = nullptr, *bw = nullptr, *bb = nullptr, *bm = nullptr, *bv = nullptr;
  float* out;
  int K, kbeg, kend, ncol, nrows, r0;                        // row length of W, this tile's reduction range, valid columns / rows, first row
  int64_t x_rs, o_rs;
  if (t < nG) {                                              // gate: x1[ks] = W1[:, chunk] [f_1..f_7][chunk]
    const int rt = t % rtB, q = t / rtB, ct = q / ksG, ks = q - ct * ksG, n0 = ct * TN;
    X = p.feats; x_rs = 7 * (int64_t)C; nrows = p.B; r0 = rt * TB;
    W = p.w1 + (int64_t)n0 * 7 * C; K = 7 * C; kbeg = ks * KCHUNK; kend = min(K, kbeg + KCHUNK); ncol = min(TN, C - n0);
    out = p.part + (int64_t)ks * p.B * C + n0; o_rs = C;
  } else if (t < nG + nK) {                                  // folded keys: t[(b, j), :] = G f_j[b]   (rows of feats viewed (7B, C))
    t -= nG;
    const int rt = t % rtK, ct = t / rtK, n0 = ct * TN;
    X = p.feats; x_rs = C; nrows = 7 * p.B; r0 = rt * TB;
    W = p.gfold + (int64_t)n0 * C; K = C; kbeg = 0; kend = C; ncol = min(TN, HC - n0);
    out = p.tkey + n0; o_rs = HC;
  } else {                                                   // expert (h, j): y = BN(QuickGELU(W f_j[h*d:(h+1)*d] + b)), whole reduction
    t -= nG + nK;
    const int rt = t % rtB, q = t / rtB, e = q / ctD, n0 = (q - e * ctD) * TN, h = e / 7, j = e - 7 * h;
    X = p.feats + (int64_t)j * C + h * d; x_rs = 7 * (int64_t)C; nrows = p.B; r0 = rt * TB;
    W = p.ew + e * p.estride + (int64_t)n0 * d; K = d; kbeg = 0; kend = d; ncol = min(TN, d - n0);
    bias = p.eb + e * p.estride + n0; bw = p.ebn_w + e * p.estride + n0; bb = p.ebn_b + e * p.estride + n0;
    bm = p.ebn_m + e * p.sstride + n0; bv = p.ebn_v + e * p.sstride + n0;
    out = p.y + (int64_t)j * C + h * d + n0; o_rs = x_rs;
  }
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;     // thread: rows 2ty, 2ty+1 x columns tx, tx+16, tx+32, tx+48
  float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k0 = kbeg; k0 < kend; k0 += TK) {
    float4 wv[8], xv[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {                            // weight tile: 64 rows x 128 k = 2048 float4, rows as they lie in memory
      const int f = tid + i * 256, n = f >> 5, kq = f & 31;
      wv[i] = z4;
      if (n < ncol && k0 + kq * 4 < kend) wv[i] = *reinterpret_cast<const float4*>(W + (int64_t)n * K + k0 + kq * 4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {                            // input tile: 32 rows x 128 k
      const int f = tid + i * 256, s = f >> 5, kq = f & 31;
      xv[i] = z4;
      if (r0 + s < nrows && k0 + kq * 4 < kend) xv[i] = *reinterpret_cast<const float4*>(X + (int64_t)(r0 + s) * x_rs + k0 + kq * 4);
    }
    __syncthreads();                                         // (previous step's reads are done)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int f = tid + i * 256;
      *reinterpret_cast<float4*>(&Ws[f >> 5][(f & 31) * 4]) = wv[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + i * 256;
      *reinterpret_cast<float4*>(&Xs[f >> 5][(f & 31) * 4]) = xv[i];
    }
    __syncthreads();
#pragma unroll 4
    for (int kq = 0; kq < TK / 4; ++kq) {
      const float4 a0 = *reinterpret_cast<const float4*>(&Xs[ty * 2][kq * 4]), a1 = *reinterpret_cast<const float4*>(&Xs[ty * 2 + 1][kq * 4]);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 w = *reinterpret_cast<const float4*>(&Ws[tx + 16 * c][kq * 4]);
        acc[0][c] += (a0.x * w.x + a0.y * w.y) + (a0.z * w.z + a0.w * w.w);
        acc[1][c] += (a1.x * w.x + a1.y * w.y) + (a1.z * w.z + a1.w * w.w);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int n = tx + 16 * c;
    if (n >= ncol) continue;
    float scale = 1.f, shift = 0.f, bs = 0.f;
    if (bias) {                                              // eval-mode BatchNorm1d: (v - mean) / sqrt(var + eps) * w + b
      bs = bias[n];
      scale = bw[n] * rsqrtf(bv[n] + p.eps);
      shift = bb[n] - bm[n] * scale;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int row = r0 + ty * 2 + r;
      if (row >= nrows) continue;
      out[(int64_t)row * o_rs + n] = bias ? fmaf(act_apply(acc[r][c] + bs, DEMO_ACT_QUICKGELU), scale, shift) : acc[r][c];
    }
  }
}

// one workgroup per sample
__global__ __launch_bounds__(256) void atmoe_eval_stage2_kernel(const atmoe_eval_args p) {
  __shared__ __attribute__((aligned(16))) float x2[CMAX];
  __shared__ float lg[128];
  const int C = p.C, H = p.H, d = C / H, b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ksG = (7 * C + KCHUNK - 1) / KCHUNK;
  for (int c = tid; c < C; c += 256) {                       // x2 = BN(QuickGELU(x1 + b1))
    float v = p.b1[c];
    for (int ks = 0; ks < ksG; ++ks) v += p.part[((int64_t)ks * p.B + b) * C + c];
    const float sc = p.bn1_w[c] * rsqrtf(p.bn1_v[c] + p.eps);
    x2[c] = fmaf(act_apply(v, DEMO_ACT_QUICKGELU), sc, p.bn1_b[c] - p.bn1_m[c] * sc);
  }
  __syncthreads();
  const float scale = rsqrtf((float)d);
  for (int pr = wave; pr < 7 * H; pr += 4) {                 // score (h, j) = x2 . t_j[h] / sqrt(d): one wave per pair, whole wave active
    const int h = pr / 7, j = pr - 7 * h;
    const float* trow = p.tkey + ((int64_t)(b * 7 + j) * H + h) * C;
    float v = 0.f;
    for (int c4 = lane; c4 < (C >> 2); c4 += 64) {
      const float4 a = *reinterpret_cast<const float4*>(&x2[c4 * 4]), w = *reinterpret_cast<const float4*>(trow + c4 * 4);
      v += (a.x * w.x + a.y * w.y) + (a.z * w.z + a.w * w.w);
    }
    v = wave_sum(v);
    if (lane == 0) lg[pr] = v * scale;
  }
  __syncthreads();
  if (tid < H) {                                             // softmax over the 7 features of head tid
    float m = lg[tid * 7];
    for (int j = 1; j < 7; ++j) m = fmaxf(m, lg[tid * 7 + j]);
    float e[7], sum = 0.f;
#pragma unroll
    for (int j = 0; j < 7; ++j) { e[j] = __expf(lg[tid * 7 + j] - m); sum += e[j]; }
    const float inv = 1.f / sum;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      lg[tid * 7 + j] = e[j] * inv;
      if (p.gates) p.gates[((int64_t)b * H + tid) * 7 + j] = e[j] * inv;
    }
  }
  __syncthreads();
  const int64_t row = (int64_t)b * 7 * C;
  for (int i4 = tid; i4 < (7 * C) >> 2; i4 += 256) {         // gate the experts' outputs (d % 4 == 0: a float4 stays inside one head)
    const int i = i4 * 4, j = i / C, h = (i - j * C) / d;
    const float g = lg[h * 7 + j];
    const float4 yv = *reinterpret_cast<const float4*>(p.y + row + i);
    *reinterpret_cast<float4*>(p.moe + row + i) = make_float4(yv.x * g, yv.y * g, yv.z * g, yv.w * g);
  }
}

}  // namespace

extern "C" int demo_atmoe_eval_supported(int32_t C, int32_t H) {
  if (H < 1 || C < 4 || C % H) return 0;
  const int d = C / H;
  return (C % 4 == 0 && C <= CMAX && d % 4 == 0 && 7 * H <= 128) ? 1 : 0;
}

extern "C" int64_t demo_atmoe_eval_part_elems(int32_t B, int32_t C) { return (int64_t)((7 * C + KCHUNK - 1) / KCHUNK) * B * C; }

extern "C" int demo_atmoe_eval_fwd(const float* feats, const float* w1, const float* b1, const float* bn1_w, const float* bn1_b,
                                   const float* bn1_mean, const float* bn1_var, const float* gfold, const float* ew, const float* eb,
                                   const float* ebn_w, const float* ebn_b, const float* ebn_mean, const float* ebn_var, int64_t expert_stride,
                                   int64_t stat_stride, float eps, float* ws_part, float* ws_tkey, float* ws_y, float* moe, float* gates,
                                   int32_t B, int32_t C, int32_t H, void* stream) {
  DEMO_REQUIRE(B >= 1 && demo_atmoe_eval_supported(C, H), "demo_atmoe_eval_fwd: unsupported B=%d C=%d H=%d", B, C, H);
  DEMO_REQUIRE(feats && w1 && b1 && bn1_w && bn1_b && bn1_mean && bn1_var && gfold && ew && eb && ebn_w && ebn_b && ebn_mean && ebn_var &&
                   ws_part && ws_tkey && ws_y && moe, "demo_atmoe_eval_fwd: null argument");
  atmoe_eval_args p;
  p.feats = feats; p.w1 = w1; p.b1 = b1; p.bn1_w = bn1_w; p.bn1_b = bn1_b; p.bn1_m = bn1_mean; p.bn1_v = bn1_var;
  p.gfold = gfold; p.ew = ew; p.eb = eb; p.ebn_w = ebn_w; p.ebn_b = ebn_b; p.ebn_m = ebn_mean; p.ebn_v = ebn_var;
  p.estride = expert_stride; p.sstride = stat_stride; p.part = ws_part; p.tkey = ws_tkey; p.y = ws_y; p.moe = moe; p.gates = gates;
  p.B = B; p.C = C; p.H = H; p.eps = eps;
  const int d = C / H, ctC = (C + TN - 1) / TN, ctD = (d + TN - 1) / TN, ctHC = (H * C + TN - 1) / TN;
  const int ksG = (7 * C + KCHUNK - 1) / KCHUNK;
  const int64_t rtB = (B + TB - 1) / TB, rtK = (7 * (int64_t)B + TB - 1) / TB;
  const int64_t tiles = ctC * ksG * rtB + ctHC * rtK + 7 * H * ctD * rtB;
  DEMO_REQUIRE(tiles < (1ll << 31), "demo_atmoe_eval_fwd: B=%d too large", B);
  hipLaunchKernelGGL(atmoe_eval_stage1_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, p);
  DEMO_LAUNCH_CHECK("demo_atmoe_eval_fwd(stage 1)");
  hipLaunchKernelGGL(atmoe_eval_stage2_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, p);
  DEMO_LAUNCH_CHECK("demo_atmoe_eval_fwd(stage 2)");
  return 0;
}
