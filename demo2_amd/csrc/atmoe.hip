// ATMoE at inference in TWO launches (reference: modeling/moe/AttnMOE.py:50-68 gate CrossAttention, :24-47 Expert / ExpertHead,
// :85-121 MoM.forward).  In eval mode every BatchNorm1d is a per-feature affine map of its running statistics, so nothing couples
// the samples and the whole module is a short dependency chain over the 7 HDM features f_1..f_7 (each (B, C)):
//
//   stage 1 -- everything that depends on the features only, one launch over independent tiles of 64 output columns x <= 256 of
//              the reduction (so that NO workgroup streams more than 64 KB of weights: the gate's 7C-long reduction is cut into
//              7C/256 pieces that add their raw sums into a zeroed accumulator with fp32 atomics):
//       x1  += W_1[:, chunk] [f_1..f_7][chunk]                                      (B, C)      gate's linear_re.0   (:52-58, :62)
//       k_j += W_k[:, chunk] f_j[chunk]                                             (B, 7, C)   gate keys            (:64)
//       y_jh = BN_hj(QuickGELU(W_hj f_j[h*d:(h+1)*d] + b_hj))                        (B, 7C)     the 7*H experts      (:27-35, :43-47)
//   stage 2 -- per head h and block of 16 samples:
//       x2 = BN_g(QuickGELU(x1 + b_1)) (applied while the tile is staged);  q_h = W_q[h*d:(h+1)*d, :] x2 ;
//       g_hj = softmax_j(q_h . k_jh / sqrt(d)) ;  out[b, j*C + h*d + n] = g_hj y_jh[n]                      (:63-68, :116-117)
//       and it zeroes the OTHER accumulator buffer for the next call (the caller alternates two: no memset launch).
//
// Both stages are small fp32 products whose cost is the weights they stream (11.3 MB at C = 512, H = 4) and, below B ~ 1k, launch
// latency: the training-time schedule issues 8 launches for the same arithmetic.  Weight tiles are staged through LDS as they
// lie in memory (row = output column, 128 reduction elements + 4 pad: 132 floats, so that ds_read_b128 along the reduction is
// conflict-free when the 16 lanes of a read group own columns 16 apart), every thread owns a 2-sample x 4-column (stage 1) or
// 1-sample x d/16-column (stage 2) block of outputs and consumes four reduction elements per LDS read.  Training keeps the
// multi-launch schedule (batch statistics couple the samples; activations are kept for backward).
#include "common.h"

struct atmoe_eval_args {
  const float* feats;                                       // (B, 7, C)
  const float *w1, *b1, *bn1_w, *bn1_b, *bn1_m, *bn1_v;     // gate.linear_re.{0,2}
  const float *wq, *wk;                                     // gate.q_ / gate.k_ (C, C), no bias
  const float *ew, *eb, *ebn_w, *ebn_b;                     // expert (h = 0, j = 0): mlp.0.{weight (d,d), bias}, mlp.2.{weight, bias}; expert (h, j) at + (7h + j) * estride
  const float *ebn_m, *ebn_v;                               // running statistics of expert (0, 0); expert (h, j) at + (7h + j) * sstride
  int64_t estride, sstride;
  float* acc;                                               // this call's accumulators, ZERO on entry: x1 (B, C) then k (B, 7, C)
  float* acc_next;                                          // the buffer the NEXT call will use (zeroed here), or NULL
  float* y;                                                 // workspace (B, 7C)
  float* moe;                                               // (B, 7C) output, layout j*C + h*d + n
  float* gates;                                             // optional (B, H, 7)
  int B, C, H;
  float eps;
};

namespace {

constexpr int TN = 64, TB = 32, TK = 128, LDK = TK + 4, KCHUNK = 256;

__global__ __launch_bounds__(256) void atmoe_eval_stage1_kernel(const atmoe_eval_args p) {
  __shared__ __attribute__((aligned(16))) float Ws[TN][LDK];
  __shared__ __attribute__((aligned(16))) float Xs[TB][LDK];
  const int C = p.C, d = C / p.H;
  const int ctC = (C + TN - 1) / TN, ctD = (d + TN - 1) / TN;
  const int ksG = (7 * C + KCHUNK - 1) / KCHUNK, ksK = (C + KCHUNK - 1) / KCHUNK;
  const int nG = ctC * ksG, nK = 7 * ctC * ksK;
  int t = blockIdx.x;
  const int b0 = blockIdx.y * TB;
  const float *X, *W, *bias = nullptr, *bw = nullptr, *bb = nullptr, *bm = nullptr, *bv = nullptr;
  float* out;
  int K, kbeg, kend, ncol;                                   // row length of W, this tile's reduction range, valid columns
  bool atomic = true;
  const int64_t x_rs = 7 * (int64_t)C;
  int64_t o_rs;
  if (t < nG) {                                              // gate: x1 += W1[:, chunk] [f_1..f_7][chunk]
    const int ct = t / ksG, ks = t - ct * ksG, n0 = ct * TN;
    X = p.feats; W = p.w1 + (int64_t)n0 * 7 * C; K = 7 * C; kbeg = ks * KCHUNK; kend = min(K, kbeg + KCHUNK); ncol = min(TN, C - n0);
    out = p.acc + n0; o_rs = C;
  } else if (t < nG + nK) {                                  // gate keys: k_j += Wk[:, chunk] f_j[chunk]
    t -= nG;
    const int j = t / (ctC * ksK), r = t - j * ctC * ksK, ct = r / ksK, ks = r - ct * ksK, n0 = ct * TN;
    X = p.feats + (int64_t)j * C; W = p.wk + (int64_t)n0 * C; K = C; kbeg = ks * KCHUNK; kend = min(K, kbeg + KCHUNK); ncol = min(TN, C - n0);
    out = p.acc + (int64_t)p.B * C + (int64_t)j * C + n0; o_rs = x_rs;
  } else {                                                   // expert (h, j): y = BN(QuickGELU(W f_j[h*d:(h+1)*d] + b)), whole reduction
    t -= nG + nK;
    const int e = t / ctD, n0 = (t - e * ctD) * TN, h = e / 7, j = e - 7 * h;
    X = p.feats + (int64_t)j * C + h * d; W = p.ew + e * p.estride + (int64_t)n0 * d; K = d; kbeg = 0; kend = d; ncol = min(TN, d - n0);
    bias = p.eb + e * p.estride + n0; bw = p.ebn_w + e * p.estride + n0; bb = p.ebn_b + e * p.estride + n0;
    bm = p.ebn_m + e * p.sstride + n0; bv = p.ebn_v + e * p.sstride + n0;
    out = p.y + (int64_t)j * C + h * d + n0; o_rs = x_rs; atomic = false;
  }
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;     // thread: samples 2ty, 2ty+1 x columns tx, tx+16, tx+32, tx+48
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
    for (int i = 0; i < 4; ++i) {                            // input tile: 32 samples x 128 k
      const int f = tid + i * 256, s = f >> 5, kq = f & 31;
      xv[i] = z4;
      if (b0 + s < p.B && k0 + kq * 4 < kend) xv[i] = *reinterpret_cast<const float4*>(X + (int64_t)(b0 + s) * x_rs + k0 + kq * 4);
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
    if (!atomic) {                                           // eval-mode BatchNorm1d: (v - mean) / sqrt(var + eps) * w + b
      bs = bias[n];
      scale = bw[n] * rsqrtf(bv[n] + p.eps);
      shift = bb[n] - bm[n] * scale;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int b = b0 + ty * 2 + r;
      if (b >= p.B) continue;
      if (atomic) atomicAdd(&out[(int64_t)b * o_rs + n], acc[r][c]);
      else out[(int64_t)b * o_rs + n] = fmaf(act_apply(acc[r][c] + bs, DEMO_ACT_QUICKGELU), scale, shift);
    }
  }
}

// grid (H, ceil(B / 16)); thread (s = tid / 16, tx = tid % 16) owns q_h[s][tx + 16 i], i < d / 16
constexpr int SB = 16, DMAX = 128;
__global__ __launch_bounds__(256) void atmoe_eval_stage2_kernel(const atmoe_eval_args p) {
  extern __shared__ __attribute__((aligned(16))) float sm2[];
  float (*Ws)[LDK] = reinterpret_cast<float (*)[LDK]>(sm2);                   // [d][LDK]
  const int C = p.C, H = p.H, d = C / H, h = blockIdx.x, b0 = blockIdx.y * SB;
  float (*Xs)[LDK] = reinterpret_cast<float (*)[LDK]>(sm2 + (size_t)d * LDK);  // [SB][LDK]
  const int tid = threadIdx.x, tx = tid & 15, s = tid >> 4, nI = d >> 4;
  const int b = b0 + s;
  const float* Wq = p.wq + (int64_t)h * d * C;
  const float* x1 = p.acc;
  const float* kg = p.acc + (int64_t)p.B * C;
  float acc[DMAX / 16];
#pragma unroll
  for (int i = 0; i < DMAX / 16; ++i) acc[i] = 0.f;
  for (int k0 = 0; k0 < C; k0 += TK) {
    __syncthreads();
    for (int f = tid; f < d * (TK / 4); f += 256) {           // weight tile: d rows x 128 k
      const int n = f >> 5, kq = f & 31;
      float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k0 + kq * 4 < C) w4 = *reinterpret_cast<const float4*>(Wq + (int64_t)n * C + k0 + kq * 4);
      *reinterpret_cast<float4*>(&Ws[n][kq * 4]) = w4;
    }
    for (int f = tid; f < SB * (TK / 4); f += 256) {          // x2 = BN(QuickGELU(x1 + b1)), applied on the way into LDS
      const int ss = f >> 5, kq = f & 31, c = k0 + kq * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b0 + ss < p.B && c < C) {
        const float4 a = *reinterpret_cast<const float4*>(x1 + (int64_t)(b0 + ss) * C + c);
        const float4 bi = *reinterpret_cast<const float4*>(p.b1 + c), gw = *reinterpret_cast<const float4*>(p.bn1_w + c);
        const float4 gb = *reinterpret_cast<const float4*>(p.bn1_b + c), gm = *reinterpret_cast<const float4*>(p.bn1_m + c);
        const float4 gv = *reinterpret_cast<const float4*>(p.bn1_v + c);
        const float in[4] = {a.x + bi.x, a.y + bi.y, a.z + bi.z, a.w + bi.w};
        const float ww[4] = {gw.x, gw.y, gw.z, gw.w}, wb[4] = {gb.x, gb.y, gb.z, gb.w}, wm[4] = {gm.x, gm.y, gm.z, gm.w}, wv[4] = {gv.x, gv.y, gv.z, gv.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float sc = ww[e] * rsqrtf(wv[e] + p.eps);
          o[e] = fmaf(act_apply(in[e], DEMO_ACT_QUICKGELU), sc, wb[e] - wm[e] * sc);
        }
        v = make_float4(o[0], o[1], o[2], o[3]);
      }
      *reinterpret_cast<float4*>(&Xs[ss][kq * 4]) = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int kq = 0; kq < TK / 4; ++kq) {
      const float4 a = *reinterpret_cast<const float4*>(&Xs[s][kq * 4]);
#pragma unroll
      for (int i = 0; i < DMAX / 16; ++i)
        if (i < nI) {
          const float4 w = *reinterpret_cast<const float4*>(&Ws[tx + 16 * i][kq * 4]);
          acc[i] += (a.x * w.x + a.y * w.y) + (a.z * w.z + a.w * w.w);
        }
    }
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
      if (i < nI) v = fmaf(acc[i], kg[row + (int64_t)j * C + h * d + tx + 16 * i], v);
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
  if (b < p.B) {
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
  // zero the next call's accumulators: this workgroup's samples x this head's column range of x1 and of the 7 key rows
  if (p.acc_next && b < p.B) {
    float* nx1 = p.acc_next + (int64_t)b * C + h * d;
    float* nk = p.acc_next + (int64_t)p.B * C + (int64_t)b * 7 * C + h * d;
#pragma unroll
    for (int i = 0; i < DMAX / 16; ++i)
      if (i < nI) {
        nx1[tx + 16 * i] = 0.f;
#pragma unroll
        for (int j = 0; j < 7; ++j) nk[(int64_t)j * C + tx + 16 * i] = 0.f;
      }
  }
}

}  // namespace

extern "C" int demo_atmoe_eval_supported(int32_t C, int32_t H) {
  if (H < 1 || C % H) return 0;
  const int d = C / H;
  return (C % 4 == 0 && d % 16 == 0 && d <= DMAX) ? 1 : 0;
}

extern "C" int demo_atmoe_eval_fwd(const float* feats, const float* w1, const float* b1, const float* bn1_w, const float* bn1_b,
                                   const float* bn1_mean, const float* bn1_var, const float* wq, const float* wk, const float* ew,
                                   const float* eb, const float* ebn_w, const float* ebn_b, const float* ebn_mean, const float* ebn_var,
                                   int64_t expert_stride, int64_t stat_stride, float eps, float* acc_zeroed, float* acc_next, float* ws_y,
                                   float* moe, float* gates, int32_t B, int32_t C, int32_t H, void* stream) {
  DEMO_REQUIRE(B >= 1 && demo_atmoe_eval_supported(C, H), "demo_atmoe_eval_fwd: unsupported B=%d C=%d H=%d", B, C, H);
  DEMO_REQUIRE(feats && w1 && b1 && bn1_w && bn1_b && bn1_mean && bn1_var && wq && wk && ew && eb && ebn_w && ebn_b && ebn_mean && ebn_var &&
                   acc_zeroed && ws_y && moe && acc_zeroed != acc_next, "demo_atmoe_eval_fwd: null argument");
  atmoe_eval_args p;
  p.feats = feats; p.w1 = w1; p.b1 = b1; p.bn1_w = bn1_w; p.bn1_b = bn1_b; p.bn1_m = bn1_mean; p.bn1_v = bn1_var;
  p.wq = wq; p.wk = wk; p.ew = ew; p.eb = eb; p.ebn_w = ebn_w; p.ebn_b = ebn_b; p.ebn_m = ebn_mean; p.ebn_v = ebn_var;
  p.estride = expert_stride; p.sstride = stat_stride; p.acc = acc_zeroed; p.acc_next = acc_next; p.y = ws_y; p.moe = moe; p.gates = gates;
  p.B = B; p.C = C; p.H = H; p.eps = eps;
  const int d = C / H, ctC = (C + TN - 1) / TN, ctD = (d + TN - 1) / TN;
  const int ksG = (7 * C + KCHUNK - 1) / KCHUNK, ksK = (C + KCHUNK - 1) / KCHUNK;
  const dim3 g1((unsigned)(ctC * ksG + 7 * ctC * ksK + 7 * H * ctD), (unsigned)((B + TB - 1) / TB));
  hipLaunchKernelGGL(atmoe_eval_stage1_kernel, g1, dim3(256), 0, (hipStream_t)stream, p);
  DEMO_LAUNCH_CHECK("demo_atmoe_eval_fwd(stage 1)");
  const dim3 g2((unsigned)H, (unsigned)((B + SB - 1) / SB));
  const size_t shm2 = sizeof(float) * (size_t)(d + SB) * LDK;
  static bool attr_set = false;                              // (idempotent; a benign race sets it twice)
  if (!attr_set && shm2 > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)atmoe_eval_stage2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (DMAX + SB) * LDK * 4);
    if (e != hipSuccess) DEMO_FAIL(-2, "demo_atmoe_eval_fwd: cannot reserve LDS: %s", hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL(atmoe_eval_stage2_kernel, g2, dim3(256), shm2, (hipStream_t)stream, p);
  DEMO_LAUNCH_CHECK("demo_atmoe_eval_fwd(stage 2)");
  return 0;
}
