// Training losses of the reference's 'softmax_triplet' branch (layers/make_loss.py:37-55), value AND gradient in one
// launch each, so the step hands d(loss)/d(score), d(loss)/d(feat) straight to the backward schedule:
//   * label-smoothed cross entropy  (layers/softmax_loss.py:23-34, epsilon = 0.1)
//   * batch-hard soft-margin triplet (layers/triplet_loss.py:16-31 euclidean_dist, :51-104 hard_example_mining,
//     :121-135 TripletLoss without margin = SoftMarginLoss(d_an - d_ap, 1))
// Sizes are tiny (B = 128 rows; K = 171 classes, D = 1536 / 3584 features): these kernels are latency-, not bandwidth-bound;
// what they replace is ~25 small library launches per (score, feat) pair.
#include "common.h"

namespace {

__device__ __forceinline__ float block_sum_256(float v, float* red) {      // 256 threads, red[4]
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_max_256(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// one workgroup per sample: loss += w/B * sum_k -t_k log p_k,  t = (1-eps) onehot + eps/K ;  dlogits = w/B * (p - t)
__global__ __launch_bounds__(256) void xent_label_smooth_kernel(const float* __restrict__ logits, int64_t ld, const int64_t* __restrict__ target,
                                                                int B, int K, float eps, float w, float* __restrict__ loss,
                                                                float* __restrict__ dlogits, int64_t ldd, int* __restrict__ err) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const float* x = logits + (int64_t)b * ld;
  float m = -INFINITY;
  for (int k = threadIdx.x; k < K; k += 256) m = fmaxf(m, x[k]);
  m = block_max_256(m, red);
  float s = 0.f, sx = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) {
    s += __expf(x[k] - m);
    sx += x[k];
  }
  s = block_sum_256(s, red);
  sx = block_sum_256(sx, red);
  const float logz = m + __logf(s);
  const int64_t t64 = target[b];
  const bool bad = t64 < 0 || t64 >= K;                   // torch's scatter_ raises (softmax_loss.py:30-31); here: NaN + error bit
  const int t = bad ? 0 : (int)t64;
  const float wb = bad ? __builtin_nanf("") : w / (float)B;
  if (threadIdx.x == 0) {
    if (bad && err) atomicOr(err, DEMO_ERR_LABEL_RANGE);
    const float lp_t = x[t] - logz;                       // log p of the label
    const float sum_lp = sx - (float)K * logz;            // sum_k log p_k
    atomicAdd(loss, wb * (-(1.f - eps) * lp_t - (eps / (float)K) * sum_lp));
  }
  if (dlogits) {
    float* dx = dlogits + (int64_t)b * ldd;
    const float inv_s = 1.f / s, tk = eps / (float)K;
    for (int k = threadIdx.x; k < K; k += 256) dx[k] = wb * (__expf(x[k] - m) * inv_s - tk - (k == t ? 1.f - eps : 0.f));
  }
}

// Batch-hard triplet, in three steps that are all tiny:
//   (1) Gram matrix G = F F^T (B x B, fp32) by the generic GEMM -- the reference forms its distances the same way
//       (dist^2 = |f_i|^2 + |f_j|^2 - 2 f_i.f_j, clamp 1e-12, sqrt; layers/triplet_loss.py:16-31);
//   (2) this kernel, one wave per anchor i: hardest positive = arg max over same-label j (self included, first index on
//       ties), hardest negative = arg min over the rest; loss += w/B * softplus(d_ap - d_an); and the gradient as a sparse
//       B x B weight matrix Wt with dF = Wt F:  with g = w/B * sigmoid(d_ap - d_an), cp = g/d_ap, cn = g/d_an
//           row i: Wt[i,i] += cp - cn, Wt[i,p] -= cp, Wt[i,n] += cn;  row p: Wt[p,i] -= cp, Wt[p,p] += cp;
//           row n: Wt[n,i] += cn, Wt[n,n] -= cn        (fp32 atomics: rows p / n belong to other anchors too)
//   (3) dF = Wt F by the generic GEMM.
// (A first version computed the B^2 dot products inside this kernel, one workgroup per anchor: 458 us at D = 3584.)
__global__ __launch_bounds__(64) void triplet_mine_kernel(const float* __restrict__ G, const int64_t* __restrict__ labels, int B, float w,
                                                          float margin, float* __restrict__ loss, float* __restrict__ Wt) {
  const int i = blockIdx.x, lane = threadIdx.x;
  const int64_t li = labels[i];
  const float gii = G[(int64_t)i * B + i];
  float bp = -INFINITY, bn = INFINITY;
  int ip = -1, in_ = -1;
  for (int j = lane; j < B; j += 64) {
    const float d2 = fmaxf(gii + G[(int64_t)j * B + j] - 2.f * G[(int64_t)i * B + j], 1e-12f);
    const float v = sqrtf(d2);
    if (labels[j] == li) {
      if (v > bp) { bp = v; ip = j; }
    } else if (v < bn) {
      bn = v; in_ = j;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {                         // arg max / arg min across lanes, first index wins on ties
    const float obp = __shfl_xor(bp, o, 64), obn = __shfl_xor(bn, o, 64);
    const int oip = __shfl_xor(ip, o, 64), oin = __shfl_xor(in_, o, 64);
    if (oip >= 0 && (ip < 0 || obp > bp || (obp == bp && oip < ip))) { bp = obp; ip = oip; }
    if (oin >= 0 && (in_ < 0 || obn < bn || (obn == bn && oin < in_))) { bn = obn; in_ = oin; }
  }
  if (lane != 0 || ip < 0 || in_ < 0) return;                // no positive / no negative in the batch: contributes nothing
  const float z = bp - bn;
  const float wb = w / (float)B;
  // margin < 0: SoftMarginLoss(d_an - d_ap, 1) = softplus(z); margin >= 0: MarginRankingLoss(d_an, d_ap, 1) = max(0, z + margin)
  if (margin < 0.f) atomicAdd(loss, wb * (z > 20.f ? z : log1pf(__expf(z))));
  else if (z + margin > 0.f) atomicAdd(loss, wb * (z + margin));
  if (!Wt) return;
  const float g = margin < 0.f ? wb / (1.f + __expf(-z)) : (z + margin > 0.f ? wb : 0.f);
  // the clamp passes no gradient where it is active (dist^2 == 1e-12: coincident points, e.g. self as hardest positive)
  const int p = ip, n = in_;
  const float d2p = gii + G[(int64_t)p * B + p] - 2.f * G[(int64_t)i * B + p];
  const float d2n = gii + G[(int64_t)n * B + n] - 2.f * G[(int64_t)i * B + n];
  const float cp = d2p > 1e-12f ? g / bp : 0.f;
  const float cn = d2n > 1e-12f ? g / bn : 0.f;
  atomicAdd(Wt + (int64_t)i * B + i, cp - cn);
  atomicAdd(Wt + (int64_t)i * B + p, -cp);
  atomicAdd(Wt + (int64_t)i * B + n, cn);
  atomicAdd(Wt + (int64_t)p * B + i, -cp);
  atomicAdd(Wt + (int64_t)p * B + p, cp);
  atomicAdd(Wt + (int64_t)n * B + i, cn);
  atomicAdd(Wt + (int64_t)n * B + n, -cn);
}

}  // namespace

extern "C" int demo_xent_label_smooth(const float* logits, int64_t ld, const int64_t* target, int32_t B, int32_t K, float eps, float weight,
                                      float* loss, float* dlogits, int64_t ldd, int32_t* err, void* stream) {
  DEMO_REQUIRE(logits && target && loss && B > 0 && K > 0 && ld >= K, "demo_xent_label_smooth: bad arguments");
  hipLaunchKernelGGL(xent_label_smooth_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, ld, target, B, K, eps, weight, loss,
                     dlogits, ldd, err);
  DEMO_LAUNCH_CHECK("demo_xent_label_smooth");
  return 0;
}

extern "C" int demo_gemm(const demo_gemm_desc* d, void* stream);

extern "C" int demo_triplet_loss(const float* feat, int64_t ld, const int64_t* labels, int32_t B, int32_t D, float weight, float margin,
                                 float* loss, float* dfeat, int64_t ldd, float* workspace, void* stream);
extern "C" int demo_triplet_soft_margin(const float* feat, int64_t ld, const int64_t* labels, int32_t B, int32_t D, float weight, float* loss,
                                        float* dfeat, int64_t ldd, float* workspace, void* stream) {
  return demo_triplet_loss(feat, ld, labels, B, D, weight, -1.f, loss, dfeat, ldd, workspace, stream);
}

extern "C" int demo_triplet_loss(const float* feat, int64_t ld, const int64_t* labels, int32_t B, int32_t D, float weight, float margin,
                                 float* loss, float* dfeat, int64_t ldd, float* workspace, void* stream) {
  DEMO_REQUIRE(feat && labels && loss && workspace && B > 0 && D > 0 && ld >= D, "demo_triplet_loss: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  float* G = workspace;                         // (B, B) Gram matrix
  float* Wt = workspace + (size_t)B * B;        // (B, B) gradient weights
  demo_gemm_desc g = {};
  g.A = feat; g.B = feat; g.C = G;
  g.M = B; g.N = B; g.K = D;
  g.a_rs = ld; g.a_cs = 1; g.b_rs = ld; g.b_cs = 1; g.c_rs = B;
  g.dtA = g.dtB = g.dtC = DEMO_F32;
  g.act = DEMO_ACT_NONE; g.alpha = 1.f; g.batch1 = g.batch2 = 1;
  int rc = demo_gemm(&g, stream);
  if (rc) return rc;
  if (dfeat) {
    hipError_t e = hipMemsetAsync(Wt, 0, sizeof(float) * (size_t)B * B, s);
    if (e != hipSuccess) DEMO_FAIL(-2, "demo_triplet_loss: memset failed: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(triplet_mine_kernel, dim3(B), dim3(64), 0, s, G, labels, B, weight, margin, loss, dfeat ? Wt : nullptr);
  DEMO_LAUNCH_CHECK("demo_triplet_loss");
  if (dfeat) {                                  // dF[i, d] = sum_j Wt[i, j] F[j, d]
    demo_gemm_desc h = {};
    h.A = Wt; h.B = feat; h.C = dfeat;
    h.M = B; h.N = D; h.K = B;
    h.a_rs = B; h.a_cs = 1; h.b_rs = 1; h.b_cs = ld; h.c_rs = ldd;
    h.dtA = h.dtB = h.dtC = DEMO_F32;
    h.act = DEMO_ACT_NONE; h.alpha = 1.f; h.batch1 = h.batch2 = 1;
    rc = demo_gemm(&h, stream);
    if (rc) return rc;
  }
  return 0;
}
