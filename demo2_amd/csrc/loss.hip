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
                                                                float* __restrict__ dlogits, int64_t ldd) {
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
  const int t = (int)target[b];
  const float wb = w / (float)B;
  if (threadIdx.x == 0) {
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

// one workgroup per anchor i.  dist(i,j) = sqrt(max(|f_i|^2 + |f_j|^2 - 2 f_i.f_j, 1e-12)) exactly as the reference forms it;
// hardest positive = arg max over same-label j (self included, first index on ties), hardest negative = arg min over the rest;
// loss += w/B * softplus(d_ap - d_an);  with g = w/B * sigmoid(d_ap - d_an):
//   df_i += g (f_i - f_p)/d_ap - g (f_i - f_n)/d_an ,  df_p -= g (f_i - f_p)/d_ap ,  df_n += g (f_i - f_n)/d_an   (fp32 atomics)
__global__ __launch_bounds__(256) void triplet_soft_margin_kernel(const float* __restrict__ feat, int64_t ld, const int64_t* __restrict__ labels,
                                                                  int B, int D, float w, float* __restrict__ loss,
                                                                  float* __restrict__ dfeat, int64_t ldd) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // f_i [D] | dist2 [B]
  float* fi = sm;
  float* d2 = sm + D;
  __shared__ float s_best[2];
  __shared__ int s_idx[2];
  const int i = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* xi = feat + (int64_t)i * ld;
  for (int d = threadIdx.x; d < D; d += 256) fi[d] = xi[d];
  __syncthreads();
  float sqi = 0.f;
  for (int d = lane; d < D; d += 64) sqi = fmaf(fi[d], fi[d], sqi);
  sqi = wave_sum(sqi);
  for (int j = wave; j < B; j += 4) {
    const float* xj = feat + (int64_t)j * ld;
    float dot = 0.f, sqj = 0.f;
    for (int d = lane; d < D; d += 64) {
      const float v = xj[d];
      dot = fmaf(fi[d], v, dot);
      sqj = fmaf(v, v, sqj);
    }
    dot = wave_sum(dot);
    sqj = wave_sum(sqj);
    if (lane == 0) d2[j] = fmaxf(sqi + sqj - 2.f * dot, 1e-12f);
  }
  __syncthreads();
  if (wave == 0) {                                          // mining by one wave: (value, index) reductions over B entries
    const int64_t li = labels[i];
    float bp = -INFINITY, bn = INFINITY;
    int ip = -1, in_ = -1;
    for (int j = lane; j < B; j += 64) {
      const float v = sqrtf(d2[j]);
      if (labels[j] == li) {
        if (v > bp) { bp = v; ip = j; }
      } else if (v < bn) {
        bn = v; in_ = j;
      }
    }
    // arg max / arg min across lanes with "first index wins" on equal values
    for (int o = 32; o > 0; o >>= 1) {
      const float obp = __shfl_xor(bp, o, 64), obn = __shfl_xor(bn, o, 64);
      const int oip = __shfl_xor(ip, o, 64), oin = __shfl_xor(in_, o, 64);
      if (oip >= 0 && (ip < 0 || obp > bp || (obp == bp && oip < ip))) { bp = obp; ip = oip; }
      if (oin >= 0 && (in_ < 0 || obn < bn || (obn == bn && oin < in_))) { bn = obn; in_ = oin; }
    }
    if (lane == 0) {
      s_best[0] = bp; s_best[1] = bn;
      s_idx[0] = ip; s_idx[1] = in_;
    }
  }
  __syncthreads();
  const float d_ap = s_best[0], d_an = s_best[1];
  const int p = s_idx[0], n = s_idx[1];
  if (p < 0 || n < 0) return;                               // no positive / no negative in the batch: contributes nothing
  const float z = d_ap - d_an;                               // softplus(z), sigmoid(z)
  const float wb = w / (float)B;
  if (threadIdx.x == 0) atomicAdd(loss, wb * (z > 20.f ? z : log1pf(__expf(z))));
  if (!dfeat) return;
  const float g = wb / (1.f + __expf(-z));
  // the clamp passes no gradient where it is active (dist^2 == 1e-12: coincident points, e.g. self as hardest positive)
  const float cp = d2[p] > 1e-12f ? g / d_ap : 0.f;
  const float cn = d2[n] > 1e-12f ? g / d_an : 0.f;
  const float* xp = feat + (int64_t)p * ld;
  const float* xn = feat + (int64_t)n * ld;
  float* gi = dfeat + (int64_t)i * ldd;
  float* gp = dfeat + (int64_t)p * ldd;
  float* gn = dfeat + (int64_t)n * ldd;
  for (int d = threadIdx.x; d < D; d += 256) {
    const float a = cp * (fi[d] - xp[d]), c = cn * (fi[d] - xn[d]);
    atomicAdd(gi + d, a - c);
    atomicAdd(gp + d, -a);
    atomicAdd(gn + d, c);
  }
}

}  // namespace

extern "C" int demo_xent_label_smooth(const float* logits, int64_t ld, const int64_t* target, int32_t B, int32_t K, float eps, float weight,
                                      float* loss, float* dlogits, int64_t ldd, void* stream) {
  DEMO_REQUIRE(logits && target && loss && B > 0 && K > 0 && ld >= K, "demo_xent_label_smooth: bad arguments");
  hipLaunchKernelGGL(xent_label_smooth_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, ld, target, B, K, eps, weight, loss,
                     dlogits, ldd);
  DEMO_LAUNCH_CHECK("demo_xent_label_smooth");
  return 0;
}

extern "C" int demo_triplet_soft_margin(const float* feat, int64_t ld, const int64_t* labels, int32_t B, int32_t D, float weight, float* loss,
                                        float* dfeat, int64_t ldd, void* stream) {
  DEMO_REQUIRE(feat && labels && loss && B > 0 && D > 0 && ld >= D, "demo_triplet_soft_margin: bad arguments");
  const size_t shm = sizeof(float) * ((size_t)D + B);
  DEMO_REQUIRE(shm <= 64 * 1024, "demo_triplet_soft_margin: D=%d B=%d need %zu B of LDS", D, B, shm);
  hipStream_t s = (hipStream_t)stream;
  if (dfeat) {
    hipError_t e = hipMemset2DAsync(dfeat, (size_t)ldd * sizeof(float), 0, (size_t)D * sizeof(float), (size_t)B, s);
    if (e != hipSuccess) DEMO_FAIL(-2, "demo_triplet_soft_margin: memset failed: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(triplet_soft_margin_kernel, dim3(B), dim3(256), shm, s, feat, ld, labels, B, D, weight, loss, dfeat, ldd);
  DEMO_LAUNCH_CHECK("demo_triplet_soft_margin");
  return 0;
}
