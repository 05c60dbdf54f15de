// Retrieval evaluator of the reference (utils/metrics.py): feature normalisation (R1_mAP_eval.compute :341-346), the squared
// euclidean distance matrix (euclidean_distance :395-401) and, per query, the market1501 protocol of eval_func (:110-169) /
// the MSVR310 scene protocol of eval_func_msrv (:12-107): sort the gallery by distance, drop gallery items with the query's
// identity AND the query's camera (or scene), then CMC and average precision over what is kept.
//   * demo_row_sqnorm / demo_l2_normalize_rows / demo_add_outer_sum: the row-wise pieces around the -2 q.g^T GEMM (demo_gemm);
//   * demo_rank_eval: one workgroup per query; (distance, gallery index) pairs are bitonic-sorted in LDS (<= 16384 gallery
//     items = 128 KB of the 160-KB LDS; the index is the tie-breaker, so the order is that of a stable argsort), then two
//     block-wide prefix sums (kept items, kept matches) give every match its rank r among the kept and its hit count h:
//         AP = (1/num_rel) sum over matches of h / r  (float64, as numpy computes it),  first_hit = min r - 1.
//     The host averages: mAP = mean AP over valid queries, cmc[k] = mean(first_hit <= k).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float* __restrict__ x, int64_t ld, int N, int D, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const float* xr = x + (int64_t)row * ld;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) s = fmaf(xr[d], xr[d], s);
  s = wave_sum(s);
  if (lane == 0) out[row] = s;
}

// torch.nn.functional.normalize(x, dim=1, p=2): x / max(||x||_2, eps), in place
__global__ __launch_bounds__(256) void l2_normalize_kernel(float* __restrict__ x, int64_t ld, int N, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  float* xr = x + (int64_t)row * ld;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) s = fmaf(xr[d], xr[d], s);
  s = wave_sum(s);
  const float inv = 1.f / fmaxf(sqrtf(s), eps);
  for (int d = lane; d < D; d += 64) xr[d] *= inv;
}

// dist[i, j] = (a[i] + b[j]) + dist[i, j]      (dist holds -2 q.g^T on entry: dist_mat.addmm_(qf, gf.t(), beta=1, alpha=-2))
__global__ __launch_bounds__(256) void add_outer_sum_kernel(float* __restrict__ dist, int64_t ld, const float* __restrict__ a,
                                                            const float* __restrict__ b, int nq, int ng) {
  const int i = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j < ng) dist[(int64_t)i * ld + j] = (a[i] + b[j]) + dist[(int64_t)i * ld + j];
}

__global__ __launch_bounds__(1024) void rank_eval_kernel(const float* __restrict__ dist, int64_t ld, const int64_t* __restrict__ q_pids,
                                                         const int64_t* __restrict__ g_pids, const int64_t* __restrict__ q_keys,
                                                         const int64_t* __restrict__ g_keys, int ng, int NP, double* __restrict__ ap,
                                                         int* __restrict__ first_hit, int* __restrict__ num_rel_out,
                                                         int* __restrict__ order_out) {
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  float* key = reinterpret_cast<float*>(smraw);               // [NP]
  int* idx = reinterpret_cast<int*>(smraw) + NP;              // [NP]
  __shared__ int s_keep[1024], s_match[1024];
  __shared__ double s_ap[16];
  __shared__ int s_first[16];
  const int q = blockIdx.x, t = threadIdx.x;
  const float* drow = dist + (int64_t)q * ld;
  for (int i = t; i < NP; i += 1024) {
    key[i] = i < ng ? drow[i] : INFINITY;
    idx[i] = i < ng ? i : 0x7fffffff;
  }
  __syncthreads();
  // bitonic sort, ascending by (key, idx)
  for (int k = 2; k <= NP; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = t; i < NP; i += 1024) {
        const int l = i ^ j;
        if (l > i) {
          const float ka = key[i], kb = key[l];
          const int ia = idx[i], ib = idx[l];
          const bool a_gt_b = ka > kb || (ka == kb && ia > ib);
          const bool up = (i & k) == 0;
          if (a_gt_b == up) {
            key[i] = kb; key[l] = ka;
            idx[i] = ib; idx[l] = ia;
          }
        }
      }
      __syncthreads();
    }
  }
  if (order_out)
    for (int i = t; i < ng; i += 1024) order_out[(int64_t)q * ng + i] = idx[i];
  // per-thread contiguous chunk of the sorted order
  const int64_t qp = q_pids[q], qk = q_keys[q];
  const int per = (ng + 1023) / 1024;
  const int p0 = t * per, p1 = min(ng, p0 + per);
  int nkeep = 0, nmatch = 0;
  for (int pos = p0; pos < p1; ++pos) {
    const int gi = idx[pos];
    const bool same = g_pids[gi] == qp;
    const bool keep = !(same && g_keys[gi] == qk);
    nkeep += keep;
    nmatch += keep && same;
  }
  s_keep[t] = nkeep;
  s_match[t] = nmatch;
  __syncthreads();
  // exclusive block scan (Hillis-Steele over 1024 entries, both arrays at once)
  int vk = nkeep, vm = nmatch;
  for (int o = 1; o < 1024; o <<= 1) {
    const int ak = t >= o ? s_keep[t - o] : 0, am = t >= o ? s_match[t - o] : 0;
    __syncthreads();
    vk += ak; vm += am;
    s_keep[t] = vk; s_match[t] = vm;
    __syncthreads();
  }
  const int total_match = s_match[1023];
  int r = vk - nkeep, h = vm - nmatch;                        // kept items / kept matches before this chunk
  double acc = 0.0;
  int first = 0x7fffffff;
  for (int pos = p0; pos < p1; ++pos) {
    const int gi = idx[pos];
    const bool same = g_pids[gi] == qp;
    const bool keep = !(same && g_keys[gi] == qk);
    if (keep) {
      ++r;
      if (same) {
        ++h;
        acc += (double)h / (double)r;
        if (first == 0x7fffffff) first = r - 1;
      }
    }
  }
  // block reductions (sum of doubles, min of ints): wave shuffles then 16 partials
  for (int o = 32; o > 0; o >>= 1) {
    acc += __shfl_xor(acc, o, 64);
    first = min(first, __shfl_xor(first, o, 64));
  }
  if ((t & 63) == 0) { s_ap[t >> 6] = acc; s_first[t >> 6] = first; }
  __syncthreads();
  if (t == 0) {
    double a = 0.0;
    int f = 0x7fffffff;
    for (int w = 0; w < 16; ++w) { a += s_ap[w]; f = min(f, s_first[w]); }
    ap[q] = total_match > 0 ? a / (double)total_match : -1.0;          // -1: query identity absent from the kept gallery
    first_hit[q] = total_match > 0 ? f : -1;
    if (num_rel_out) num_rel_out[q] = total_match;
  }
}

}  // namespace

extern "C" int demo_row_sqnorm(const float* x, int64_t ld, int32_t N, int32_t D, float* out, void* stream) {
  DEMO_REQUIRE(x && out && N > 0 && D > 0, "demo_row_sqnorm: bad arguments");
  hipLaunchKernelGGL(row_sqnorm_kernel, dim3((unsigned)cdiv64(N, 4)), dim3(256), 0, (hipStream_t)stream, x, ld, N, D, out);
  DEMO_LAUNCH_CHECK("demo_row_sqnorm");
  return 0;
}

extern "C" int demo_l2_normalize_rows(float* x, int64_t ld, int32_t N, int32_t D, float eps, void* stream) {
  DEMO_REQUIRE(x && N > 0 && D > 0, "demo_l2_normalize_rows: bad arguments");
  hipLaunchKernelGGL(l2_normalize_kernel, dim3((unsigned)cdiv64(N, 4)), dim3(256), 0, (hipStream_t)stream, x, ld, N, D, eps);
  DEMO_LAUNCH_CHECK("demo_l2_normalize_rows");
  return 0;
}

extern "C" int demo_add_outer_sum(float* dist, int64_t ld, const float* a, const float* b, int32_t nq, int32_t ng, void* stream) {
  DEMO_REQUIRE(dist && a && b && nq > 0 && ng > 0 && nq <= 65535, "demo_add_outer_sum: bad arguments");
  hipLaunchKernelGGL(add_outer_sum_kernel, dim3((unsigned)cdiv64(ng, 256), (unsigned)nq), dim3(256), 0, (hipStream_t)stream, dist, ld, a, b,
                     nq, ng);
  DEMO_LAUNCH_CHECK("demo_add_outer_sum");
  return 0;
}

extern "C" int demo_rank_eval(const float* dist, int64_t ld, const int64_t* q_pids, const int64_t* g_pids, const int64_t* q_keys,
                              const int64_t* g_keys, int32_t nq, int32_t ng, double* ap, int32_t* first_hit, int32_t* num_rel,
                              int32_t* order_out, void* stream) {
  DEMO_REQUIRE(dist && q_pids && g_pids && q_keys && g_keys && ap && first_hit && nq > 0 && ng > 0, "demo_rank_eval: bad arguments");
  int NP = 1;
  while (NP < ng) NP <<= 1;
  DEMO_REQUIRE(NP <= 16384, "demo_rank_eval: gallery of %d exceeds the 16384 items one workgroup sorts in LDS", ng);
  const size_t shm = (size_t)NP * 8;
  hipError_t e = hipFuncSetAttribute((const void*)rank_eval_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  if (e != hipSuccess) DEMO_FAIL(-2, "demo_rank_eval: cannot reserve %zu B of LDS: %s", shm, hipGetErrorString(e));
  hipLaunchKernelGGL(rank_eval_kernel, dim3(nq), dim3(1024), shm, (hipStream_t)stream, dist, ld, q_pids, g_pids, q_keys, g_keys, ng, NP, ap,
                     first_hit, num_rel, order_out);
  DEMO_LAUNCH_CHECK("demo_rank_eval");
  return 0;
}
