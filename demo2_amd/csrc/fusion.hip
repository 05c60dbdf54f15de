// HDM (folded single-query cross-attention) softmax + pooling, and the ATMoE gate softmax.
//
// HDM algebra (AttnMOE.py:153-191).  Each of the 7 branches is nn.MultiheadAttention with a LEARNED
// query token shared by every sample, so with q_h = (Wq tok + bq)_h / sqrt(dh):
//     score_h(x_l) = q_h . (Wk_h x_l + bk_h) = x_l . (Wk_h^T q_h) + const      (const cancels in softmax)
//     out_h        = sum_l p_l (Wv_h x_l + bv_h) = Wv_h (sum_l p_l x_l) + bv_h sum_l p_l
// The (Lk, C) x (C, 2C) key/value projections (1.62 of the 1.63 GFLOP/sample) disappear: scores are one
// skinny GEMM of the token stream against the folded keys (C x 4*heads per modality), and this kernel
// does softmax over the concatenated streams of a branch plus the probability-weighted token pooling.
#include "common.h"

namespace {

constexpr int HH_MAX = 12;

// stream lists per branch (r,n,t,rn,rt,nt,rnt) and the slot of each branch in a modality's score columns
__constant__ int c_nstreams[7] = {1, 1, 1, 2, 2, 2, 3};
__constant__ int c_streams[7][3] = {{0, 0, 0}, {1, 0, 0}, {2, 0, 0}, {0, 1, 0}, {0, 2, 0}, {1, 2, 0}, {0, 1, 2}};

__device__ __forceinline__ uint32_t rng_hash(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ULL * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return (uint32_t)((z ^ (z >> 31)) >> 32);
}

template <typename T, int U>
__global__ __launch_bounds__(512) void hdm_softmax_pool_kernel(const float* __restrict__ S, int64_t ldS, const T* __restrict__ tok,
                                                               float* __restrict__ pooled, float* __restrict__ psum,
                                                               float* __restrict__ probs, float* __restrict__ probs_post, int Bs, int L, int C, int Hh,
                                                               float p_drop, uint64_t seed, const uint64_t* __restrict__ seed_dev) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  if (seed_dev) seed += *seed_dev;            // device-resident part of the seed: advances between replays of a captured step
  const int b = blockIdx.x, br = blockIdx.y;
  const int ns = c_nstreams[br];
  const int Lk = ns * L, L3 = 3 * L;
  float* P = sm;                              // [Hh][L3]
  float* red = sm + Hh * L3;                  // [blockDim][4] scratch for the split reduction (reused per head)

  // ---- phase 1: gather scores --------------------------------------------------------------
  for (int i = threadIdx.x; i < Hh * Lk; i += blockDim.x) {
    const int h = i / Lk, lk = i - h * Lk;
    const int si = lk / L, l = lk - si * L;
    const int m = c_streams[br][si];
    P[h * L3 + lk] = S[((int64_t)(m * Bs + b) * L + l) * ldS + br * Hh + h];
  }
  __syncthreads();
  // ---- phase 2: softmax (+dropout) per head, one wave per head -----------------------------------
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const float keep_scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  const uint32_t thresh = p_drop > 0.f ? (uint32_t)(p_drop * 4294967296.0) : 0u;
  for (int h = wave; h < Hh; h += nw) {
    float* ph = P + h * L3;
    float m = -INFINITY;
    for (int i = lane; i < Lk; i += 64) m = fmaxf(m, ph[i]);
    m = wave_max(m);
    float s = 0.f;
    for (int i = lane; i < Lk; i += 64) {
      const float e = __expf(ph[i] - m);
      ph[i] = e;
      s += e;
    }
    s = wave_sum(s);
    const float inv = 1.f / s;
    float kept = 0.f;
    for (int i = lane; i < Lk; i += 64) {
      float p = ph[i] * inv;
      if (p_drop > 0.f) {
        const uint32_t u = rng_hash(seed, ((uint64_t)(b * 7 + br) * Hh + h) * L3 + i);
        p = (u >= thresh) ? p * keep_scale : 0.f;
      }
      if (probs && blockIdx.z == 0) probs[((int64_t)(b * 7 + br) * Hh + h) * L3 + i] = ph[i] * inv;     // softmax BEFORE dropout (backward needs it)
      ph[i] = p;
      kept += p;
      if (probs_post && blockIdx.z == 0) probs_post[((int64_t)(b * 7 + br) * Hh + h) * L3 + i] = p;
    }
    for (int i = Lk + lane; i < L3; i += 64) {
      if (probs && blockIdx.z == 0) probs[((int64_t)(b * 7 + br) * Hh + h) * L3 + i] = 0.f;
      if (probs_post && blockIdx.z == 0) probs_post[((int64_t)(b * 7 + br) * Hh + h) * L3 + i] = 0.f;
    }
    kept = wave_sum(kept);
    if (lane == 0 && psum && blockIdx.z == 0) psum[(int64_t)(b * 7 + br) * Hh + h] = kept;
  }
  __syncthreads();
  // ---- phase 3: pooled[h][c] = sum_lk P[h][lk] * X[lk][c] -------------------------------------------
  // gridDim.z channel chunks (small grids: the (sample, branch) workgroups alone leave most CUs idle -- 7 workgroups at B = 1 -- so the
  // pooling of a branch is cut across z; phases 1-2 are repeated per chunk, only chunk 0 writes the probabilities)
  const int ngroups = (C >> 2) / gridDim.z;              // channel groups of 4 in this chunk
  const int nsplit = blockDim.x / ngroups;               // >= 1 (C <= 1024)
  const int gl = threadIdx.x % ngroups, split = threadIdx.x / ngroups;
  const int grp = blockIdx.z * ngroups + gl;
  float4 acc[HH_MAX];
#pragma unroll
  for (int h = 0; h < HH_MAX; ++h) acc[h] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (split < nsplit) {
    // U tokens per step, all U loads requested before the first is used: a thread walks ~Lk / nsplit tokens (97 at Lk = 387), and one
    // dependent HBM/L2 round trip per token made this kernel 99 us at B = 1 (U = 8 on small grids: 130 VGPRs, one workgroup per CU; on chip-filling
    // grids that occupancy costs more than it hides -- 147 -> 214 us at B = 128 -- so U = 4 there: 96 VGPRs, two workgroups per CU, 122 us)
    for (int lk0 = split; lk0 < Lk; lk0 += U * nsplit) {
      typename io<T>::raw4 xr[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int lk = lk0 + u * nsplit;
        const int lkc = lk < Lk ? lk : split;                // (clamped: the contribution is masked below)
        const int si = lkc / L, l = lkc - si * L;
        const int m = c_streams[br][si];
        xr[u] = io<T>::ld4_raw(tok + ((int64_t)(m * Bs + b) * L + l) * C + grp * 4);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int lk = lk0 + u * nsplit;
        if (lk >= Lk) break;
        const float4 x = io<T>::cvt4(xr[u]);
#pragma unroll
        for (int h = 0; h < HH_MAX; ++h) {
          if (h < Hh) {
            const float p = P[h * L3 + lk];
            acc[h].x = fmaf(p, x.x, acc[h].x); acc[h].y = fmaf(p, x.y, acc[h].y);
            acc[h].z = fmaf(p, x.z, acc[h].z); acc[h].w = fmaf(p, x.w, acc[h].w);
          }
        }
      }
    }
  }
  float* dst = pooled + (int64_t)(b * 7 + br) * Hh * C;
#pragma unroll
  for (int h = 0; h < HH_MAX; ++h) {
    if (h < Hh) {                                          // uniform
      float4 v = acc[h];
      if (nsplit > 1) {
        __syncthreads();
        if (split < nsplit) *reinterpret_cast<float4*>(red + threadIdx.x * 4) = v;
        __syncthreads();
        if (split == 0) {
          for (int sp = 1; sp < nsplit; ++sp) {
            const float4 o = *reinterpret_cast<const float4*>(red + (sp * ngroups + gl) * 4);
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
          }
        }
      }
      if (split == 0) *reinterpret_cast<float4*>(dst + (int64_t)h * C + grp * 4) = v;
    }
  }
}

__global__ __launch_bounds__(256) void hdm_vbias_kernel(float* __restrict__ y, const float* __restrict__ bias, int64_t bias_stride,
                                                        const float* __restrict__ psum, int Bs, int C, int Hh) {
  const int64_t n = (int64_t)Bs * 7 * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t bb = i / C;            // b*7 + br
    const int br = (int)(bb % 7);
    y[i] += bias[br * bias_stride + c] * psum[bb * Hh + c / (C / Hh)];
  }
}

// gates[b,h,j] = softmax_j(q[b,h,:] . k[b,j,h,:] * d^-0.5); block per sample, wave per head.
__global__ __launch_bounds__(256) void atmoe_gate_softmax_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                 float* __restrict__ gates, int C, int H) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = C / H;
  const float scale = rsqrtf((float)d);
  for (int h = wave; h < H; h += 4) {
    float lg[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      float s = 0.f;
      for (int e = lane; e < d; e += 64)
        s = fmaf(q[(int64_t)b * C + h * d + e], k[((int64_t)b * 7 + j) * C + h * d + e], s);
      lg[j] = wave_sum(s) * scale;
    }
    float m = lg[0];
#pragma unroll
    for (int j = 1; j < 7; ++j) m = fmaxf(m, lg[j]);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 7; ++j) { lg[j] = __expf(lg[j] - m); sum += lg[j]; }
    if (lane < 7) {
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < 7; ++j) v = (lane == j) ? lg[j] : v;
      gates[((int64_t)b * H + h) * 7 + lane] = v / sum;
    }
  }
}

}  // namespace

extern "C" int demo_hdm_softmax_pool(const float* S, int64_t ldS, const void* tokens, int32_t dt, float* pooled, float* psum, float* probs, float* probs_post,
                                     int32_t Bs, int32_t L, int32_t C, int32_t Hh, float p_drop, uint64_t seed, const uint64_t* seed_dev,
                                     void* stream) {
  DEMO_REQUIRE(Hh >= 1 && Hh <= HH_MAX, "demo_hdm_softmax_pool: heads=%d unsupported (<=%d)", Hh, HH_MAX);
  DEMO_REQUIRE(C % 4 == 0 && C / 4 <= 256 && C >= 4, "demo_hdm_softmax_pool: C=%d unsupported", C);
  DEMO_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "demo_hdm_softmax_pool: p_drop=%f", p_drop);
  DEMO_REQUIRE(ldS >= 7 * Hh, "demo_hdm_softmax_pool: ldS too small");
  const size_t shm = sizeof(float) * ((size_t)Hh * 3 * L + 512 * 4);
  DEMO_REQUIRE(shm <= 64 * 1024, "demo_hdm_softmax_pool: L=%d too long", L);
  hipStream_t s = (hipStream_t)stream;
  const bool small = (int64_t)Bs * 7 < 512;
  int nz = 1;                                          // channel chunks: fill ~256 CUs on small grids (a chunk keeps >= 16 channel groups)
  while ((int64_t)Bs * 7 * nz < 256 && nz < 8 && (C / 4) % (nz * 2) == 0 && (C / 4) / (nz * 2) >= 16 && 512 % ((C / 4) / (nz * 2)) == 0) nz *= 2;
  dim3 grid(Bs, 7, nz);
#define HDM_POOL(T_, U_) hipLaunchKernelGGL((hdm_softmax_pool_kernel<T_, U_>), grid, dim3(512), shm, s, S, ldS, (const T_*)tokens, pooled, psum, probs, probs_post, Bs, L, C, Hh, p_drop, seed, seed_dev)
  if (dt == DEMO_BF16) { if (small) HDM_POOL(bf16_t, 8); else HDM_POOL(bf16_t, 4); }
  else { if (small) HDM_POOL(float, 8); else HDM_POOL(float, 1); }
#undef HDM_POOL
  DEMO_LAUNCH_CHECK("demo_hdm_softmax_pool");
  return 0;
}

extern "C" int demo_hdm_vbias(float* y, const float* bias, int64_t bias_stride, const float* psum, int32_t Bs, int32_t C, int32_t Hh, void* stream) {
  DEMO_REQUIRE(C % Hh == 0, "demo_hdm_vbias: C %% Hh");
  const int64_t n = (int64_t)Bs * 7 * C;
  int64_t g = cdiv64(n, 256);
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(hdm_vbias_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, y, bias, bias_stride, psum, Bs, C, Hh);
  DEMO_LAUNCH_CHECK("demo_hdm_vbias");
  return 0;
}

extern "C" int demo_atmoe_gate_softmax(const float* q, const float* k, float* gates, int32_t Bs, int32_t C, int32_t H, void* stream) {
  DEMO_REQUIRE(H >= 1 && C % H == 0, "demo_atmoe_gate_softmax: bad C/H");
  hipLaunchKernelGGL(atmoe_gate_softmax_kernel, dim3(Bs), dim3(256), 0, (hipStream_t)stream, q, k, gates, C, H);
  DEMO_LAUNCH_CHECK("demo_atmoe_gate_softmax");
  return 0;
}
