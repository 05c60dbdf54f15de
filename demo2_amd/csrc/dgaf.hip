// DGAF-V3 (dual-gated adaptive fusion on patch tokens; reference modeling/dual_gated_fusion.py:558-745).
//
// Attention pooling (:661-679) is nn.MultiheadAttention with a LEARNED query per modality and shared projections, so it
// folds exactly like HDM (fusion.hip): with q_h = (Wq query + bq)_h / sqrt(dh)
//     score_h(x_n) = x_n . (Wk_h^T q_h) + const ,   out_h = Wv_h (sum_n p_n x_n) + bv_h        (no dropout: sum_n p_n = 1)
// -- the scores are one skinny MFMA GEMM of the token stream against the folded keys, and the kernels here do the softmax
// over a modality's PATCH tokens (the cls row is not part of DGAF's input) plus the probability-weighted pooling.
// The two gates (entropy gate IEG :681-719, importance gate MIG :721-730), the sigmoid(alpha) blend (:732-734) and their
// gradients are one workgroup per sample: everything is a handful of C-long reductions.
#include "common.h"

namespace {

constexpr int NH_MAX = 12;

// sums NV per-thread values over the 256 threads of the workgroup; every thread gets the totals.  red: 4*NV floats of LDS.
template <int NV> __device__ __forceinline__ void block_sum(float (&v)[NV], float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float s = wave_sum(v[i]);
    if (lane == 0) red[wave * NV + i] = s;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = (red[i] + red[NV + i]) + (red[2 * NV + i] + red[3 * NV + i]);
  __syncthreads();
}

// ---- attention pooling, forward: grid (Bs, 3), 512 threads ------------------------------------------------------------------
// S (3*Bs*L, ldS) fp32: column s*nh + h of the rows of modality s holds the folded score of head h.
// probs (Bs,3,nh,L-1), pooled (Bs,3,nh,C): pooled[b,s,h,:] = sum_n p[n] tok[(s*Bs+b)*L + 1 + n, :]
template <typename T>
__global__ __launch_bounds__(512) void dgaf_attn_pool_kernel(const float* __restrict__ S, int64_t ldS, const T* __restrict__ tok,
                                                             float* __restrict__ pooled, float* __restrict__ probs, int Bs, int L, int C,
                                                             int nh) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int b = blockIdx.x, s = blockIdx.y, N = L - 1;
  float* P = sm;                               // [nh][N]
  float* red = sm + nh * N;                    // [blockDim][4]
  const int64_t row0 = (int64_t)(s * Bs + b) * L + 1;
  for (int i = threadIdx.x; i < nh * N; i += blockDim.x) {
    const int h = i / N, n = i - h * N;
    P[i] = S[(row0 + n) * ldS + s * nh + h];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int h = wave; h < nh; h += nw) {
    float* ph = P + h * N;
    float m = -INFINITY;
    for (int i = lane; i < N; i += 64) m = fmaxf(m, ph[i]);
    m = wave_max(m);
    float sum = 0.f;
    for (int i = lane; i < N; i += 64) {
      const float e = __expf(ph[i] - m);
      ph[i] = e;
      sum += e;
    }
    const float inv = 1.f / wave_sum(sum);
    for (int i = lane; i < N; i += 64) {
      const float p = ph[i] * inv;
      ph[i] = p;
      if (probs) probs[((int64_t)(b * 3 + s) * nh + h) * N + i] = p;
    }
  }
  __syncthreads();
  const int ngroups = C >> 2, nsplit = blockDim.x / ngroups;
  const int grp = threadIdx.x % ngroups, split = threadIdx.x / ngroups;
  float4 acc[NH_MAX];
#pragma unroll
  for (int h = 0; h < NH_MAX; ++h) acc[h] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (split < nsplit) {
    for (int n = split; n < N; n += nsplit) {
      const float4 x = io<T>::ld4(tok + (row0 + n) * C + grp * 4);
#pragma unroll
      for (int h = 0; h < NH_MAX; ++h) {
        if (h < nh) {
          const float p = P[h * N + n];
          acc[h].x = fmaf(p, x.x, acc[h].x); acc[h].y = fmaf(p, x.y, acc[h].y);
          acc[h].z = fmaf(p, x.z, acc[h].z); acc[h].w = fmaf(p, x.w, acc[h].w);
        }
      }
    }
  }
  float* dst = pooled + (int64_t)(b * 3 + s) * nh * C;
#pragma unroll
  for (int h = 0; h < NH_MAX; ++h) {
    if (h < nh) {                                            // uniform
      float4 v = acc[h];
      if (nsplit > 1) {
        __syncthreads();
        if (split < nsplit) *reinterpret_cast<float4*>(red + threadIdx.x * 4) = v;
        __syncthreads();
        if (split == 0)
          for (int sp = 1; sp < nsplit; ++sp) {
            const float4 o = *reinterpret_cast<const float4*>(red + (sp * ngroups + grp) * 4);
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
          }
      }
      if (split == 0) *reinterpret_cast<float4*>(dst + (int64_t)h * C + grp * 4) = v;
    }
  }
}

// ---- attention pooling, backward: grid (Bs, 3), 512 threads; a wave walks tokens ------------------------------------------------
// dS rows of this (sample, modality) are written over all ldS columns (zeros outside the modality's nh columns and on the
// cls row); dtok rows likewise (cls row = 0).  sum_m p_m dp_m = dpooled_h . pooled_h, so no second pass over the tokens.
template <typename T>
__global__ __launch_bounds__(512) void dgaf_attn_pool_bwd_kernel(const float* __restrict__ dpooled, const float* __restrict__ pooled,
                                                                 const float* __restrict__ probs, const T* __restrict__ tok,
                                                                 float* __restrict__ dS, int64_t ldS, float* __restrict__ dtok, int Bs, int L,
                                                                 int C, int nh) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int b = blockIdx.x, s = blockIdx.y, N = L - 1;
  float* dP = sm;                              // [nh][C]
  float* P = dP + nh * C;                      // [nh][N]
  float* rowdot = P + nh * N;                  // [NH_MAX]
  const int64_t bs = (int64_t)(b * 3 + s);
  for (int i = threadIdx.x; i < nh * C; i += blockDim.x) dP[i] = dpooled[bs * nh * C + i];
  for (int i = threadIdx.x; i < nh * N; i += blockDim.x) P[i] = probs[bs * nh * N + i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int h = wave; h < nh; h += nw) {
    float a = 0.f;
    for (int c = lane; c < C; c += 64) a = fmaf(dP[h * C + c], pooled[(bs * nh + h) * C + c], a);
    a = wave_sum(a);
    if (lane == 0) rowdot[h] = a;
  }
  __syncthreads();
  const int64_t rowc = (int64_t)(s * Bs + b) * L;
  for (int n = wave; n < L; n += nw) {         // n == 0: the cls row (zeros), n >= 1: patch n-1
    float* ds = dS + (rowc + n) * ldS;
    float* dt = dtok + (rowc + n) * C;
    if (n == 0) {
      for (int c = lane; c < ldS; c += 64) ds[c] = 0.f;
      for (int c = lane * 4; c < C; c += 256) *reinterpret_cast<float4*>(dt + c) = make_float4(0.f, 0.f, 0.f, 0.f);
      continue;
    }
    const T* x = tok + (rowc + n) * C;
    float dp[NH_MAX];
#pragma unroll
    for (int h = 0; h < NH_MAX; ++h) dp[h] = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
      const float4 xv = io<T>::ld4(x + c);
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int h = 0; h < NH_MAX; ++h) {
        if (h < nh) {
          const float4 g = *reinterpret_cast<const float4*>(dP + h * C + c);
          dp[h] = fmaf(g.x, xv.x, fmaf(g.y, xv.y, fmaf(g.z, xv.z, fmaf(g.w, xv.w, dp[h]))));
          const float p = P[h * N + n - 1];
          o.x = fmaf(p, g.x, o.x); o.y = fmaf(p, g.y, o.y); o.z = fmaf(p, g.z, o.z); o.w = fmaf(p, g.w, o.w);
        }
      }
      *reinterpret_cast<float4*>(dt + c) = o;
    }
#pragma unroll
    for (int h = 0; h < NH_MAX; ++h)
      if (h < nh) dp[h] = P[h * N + n - 1] * (wave_sum(dp[h]) - rowdot[h]);
    for (int c = lane; c < ldS; c += 64) {
      const int hh = c - s * nh;
      float v = 0.f;
#pragma unroll
      for (int h = 0; h < NH_MAX; ++h) v = (h == hh && h < nh) ? dp[h] : v;
      ds[c] = v;
    }
  }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// ---- gates + blend, forward: one workgroup per sample ----------------------------------------------------------------------------
// h (Bs,3,C) pooled modality features, E (Bs,3,C) = entropy_proj(h), g1n (Bs,C) = LayerNorm(gate_net.0(cat h)) BEFORE the ReLU.
// fused (Bs,C) = a*sum_s ew_s h_s + (1-a)*sum_s gate_s h_s;  relu_out (optional, Bs,C) = relu(g1n);
// aux (Bs,16) = {ew[3], gate[3], H[3], z[3], A[3], -}  (kept for the backward)
__global__ __launch_bounds__(256) void dgaf_gate_fuse_kernel(const float* __restrict__ h, const float* __restrict__ E, const float* __restrict__ g1n,
                                                             const float* __restrict__ W2, const float* __restrict__ b2,
                                                             const float* __restrict__ alpha_raw, float tau, float* __restrict__ fused,
                                                             float* __restrict__ relu_out, float* __restrict__ aux, int C) {
  __shared__ float red[4 * 9];
  const int b = blockIdx.x;
  const float* hb = h + (int64_t)b * 3 * C;
  const float* Eb = E + (int64_t)b * 3 * C;
  const float* gb = g1n + (int64_t)b * C;
  float v[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) v[i] = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float r = fmaxf(gb[c], 0.f);
    if (relu_out) relu_out[(int64_t)b * C + c] = r;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      v[s] += fabsf(hb[s * C + c]) + 1e-8f;
      v[3 + s] += Eb[s * C + c];
      v[6 + s] = fmaf(W2[s * C + c], r, v[6 + s]);
    }
  }
  block_sum<9>(v, red);
  float ent[3] = {0.f, 0.f, 0.f};
  for (int c = threadIdx.x; c < C; c += 256) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const float p = (fabsf(hb[s * C + c]) + 1e-8f) / v[s];
      ent[s] -= p * __logf(p + 1e-8f);
    }
  }
  block_sum<3>(ent, red);
  float z[3], sc[3], gate[3];
  float m = -INFINITY;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    z[s] = v[3 + s] / (float)C;
    sc[s] = z[s] * __expf(-ent[s] / tau);
    m = fmaxf(m, sc[s]);
    gate[s] = sigmoidf_(v[6 + s] + b2[s]);
  }
  float den = 0.f;
#pragma unroll
  for (int s = 0; s < 3; ++s) { sc[s] = __expf(sc[s] - m); den += sc[s]; }
  const float a = sigmoidf_(*alpha_raw);
  float wgt[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) { sc[s] /= den; wgt[s] = a * sc[s] + (1.f - a) * gate[s]; }
  for (int c = threadIdx.x; c < C; c += 256)
    fused[(int64_t)b * C + c] = wgt[0] * hb[c] + wgt[1] * hb[C + c] + wgt[2] * hb[2 * C + c];
  if (aux && threadIdx.x < 3) {
    const int s = threadIdx.x;
    float* ab = aux + (int64_t)b * 16;
    ab[s] = s == 0 ? sc[0] : s == 1 ? sc[1] : sc[2];
    ab[3 + s] = s == 0 ? gate[0] : s == 1 ? gate[1] : gate[2];
    ab[6 + s] = s == 0 ? ent[0] : s == 1 ? ent[1] : ent[2];
    ab[9 + s] = s == 0 ? z[0] : s == 1 ? z[1] : z[2];
    ab[12 + s] = s == 0 ? v[0] : s == 1 ? v[1] : v[2];
  }
}

// ---- gates + blend, backward ---------------------------------------------------------------------------------------------------------
// dh (Bs,3,C) = dres (optional: the gradient arriving on h directly, h_s + enh) + blend + entropy paths;
// dE (Bs,3,C) = dz_s / C; dg1n (Bs,C) through the ReLU and gate_net.3; dlogit (Bs,3) (-> dW2, db2 by the caller's GEMMs);
// dalpha (Bs) per-sample gradient of the raw alpha parameter.
__global__ __launch_bounds__(256) void dgaf_gate_fuse_bwd_kernel(const float* __restrict__ dfused, const float* __restrict__ dres,
                                                                 const float* __restrict__ h, const float* __restrict__ g1n,
                                                                 const float* __restrict__ W2, const float* __restrict__ aux,
                                                                 const float* __restrict__ alpha_raw, float tau, float* __restrict__ dh,
                                                                 float* __restrict__ dE, float* __restrict__ dg1n, float* __restrict__ dlogit,
                                                                 float* __restrict__ dalpha, int C) {
  __shared__ float red[4 * 6];
  const int b = blockIdx.x;
  const float* hb = h + (int64_t)b * 3 * C;
  const float* df = dfused + (int64_t)b * C;
  const float* ab = aux + (int64_t)b * 16;
  float ew[3], gate[3], ent[3], z[3], A[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) { ew[s] = ab[s]; gate[s] = ab[3 + s]; ent[s] = ab[6 + s]; z[s] = ab[9 + s]; A[s] = ab[12 + s]; }
  float v[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) v[i] = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float d = df[c];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const float x = hb[s * C + c];
      v[s] = fmaf(d, x, v[s]);
      const float p = (fabsf(x) + 1e-8f) / A[s];
      const float u = -(__logf(p + 1e-8f) + p / (p + 1e-8f));
      v[3 + s] = fmaf(u, p, v[3 + s]);
    }
  }
  block_sum<6>(v, red);
  const float a = sigmoidf_(*alpha_raw);
  float dew[3], dot = 0.f, dal = 0.f;
#pragma unroll
  for (int s = 0; s < 3; ++s) { dew[s] = a * v[s]; dot += ew[s] * dew[s]; dal += (ew[s] - gate[s]) * v[s]; }
  float dz[3], dH[3], dlg[3], wgt[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const float dsc = ew[s] * (dew[s] - dot);
    const float t = __expf(-ent[s] / tau);
    dz[s] = dsc * t;
    dH[s] = -dsc * z[s] * t / tau;
    dlg[s] = (1.f - a) * v[s] * gate[s] * (1.f - gate[s]);
    wgt[s] = a * ew[s] + (1.f - a) * gate[s];
  }
  for (int c = threadIdx.x; c < C; c += 256) {
    const float d = df[c];
    const float gpre = g1n[(int64_t)b * C + c];
    float dg = 0.f;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const float x = hb[s * C + c];
      const float p = (fabsf(x) + 1e-8f) / A[s];
      const float u = -(__logf(p + 1e-8f) + p / (p + 1e-8f));
      const float sgn = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
      float g = d * wgt[s] + dH[s] * (u - v[3 + s]) / A[s] * sgn;
      if (dres) g += dres[(int64_t)b * 3 * C + s * C + c];
      dh[(int64_t)b * 3 * C + s * C + c] = g;
      dE[(int64_t)b * 3 * C + s * C + c] = dz[s] / (float)C;
      dg = fmaf(dlg[s], W2[s * C + c], dg);
    }
    dg1n[(int64_t)b * C + c] = gpre > 0.f ? dg : 0.f;
  }
  if (threadIdx.x < 3) dlogit[(int64_t)b * 3 + threadIdx.x] = threadIdx.x == 0 ? dlg[0] : threadIdx.x == 1 ? dlg[1] : dlg[2];
  if (threadIdx.x == 0) dalpha[b] = dal * a * (1.f - a);
}

// out[b, s*C + c] = h[b,s,c] + enh[b,c]
__global__ __launch_bounds__(256) void dgaf_combine_kernel(const float* __restrict__ h, const float* __restrict__ enh, float* __restrict__ out,
                                                           int64_t n, int C) {
  for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int64_t b = i / (3LL * C);
    out[i] = h[i] + enh[b * C + c];
  }
}

// denh[b,c] = sum_s dout[b, s*C + c]
__global__ __launch_bounds__(256) void dgaf_sum3_kernel(const float* __restrict__ dout, float* __restrict__ denh, int64_t n, int C) {
  for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int64_t b = i / C;
    const float* d = dout + b * 3 * C + c;
    denh[i] = d[0] + d[C] + d[2 * C];
  }
}

inline unsigned grid_for(int64_t n) {
  int64_t g = cdiv64(n, 256);
  return (unsigned)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int demo_dgaf_attn_pool(const float* S, int64_t ldS, const void* tokens, int32_t dt, float* pooled, float* probs, int32_t Bs,
                                   int32_t L, int32_t C, int32_t nh, void* stream) {
  DEMO_REQUIRE(S && tokens && pooled, "demo_dgaf_attn_pool: null operand");
  DEMO_REQUIRE(Bs > 0 && L >= 2, "demo_dgaf_attn_pool: Bs=%d L=%d", Bs, L);
  DEMO_REQUIRE(nh >= 1 && nh <= NH_MAX, "demo_dgaf_attn_pool: heads=%d unsupported (<=%d)", nh, NH_MAX);
  DEMO_REQUIRE(C % 4 == 0 && C / 4 <= 256 && C >= 4, "demo_dgaf_attn_pool: C=%d unsupported", C);
  DEMO_REQUIRE(ldS >= 3 * nh, "demo_dgaf_attn_pool: ldS too small");
  const size_t shm = sizeof(float) * ((size_t)nh * (L - 1) + 512 * 4);
  DEMO_REQUIRE(shm <= 64 * 1024, "demo_dgaf_attn_pool: L=%d too long", L);
  hipStream_t s = (hipStream_t)stream;
  if (dt == DEMO_BF16)
    hipLaunchKernelGGL(dgaf_attn_pool_kernel<bf16_t>, dim3(Bs, 3), dim3(512), shm, s, S, ldS, (const bf16_t*)tokens, pooled, probs, Bs, L, C, nh);
  else
    hipLaunchKernelGGL(dgaf_attn_pool_kernel<float>, dim3(Bs, 3), dim3(512), shm, s, S, ldS, (const float*)tokens, pooled, probs, Bs, L, C, nh);
  DEMO_LAUNCH_CHECK("demo_dgaf_attn_pool");
  return 0;
}

extern "C" int demo_dgaf_attn_pool_bwd(const float* dpooled, const float* pooled, const float* probs, const void* tokens, int32_t dt,
                                       float* dS, int64_t ldS, float* dtokens, int32_t Bs, int32_t L, int32_t C, int32_t nh, void* stream) {
  DEMO_REQUIRE(dpooled && pooled && probs && tokens && dS && dtokens, "demo_dgaf_attn_pool_bwd: null operand");
  DEMO_REQUIRE(Bs > 0 && L >= 2, "demo_dgaf_attn_pool_bwd: Bs=%d L=%d", Bs, L);
  DEMO_REQUIRE(nh >= 1 && nh <= NH_MAX, "demo_dgaf_attn_pool_bwd: heads=%d unsupported (<=%d)", nh, NH_MAX);
  DEMO_REQUIRE(C % 4 == 0 && C >= 4, "demo_dgaf_attn_pool_bwd: C=%d unsupported", C);
  DEMO_REQUIRE(ldS >= 3 * nh, "demo_dgaf_attn_pool_bwd: ldS too small");
  const size_t shm = sizeof(float) * ((size_t)nh * C + (size_t)nh * (L - 1) + NH_MAX);
  DEMO_REQUIRE(shm <= 64 * 1024, "demo_dgaf_attn_pool_bwd: heads*C=%d too large for LDS", nh * C);
  hipStream_t s = (hipStream_t)stream;
  if (dt == DEMO_BF16)
    hipLaunchKernelGGL(dgaf_attn_pool_bwd_kernel<bf16_t>, dim3(Bs, 3), dim3(512), shm, s, dpooled, pooled, probs, (const bf16_t*)tokens, dS, ldS,
                       dtokens, Bs, L, C, nh);
  else
    hipLaunchKernelGGL(dgaf_attn_pool_bwd_kernel<float>, dim3(Bs, 3), dim3(512), shm, s, dpooled, pooled, probs, (const float*)tokens, dS, ldS,
                       dtokens, Bs, L, C, nh);
  DEMO_LAUNCH_CHECK("demo_dgaf_attn_pool_bwd");
  return 0;
}

extern "C" int demo_dgaf_gate_fuse(const float* h, const float* E, const float* g1n, const float* W2, const float* b2, const float* alpha_raw,
                                   float tau, float* fused, float* relu_out, float* aux, int32_t Bs, int32_t C, void* stream) {
  DEMO_REQUIRE(h && E && g1n && W2 && b2 && alpha_raw && fused, "demo_dgaf_gate_fuse: null operand");
  DEMO_REQUIRE(Bs > 0 && C > 0 && tau > 0.f, "demo_dgaf_gate_fuse: Bs=%d C=%d tau=%f", Bs, C, tau);
  hipLaunchKernelGGL(dgaf_gate_fuse_kernel, dim3(Bs), dim3(256), 0, (hipStream_t)stream, h, E, g1n, W2, b2, alpha_raw, tau, fused, relu_out, aux, C);
  DEMO_LAUNCH_CHECK("demo_dgaf_gate_fuse");
  return 0;
}

extern "C" int demo_dgaf_gate_fuse_bwd(const float* dfused, const float* dres, const float* h, const float* g1n, const float* W2,
                                       const float* aux, const float* alpha_raw, float tau, float* dh, float* dE, float* dg1n, float* dlogit,
                                       float* dalpha, int32_t Bs, int32_t C, void* stream) {
  DEMO_REQUIRE(dfused && h && g1n && W2 && aux && alpha_raw && dh && dE && dg1n && dlogit && dalpha, "demo_dgaf_gate_fuse_bwd: null operand");
  DEMO_REQUIRE(Bs > 0 && C > 0 && tau > 0.f, "demo_dgaf_gate_fuse_bwd: Bs=%d C=%d tau=%f", Bs, C, tau);
  hipLaunchKernelGGL(dgaf_gate_fuse_bwd_kernel, dim3(Bs), dim3(256), 0, (hipStream_t)stream, dfused, dres, h, g1n, W2, aux, alpha_raw, tau, dh, dE,
                     dg1n, dlogit, dalpha, C);
  DEMO_LAUNCH_CHECK("demo_dgaf_gate_fuse_bwd");
  return 0;
}

extern "C" int demo_dgaf_combine(const float* h, const float* enh, float* out, int32_t Bs, int32_t C, void* stream) {
  DEMO_REQUIRE(h && enh && out && Bs > 0 && C > 0, "demo_dgaf_combine: bad operand");
  const int64_t n = (int64_t)Bs * 3 * C;
  hipLaunchKernelGGL(dgaf_combine_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, h, enh, out, n, C);
  DEMO_LAUNCH_CHECK("demo_dgaf_combine");
  return 0;
}

extern "C" int demo_dgaf_sum3(const float* dout, float* denh, int32_t Bs, int32_t C, void* stream) {
  DEMO_REQUIRE(dout && denh && Bs > 0 && C > 0, "demo_dgaf_sum3: bad operand");
  const int64_t n = (int64_t)Bs * C;
  hipLaunchKernelGGL(dgaf_sum3_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dout, denh, n, C);
  DEMO_LAUNCH_CHECK("demo_dgaf_sum3");
  return 0;
}
