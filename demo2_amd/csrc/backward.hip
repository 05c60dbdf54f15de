// Backward kernels of the small (HBM / launch bound) stages: token embedding, activations, BatchNorm1d (plain and the
// grouped + gated ATMoE form), ATMoE gate softmax, HDM softmax-pool, PIFE pooling.  All math fp32.
#include "common.h"

namespace {

constexpr int HH_MAX = 12;
__constant__ int b_nstreams[7] = {1, 1, 1, 2, 2, 2, 3};
__constant__ int b_streams[7][3] = {{0, 0, 0}, {1, 0, 0}, {2, 0, 0}, {0, 1, 0}, {0, 2, 0}, {1, 2, 0}, {0, 1, 2}};
// branches attending modality m, and the position of m inside that branch's concatenation
__constant__ int b_mbranch[3][4] = {{0, 3, 4, 6}, {1, 3, 5, 6}, {2, 4, 5, 6}};
__constant__ int b_mpos[3][4] = {{0, 0, 0, 0}, {0, 1, 0, 1}, {0, 1, 1, 2}};

__device__ __forceinline__ uint32_t rng_hash(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ULL * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return (uint32_t)((z ^ (z >> 31)) >> 32);
}

// ---- token embedding backward: block per token position t, threads over channels --------------------------------
template <typename TP>
__global__ __launch_bounds__(256) void embed_tokens_bwd_kernel(const float* __restrict__ dpre, const int64_t* __restrict__ cam,
                                                               int n_cam, float cam_scale, int cam_all, TP* __restrict__ dpatch,
                                                               float* __restrict__ dcls, float* __restrict__ dpos,
                                                               float* __restrict__ dcam, int Bn, int np, int W, int* __restrict__ err) {
  // grid (token position, batch slab): partial sums over the slab's images, combined with fp32 atomics
  const int t = blockIdx.x, L = np + 1;
  const int per = (Bn + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(Bn, b0 + per);
  if ((W & 3) == 0) {
    // four consecutive channels per thread (16-B loads, 8-B bf16 stores), four images in flight: the scalar form below walks its slab's
    // images one dependent 4-B load at a time (115 us for 228 MB at B = 128)
    for (int c = threadIdx.x * 4; c < W; c += blockDim.x * 4) {
      float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int bb = b0; bb < b1; bb += 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          v[u] = bb + u < b1 ? *reinterpret_cast<const float4*>(dpre + ((int64_t)(bb + u) * L + t) * W + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int b = bb + u;
          if (b >= b1) break;
          s4.x += v[u].x; s4.y += v[u].y; s4.z += v[u].z; s4.w += v[u].w;
          if (t > 0) io<TP>::st4(dpatch + ((int64_t)b * np + (t - 1)) * W + c, v[u]);
          if (dcam && cam && (cam_all || t == 0)) {
            const int64_t ci = cam[b];
            if (ci >= 0 && ci < n_cam) {
              float* dc = dcam + ci * (int64_t)W + c;
              atomicAdd(dc, cam_scale * v[u].x); atomicAdd(dc + 1, cam_scale * v[u].y);
              atomicAdd(dc + 2, cam_scale * v[u].z); atomicAdd(dc + 3, cam_scale * v[u].w);
            } else if (err && c == 0) atomicOr(err, DEMO_ERR_CAM_RANGE);       // out of range: nothing is written (demo_hip.h)
          }
        }
      }
      if (dpos) {
        float* dp = dpos + (int64_t)t * W + c;
        atomicAdd(dp, s4.x); atomicAdd(dp + 1, s4.y); atomicAdd(dp + 2, s4.z); atomicAdd(dp + 3, s4.w);
      }
      if (t == 0 && dcls) { atomicAdd(dcls + c, s4.x); atomicAdd(dcls + c + 1, s4.y); atomicAdd(dcls + c + 2, s4.z); atomicAdd(dcls + c + 3, s4.w); }
    }
    return;
  }
  for (int c = threadIdx.x; c < W; c += blockDim.x) {
    float s = 0.f;
    for (int b = b0; b < b1; ++b) {
      const float v = dpre[((int64_t)b * L + t) * W + c];
      s += v;
      if (t > 0) io<TP>::st(dpatch + ((int64_t)b * np + (t - 1)) * W + c, v);
      if (dcam && cam && (cam_all || t == 0)) {
        const int64_t ci = cam[b];
        if (ci >= 0 && ci < n_cam) atomicAdd(&dcam[ci * (int64_t)W + c], cam_scale * v);
        else if (err && c == 0) atomicOr(err, DEMO_ERR_CAM_RANGE);       // out of range: nothing is written (demo_hip.h)
      }
    }
    if (dpos) atomicAdd(&dpos[(int64_t)t * W + c], s);
    if (t == 0 && dcls) atomicAdd(&dcls[c], s);
  }
}

__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ pre, int kind, float* __restrict__ y, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = act_apply(pre[i], kind);
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ pre, int kind,
                                                      float* __restrict__ dx, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dx[i] = dy[i] * act_grad(pre[i], kind);
}

// ---- BatchNorm1d backward (training statistics); thread per feature ----------------------------------------------
// forward: y = ((x - mean) * invstd * w + b) * gate
__global__ __launch_bounds__(64) void batchnorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ w, const float* __restrict__ b,
                                                           const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                                                           float* __restrict__ dx, int accumulate_dx, float* __restrict__ dw,
                                                           float* __restrict__ db, int Bn, int F, int gsz, int n_inner, int n_outer,
                                                           int64_t pstride, const float* __restrict__ gate, float* __restrict__ dgate) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  int64_t pi = f;
  int gi = 0, go = 0;
  if (gsz > 0) {
    const int g = f / gsz, k = f - g * gsz;
    gi = g % n_inner; go = g / n_inner;
    pi = ((int64_t)gi * n_outer + go) * pstride + k;
  }
  const float mean = save_mean[f], invstd = save_invstd[f], wv = w[pi], bv = b ? b[pi] : 0.f;
  float sdy = 0.f, sdyx = 0.f;
  for (int r = 0; r < Bn; ++r) {
    const float xh = (x[(int64_t)r * F + f] - mean) * invstd;
    float d = dy[(int64_t)r * F + f];
    if (gate) {
      const int64_t gidx = ((int64_t)r * n_inner + gi) * n_outer + go;
      if (dgate) atomicAdd(&dgate[gidx], d * (xh * wv + bv));
      d *= gate[gidx];
    }
    sdy += d;
    sdyx += d * xh;
  }
  if (dw) dw[pi] += sdyx;
  if (db) db[pi] += sdy;
  const float inv = 1.f / (float)Bn;
  for (int r = 0; r < Bn; ++r) {
    const float xh = (x[(int64_t)r * F + f] - mean) * invstd;
    float d = dy[(int64_t)r * F + f];
    if (gate) d *= gate[((int64_t)r * n_inner + gi) * n_outer + go];
    float v = wv * invstd * (d - sdy * inv - xh * sdyx * inv);
    if (accumulate_dx) v += dx[(int64_t)r * F + f];
    dx[(int64_t)r * F + f] = v;
  }
}

// Sliced form for Bn <= 128 (the training batch): workgroup = 64 features x 4 row slices, every x / dy value of the slice is
// loaded once into registers (all loads in flight together instead of Bn serial latencies per thread), the per-feature
// sums cross the 4 waves through LDS.  The gate gradient is reduced over the wave's 64 features first, so a (row, group)
// word receives gsz/64 atomics instead of gsz.
__global__ __launch_bounds__(256) void batchnorm_bwd_sliced_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                   const float* __restrict__ w, const float* __restrict__ b,
                                                                   const float* __restrict__ save_mean,
                                                                   const float* __restrict__ save_invstd, float* __restrict__ dx,
                                                                   int accumulate_dx, float* __restrict__ dw, float* __restrict__ db,
                                                                   int Bn, int F, int gsz, int n_inner, int n_outer, int64_t pstride,
                                                                   const float* __restrict__ gate, float* __restrict__ dgate) {
  __shared__ float red[2][4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int f = blockIdx.x * 64 + lane;
  const bool fok = f < F;
  const int fc = fok ? f : F - 1;
  int64_t pi = fc;
  int gi = 0, go = 0;
  if (gsz > 0) {
    const int g = fc / gsz, k = fc - g * gsz;
    gi = g % n_inner; go = g / n_inner;
    pi = ((int64_t)gi * n_outer + go) * pstride + k;
  }
  const float mean = save_mean[fc], invstd = save_invstd[fc], wv_ = w[pi], bv = b ? b[pi] : 0.f;
  const bool wave_one_group = gsz > 0 && gsz % 64 == 0;
  float xh[32], d[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int r = wv + 4 * i;
    const bool ok = fok && r < Bn;
    xh[i] = ok ? x[(int64_t)r * F + f] : mean;
    d[i] = ok ? dy[(int64_t)r * F + f] : 0.f;
  }
  float sdy = 0.f, sdyx = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int r = wv + 4 * i;
    xh[i] = (xh[i] - mean) * invstd;
    if (gate && r < Bn) {                                   // r < Bn is wave-uniform
      const int64_t gidx = ((int64_t)r * n_inner + gi) * n_outer + go;
      if (dgate) {
        const float gv = d[i] * (xh[i] * wv_ + bv);
        if (wave_one_group) {
          const float t = wave_sum(gv);
          if (lane == 0) atomicAdd(&dgate[gidx], t);
        } else if (fok) {
          atomicAdd(&dgate[gidx], gv);
        }
      }
      d[i] *= gate[gidx];
    }
    sdy += d[i];
    sdyx += d[i] * xh[i];
  }
  red[0][wv][lane] = sdy;
  red[1][wv][lane] = sdyx;
  __syncthreads();
  sdy = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
  sdyx = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
  if (wv == 0 && fok) {
    if (dw) dw[pi] += sdyx;
    if (db) db[pi] += sdy;
  }
  const float inv = 1.f / (float)Bn;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int r = wv + 4 * i;
    if (fok && r < Bn) {
      float v = wv_ * invstd * (d[i] - sdy * inv - xh[i] * sdyx * inv);
      if (accumulate_dx) v += dx[(int64_t)r * F + f];
      dx[(int64_t)r * F + f] = v;
    }
  }
}

// ---- ATMoE gate softmax backward: block per sample, wave per head ---------------------------------------------------
__global__ __launch_bounds__(256) void atmoe_gate_softmax_bwd_kernel(const float* __restrict__ dgates, const float* __restrict__ gates,
                                                                     const float* __restrict__ q, const float* __restrict__ k,
                                                                     float* __restrict__ dq, float* __restrict__ dk, int C, int H) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = C / H;
  const float scale = rsqrtf((float)d);
  for (int h = wave; h < H; h += 4) {
    float gt[7], dl[7];
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      gt[j] = gates[((int64_t)b * H + h) * 7 + j];
      dl[j] = dgates[((int64_t)b * H + h) * 7 + j];
      dot += gt[j] * dl[j];
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) dl[j] = gt[j] * (dl[j] - dot) * scale;     // d logits (scaled)
    for (int e = lane; e < d; e += 64) {
      const float qv = q[(int64_t)b * C + h * d + e];
      float aq = 0.f;
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const int64_t ki = ((int64_t)b * 7 + j) * C + h * d + e;
        aq = fmaf(dl[j], k[ki], aq);
        dk[ki] = dl[j] * qv;
      }
      dq[(int64_t)b * C + h * d + e] = aq;
    }
  }
}

// ---- HDM softmax-pool backward: block per (sample, modality) ----------------------------------------------------------
// forward (per branch br, head h over the concatenated streams): p = softmax(s); P = dropout(p); pooled = sum P_l x_l; psum = sum P_l
// backward:  dP_l = dpooled . x_l + dpsum ;  rowdot = sum_l P_l dP_l = dpooled . pooled + dpsum * psum ;
//            ds_l = P_l dP_l - p_l rowdot ;  dx_l = sum_{br,h} P_l dpooled
template <typename T>
__global__ __launch_bounds__(512) void hdm_softmax_pool_bwd_kernel(const float* __restrict__ dpooled, const float* __restrict__ dpsum,
                                                                   const float* __restrict__ probs, const T* __restrict__ tok,
                                                                   const float* __restrict__ pooled, const float* __restrict__ psum,
                                                                   float* __restrict__ dS, int64_t ldS, float* __restrict__ dtok,
                                                                   int Bs, int L, int C, int Hh, float p_drop, uint64_t seed,
                                                                   const uint64_t* __restrict__ seed_dev) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  if (seed_dev) seed += *seed_dev;            // same device-resident seed part as the forward read
  const int b = blockIdx.x, m = blockIdx.y;
  const int L3 = 3 * L;
  float* dpl = sm;                          // [4][Hh][C]
  float* rowdot = sm + 4 * Hh * C;          // [4][Hh]
  float* dps = rowdot + 4 * HH_MAX;         // [4][Hh]
  for (int i = threadIdx.x; i < 4 * Hh * C; i += blockDim.x) {
    const int s = i / (Hh * C), rem = i - s * Hh * C;
    dpl[i] = dpooled[((int64_t)(b * 7 + b_mbranch[m][s]) * Hh) * C + rem];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int i = wave; i < 4 * Hh; i += nw) {
    const int s = i / Hh, h = i - s * Hh;
    const int br = b_mbranch[m][s];
    const float* pl = pooled + ((int64_t)(b * 7 + br) * Hh + h) * C;
    float a = 0.f;
    for (int c = lane; c < C; c += 64) a = fmaf(dpl[(s * Hh + h) * C + c], pl[c], a);
    a = wave_sum(a);
    if (lane == 0) {
      const float dp_ = dpsum ? dpsum[(int64_t)(b * 7 + br) * Hh + h] : 0.f;
      rowdot[s * HH_MAX + h] = a + dp_ * psum[(int64_t)(b * 7 + br) * Hh + h];
      dps[s * HH_MAX + h] = dp_;
    }
  }
  __syncthreads();
  const float keep_scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  const uint32_t thresh = p_drop > 0.f ? (uint32_t)(p_drop * 4294967296.0) : 0u;
  // Lane roles: for the dot products / dx a lane owns channel quads (i*64 + lane)*4; for dS a lane owns score columns
  // lane and lane + 64 (column = branch*Hh + head; 7*Hh <= 84).  Everything a column needs that does not depend on the
  // token is fetched once here; per token the column's probability is ONE coalesced load issued with the token's x row
  // (the earlier form loaded it inside the (branch, head) loop: 16 dependent HBM latencies per token).
  const int nch = (C + 255) / 256;
  const int ncol = 7 * Hh;
  int col_s[2];
  float col_rd[2], col_dps[2];
  int64_t col_pbase[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int col = lane + 64 * k;
    col_s[k] = -1;
    col_rd[k] = col_dps[k] = 0.f;
    col_pbase[k] = 0;
    if (col < ncol) {
      const int br = col / Hh, h = col - br * Hh;
#pragma unroll
      for (int ss = 0; ss < 4; ++ss)
        if (b_mbranch[m][ss] == br) {
          col_s[k] = ss;
          col_rd[k] = rowdot[ss * HH_MAX + h];
          col_dps[k] = dps[ss * HH_MAX + h];
          col_pbase[k] = ((int64_t)(b * 7 + br) * Hh + h) * L3 + (int64_t)b_mpos[m][ss] * L;
        }
    }
  }
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int l0 = wave * 2; l0 < L; l0 += 2 * nw) {     // two tokens per wave per pass: each dpooled row read serves both
    const bool has2 = l0 + 1 < L;
    const int64_t row0 = (int64_t)(m * Bs + b) * L + l0;
    float4 x0[4], x1[4], dx0[4], dx1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c4 = (i * 64 + lane) * 4;
      const bool in = i < nch && c4 < C;
      x0[i] = in ? io<T>::ld4(tok + row0 * C + c4) : zero4;
      x1[i] = (in && has2) ? io<T>::ld4(tok + (row0 + 1) * C + c4) : zero4;
      dx0[i] = dx1[i] = zero4;
    }
    float p0[2], p1[2], a0[2] = {0.f, 0.f}, a1[2] = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      p0[k] = col_s[k] >= 0 ? probs[col_pbase[k] + l0] : 0.f;
      p1[k] = (col_s[k] >= 0 && has2) ? probs[col_pbase[k] + l0 + 1] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int cb = b_mbranch[m][s] * Hh;
      for (int h = 0; h < Hh; ++h) {
        const float* dp_h = dpl + (s * Hh + h) * C;
        float d0 = 0.f, d1 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c4 = (i * 64 + lane) * 4;
          if (i < nch && c4 < C) {
            const float4 wv = *reinterpret_cast<const float4*>(dp_h + c4);
            d0 += (wv.x * x0[i].x + wv.y * x0[i].y) + (wv.z * x0[i].z + wv.w * x0[i].w);
            d1 += (wv.x * x1[i].x + wv.y * x1[i].y) + (wv.z * x1[i].z + wv.w * x1[i].w);
          }
        }
        d0 = wave_sum(d0);
        d1 = wave_sum(d1);
        const int col = cb + h;                        // wave-uniform
        if (col < 64) { if (lane == col) { a0[0] = d0; a1[0] = d1; } }
        else if (lane == col - 64) { a0[1] = d0; a1[1] = d1; }
      }
    }
    // dS columns of this lane for both tokens (columns of branches that do not attend this modality are zero)
    float P0[2], P1[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int col = lane + 64 * k;
      float ds0 = 0.f, ds1 = 0.f;
      P0[k] = P1[k] = 0.f;
      if (col_s[k] >= 0) {
        P0[k] = p0[k];
        P1[k] = p1[k];
        if (p_drop > 0.f) {
          P0[k] = (rng_hash(seed, (uint64_t)(col_pbase[k] + l0)) >= thresh) ? p0[k] * keep_scale : 0.f;
          P1[k] = (rng_hash(seed, (uint64_t)(col_pbase[k] + l0 + 1)) >= thresh) ? p1[k] * keep_scale : 0.f;
        }
        ds0 = P0[k] * (a0[k] + col_dps[k]) - p0[k] * col_rd[k];
        ds1 = P1[k] * (a1[k] + col_dps[k]) - p1[k] * col_rd[k];
      }
      if (col < ldS) {
        dS[row0 * ldS + col] = ds0;
        if (has2) dS[(row0 + 1) * ldS + col] = ds1;
      }
    }
    for (int c = lane + 128; c < ldS; c += 64) {
      dS[row0 * ldS + c] = 0.f;
      if (has2) dS[(row0 + 1) * ldS + c] = 0.f;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int cb = b_mbranch[m][s] * Hh;
      for (int h = 0; h < Hh; ++h) {
        const float* dp_h = dpl + (s * Hh + h) * C;
        const int col = cb + h;
        const float Pa = col < 64 ? lane_value(P0[0], col) : lane_value(P0[1], col - 64);
        const float Pb = col < 64 ? lane_value(P1[0], col) : lane_value(P1[1], col - 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c4 = (i * 64 + lane) * 4;
          if (i < nch && c4 < C) {
            const float4 wv = *reinterpret_cast<const float4*>(dp_h + c4);
            dx0[i].x = fmaf(Pa, wv.x, dx0[i].x); dx0[i].y = fmaf(Pa, wv.y, dx0[i].y);
            dx0[i].z = fmaf(Pa, wv.z, dx0[i].z); dx0[i].w = fmaf(Pa, wv.w, dx0[i].w);
            dx1[i].x = fmaf(Pb, wv.x, dx1[i].x); dx1[i].y = fmaf(Pb, wv.y, dx1[i].y);
            dx1[i].z = fmaf(Pb, wv.z, dx1[i].z); dx1[i].w = fmaf(Pb, wv.w, dx1[i].w);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c4 = (i * 64 + lane) * 4;
      if (i < nch && c4 < C) {
        *reinterpret_cast<float4*>(dtok + row0 * C + c4) = dx0[i];
        if (has2) *reinterpret_cast<float4*>(dtok + (row0 + 1) * C + c4) = dx1[i];
      }
    }
  }
}

// do (Bs,7,C): dbias[br*stride + c] += sum_b do*psum ; dpsum[b,br,h] = sum_{c in head h} do * bias
// blocks [0, 7): dbias of branch br (thread per channel, loop over samples, coalesced);
// blocks [7, ...): dpsum, one wave per (sample, branch, head): dot over the head's channels
__global__ __launch_bounds__(256) void hdm_vbias_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ bias, int64_t bias_stride,
                                                            const float* __restrict__ psum, float* __restrict__ dbias,
                                                            float* __restrict__ dpsum, int Bs, int C, int Hh) {
  const int dh = C / Hh;
  constexpr int SLABS = 8;                 // dbias: 7 branches x SLABS sample slabs, combined with fp32 atomics (one block per branch walked all
                                           // samples with one dependent load each: 52 us at B = 128)
  if (blockIdx.x < 7 * SLABS) {
    const int br = blockIdx.x % 7, slab = blockIdx.x / 7;
    const int per = (Bs + SLABS - 1) / SLABS, b0 = slab * per, b1 = min(Bs, b0 + per);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      float s = 0.f;
#pragma unroll 4
      for (int b = b0; b < b1; ++b) s = fmaf(dy[((int64_t)b * 7 + br) * C + c], psum[((int64_t)b * 7 + br) * Hh + c / dh], s);
      atomicAdd(&dbias[br * bias_stride + c], s);
    }
    return;
  }
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)(blockIdx.x - 7 * SLABS) * 4 + (threadIdx.x >> 6);       // (b*7 + br)*Hh + h
  if (item >= (int64_t)Bs * 7 * Hh) return;
  const int h = (int)(item % Hh);
  const int64_t bb = item / Hh;
  const int br = (int)(bb % 7);
  float a = 0.f;
  for (int e = lane; e < dh; e += 64) a = fmaf(dy[bb * C + h * dh + e], bias[br * bias_stride + h * dh + e], a);
  a = wave_sum(a);
  if (lane == 0) dpsum[item] += a;
}

// out[b, c] += src[(b*L + 0), c]   (row-0 gradient of a token stream -> the (B, ld) global-feature gradient)
__global__ __launch_bounds__(128) void token0_grad_kernel(const float* __restrict__ dtok, float* __restrict__ out, int64_t ld, int L, int C) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) out[(int64_t)b * ld + c] += dtok[(int64_t)b * L * C + c];
}

// final token gradient (dtype T):  row 0 = dcls[b,:]   (+ dtok row 0 when keep_row0)
//                                  row l>=1 = dtok[b,l,:] + dmean[b,:] / (L-1)
template <typename T>
__global__ __launch_bounds__(256) void tokens_grad_kernel(const float* __restrict__ dtok, const float* __restrict__ dcls, int64_t cls_ld,
                                                          const float* __restrict__ dmean, int64_t mean_ld, int keep_row0,
                                                          T* __restrict__ out, int L, int C) {
  const int b = blockIdx.x;
  const float inv = 1.f / (float)(L - 1);
  // blockIdx.y: slice of the token rows (a grid of Bn workgroups alone leaves half the CUs idle at B = 128)
  const int lper = (L + gridDim.y - 1) / gridDim.y, l0 = blockIdx.y * lper, l1 = min(L, l0 + lper);
  for (int i = l0 * (C >> 2) + threadIdx.x; i < l1 * (C >> 2); i += blockDim.x) {
    const int l = i / (C >> 2), c4 = (i - l * (C >> 2)) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (dtok && (l > 0 || keep_row0)) v = *reinterpret_cast<const float4*>(dtok + ((int64_t)b * L + l) * C + c4);
    if (l == 0) {
      if (dcls) {
        const float4 a = *reinterpret_cast<const float4*>(dcls + (int64_t)b * cls_ld + c4);
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
      }
    } else if (dmean) {
      const float4 a = *reinterpret_cast<const float4*>(dmean + (int64_t)b * mean_ld + c4);
      v.x += a.x * inv; v.y += a.y * inv; v.z += a.z * inv; v.w += a.w * inv;
    }
    io<T>::st4(out + ((int64_t)b * L + l) * C + c4, v);
  }
}

// Fused flat Adam: one pass over the flat fp32 parameter / gradient / moment buffers; also emits the bf16 weight
// shadow for the next forward and clears the gradient (torch.optim.Adam update rule, L2 decay added to the gradient).
// KIND: 0 torch.optim.Adam (L2 decay folded into the gradient), 1 AdamW (decoupled decay), 2 SGD with momentum b1 (v unused).
// Parameter groups (solver/make_optimizer.py:5-17: bias decay, doubled classifier lr; plus 'no update' for parameters the reference's
// optimizer never moves): cls[i >> 4] is the class of the 64-element block a float4 belongs to (parameters start on 64-element
// boundaries of the flat buffer), tab[2c] its lr multiplier and tab[2c+1] its weight decay; class data absent = one group.
template <int KIND>
__global__ __launch_bounds__(256) void optim_step_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, bf16_t* __restrict__ shadow, int64_t n4, float lr,
                                                         float b1, float b2, float eps, float wd, float bc1, float bc2_rsqrt,
                                                         int zero_grad, const float* __restrict__ hyper,
                                                         const uint8_t* __restrict__ cls, const float* __restrict__ tab) {
  if (hyper) {              // {lr, update count t} in device memory: a captured step replays with fresh values
    lr = hyper[0];
    const float t = hyper[1];
    bc1 = 1.f - powf(b1, t);
    bc2_rsqrt = rsqrtf(1.f - powf(b2, t));
  }
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float lre = lr, wde = wd;
    bool active = true;
    if (cls) {
      const int c = cls[i >> 4];
      const float mult = tab[2 * c];
      lre = lr * mult;
      wde = tab[2 * c + 1];
      active = mult != 0.f || wde != 0.f;      // class {0, 0}: frozen / never-used parameters keep their value and an all-zero state
    }
    float4 pv = *reinterpret_cast<const float4*>(p + i * 4);
    const float4 gv = *reinterpret_cast<const float4*>(g + i * 4);
    float4 mv = *reinterpret_cast<const float4*>(m + i * 4);
    float4 vv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KIND != 2) vv = *reinterpret_cast<const float4*>(v + i * 4);
    float pa[4] = {pv.x, pv.y, pv.z, pv.w}, ga[4] = {gv.x, gv.y, gv.z, gv.w};
    float ma[4] = {mv.x, mv.y, mv.z, mv.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
    if (active) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (KIND == 2) {                       // torch.optim.SGD: buf = momentum * buf + (g + wd p); p -= lr buf  (dampening 0)
          const float gg = ga[e] + wde * pa[e];
          ma[e] = b1 * ma[e] + gg;
          pa[e] -= lre * ma[e];
        } else {
          float gg = ga[e];
          if (KIND == 0) gg += wde * pa[e];
          else pa[e] *= 1.f - lre * wde;       // AdamW: decoupled decay first
          ma[e] = b1 * ma[e] + (1.f - b1) * gg;
          va[e] = b2 * va[e] + (1.f - b2) * gg * gg;
          const float denom = sqrtf(va[e]) * bc2_rsqrt + eps;
          pa[e] -= (lre / bc1) * (ma[e] / denom);
        }
      }
    }
    pv = make_float4(pa[0], pa[1], pa[2], pa[3]);
    *reinterpret_cast<float4*>(p + i * 4) = pv;
    *reinterpret_cast<float4*>(m + i * 4) = make_float4(ma[0], ma[1], ma[2], ma[3]);
    if (KIND != 2) *reinterpret_cast<float4*>(v + i * 4) = make_float4(va[0], va[1], va[2], va[3]);
    if (shadow) io<bf16_t>::st4(shadow + i * 4, pv);
    if (zero_grad) *reinterpret_cast<float4*>(g + i * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

}  // namespace

extern "C" int demo_optim_step(int32_t kind, float* p, float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr, float beta1,
                               float beta2, float eps, float weight_decay, int32_t step, int32_t zero_grad, const float* hyper_dev,
                               const uint8_t* block_class, const float* class_table, void* stream) {
  DEMO_REQUIRE(kind >= 0 && kind <= 2, "demo_optim_step: kind %d (0 Adam, 1 AdamW, 2 SGD)", kind);
  DEMO_REQUIRE(n % 4 == 0 && (!block_class || n % 64 == 0) && (step >= 1 || hyper_dev),
               "demo_optim_step: n must be a multiple of 4 (64 with parameter classes) and step >= 1");
  DEMO_REQUIRE(p && g && m && (v || kind == 2), "demo_optim_step: null buffer");
  DEMO_REQUIRE((block_class == nullptr) == (class_table == nullptr), "demo_optim_step: block_class and class_table go together");
  const float bc1 = (step >= 1 && kind != 2) ? 1.f - powf(beta1, (float)step) : 1.f;
  const float bc2 = (step >= 1 && kind != 2) ? 1.f - powf(beta2, (float)step) : 1.f;
  int64_t grid = cdiv64(n / 4, 256);
  if (grid > 4096) grid = 4096;
  const float* hyper = kind == 2 ? nullptr : hyper_dev;     // (SGD has no bias correction; a captured SGD step is not built)
#define DEMO_OPT(K)                                                                                                                      \
  hipLaunchKernelGGL(optim_step_kernel<K>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)shadow_bf16,   \
                     n / 4, lr, beta1, beta2, eps, weight_decay, bc1, 1.f / sqrtf(bc2), zero_grad, hyper, block_class, class_table)
  if (kind == 0) DEMO_OPT(0);
  else if (kind == 1) DEMO_OPT(1);
  else DEMO_OPT(2);
#undef DEMO_OPT
  DEMO_LAUNCH_CHECK("demo_optim_step");
  return 0;
}

extern "C" int demo_adam_step(float* p, float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, int32_t step, int32_t zero_grad, const float* hyper_dev, void* stream) {
  return demo_optim_step(0, p, g, m, v, shadow_bf16, n, lr, beta1, beta2, eps, weight_decay, step, zero_grad, hyper_dev, nullptr, nullptr,
                         stream);
}

extern "C" int demo_embed_tokens_bwd(const float* dpre, const int64_t* cam, int32_t n_cam, float cam_scale, int32_t cam_all_tokens,
                                     void* dpatch, int32_t dt_patch, float* dcls, float* dpos, float* dcam, int32_t Bn, int32_t np,
                                     int32_t W, int32_t* err, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const dim3 egrid(np + 1, (unsigned)(Bn >= 64 ? 16 : (Bn >= 8 ? 4 : 1)));
  if (dt_patch == DEMO_BF16)
    hipLaunchKernelGGL(embed_tokens_bwd_kernel<bf16_t>, egrid, dim3(256), 0, s, dpre, cam, n_cam, cam_scale, cam_all_tokens,
                       (bf16_t*)dpatch, dcls, dpos, dcam, Bn, np, W, err);
  else
    hipLaunchKernelGGL(embed_tokens_bwd_kernel<float>, egrid, dim3(256), 0, s, dpre, cam, n_cam, cam_scale, cam_all_tokens,
                       (float*)dpatch, dcls, dpos, dcam, Bn, np, W, err);
  DEMO_LAUNCH_CHECK("demo_embed_tokens_bwd");
  return 0;
}

extern "C" int demo_act_fwd(const float* pre, int32_t kind, float* y, int64_t n, void* stream) {
  int64_t g = cdiv64(n, 256);
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(act_fwd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, pre, kind, y, n);
  DEMO_LAUNCH_CHECK("demo_act_fwd");
  return 0;
}

extern "C" int demo_act_bwd(const float* dy, const float* pre, int32_t kind, float* dx, int64_t n, void* stream) {
  int64_t g = cdiv64(n, 256);
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, dy, pre, kind, dx, n);
  DEMO_LAUNCH_CHECK("demo_act_bwd");
  return 0;
}

extern "C" int demo_batchnorm_bwd(const float* dy, const float* x, const float* w, const float* b, const float* save_mean,
                                  const float* save_invstd, float* dx, int32_t accumulate_dx, float* dw, float* db, int32_t Bn,
                                  int32_t F, int32_t gsz, int32_t n_inner, int32_t n_outer, int64_t pstride, const float* gate,
                                  float* dgate, void* stream) {
  DEMO_REQUIRE(gsz == 0 || (F % gsz == 0 && (F / gsz) == n_inner * n_outer), "demo_batchnorm_bwd: bad grouping");
  DEMO_REQUIRE(!gate || (gsz > 0 && b), "demo_batchnorm_bwd: gate requires grouping and the bias");
  if (Bn <= 128)
    hipLaunchKernelGGL(batchnorm_bwd_sliced_kernel, dim3((unsigned)cdiv64(F, 64)), dim3(256), 0, (hipStream_t)stream, dy, x, w, b,
                       save_mean, save_invstd, dx, accumulate_dx, dw, db, Bn, F, gsz, n_inner, n_outer, pstride, gate, dgate);
  else
    hipLaunchKernelGGL(batchnorm_bwd_kernel, dim3((unsigned)cdiv64(F, 64)), dim3(64), 0, (hipStream_t)stream, dy, x, w, b, save_mean,
                       save_invstd, dx, accumulate_dx, dw, db, Bn, F, gsz, n_inner, n_outer, pstride, gate, dgate);
  DEMO_LAUNCH_CHECK("demo_batchnorm_bwd");
  return 0;
}

extern "C" int demo_atmoe_gate_softmax_bwd(const float* dgates, const float* gates, const float* q, const float* k, float* dq,
                                           float* dk, int32_t Bs, int32_t C, int32_t H, void* stream) {
  hipLaunchKernelGGL(atmoe_gate_softmax_bwd_kernel, dim3(Bs), dim3(256), 0, (hipStream_t)stream, dgates, gates, q, k, dq, dk, C, H);
  DEMO_LAUNCH_CHECK("demo_atmoe_gate_softmax_bwd");
  return 0;
}

extern "C" int demo_hdm_softmax_pool_bwd(const float* dpooled, const float* dpsum, const float* probs, const void* tokens, int32_t dt,
                                         const float* pooled, const float* psum, float* dS, int64_t ldS, float* dtokens, int32_t Bs,
                                         int32_t L, int32_t C, int32_t Hh, float p_drop, uint64_t seed, const uint64_t* seed_dev,
                                         void* stream) {
  DEMO_REQUIRE(Hh >= 1 && Hh <= HH_MAX && C <= 1024 && C % 4 == 0 && ldS >= 7 * Hh, "demo_hdm_softmax_pool_bwd: unsupported shape");
  const size_t shm = sizeof(float) * (4 * (size_t)Hh * C + 8 * HH_MAX);
  DEMO_REQUIRE(shm <= 160 * 1024, "demo_hdm_softmax_pool_bwd: needs %zu B of LDS", shm);
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = dt == DEMO_BF16
      ? hipFuncSetAttribute((const void*)hdm_softmax_pool_bwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)
      : hipFuncSetAttribute((const void*)hdm_softmax_pool_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  if (e != hipSuccess) DEMO_FAIL(-2, "demo_hdm_softmax_pool_bwd: cannot reserve %zu B of LDS: %s", shm, hipGetErrorString(e));
  dim3 grid(Bs, 3);
  if (dt == DEMO_BF16)
    hipLaunchKernelGGL(hdm_softmax_pool_bwd_kernel<bf16_t>, grid, dim3(512), shm, s, dpooled, dpsum, probs, (const bf16_t*)tokens, pooled,
                       psum, dS, ldS, dtokens, Bs, L, C, Hh, p_drop, seed, seed_dev);
  else
    hipLaunchKernelGGL(hdm_softmax_pool_bwd_kernel<float>, grid, dim3(512), shm, s, dpooled, dpsum, probs, (const float*)tokens, pooled,
                       psum, dS, ldS, dtokens, Bs, L, C, Hh, p_drop, seed, seed_dev);
  DEMO_LAUNCH_CHECK("demo_hdm_softmax_pool_bwd");
  return 0;
}

extern "C" int demo_hdm_vbias_bwd(const float* dy, const float* bias, int64_t bias_stride, const float* psum, float* dbias, float* dpsum,
                                  int32_t Bs, int32_t C, int32_t Hh, void* stream) {
  const unsigned nblk = 7 * 8 + (unsigned)cdiv64((int64_t)Bs * 7 * Hh, 4);      // (8 = SLABS of the kernel)
  hipLaunchKernelGGL(hdm_vbias_bwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, dy, bias, bias_stride, psum, dbias, dpsum, Bs, C, Hh);
  DEMO_LAUNCH_CHECK("demo_hdm_vbias_bwd");
  return 0;
}

extern "C" int demo_token0_grad(const float* dtokens, float* out, int64_t out_ld, int32_t Bn, int32_t L, int32_t C, void* stream) {
  hipLaunchKernelGGL(token0_grad_kernel, dim3(Bn), dim3(128), 0, (hipStream_t)stream, dtokens, out, out_ld, L, C);
  DEMO_LAUNCH_CHECK("demo_token0_grad");
  return 0;
}

extern "C" int demo_tokens_grad(const float* dtok, const float* dcls, int64_t cls_ld, const float* dmean, int64_t mean_ld,
                                int32_t keep_row0, void* out, int32_t dt, int32_t Bn, int32_t L, int32_t C, void* stream) {
  DEMO_REQUIRE(C % 4 == 0 && L >= 2, "demo_tokens_grad: bad shape");
  hipStream_t s = (hipStream_t)stream;
  if (dt == DEMO_BF16)
    hipLaunchKernelGGL(tokens_grad_kernel<bf16_t>, dim3(Bn, Bn >= 1024 ? 1 : 8), dim3(256), 0, s, dtok, dcls, cls_ld, dmean, mean_ld, keep_row0, (bf16_t*)out, L, C);
  else
    hipLaunchKernelGGL(tokens_grad_kernel<float>, dim3(Bn, Bn >= 1024 ? 1 : 8), dim3(256), 0, s, dtok, dcls, cls_ld, dmean, mean_ld, keep_row0, (float*)out, L, C);
  DEMO_LAUNCH_CHECK("demo_tokens_grad");
  return 0;
}
