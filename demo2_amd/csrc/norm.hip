// LayerNorm (one 64-lane wave per row, shuffle reductions, 16-B accesses) and BatchNorm1d.
#include "common.h"

namespace {

constexpr int LN_MAXV = 8;  // float4 per lane -> D <= 64*4*8 = 2048

template <typename TY>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, float eps, TY* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            int M, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (int64_t)row * D;
  const int nv = D >> 2;  // float4 count
  float4 v[LN_MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      v[i] = *reinterpret_cast<const float4*>(xr + c * 4);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      const float a = v[i].x - mean, bb = v[i].y - mean, cc = v[i].z - mean, dd = v[i].w - mean;
      q += (a * a + bb * bb) + (cc * cc + dd * dd);
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
  TY* yr = y + (int64_t)row * D;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      const float4 g = *reinterpret_cast<const float4*>(w + c * 4);
      const float4 be = *reinterpret_cast<const float4*>(b + c * 4);
      float4 o;
      o.x = (v[i].x - mean) * rstd * g.x + be.x;
      o.y = (v[i].y - mean) * rstd * g.y + be.y;
      o.z = (v[i].z - mean) * rstd * g.z + be.z;
      o.w = (v[i].w - mean) * rstd * g.w + be.w;
      io<TY>::st4(yr + c * 4, o);
    }
  }
}

// Residual add fused into the LayerNorm that follows it (bf16 residual stream of the bf16 mode): s = x (+ add), rounded to the
// stream's dtype TX FIRST (so that the saved stream, its statistics and y all describe the same values), xsum_out = s (may alias
// x), y = LN(s).  The branch GEMM (attention projection / fc2) then has a plain bf16 epilogue and the fp32 read-modify-write of the
// stream in its epilogue disappears: per pair 380 MB instead of 532 MB at 49536 x 768.
template <typename TX, typename TA, typename TY>
__global__ __launch_bounds__(256) void layernorm_add_fwd_kernel(const TX* __restrict__ x, const TA* __restrict__ add, TX* __restrict__ xsum_out,
                                                                const float* __restrict__ w, const float* __restrict__ b, float eps,
                                                                TY* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                                int M, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int64_t ro = (int64_t)row * D;
  const int nv = D >> 2;
  float4 v[LN_MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      float4 a = io<TX>::ld4(x + ro + c * 4);
      if (add) {
        const float4 t = io<TA>::ld4(add + ro + c * 4);
        a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
      }
      if (sizeof(TX) == 2) {                         // the stream's rounding, applied before anything is derived from the sum
        a.x = bf2f(f2bf(a.x)); a.y = bf2f(f2bf(a.y)); a.z = bf2f(f2bf(a.z)); a.w = bf2f(f2bf(a.w));
      }
      v[i] = a;
      if (xsum_out) io<TX>::st4(xsum_out + ro + c * 4, a);
      s += (a.x + a.y) + (a.z + a.w);
    }
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      const float a = v[i].x - mean, bb = v[i].y - mean, cc = v[i].z - mean, dd = v[i].w - mean;
      q += (a * a + bb * bb) + (cc * cc + dd * dd);
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      const float4 g = *reinterpret_cast<const float4*>(w + c * 4);
      const float4 be = *reinterpret_cast<const float4*>(b + c * 4);
      float4 o;
      o.x = (v[i].x - mean) * rstd * g.x + be.x;
      o.y = (v[i].y - mean) * rstd * g.y + be.y;
      o.z = (v[i].z - mean) * rstd * g.z + be.z;
      o.w = (v[i].w - mean) * rstd * g.w + be.w;
      io<TY>::st4(y + ro + c * 4, o);
    }
  }
}

// Backward, column-owner form (D <= 1024): thread t owns columns 4t..4t+3 and walks the workgroup's rows R at a time.
// All 3R loads of a chunk (x, dy, running dx) are issued back to back, so a wave keeps R*40 B per lane in flight with
// ~110 VGPRs at R = 4 (4 waves/SIMD; R = 8 at 2 waves/SIMD measures the same); measured loaded-HBM latency is ~5 us, i.e. this kernel lives on bytes in flight.  The two
// row sums cross the waves through a parity-double-buffered LDS slot (one barrier per chunk).  dw/db are per-thread
// column accumulators (8 registers, no LDS atomics) stored once per workgroup into the partial-row workspace.
// LPS ("low-precision stream", demo_layernorm_bwd_lp): the running residual gradient arrives as bf16 (dres_lp, may be null = none) and
// only the bf16 result is written -- 380 instead of 608 MB per launch on the trunk's shape, at the price of one bf16 rounding of
// the gradient stream per LayerNorm instead of one at the end.
template <typename TDY, int R, bool ACC, bool LPS = false, typename TX = float>
__global__ __launch_bounds__(256) void layernorm_bwd_cols_kernel(const TDY* __restrict__ dy, const TX* __restrict__ x,
                                                                 const float* __restrict__ w, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, float* __restrict__ dx,
                                                                 bf16_t* __restrict__ dx_lp, float* __restrict__ dw,
                                                                 float* __restrict__ db, float* __restrict__ part, int M, int D,
                                                                 int rows_per_block, const bf16_t* __restrict__ dres_lp = nullptr) {
  typedef typename io<TDY>::raw4 dyraw;
  __shared__ float red[2][R][2][4];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, nw = blockDim.x >> 6;
  const int c4 = t * 4;
  const bool colok = c4 < D;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 w4 = colok ? *reinterpret_cast<const float4*>(w + c4) : zero4;
  float4 adw = zero4, adb = zero4;
  const float inv_d = 1.f / (float)D;
  const int r0 = blockIdx.x * rows_per_block;
  int par = 0;
  for (int rc = 0; rc < rows_per_block; rc += R, par ^= 1) {
    const int rbase = r0 + rc;
    if (rbase >= M) break;
    float4 xv[R], pv[R];
    dyraw dv[R];
    float mu[R], rs[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = rbase + r;
      const bool rowok = row < M;                       // workgroup-uniform
      const bool ok = rowok && colok;
      const int64_t o = (int64_t)row * D;               // uniform row base (SGPR pair) + per-thread 32-bit column offset
      mu[r] = rowok ? mean[row] : 0.f;
      rs[r] = rowok ? rstd[row] : 0.f;
      xv[r] = ok ? io<TX>::ld4((x + o) + c4) : zero4;
      if (ok) dv[r] = io<TDY>::ld4_raw((dy + o) + c4); else dv[r] = dyraw();
      if constexpr (LPS) pv[r] = (dres_lp && ok) ? io<bf16_t>::ld4((dres_lp + o) + c4) : zero4;
      else pv[r] = (ACC && ok) ? *reinterpret_cast<const float4*>((dx + o) + c4) : zero4;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float4 d = io<TDY>::cvt4(dv[r]);
      xv[r] = make_float4((xv[r].x - mu[r]) * rs[r], (xv[r].y - mu[r]) * rs[r], (xv[r].z - mu[r]) * rs[r], (xv[r].w - mu[r]) * rs[r]);
      const float4 g = make_float4(d.x * w4.x, d.y * w4.y, d.z * w4.z, d.w * w4.w);
      float s1 = (g.x + g.y) + (g.z + g.w);
      float s2 = (g.x * xv[r].x + g.y * xv[r].y) + (g.z * xv[r].z + g.w * xv[r].w);
      s1 = wave_sum(s1);
      s2 = wave_sum(s2);
      if (lane == 0) {
        red[par][r][0][wv] = s1;
        red[par][r][1][wv] = s2;
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = rbase + r;
      float s1 = 0.f, s2 = 0.f;
      for (int k = 0; k < nw; ++k) {
        s1 += red[par][r][0][k];
        s2 += red[par][r][1][k];
      }
      s1 *= inv_d;
      s2 *= inv_d;
      const float4 d = io<TDY>::cvt4(dv[r]);
      const float4 xh = xv[r];
      adw.x += d.x * xh.x; adw.y += d.y * xh.y; adw.z += d.z * xh.z; adw.w += d.w * xh.w;
      adb.x += d.x; adb.y += d.y; adb.z += d.z; adb.w += d.w;
      if (row < M && colok) {
        float4 o;
        o.x = pv[r].x + rs[r] * (d.x * w4.x - s1 - xh.x * s2);
        o.y = pv[r].y + rs[r] * (d.y * w4.y - s1 - xh.y * s2);
        o.z = pv[r].z + rs[r] * (d.z * w4.z - s1 - xh.z * s2);
        o.w = pv[r].w + rs[r] * (d.w * w4.w - s1 - xh.w * s2);
        const int64_t oo = (int64_t)row * D;
        if constexpr (!LPS) *reinterpret_cast<float4*>((dx + oo) + c4) = o;
        if (dx_lp) io<bf16_t>::st4((dx_lp + oo) + c4, o);      // bf16 copy = A operand of the next backward GEMM
      }
    }
  }
  if (!colok) return;
  if (part) {            // two-stage: per-workgroup partial rows, summed by ln_bwd_reduce_kernel
    float* pr = part + (int64_t)blockIdx.x * 2 * D;
    *reinterpret_cast<float4*>(pr + c4) = adw;
    *reinterpret_cast<float4*>(pr + D + c4) = adb;
    return;
  }
  if (dw) { atomicAdd(&dw[c4], adw.x); atomicAdd(&dw[c4 + 1], adw.y); atomicAdd(&dw[c4 + 2], adw.z); atomicAdd(&dw[c4 + 3], adw.w); }
  if (db) { atomicAdd(&db[c4], adb.x); atomicAdd(&db[c4 + 1], adb.y); atomicAdd(&db[c4 + 2], adb.z); atomicAdd(&db[c4 + 3], adb.w); }
}

// All-bf16 stream form of the column-owner backward (bf16 residual stream AND bf16 gradient stream: dy, x, the running gradient and
// the result are all bf16): a thread owns EIGHT columns, so every access is 16 B, and walks R = 4 rows per step (R = 8 spills at three waves per SIMD) with all 3R loads in
// flight -- 192 B per thread against 96 B for the 4-column form on the same operands, which on this latency-bound kernel (see above)
// ran SLOWER with bf16 x than with fp32 x (106 vs 90 us at 49536 x 768) although it moves fewer bytes.  128-thread workgroups (D <= 1024).
__device__ __forceinline__ void bf16x8_unpack(const uint4 r, float (&f)[8]) {
  f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
  f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
  f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
  f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
}
template <int R>
__global__ __launch_bounds__(128, 3) void layernorm_bwd_lp8_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                                const float* __restrict__ w, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, const bf16_t* __restrict__ dres,
                                                                bf16_t* __restrict__ out, float* __restrict__ dw, float* __restrict__ db,
                                                                float* __restrict__ part, int M, int D, int rows_per_block) {
  __shared__ float red[2][R][2][2];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int c8 = t * 8;
  const bool colok = c8 < D;
  float w8[8], adw[8], adb[8];
  {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 a = colok ? *reinterpret_cast<const float4*>(w + c8) : z, b = colok ? *reinterpret_cast<const float4*>(w + c8 + 4) : z;
    w8[0] = a.x; w8[1] = a.y; w8[2] = a.z; w8[3] = a.w; w8[4] = b.x; w8[5] = b.y; w8[6] = b.z; w8[7] = b.w;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) adw[e] = adb[e] = 0.f;
  const float inv_d = 1.f / (float)D;
  const int r0 = blockIdx.x * rows_per_block;
  const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
  int par = 0;
  for (int rc = 0; rc < rows_per_block; rc += R, par ^= 1) {
    const int rbase = r0 + rc;
    if (rbase >= M) break;
    uint4 xr[R], dr[R], pr[R];
    float mu[R], rs[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = rbase + r;
      const bool rowok = row < M, ok = rowok && colok;      // rowok is workgroup-uniform
      const int64_t o = (int64_t)row * D;
      mu[r] = rowok ? mean[row] : 0.f;
      rs[r] = rowok ? rstd[row] : 0.f;
      xr[r] = ok ? *reinterpret_cast<const uint4*>((x + o) + c8) : z4;
      dr[r] = ok ? *reinterpret_cast<const uint4*>((dy + o) + c8) : z4;
      pr[r] = (dres && ok) ? *reinterpret_cast<const uint4*>((dres + o) + c8) : z4;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float xf[8], df[8];
      bf16x8_unpack(xr[r], xf);
      bf16x8_unpack(dr[r], df);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float g = df[e] * w8[e];
        s1 += g;
        s2 = fmaf(g, (xf[e] - mu[r]) * rs[r], s2);
      }
      s1 = wave_sum(s1);
      s2 = wave_sum(s2);
      if (lane == 0) {
        red[par][r][0][wv] = s1;
        red[par][r][1][wv] = s2;
      }
      __builtin_amdgcn_sched_barrier(0);            // one row at a time: the raw rows stay packed (12 registers each) until they are used
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = rbase + r;
      const float s1 = (red[par][r][0][0] + red[par][r][0][1]) * inv_d, s2 = (red[par][r][1][0] + red[par][r][1][1]) * inv_d;
      float xf[8], df[8], pf[8], o[8];
      bf16x8_unpack(xr[r], xf);
      bf16x8_unpack(dr[r], df);
      bf16x8_unpack(pr[r], pf);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (xf[e] - mu[r]) * rs[r];
        adw[e] = fmaf(df[e], xh, adw[e]);
        adb[e] += df[e];
        o[e] = pf[e] + rs[r] * (df[e] * w8[e] - s1 - xh * s2);
      }
      if (row < M && colok) {
        uint4 q;
        q.x = pack_bf16x2(o[0], o[1]); q.y = pack_bf16x2(o[2], o[3]); q.z = pack_bf16x2(o[4], o[5]); q.w = pack_bf16x2(o[6], o[7]);
        *reinterpret_cast<uint4*>((out + (int64_t)row * D) + c8) = q;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (!colok) return;
  if (part) {
    float* prow = part + (int64_t)blockIdx.x * 2 * D;
    *reinterpret_cast<float4*>(prow + c8) = make_float4(adw[0], adw[1], adw[2], adw[3]);
    *reinterpret_cast<float4*>(prow + c8 + 4) = make_float4(adw[4], adw[5], adw[6], adw[7]);
    *reinterpret_cast<float4*>(prow + D + c8) = make_float4(adb[0], adb[1], adb[2], adb[3]);
    *reinterpret_cast<float4*>(prow + D + c8 + 4) = make_float4(adb[4], adb[5], adb[6], adb[7]);
    return;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if (dw) atomicAdd(&dw[c8 + e], adw[e]);
    if (db) atomicAdd(&db[c8 + e], adb[e]);
  }
}

// LayerNorm forward on a bf16 stream, bf16 output (bf16 residual stream): HALF a wave per row, a lane owns 8-column chunks
// lane32 + 32 i (16-B loads and stores; D = 768: three chunks), two rows per wave, all of a lane's loads issued first.  The
// wave-per-row form above moves 24 B per lane and row on this operand and takes as long on 152 MB as on the fp32 stream's 229 MB.
constexpr int LN8_MAXC = 4;                        // D <= 32 * 8 * 4 = 1024
__device__ __forceinline__ float half_wave_sum(float v) {      // sum over the 32 lanes of this half of the wave
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return v + __shfl_xor(v, 16, 64);
}
__global__ __launch_bounds__(256) void layernorm_fwd_lp8_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ b, float eps, bf16_t* __restrict__ y,
                                                                float* __restrict__ mean_out, float* __restrict__ rstd_out, int M, int D) {
  const int l32 = threadIdx.x & 31;
  const int row = blockIdx.x * 8 + (threadIdx.x >> 5);
  const bool rowok = row < M;
  const int64_t ro = (int64_t)(rowok ? row : M - 1) * D;      // (clamped: the DPP reductions need every lane of the wave)
  const int nch = D >> 3;
  uint4 raw[LN8_MAXC];
#pragma unroll
  for (int i = 0; i < LN8_MAXC; ++i) {
    const int c = l32 + 32 * i;
    raw[i] = c < nch ? *reinterpret_cast<const uint4*>(x + ro + c * 8) : make_uint4(0u, 0u, 0u, 0u);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN8_MAXC; ++i) {
    float f[8];
    bf16x8_unpack(raw[i], f);
    s += ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
  }
  const float mean = half_wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN8_MAXC; ++i) {
    if (l32 + 32 * i < nch) {
      float f[8];
      bf16x8_unpack(raw[i], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) q = fmaf(f[e] - mean, f[e] - mean, q);
    }
  }
  const float rstd = rsqrtf(half_wave_sum(q) / (float)D + eps);
  if (!rowok) return;
  if (l32 == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
#pragma unroll
  for (int i = 0; i < LN8_MAXC; ++i) {
    const int c = l32 + 32 * i;
    if (c < nch) {
      float f[8];
      bf16x8_unpack(raw[i], f);
      const float4 g0 = *reinterpret_cast<const float4*>(w + c * 8), g1 = *reinterpret_cast<const float4*>(w + c * 8 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(b + c * 8), b1 = *reinterpret_cast<const float4*>(b + c * 8 + 4);
      const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = fmaf((f[e] - mean) * rstd, gg[e], bb[e]);
      uint4 r;
      r.x = pack_bf16x2(o[0], o[1]); r.y = pack_bf16x2(o[2], o[3]); r.z = pack_bf16x2(o[4], o[5]); r.w = pack_bf16x2(o[6], o[7]);
      *reinterpret_cast<uint4*>(y + (int64_t)row * D + c * 8) = r;
    }
  }
}

// LayerNorm backward on all-bf16 streams, HALF a wave per row (as the forward above): a lane owns the 8-column chunks lane32 + 32 i of
// its row (16-B accesses; D = 768: three), the row sums are two half-wave DPP reductions -- no LDS, no barrier inside the row loop
// (the 128-thread form above spends a workgroup barrier per four rows and leaves 32 of its 128 lanes idle at D = 768) -- and the next
// row of all three operands is requested before the current one is worked.  Column sums (dw, db) stay in registers over the
// workgroup's rows and are combined once at the end (half-waves by shuffle, waves through LDS) into one partial row per workgroup.
template <int NC>
__global__ __launch_bounds__(256, 2) void layernorm_bwd_hw_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                                  const float* __restrict__ w, const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, const bf16_t* __restrict__ dres,
                                                                  bf16_t* __restrict__ out, float* __restrict__ dw, float* __restrict__ db,
                                                                  float* __restrict__ part, int M, int D, int rows_per_block) {
  __shared__ float red[4][2][NC * 256];
  const int l32 = threadIdx.x & 31, hw = threadIdx.x >> 5, wave = threadIdx.x >> 6;
  const int nch = D >> 3;
  float wv[NC][8], adw[NC][8], adb[NC][8];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = l32 + 32 * i;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 a = c < nch ? *reinterpret_cast<const float4*>(w + c * 8) : z, b = c < nch ? *reinterpret_cast<const float4*>(w + c * 8 + 4) : z;
    wv[i][0] = a.x; wv[i][1] = a.y; wv[i][2] = a.z; wv[i][3] = a.w; wv[i][4] = b.x; wv[i][5] = b.y; wv[i][6] = b.z; wv[i][7] = b.w;
#pragma unroll
    for (int e = 0; e < 8; ++e) adw[i][e] = adb[i][e] = 0.f;
  }
  const float inv_d = 1.f / (float)D;
  const int r0 = blockIdx.x * rows_per_block;
  const int rend = min(M, r0 + rows_per_block);
  const int iters = (rend - r0 + 7) >> 3;            // workgroup-uniform: every half-wave runs every iteration (rows past the end masked)
  const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
  uint4 xn[NC], dn[NC], pn[NC];
  float mun = 0.f, rsn = 0.f;
  auto fetch = [&](int row) {
    const bool rowok = row < rend;
    const int64_t o = (int64_t)(rowok ? row : r0) * D;
    mun = mean[rowok ? row : r0];
    rsn = rowok ? rstd[row] : 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = l32 + 32 * i;
      const bool ok = rowok && c < nch;
      xn[i] = ok ? *reinterpret_cast<const uint4*>(x + o + c * 8) : z4;
      dn[i] = ok ? *reinterpret_cast<const uint4*>(dy + o + c * 8) : z4;
      pn[i] = (ok && dres) ? *reinterpret_cast<const uint4*>(dres + o + c * 8) : z4;
    }
  };
  fetch(r0 + hw);
  for (int it = 0; it < iters; ++it) {
    const int row = r0 + hw + it * 8;
    uint4 xr[NC], dr[NC], pr[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) { xr[i] = xn[i]; dr[i] = dn[i]; pr[i] = pn[i]; }
    const float mu = mun, rs = rsn;
    if (it + 1 < iters) fetch(row + 8);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      float xf[8], df[8];
      bf16x8_unpack(xr[i], xf);
      bf16x8_unpack(dr[i], df);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float g = df[e] * wv[i][e];
        s1 += g;
        s2 = fmaf(g, (xf[e] - mu) * rs, s2);
      }
    }
    s1 = half_wave_sum(s1) * inv_d;
    s2 = half_wave_sum(s2) * inv_d;
    const bool rowok = row < rend;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      float xf[8], df[8], pf[8], o[8];
      bf16x8_unpack(xr[i], xf);
      bf16x8_unpack(dr[i], df);
      bf16x8_unpack(pr[i], pf);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (xf[e] - mu) * rs;           // (masked rows: dy = 0 and rstd = 0, so nothing is added)
        adw[i][e] = fmaf(df[e], xh, adw[i][e]);
        adb[i][e] += df[e];
        o[e] = pf[e] + rs * (df[e] * wv[i][e] - s1 - xh * s2);
      }
      const int c = l32 + 32 * i;
      if (rowok && c < nch) {
        uint4 q;
        q.x = pack_bf16x2(o[0], o[1]); q.y = pack_bf16x2(o[2], o[3]); q.z = pack_bf16x2(o[4], o[5]); q.w = pack_bf16x2(o[6], o[7]);
        *reinterpret_cast<uint4*>(out + (int64_t)row * D + c * 8) = q;
      }
    }
  }
  if (!dw && !db) return;
  // column sums: the two half-waves of a wave own the same columns
#pragma unroll
  for (int i = 0; i < NC; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      adw[i][e] += __shfl_xor(adw[i][e], 32, 64);
      adb[i][e] += __shfl_xor(adb[i][e], 32, 64);
    }
  if ((threadIdx.x & 63) < 32) {
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = l32 + 32 * i;
      *reinterpret_cast<float4*>(&red[wave][0][c * 8]) = make_float4(adw[i][0], adw[i][1], adw[i][2], adw[i][3]);
      *reinterpret_cast<float4*>(&red[wave][0][c * 8 + 4]) = make_float4(adw[i][4], adw[i][5], adw[i][6], adw[i][7]);
      *reinterpret_cast<float4*>(&red[wave][1][c * 8]) = make_float4(adb[i][0], adb[i][1], adb[i][2], adb[i][3]);
      *reinterpret_cast<float4*>(&red[wave][1][c * 8 + 4]) = make_float4(adb[i][4], adb[i][5], adb[i][6], adb[i][7]);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * D; c += 256) {
    const int k = c >= D ? 1 : 0, col = c - k * D;
    const float v = (red[0][k][col] + red[1][k][col]) + (red[2][k][col] + red[3][k][col]);
    if (part) part[(int64_t)blockIdx.x * 2 * D + c] = v;
    else if (k == 0) { if (dw) atomicAdd(&dw[col], v); }
    else if (db) atomicAdd(&db[col], v);
  }
}

// LayerNorm backward on all-bf16 streams, ONE WAVE per row, D = 512 * N8 + 256 * N4 (768: N8 = N4 = 1): lane l owns columns
// [8l, 8l + 8) of every 512-column block (a 16-B access) and [512 N8 + 4l, + 4) of the 256-column tail (an 8-B access).  12 columns per
// lane at D = 768 instead of the half-wave form's 24: 24 column accumulators + 12 weights per lane leave room for TWO rows of prefetch
// and four waves per SIMD, which the half-wave form (238 VGPRs, one row ahead, two waves per SIMD) does not have.
template <int N8, int N4>
__global__ __launch_bounds__(256, 3) void layernorm_bwd_w_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                                 const float* __restrict__ w, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, const bf16_t* __restrict__ dres,
                                                                 bf16_t* __restrict__ out, float* __restrict__ dw, float* __restrict__ db,
                                                                 float* __restrict__ part, int M, int rows_per_block) {
  constexpr int D = 512 * N8 + 256 * N4, NCOL = 8 * N8 + 4 * N4, T0 = 512 * N8;
  __shared__ float red[4][2][D];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float wv[NCOL], adw[NCOL], adb[NCOL];
#pragma unroll
  for (int i = 0; i < N8; ++i) {
    const float4 a = *reinterpret_cast<const float4*>(w + 512 * i + 8 * lane), b = *reinterpret_cast<const float4*>(w + 512 * i + 8 * lane + 4);
    wv[8 * i + 0] = a.x; wv[8 * i + 1] = a.y; wv[8 * i + 2] = a.z; wv[8 * i + 3] = a.w;
    wv[8 * i + 4] = b.x; wv[8 * i + 5] = b.y; wv[8 * i + 6] = b.z; wv[8 * i + 7] = b.w;
  }
  if constexpr (N4) {
    const float4 a = *reinterpret_cast<const float4*>(w + T0 + 4 * lane);
    wv[8 * N8 + 0] = a.x; wv[8 * N8 + 1] = a.y; wv[8 * N8 + 2] = a.z; wv[8 * N8 + 3] = a.w;
  }
#pragma unroll
  for (int e = 0; e < NCOL; ++e) adw[e] = adb[e] = 0.f;
  struct Row {                                       // one row of the three operands, still packed
    uint4 x8[N8 ? N8 : 1], d8[N8 ? N8 : 1], p8[N8 ? N8 : 1];
    uint2 x4, d4, p4;
    float mu, rs;
  };
  const float inv_d = 1.f / (float)D;
  const int r0 = blockIdx.x * rows_per_block;
  const int rend = min(M, r0 + rows_per_block);
  const int iters = (rend - r0 + 3) >> 2;            // workgroup-uniform; a wave takes rows r0 + wave + 4 j
  auto fetch = [&](Row& r, int row) {
    const bool ok = row < rend;
    const int64_t o = (int64_t)(ok ? row : r0) * D;
    r.mu = mean[ok ? row : r0];
    r.rs = ok ? rstd[row] : 0.f;
    const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
    const uint2 z2 = make_uint2(0u, 0u);
#pragma unroll
    for (int i = 0; i < N8; ++i) {
      r.x8[i] = ok ? *reinterpret_cast<const uint4*>(x + o + 512 * i + 8 * lane) : z4;
      r.d8[i] = ok ? *reinterpret_cast<const uint4*>(dy + o + 512 * i + 8 * lane) : z4;
      r.p8[i] = (ok && dres) ? *reinterpret_cast<const uint4*>(dres + o + 512 * i + 8 * lane) : z4;
    }
    if constexpr (N4) {
      r.x4 = ok ? *reinterpret_cast<const uint2*>(x + o + T0 + 4 * lane) : z2;
      r.d4 = ok ? *reinterpret_cast<const uint2*>(dy + o + T0 + 4 * lane) : z2;
      r.p4 = (ok && dres) ? *reinterpret_cast<const uint2*>(dres + o + T0 + 4 * lane) : z2;
    }
  };
  auto unpack = [&](const Row& r, float (&xf)[NCOL], float (&df)[NCOL]) {
#pragma unroll
    for (int i = 0; i < N8; ++i) {
      float a[8], b[8];
      bf16x8_unpack(r.x8[i], a);
      bf16x8_unpack(r.d8[i], b);
#pragma unroll
      for (int e = 0; e < 8; ++e) { xf[8 * i + e] = a[e]; df[8 * i + e] = b[e]; }
    }
    if constexpr (N4) {
      xf[8 * N8 + 0] = __uint_as_float(r.x4.x << 16); xf[8 * N8 + 1] = __uint_as_float(r.x4.x & 0xffff0000u);
      xf[8 * N8 + 2] = __uint_as_float(r.x4.y << 16); xf[8 * N8 + 3] = __uint_as_float(r.x4.y & 0xffff0000u);
      df[8 * N8 + 0] = __uint_as_float(r.d4.x << 16); df[8 * N8 + 1] = __uint_as_float(r.d4.x & 0xffff0000u);
      df[8 * N8 + 2] = __uint_as_float(r.d4.y << 16); df[8 * N8 + 3] = __uint_as_float(r.d4.y & 0xffff0000u);
    }
  };
  auto work = [&](const Row& r, int row) {
    float xf[NCOL], df[NCOL];
    unpack(r, xf, df);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < NCOL; ++e) {
      xf[e] = (xf[e] - r.mu) * r.rs;                  // x-hat (masked rows: dy = 0, rstd = 0)
      const float g = df[e] * wv[e];
      s1 += g;
      s2 = fmaf(g, xf[e], s2);
    }
    s1 = wave_sum(s1) * inv_d;
    s2 = wave_sum(s2) * inv_d;
    float o[NCOL];
#pragma unroll
    for (int e = 0; e < NCOL; ++e) {
      adw[e] = fmaf(df[e], xf[e], adw[e]);
      adb[e] += df[e];
      o[e] = r.rs * (df[e] * wv[e] - s1 - xf[e] * s2);
    }
    if (row < rend) {
#pragma unroll
      for (int i = 0; i < N8; ++i) {
        float pf[8];
        bf16x8_unpack(r.p8[i], pf);
        uint4 q;
        q.x = pack_bf16x2(o[8 * i + 0] + pf[0], o[8 * i + 1] + pf[1]); q.y = pack_bf16x2(o[8 * i + 2] + pf[2], o[8 * i + 3] + pf[3]);
        q.z = pack_bf16x2(o[8 * i + 4] + pf[4], o[8 * i + 5] + pf[5]); q.w = pack_bf16x2(o[8 * i + 6] + pf[6], o[8 * i + 7] + pf[7]);
        *reinterpret_cast<uint4*>(out + (int64_t)row * D + 512 * i + 8 * lane) = q;
      }
      if constexpr (N4) {
        const float p0 = __uint_as_float(r.p4.x << 16), p1 = __uint_as_float(r.p4.x & 0xffff0000u);
        const float p2 = __uint_as_float(r.p4.y << 16), p3 = __uint_as_float(r.p4.y & 0xffff0000u);
        uint2 q;
        q.x = pack_bf16x2(o[8 * N8 + 0] + p0, o[8 * N8 + 1] + p1);
        q.y = pack_bf16x2(o[8 * N8 + 2] + p2, o[8 * N8 + 3] + p3);
        *reinterpret_cast<uint2*>(out + (int64_t)row * D + T0 + 4 * lane) = q;
      }
    }
  };
  // two rows ahead: ra is worked while rb and the next ra are in flight
  Row ra, rb;
  fetch(ra, r0 + wave);
  fetch(rb, r0 + wave + 4);
  for (int it = 0; it < iters; it += 2) {
    const int row = r0 + wave + 4 * it;
    Row rc;
    fetch(rc, row + 8);
    work(ra, row);
    ra = rc;
    if (it + 1 < iters) {
      Row rd;
      fetch(rd, row + 12);
      work(rb, row + 4);
      rb = rd;
    }
  }
  if (!dw && !db) return;
  // column sums: waves through LDS, one partial row per workgroup
#pragma unroll
  for (int i = 0; i < N8; ++i) {
    *reinterpret_cast<float4*>(&red[wave][0][512 * i + 8 * lane]) = make_float4(adw[8 * i], adw[8 * i + 1], adw[8 * i + 2], adw[8 * i + 3]);
    *reinterpret_cast<float4*>(&red[wave][0][512 * i + 8 * lane + 4]) = make_float4(adw[8 * i + 4], adw[8 * i + 5], adw[8 * i + 6], adw[8 * i + 7]);
    *reinterpret_cast<float4*>(&red[wave][1][512 * i + 8 * lane]) = make_float4(adb[8 * i], adb[8 * i + 1], adb[8 * i + 2], adb[8 * i + 3]);
    *reinterpret_cast<float4*>(&red[wave][1][512 * i + 8 * lane + 4]) = make_float4(adb[8 * i + 4], adb[8 * i + 5], adb[8 * i + 6], adb[8 * i + 7]);
  }
  if constexpr (N4) {
    *reinterpret_cast<float4*>(&red[wave][0][T0 + 4 * lane]) = make_float4(adw[8 * N8], adw[8 * N8 + 1], adw[8 * N8 + 2], adw[8 * N8 + 3]);
    *reinterpret_cast<float4*>(&red[wave][1][T0 + 4 * lane]) = make_float4(adb[8 * N8], adb[8 * N8 + 1], adb[8 * N8 + 2], adb[8 * N8 + 3]);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * D; c += 256) {
    const int k = c >= D ? 1 : 0, col = c - k * D;
    const float v = (red[0][k][col] + red[1][k][col]) + (red[2][k][col] + red[3][k][col]);
    if (part) part[(int64_t)blockIdx.x * 2 * D + c] = v;
    else if (k == 0) { if (dw) atomicAdd(&dw[col], v); }
    else if (db) atomicAdd(&db[col], v);
  }
}

// Backward, row-owner form (any D up to 2048): one wave per row for dx; dw/db accumulated per block in LDS then atomically
// added into fp32 dw/db (D floats each; D*gridDim atomics spread over D addresses).
// NV = float4 per lane (compile time: D = 768 -> 3), so that only the registers a row needs are allocated and 8 waves per
// SIMD hide the HBM latency of this pure streaming kernel.
template <typename TDY, int NV, bool FULL, bool ACC>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const TDY* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ w, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ dx,
                                                            bf16_t* __restrict__ dx_lp, float* __restrict__ dw,
                                                            float* __restrict__ db, float* __restrict__ part, int M, int D,
                                                            int rows_per_block) {
  // FULL: D == NV*256, every lane owns NV float4 of the row and the loads below carry no bounds branches -- with a
  // branch per float4 the compiler waits for each chunk's loads before issuing the next chunk's (3 latencies per row).
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [2][D]
  float* s_dw = sm;
  float* s_db = sm + D;
  for (int i = threadIdx.x; i < 2 * D; i += blockDim.x) sm[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nv = D >> 2;
  const int r0 = blockIdx.x * rows_per_block;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 adw[NV], adb[NV], w4[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    adw[i] = adb[i] = zero4;
    const int c = lane + i * 64;
    w4[i] = (FULL || c < nv) ? *reinterpret_cast<const float4*>(w + c * 4) : zero4;
  }

  // One row ahead: the next row's loads are issued before this row's arithmetic and stores, so a wave always has a row
  // of loads in flight and never waits for its own stores to be acknowledged before it can see new data (loads and
  // stores share the in-order vmcnt counter on gfx9).
  typedef typename io<TDY>::raw4 dyraw;
  float4 nx[NV], np[NV];
  dyraw nd[NV];
  float nmu = 0.f, nrs = 0.f;
  auto fetch = [&](int row) {
    nmu = mean[row];
    nrs = rstd[row];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + i * 64;
      const bool in = FULL || c < nv;
      const int64_t o = (int64_t)row * D + c * 4;
      nx[i] = in ? *reinterpret_cast<const float4*>(x + o) : zero4;
      if (in) nd[i] = io<TDY>::ld4_raw(dy + o); else nd[i] = dyraw();
      np[i] = (ACC && in) ? *reinterpret_cast<const float4*>(dx + o) : zero4;
    }
  };
  const int rend = min(M, r0 + rows_per_block);
  if (r0 + wv < rend) fetch(r0 + wv);
  for (int row = r0 + wv; row < rend; row += 4) {
    const float mu = nmu, rs = nrs;
    float* dxr = dx + (int64_t)row * D;
    float4 xh[NV], g[NV], prev[NV], d[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      xh[i] = nx[i];
      d[i] = io<TDY>::cvt4(nd[i]);
      prev[i] = np[i];
    }
    if (row + 4 < rend) fetch(row + 4);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      xh[i] = make_float4((xh[i].x - mu) * rs, (xh[i].y - mu) * rs, (xh[i].z - mu) * rs, (xh[i].w - mu) * rs);
      g[i] = make_float4(d[i].x * w4[i].x, d[i].y * w4[i].y, d[i].z * w4[i].z, d[i].w * w4[i].w);
      s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
      s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
      adw[i].x += d[i].x * xh[i].x; adw[i].y += d[i].y * xh[i].y; adw[i].z += d[i].z * xh[i].z; adw[i].w += d[i].w * xh[i].w;
      adb[i].x += d[i].x; adb[i].y += d[i].y; adb[i].z += d[i].z; adb[i].w += d[i].w;
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + i * 64;
      if (FULL || c < nv) {
        float4 o;
        o.x = prev[i].x + rs * (g[i].x - s1 - xh[i].x * s2);
        o.y = prev[i].y + rs * (g[i].y - s1 - xh[i].y * s2);
        o.z = prev[i].z + rs * (g[i].z - s1 - xh[i].z * s2);
        o.w = prev[i].w + rs * (g[i].w - s1 - xh[i].w * s2);
        *reinterpret_cast<float4*>(dxr + c * 4) = o;
        if (dx_lp) io<bf16_t>::st4(dx_lp + (int64_t)row * D + c * 4, o);   // bf16 copy = A operand of the next backward GEMM
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      atomicAdd(&s_dw[c * 4 + 0], adw[i].x); atomicAdd(&s_dw[c * 4 + 1], adw[i].y);
      atomicAdd(&s_dw[c * 4 + 2], adw[i].z); atomicAdd(&s_dw[c * 4 + 3], adw[i].w);
      atomicAdd(&s_db[c * 4 + 0], adb[i].x); atomicAdd(&s_db[c * 4 + 1], adb[i].y);
      atomicAdd(&s_db[c * 4 + 2], adb[i].z); atomicAdd(&s_db[c * 4 + 3], adb[i].w);
    }
  }
  __syncthreads();
  if (part) {            // two-stage: per-workgroup partials, summed by ln_bwd_reduce_kernel (no contended global atomics)
    float* pr = part + (int64_t)blockIdx.x * 2 * D;
    for (int i = threadIdx.x; i < 2 * D; i += blockDim.x) pr[i] = sm[i];
    return;
  }
  for (int i = threadIdx.x; i < D; i += blockDim.x) {
    if (dw) atomicAdd(&dw[i], s_dw[i]);
    if (db) atomicAdd(&db[i], s_db[i]);
  }
}

__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ part, int nblk, int D, float* __restrict__ dw,
                                                            float* __restrict__ db) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;      // column of the [2][D] partial rows
  if (c >= 2 * D) return;
  const int per = (nblk + gridDim.y - 1) / gridDim.y;       // grid.y slices of the partial rows, combined with a few atomics
  const int b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = b0;
  for (; b + 4 <= b1; b += 4) {
    s0 += part[(int64_t)b * 2 * D + c];
    s1 += part[(int64_t)(b + 1) * 2 * D + c];
    s2 += part[(int64_t)(b + 2) * 2 * D + c];
    s3 += part[(int64_t)(b + 3) * 2 * D + c];
  }
  for (; b < b1; ++b) s0 += part[(int64_t)b * 2 * D + c];
  const float s = (s0 + s1) + (s2 + s3);
  if (c < D) { if (dw) atomicAdd(&dw[c], s); }
  else if (db) atomicAdd(&db[c - D], s);
}

// BatchNorm1d: one thread per feature, coalesced across features; Bn <= a few thousand rows.
template <typename TY>
__global__ __launch_bounds__(256) void batchnorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, float* __restrict__ rmean,
                                                            float* __restrict__ rvar, int training, float momentum,
                                                            float eps, TY* __restrict__ y, float* __restrict__ save_mean,
                                                            float* __restrict__ save_invstd, int Bn, int F, int gsz,
                                                            int n_inner, int n_outer, int64_t pstride, int64_t sstride,
                                                            const float* __restrict__ gate) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  int64_t pi = f, si = f;
  int gi = 0, go = 0;
  if (gsz > 0) {
    const int g = f / gsz, k = f - g * gsz;
    gi = g % n_inner; go = g / n_inner;
    const int64_t blk = (int64_t)gi * n_outer + go;
    pi = blk * pstride + k;
    si = blk * sstride + k;
  }
  float mean, invstd;
  if (training) {
    float s = 0.f;
    for (int r = 0; r < Bn; ++r) s += x[(int64_t)r * F + f];
    mean = s / (float)Bn;
    float q = 0.f;
    for (int r = 0; r < Bn; ++r) {
      const float d = x[(int64_t)r * F + f] - mean;
      q += d * d;
    }
    const float var = q / (float)Bn;
    invstd = rsqrtf(var + eps);
    rmean[si] = (1.f - momentum) * rmean[si] + momentum * mean;
    rvar[si] = (1.f - momentum) * rvar[si] + momentum * (var * (float)Bn / (float)(Bn - 1));
    if (save_mean) save_mean[f] = mean;
    if (save_invstd) save_invstd[f] = invstd;
  } else {
    mean = rmean[si];
    invstd = rsqrtf(rvar[si] + eps);
  }
  const float sc = invstd * w[pi], sh = b[pi] - mean * sc;
  for (int r = 0; r < Bn; ++r) {
    float v = x[(int64_t)r * F + f] * sc + sh;
    if (gate) v *= gate[((int64_t)r * n_inner + gi) * n_outer + go];
    io<TY>::st(y + (int64_t)r * F + f, v);
  }
}

// Sliced form for Bn <= 128: 64 features x 4 row slices per workgroup, the slice's rows live in registers (one memory
// latency for the whole column instead of three serial passes of Bn dependent loads), sums cross the waves through LDS.
template <typename TY>
__global__ __launch_bounds__(256) void batchnorm_fwd_sliced_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                   const float* __restrict__ b, float* __restrict__ rmean,
                                                                   float* __restrict__ rvar, int training, float momentum, float eps,
                                                                   TY* __restrict__ y, float* __restrict__ save_mean,
                                                                   float* __restrict__ save_invstd, int Bn, int F, int gsz,
                                                                   int n_inner, int n_outer, int64_t pstride, int64_t sstride,
                                                                   const float* __restrict__ gate) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int f = blockIdx.x * 64 + lane;
  const bool fok = f < F;
  const int fc = fok ? f : F - 1;
  int64_t pi = fc, si = fc;
  int gi = 0, go = 0;
  if (gsz > 0) {
    const int g = fc / gsz, k = fc - g * gsz;
    gi = g % n_inner; go = g / n_inner;
    const int64_t blk = (int64_t)gi * n_outer + go;
    pi = blk * pstride + k;
    si = blk * sstride + k;
  }
  float xr[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int r = wv + 4 * i;
    xr[i] = (fok && r < Bn) ? x[(int64_t)r * F + f] : 0.f;
  }
  float mean, invstd;
  if (training) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += xr[i];
    red[wv][lane] = s;
    __syncthreads();
    mean = ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) / (float)Bn;
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const float dlt = xr[i] - mean;
      if (wv + 4 * i < Bn) q += dlt * dlt;
    }
    red[wv][lane] = q;
    __syncthreads();
    const float var = ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) / (float)Bn;
    invstd = rsqrtf(var + eps);
    if (wv == 0 && fok) {
      rmean[si] = (1.f - momentum) * rmean[si] + momentum * mean;
      rvar[si] = (1.f - momentum) * rvar[si] + momentum * (var * (float)Bn / (float)(Bn - 1));
      if (save_mean) save_mean[f] = mean;
      if (save_invstd) save_invstd[f] = invstd;
    }
  } else {
    mean = rmean[si];
    invstd = rsqrtf(rvar[si] + eps);
  }
  const float sc = invstd * w[pi], sh = b[pi] - mean * sc;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int r = wv + 4 * i;
    if (fok && r < Bn) {
      float v = xr[i] * sc + sh;
      if (gate) v *= gate[((int64_t)r * n_inner + gi) * n_outer + go];
      io<TY>::st(y + (int64_t)r * F + f, v);
    }
  }
}

}  // namespace

extern "C" int demo_layernorm_fwd(const float* x, const float* w, const float* b, float eps, void* y, int32_t dtY,
                                  float* mean, float* rstd, int32_t M, int32_t D, void* stream) {
  DEMO_REQUIRE(D % 4 == 0 && D <= 64 * 4 * LN_MAXV, "demo_layernorm_fwd: D=%d must be a multiple of 4 and <= %d", D, 64 * 4 * LN_MAXV);
  DEMO_REQUIRE(M > 0, "demo_layernorm_fwd: M=%d", M);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)cdiv64(M, 4));
  if (dtY == DEMO_BF16)
    hipLaunchKernelGGL(layernorm_fwd_kernel<bf16_t>, grid, dim3(256), 0, s, x, w, b, eps, (bf16_t*)y, mean, rstd, M, D);
  else
    hipLaunchKernelGGL(layernorm_fwd_kernel<float>, grid, dim3(256), 0, s, x, w, b, eps, (float*)y, mean, rstd, M, D);
  DEMO_LAUNCH_CHECK("demo_layernorm_fwd");
  return 0;
}

static int g_ln_lp8 = 1;      // tuning hook: 0 = the wave-per-row / 4-column kernels also for all-bf16 streams, 2 = backward in the 128-thread 8-column form, 3 = backward in the half-wave form
extern "C" int demo_layernorm_set_lp8(int on) { g_ln_lp8 = on < 0 ? 0 : on; return 0; }

extern "C" int demo_layernorm_add_fwd(const void* x, int32_t dtX, const void* add, int32_t dtAdd, void* xsum_out, const float* w, const float* b,
                                      float eps, void* y, int32_t dtY, float* mean, float* rstd, int32_t M, int32_t D, void* stream) {
  DEMO_REQUIRE(D % 4 == 0 && D <= 64 * 4 * LN_MAXV, "demo_layernorm_add_fwd: D=%d must be a multiple of 4 and <= %d", D, 64 * 4 * LN_MAXV);
  DEMO_REQUIRE(x && w && b && y && M > 0, "demo_layernorm_add_fwd: null operand");
  DEMO_REQUIRE(dtY == DEMO_BF16 && (!add || dtAdd == DEMO_BF16), "demo_layernorm_add_fwd: y and add are bf16 (the bf16 compute mode's activations)");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)cdiv64(M, 4));
  if (dtX == DEMO_BF16 && !add && !xsum_out && D % 8 == 0 && D <= 32 * 8 * LN8_MAXC && g_ln_lp8)       // plain LayerNorm of a bf16 stream
    hipLaunchKernelGGL(layernorm_fwd_lp8_kernel, dim3((unsigned)cdiv64(M, 8)), dim3(256), 0, s, (const bf16_t*)x, w, b, eps, (bf16_t*)y, mean, rstd,
                       M, D);
  else if (dtX == DEMO_BF16)
    hipLaunchKernelGGL((layernorm_add_fwd_kernel<bf16_t, bf16_t, bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)add, (bf16_t*)xsum_out,
                       w, b, eps, (bf16_t*)y, mean, rstd, M, D);
  else
    hipLaunchKernelGGL((layernorm_add_fwd_kernel<float, bf16_t, bf16_t>), grid, dim3(256), 0, s, (const float*)x, (const bf16_t*)add, (float*)xsum_out,
                       w, b, eps, (bf16_t*)y, mean, rstd, M, D);
  DEMO_LAUNCH_CHECK("demo_layernorm_add_fwd");
  return 0;
}

extern "C" int demo_layernorm_bwd(const void* dy, int32_t dtDy, const float* x, const float* w, const float* mean,
                                  const float* rstd, float* dx, int32_t accumulate_dx, void* dx_bf16, float* dw, float* db,
                                  float* workspace, int64_t workspace_elems, int32_t M, int32_t D, void* stream) {
  DEMO_REQUIRE(D % 4 == 0 && D <= 64 * 4 * LN_MAXV, "demo_layernorm_bwd: D=%d unsupported", D);
  hipStream_t s = (hipStream_t)stream;
  const int rows_per_block = M >= 16384 ? 32 : 16;      // ~1500 workgroups at 3B*L = 49536 rows; fewer partial dw/db rows for the reduce kernel
  const int nblk = (int)cdiv64(M, rows_per_block);
  dim3 grid((unsigned)nblk);
  const size_t shm = 2 * (size_t)D * sizeof(float);
  // thousands of workgroups atomically adding into the same 2*D words serialise; with a workspace each workgroup stores
  // its partial dw/db row and a second tiny kernel sums the rows
  float* part = (workspace && workspace_elems >= (int64_t)nblk * 2 * D && (dw || db)) ? workspace : nullptr;
  if (D <= 1024) {
    const dim3 blk((unsigned)(cdiv64(D, 256) * 64));
#define LN_BWD_COLS(T_, ACC_)                                                                                                   \
  hipLaunchKernelGGL((layernorm_bwd_cols_kernel<T_, 4, ACC_>), grid, blk, 0, s, (const T_*)dy, x, w, mean, rstd, dx,            \
                     (bf16_t*)dx_bf16, dw, db, part, M, D, rows_per_block)
    if (dtDy == DEMO_BF16) { if (accumulate_dx) LN_BWD_COLS(bf16_t, true); else LN_BWD_COLS(bf16_t, false); }
    else { if (accumulate_dx) LN_BWD_COLS(float, true); else LN_BWD_COLS(float, false); }
#undef LN_BWD_COLS
    DEMO_LAUNCH_CHECK("demo_layernorm_bwd");
    if (part) {
      hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((unsigned)cdiv64(2 * D, 256), (unsigned)(nblk >= 64 ? 32 : 1)), dim3(256), 0, s,
                         part, nblk, D, dw, db);
      DEMO_LAUNCH_CHECK("demo_layernorm_bwd(reduce)");
    }
    return 0;
  }
  const int nvl = (int)cdiv64(D, 256);      // float4 per lane actually needed
#define LN_BWD_LAUNCH2(NV_, FULL_, ACC_)                                                                                        \
  do {                                                                                                                          \
    if (dtDy == DEMO_BF16)                                                                                                      \
      hipLaunchKernelGGL((layernorm_bwd_kernel<bf16_t, NV_, FULL_, ACC_>), grid, dim3(256), shm, s, (const bf16_t*)dy, x, w,    \
                         mean, rstd, dx, (bf16_t*)dx_bf16, dw, db, part, M, D, rows_per_block);                                 \
    else                                                                                                                        \
      hipLaunchKernelGGL((layernorm_bwd_kernel<float, NV_, FULL_, ACC_>), grid, dim3(256), shm, s, (const float*)dy, x, w,      \
                         mean, rstd, dx, (bf16_t*)dx_bf16, dw, db, part, M, D, rows_per_block);                                 \
  } while (0)
#define LN_BWD_LAUNCH(NV_)                                                                                                      \
  do {                                                                                                                          \
    const bool full = D == NV_ * 256;                                                                                           \
    if (full && accumulate_dx) LN_BWD_LAUNCH2(NV_, true, true);                                                                 \
    else if (full) LN_BWD_LAUNCH2(NV_, true, false);                                                                            \
    else if (accumulate_dx) LN_BWD_LAUNCH2(NV_, false, true);                                                                   \
    else LN_BWD_LAUNCH2(NV_, false, false);                                                                                     \
  } while (0)
  if (nvl <= 1) LN_BWD_LAUNCH(1);
  else if (nvl <= 2) LN_BWD_LAUNCH(2);
  else if (nvl <= 3) LN_BWD_LAUNCH(3);
  else if (nvl <= 4) LN_BWD_LAUNCH(4);
  else if (nvl <= 6) LN_BWD_LAUNCH(6);
  else LN_BWD_LAUNCH(8);
#undef LN_BWD_LAUNCH2
#undef LN_BWD_LAUNCH
  DEMO_LAUNCH_CHECK("demo_layernorm_bwd");
  if (part) {
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((unsigned)cdiv64(2 * D, 256), (unsigned)(nblk >= 64 ? 32 : 1)), dim3(256), 0, s, part,
                       nblk, D, dw, db);
    DEMO_LAUNCH_CHECK("demo_layernorm_bwd(reduce)");
  }
  return 0;
}

extern "C" int demo_layernorm_bwd_lp(const void* dy, int32_t dtDy, const void* x, int32_t dtX, const float* w, const float* mean, const float* rstd,
                                     const void* dres_bf16, void* dx_bf16, float* dw, float* db, float* workspace, int64_t workspace_elems,
                                     int32_t M, int32_t D, void* stream) {
  DEMO_REQUIRE(D % 4 == 0 && D <= 1024, "demo_layernorm_bwd_lp: D=%d unsupported (<= 1024)", D);
  DEMO_REQUIRE(dy && x && w && mean && rstd && dx_bf16 && M > 0, "demo_layernorm_bwd_lp: null operand");
  hipStream_t s = (hipStream_t)stream;
  const bool w_form = dtDy == DEMO_BF16 && dtX == DEMO_BF16 && (D == 768 || D == 512) && g_ln_lp8 == 1;      // wave per row, two rows ahead
  const bool hw_form = dtDy == DEMO_BF16 && dtX == DEMO_BF16 && D % 8 == 0 && D <= 768 && (g_ln_lp8 == 3 || (g_ln_lp8 == 1 && !w_form));      // all-bf16 streams: half a wave per row
  // half-wave form: two 256-thread workgroups fit a CU (238 VGPRs), so the grid is ONE round of 512 workgroups where M allows (774 workgroups
  // of 64 rows ran as 1.5 rounds: the second half empty)
  // wave form: three 256-thread workgroups fit a CU (152 VGPRs) -> ONE round of 768 workgroups where M allows
  const int rows_per_block = w_form ? (int)(cdiv64(M, 768) < 8 ? 8 : cdiv64(M, 768)) : hw_form ? (int)(cdiv64(M, 512) < 8 ? 8 : cdiv64(M, 512)) : (M >= 16384 ? 32 : 16);
  const int nblk = (int)cdiv64(M, rows_per_block);
  float* part = (workspace && workspace_elems >= (int64_t)nblk * 2 * D && (dw || db)) ? workspace : nullptr;
  const dim3 blk((unsigned)(cdiv64(D, 256) * 64));
#define LN_LP(TDY_, TX_)                                                                                                               \
  hipLaunchKernelGGL((layernorm_bwd_cols_kernel<TDY_, 4, false, true, TX_>), dim3((unsigned)nblk), blk, 0, s, (const TDY_*)dy, (const TX_*)x, w, \
                     mean, rstd, (float*)nullptr, (bf16_t*)dx_bf16, dw, db, part, M, D, rows_per_block, (const bf16_t*)dres_bf16)
  if (w_form) {
    if (D == 768)
      hipLaunchKernelGGL((layernorm_bwd_w_kernel<1, 1>), dim3((unsigned)nblk), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, w, mean, rstd,
                         (const bf16_t*)dres_bf16, (bf16_t*)dx_bf16, dw, db, part, M, rows_per_block);
    else
      hipLaunchKernelGGL((layernorm_bwd_w_kernel<1, 0>), dim3((unsigned)nblk), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, w, mean, rstd,
                         (const bf16_t*)dres_bf16, (bf16_t*)dx_bf16, dw, db, part, M, rows_per_block);
  } else if (hw_form) {                              // (wider rows spill at four chunks per lane: they keep the 128-thread form)
    hipLaunchKernelGGL(layernorm_bwd_hw_kernel<3>, dim3((unsigned)nblk), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, w, mean, rstd,
                       (const bf16_t*)dres_bf16, (bf16_t*)dx_bf16, dw, db, part, M, D, rows_per_block);
  } else if (dtDy == DEMO_BF16 && dtX == DEMO_BF16 && D % 8 == 0 && g_ln_lp8)      // (hook value 2) the 128-thread form: 8 columns per thread
    hipLaunchKernelGGL(layernorm_bwd_lp8_kernel<4>, dim3((unsigned)nblk), dim3(128), 0, s, (const bf16_t*)dy, (const bf16_t*)x, w, mean, rstd,
                       (const bf16_t*)dres_bf16, (bf16_t*)dx_bf16, dw, db, part, M, D, rows_per_block);
  else if (dtDy == DEMO_BF16) { if (dtX == DEMO_BF16) LN_LP(bf16_t, bf16_t); else LN_LP(bf16_t, float); }
  else { if (dtX == DEMO_BF16) LN_LP(float, bf16_t); else LN_LP(float, float); }
#undef LN_LP
  DEMO_LAUNCH_CHECK("demo_layernorm_bwd_lp");
  if (part) {
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((unsigned)cdiv64(2 * D, 256), (unsigned)(nblk >= 64 ? 32 : 1)), dim3(256), 0, s, part, nblk, D, dw, db);
    DEMO_LAUNCH_CHECK("demo_layernorm_bwd_lp(reduce)");
  }
  return 0;
}

extern "C" int64_t demo_layernorm_bwd_workspace(int32_t M, int32_t D) { return cdiv64(M, 8) * 2 * (int64_t)D; }

extern "C" int demo_batchnorm_fwd(const float* x, const float* w, const float* b, float* running_mean, float* running_var,
                                  int32_t training, float momentum, float eps, void* y, int32_t dtY, float* save_mean,
                                  float* save_invstd, int32_t Bn, int32_t F, int32_t gsz, int32_t n_inner, int32_t n_outer,
                                  int64_t pstride, int64_t sstride, const float* gate, void* stream) {
  DEMO_REQUIRE(!(training && Bn < 2), "demo_batchnorm_fwd: Expected more than 1 value per channel when training (Bn=%d)", Bn);
  DEMO_REQUIRE(gsz == 0 || (F % gsz == 0 && (F / gsz) == n_inner * n_outer), "demo_batchnorm_fwd: bad grouping");
  DEMO_REQUIRE(!gate || gsz > 0, "demo_batchnorm_fwd: gate requires grouping");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)cdiv64(F, 64));
  if (Bn <= 128) {
    if (dtY == DEMO_BF16)
      hipLaunchKernelGGL(batchnorm_fwd_sliced_kernel<bf16_t>, grid, dim3(256), 0, s, x, w, b, running_mean, running_var, training,
                         momentum, eps, (bf16_t*)y, save_mean, save_invstd, Bn, F, gsz, n_inner, n_outer, pstride, sstride, gate);
    else
      hipLaunchKernelGGL(batchnorm_fwd_sliced_kernel<float>, grid, dim3(256), 0, s, x, w, b, running_mean, running_var, training,
                         momentum, eps, (float*)y, save_mean, save_invstd, Bn, F, gsz, n_inner, n_outer, pstride, sstride, gate);
    DEMO_LAUNCH_CHECK("demo_batchnorm_fwd");
    return 0;
  }
  if (dtY == DEMO_BF16)
    hipLaunchKernelGGL(batchnorm_fwd_kernel<bf16_t>, grid, dim3(64), 0, s, x, w, b, running_mean, running_var, training,
                       momentum, eps, (bf16_t*)y, save_mean, save_invstd, Bn, F, gsz, n_inner, n_outer, pstride, sstride, gate);
  else
    hipLaunchKernelGGL(batchnorm_fwd_kernel<float>, grid, dim3(64), 0, s, x, w, b, running_mean, running_var, training,
                       momentum, eps, (float*)y, save_mean, save_invstd, Bn, F, gsz, n_inner, n_outer, pstride, sstride, gate);
  DEMO_LAUNCH_CHECK("demo_batchnorm_fwd");
  return 0;
}
