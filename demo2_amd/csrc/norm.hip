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

// Backward: one wave per row for dx; dw/db accumulated per block in LDS then atomically added
// into fp32 dw/db (D floats each; D*gridDim atomics spread over D addresses).
// NV = float4 per lane (compile time: D = 768 -> 3), so that only the registers a row needs are allocated and 8 waves per
// SIMD hide the HBM latency of this pure streaming kernel.
template <typename TDY, int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const TDY* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ w, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ dx,
                                                            int accumulate_dx, bf16_t* __restrict__ dx_lp, float* __restrict__ dw,
                                                            float* __restrict__ db, float* __restrict__ part, int M, int D,
                                                            int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [2][D]
  float* s_dw = sm;
  float* s_db = sm + D;
  for (int i = threadIdx.x; i < 2 * D; i += blockDim.x) sm[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nv = D >> 2;
  const int r0 = blockIdx.x * rows_per_block;
  float4 adw[NV], adb[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) adw[i] = adb[i] = make_float4(0.f, 0.f, 0.f, 0.f);

  for (int rr = wv; rr < rows_per_block; rr += 4) {
    const int row = r0 + rr;
    if (row >= M) break;
    const float mu = mean[row], rs = rstd[row];
    const float* xr = x + (int64_t)row * D;
    const TDY* dyr = dy + (int64_t)row * D;
    float4 xh[NV], g[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
        const float4 xv = *reinterpret_cast<const float4*>(xr + c * 4);
        const float4 d = io<TDY>::ld4(dyr + c * 4);
        const float4 wv4 = *reinterpret_cast<const float4*>(w + c * 4);
        xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
        g[i] = make_float4(d.x * wv4.x, d.y * wv4.y, d.z * wv4.z, d.w * wv4.w);
        s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
        s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
        adw[i].x += d.x * xh[i].x; adw[i].y += d.y * xh[i].y; adw[i].z += d.z * xh[i].z; adw[i].w += d.w * xh[i].w;
        adb[i].x += d.x; adb[i].y += d.y; adb[i].z += d.z; adb[i].w += d.w;
      }
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
    float* dxr = dx + (int64_t)row * D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
        float4 o;
        o.x = rs * (g[i].x - s1 - xh[i].x * s2);
        o.y = rs * (g[i].y - s1 - xh[i].y * s2);
        o.z = rs * (g[i].z - s1 - xh[i].z * s2);
        o.w = rs * (g[i].w - s1 - xh[i].w * s2);
        if (accumulate_dx) {
          const float4 p = *reinterpret_cast<const float4*>(dxr + c * 4);
          o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
        }
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

extern "C" int demo_layernorm_bwd(const void* dy, int32_t dtDy, const float* x, const float* w, const float* mean,
                                  const float* rstd, float* dx, int32_t accumulate_dx, void* dx_bf16, float* dw, float* db,
                                  float* workspace, int64_t workspace_elems, int32_t M, int32_t D, void* stream) {
  DEMO_REQUIRE(D % 4 == 0 && D <= 64 * 4 * LN_MAXV, "demo_layernorm_bwd: D=%d unsupported", D);
  hipStream_t s = (hipStream_t)stream;
  const int rows_per_block = 16;      // 4 rows per wave: ~3000 workgroups at 3B*L = 49536 rows keep every CU's memory pipes full
  const int nblk = (int)cdiv64(M, rows_per_block);
  dim3 grid((unsigned)nblk);
  const size_t shm = 2 * (size_t)D * sizeof(float);
  // thousands of workgroups atomically adding into the same 2*D words serialise; with a workspace each workgroup stores
  // its partial dw/db row and a second tiny kernel sums the rows
  float* part = (workspace && workspace_elems >= (int64_t)nblk * 2 * D && (dw || db)) ? workspace : nullptr;
  const int nvl = (int)cdiv64(D, 256);      // float4 per lane actually needed
#define LN_BWD_LAUNCH(NV_)                                                                                                      \
  do {                                                                                                                          \
    if (dtDy == DEMO_BF16)                                                                                                      \
      hipLaunchKernelGGL((layernorm_bwd_kernel<bf16_t, NV_>), grid, dim3(256), shm, s, (const bf16_t*)dy, x, w, mean, rstd, dx, \
                         accumulate_dx, (bf16_t*)dx_bf16, dw, db, part, M, D, rows_per_block);                                  \
    else                                                                                                                        \
      hipLaunchKernelGGL((layernorm_bwd_kernel<float, NV_>), grid, dim3(256), shm, s, (const float*)dy, x, w, mean, rstd, dx,   \
                         accumulate_dx, (bf16_t*)dx_bf16, dw, db, part, M, D, rows_per_block);                                  \
  } while (0)
  if (nvl <= 1) LN_BWD_LAUNCH(1);
  else if (nvl <= 2) LN_BWD_LAUNCH(2);
  else if (nvl <= 3) LN_BWD_LAUNCH(3);
  else if (nvl <= 4) LN_BWD_LAUNCH(4);
  else if (nvl <= 6) LN_BWD_LAUNCH(6);
  else LN_BWD_LAUNCH(8);
#undef LN_BWD_LAUNCH
  DEMO_LAUNCH_CHECK("demo_layernorm_bwd");
  if (part) {
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((unsigned)cdiv64(2 * D, 256), (unsigned)(nblk >= 64 ? 32 : 1)), dim3(256), 0, s, part,
                       nblk, D, dw, db);
    DEMO_LAUNCH_CHECK("demo_layernorm_bwd(reduce)");
  }
  return 0;
}

extern "C" int64_t demo_layernorm_bwd_workspace(int32_t M, int32_t D) { return cdiv64(M, 16) * 2 * (int64_t)D; }

extern "C" int demo_batchnorm_fwd(const float* x, const float* w, const float* b, float* running_mean, float* running_var,
                                  int32_t training, float momentum, float eps, void* y, int32_t dtY, float* save_mean,
                                  float* save_invstd, int32_t Bn, int32_t F, int32_t gsz, int32_t n_inner, int32_t n_outer,
                                  int64_t pstride, int64_t sstride, const float* gate, void* stream) {
  DEMO_REQUIRE(!(training && Bn < 2), "demo_batchnorm_fwd: Expected more than 1 value per channel when training (Bn=%d)", Bn);
  DEMO_REQUIRE(gsz == 0 || (F % gsz == 0 && (F / gsz) == n_inner * n_outer), "demo_batchnorm_fwd: bad grouping");
  DEMO_REQUIRE(!gate || gsz > 0, "demo_batchnorm_fwd: gate requires grouping");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)cdiv64(F, 64));
  if (dtY == DEMO_BF16)
    hipLaunchKernelGGL(batchnorm_fwd_kernel<bf16_t>, grid, dim3(64), 0, s, x, w, b, running_mean, running_var, training,
                       momentum, eps, (bf16_t*)y, save_mean, save_invstd, Bn, F, gsz, n_inner, n_outer, pstride, sstride, gate);
  else
    hipLaunchKernelGGL(batchnorm_fwd_kernel<float>, grid, dim3(64), 0, s, x, w, b, running_mean, running_var, training,
                       momentum, eps, (float*)y, save_mean, save_invstd, Bn, F, gsz, n_inner, n_outer, pstride, sstride, gate);
  DEMO_LAUNCH_CHECK("demo_batchnorm_fwd");
  return 0;
}
