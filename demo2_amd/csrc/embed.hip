// Patch gather (im2col), token assembly (+ln_pre), PIFE pooling, and small layout kernels.
#include "common.h"

namespace {

// One thread per 4 consecutive kx of one (patch, c, ky): reads 16 B of the image row, writes 4 elements.
template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, T* __restrict__ cols, int Bn, int H,
                                                     int W, int patch, int stride, int ny, int nx) {
  const int pk = 3 * patch * patch;
  const int q = patch >> 2;                               // float4 per patch row
  const int64_t total = (int64_t)Bn * ny * nx * 3 * patch * q;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i;
    const int kx4 = (int)(r % q); r /= q;
    const int ky = (int)(r % patch); r /= patch;
    const int c = (int)(r % 3); r /= 3;
    const int px = (int)(r % nx); r /= nx;
    const int py = (int)(r % ny); r /= ny;
    const int b = (int)r;
    const float* src = img + (((int64_t)b * 3 + c) * H + (py * stride + ky)) * W + px * stride + kx4 * 4;
    float4 v;
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) v = *reinterpret_cast<const float4*>(src);
    else v = make_float4(src[0], src[1], src[2], src[3]);
    T* dst = cols + ((int64_t)(b * ny + py) * nx + px) * pk + (c * patch + ky) * patch + kx4 * 4;
    io<T>::st4(dst, v);
  }
}

// One wave per token row.  W <= 2048.
template <typename TP>
__global__ __launch_bounds__(256) void embed_tokens_kernel(const TP* __restrict__ patch, const float* __restrict__ cls,
                                                           const float* __restrict__ pos, const float* __restrict__ cam_embed,
                                                           const int64_t* __restrict__ cam, int n_cam, float cam_scale, int cam_all,
                                                           const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                           float eps, float* __restrict__ x, float* __restrict__ pre,
                                                           float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                           int Bn, int np, int W, int* __restrict__ err) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int L = np + 1;
  if (row >= (int64_t)Bn * L) return;
  const int b = (int)(row / L), t = (int)(row % L);
  const int nv = W >> 2;
  const float* ce = nullptr;
  if (cam_embed && cam) {
    int64_t ci = cam[b];
    if (ci < 0 || ci >= n_cam) {                    // torch indexing raises here; a kernel clamps and reports (demo_hip.h)
      if (err && t == 0 && lane == 0) atomicOr(err, DEMO_ERR_CAM_RANGE);
      ci = ci < 0 ? 0 : n_cam - 1;
    }
    ce = cam_embed + ci * (int64_t)W;
  }
  const bool add_cam = ce && (cam_all || t == 0);
  float4 v[8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      float4 a;
      if (t == 0) a = *reinterpret_cast<const float4*>(cls + c * 4);
      else a = io<TP>::ld4(patch + ((int64_t)b * np + (t - 1)) * W + c * 4);
      const float4 p = *reinterpret_cast<const float4*>(pos + (int64_t)t * W + c * 4);
      a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
      if (add_cam) {
        const float4 e = *reinterpret_cast<const float4*>(ce + c * 4);
        a.x += cam_scale * e.x; a.y += cam_scale * e.y; a.z += cam_scale * e.z; a.w += cam_scale * e.w;
      }
      v[i] = a;
      s += (a.x + a.y) + (a.z + a.w);
      if (pre) *reinterpret_cast<float4*>(pre + row * W + c * 4) = a;
    }
  }
  float* xr = x + row * W;
  if (!ln_w) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = lane + i * 64;
      if (c < nv) *reinterpret_cast<float4*>(xr + c * 4) = v[i];
    }
    return;
  }
  const float mean = wave_sum(s) / (float)W;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      const float a = v[i].x - mean, bb = v[i].y - mean, cc = v[i].z - mean, dd = v[i].w - mean;
      q += (a * a + bb * bb) + (cc * cc + dd * dd);
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)W + eps);
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      const float4 g = *reinterpret_cast<const float4*>(ln_w + c * 4);
      const float4 be = *reinterpret_cast<const float4*>(ln_b + c * 4);
      float4 o;
      o.x = (v[i].x - mean) * rstd * g.x + be.x;
      o.y = (v[i].y - mean) * rstd * g.y + be.y;
      o.z = (v[i].z - mean) * rstd * g.z + be.z;
      o.w = (v[i].w - mean) * rstd * g.w + be.w;
      *reinterpret_cast<float4*>(xr + c * 4) = o;
    }
  }
}

// z[b] = [tokens[b,0,:], mean_{i>=1} tokens[b,i,:]]; block per sample, thread per 4 channels.
template <typename T>
__global__ __launch_bounds__(256) void pife_pool_cat_kernel(const T* __restrict__ tok, float* __restrict__ z, int L, int C) {
  // grid (images, chunks of 256 columns); thread = (4-column group g of the chunk, token split sp of 4): 8 tokens requested per step.
  // (One 128-thread workgroup per image walking all L - 1 tokens in sequence took 34 us at 3 images.)
  __shared__ float4 red[4][64];
  const int b = blockIdx.x, g = threadIdx.x & 63, sp = threadIdx.x >> 6;
  const int c4 = blockIdx.y * 64 + g;
  const bool ok = c4 < (C >> 2);
  const T* tb = tok + (int64_t)b * L * C + (ok ? c4 * 4 : 0);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int U = 8;
  for (int l0 = 1 + sp; l0 < L; l0 += 4 * U) {
    typename io<T>::raw4 r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int l = l0 + 4 * u;
      r[u] = io<T>::ld4_raw(tb + (int64_t)(l < L ? l : 1) * C);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (l0 + 4 * u >= L) break;
      const float4 v = io<T>::cvt4(r[u]);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  red[sp][g] = s;
  __syncthreads();
  if (sp == 0 && ok) {
    const float4 a = red[0][g], b1 = red[1][g], c1 = red[2][g], d1 = red[3][g];
    const float inv = 1.f / (float)(L - 1);
    float* zr = z + (int64_t)b * 2 * C;
    *reinterpret_cast<float4*>(zr + c4 * 4) = io<T>::ld4(tb);
    *reinterpret_cast<float4*>(zr + C + c4 * 4) = make_float4(((a.x + b1.x) + (c1.x + d1.x)) * inv, ((a.y + b1.y) + (c1.y + d1.y)) * inv,
                                                              ((a.z + b1.z) + (c1.z + d1.z)) * inv, ((a.w + b1.w) + (c1.w + d1.w)) * inv);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void set_token0_kernel(T* __restrict__ tok, const float* __restrict__ glob, int64_t ld, int L, int C) {
  const int b = blockIdx.x;
  for (int c4 = threadIdx.x; c4 < (C >> 2); c4 += blockDim.x)
    io<T>::st4(tok + (int64_t)b * L * C + c4 * 4, *reinterpret_cast<const float4*>(glob + (int64_t)b * ld + c4 * 4));
}
template <typename T>
__global__ __launch_bounds__(256) void get_token0_kernel(const T* __restrict__ tok, float* __restrict__ out, int64_t ld, int L, int C) {
  const int b = blockIdx.x;
  for (int c4 = threadIdx.x; c4 < (C >> 2); c4 += blockDim.x)
    *reinterpret_cast<float4*>(out + (int64_t)b * ld + c4 * 4) = io<T>::ld4(tok + (int64_t)b * L * C + c4 * 4);
}

// ---- DropPath / stochastic depth of the ImageNet ViT blocks (backbones/vit_pytorch.py:48-65,186-187) ----------------------------
// scale[b] = floor(keep + u_b) / keep, u_b uniform in [0,1) from a counter hash of (seed, b): 0 or 1/keep per image and branch
__device__ __forceinline__ uint32_t dp_hash(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ULL * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return (uint32_t)((z ^ (z >> 31)) >> 32);
}
__global__ __launch_bounds__(256) void droppath_mask_kernel(float* __restrict__ scale, int n, float keep, uint64_t seed,
                                                            const uint64_t* __restrict__ seed_dev) {
  if (seed_dev) seed += *seed_dev;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float u = (float)(dp_hash(seed, (uint64_t)i) >> 8) * (1.0f / 16777216.0f);
  scale[i] = floorf(keep + u) / keep;
}
// out[r, :] = x[r, :] + scale[r / L] * t[r, :]      (x, out fp32; t in the compute dtype; out may alias x)
template <typename T>
__global__ __launch_bounds__(256) void rowscale_add_kernel(const float* __restrict__ x, const T* __restrict__ t, const float* __restrict__ scale,
                                                           float* __restrict__ out, int64_t n4, int W4, int L) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float s = scale[(i / W4) / L];
    const float4 a = *reinterpret_cast<const float4*>(x + i * 4);
    const float4 b = io<T>::ld4(t + i * 4);
    *reinterpret_cast<float4*>(out + i * 4) = make_float4(fmaf(s, b.x, a.x), fmaf(s, b.y, a.y), fmaf(s, b.z, a.z), fmaf(s, b.w, a.w));
  }
}
// out[r, :] = scale[r / L] * x[r, :]   (x fp32 -> out in the compute dtype): the branch gradient of a DropPath residual add
template <typename T>
__global__ __launch_bounds__(256) void rowscale_cast_kernel(const float* __restrict__ x, const float* __restrict__ scale, T* __restrict__ out,
                                                            int64_t n4, int W4, int L) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float s = scale[(i / W4) / L];
    const float4 a = *reinterpret_cast<const float4*>(x + i * 4);
    io<T>::st4(out + i * 4, make_float4(s * a.x, s * a.y, s * a.z, s * a.w));
  }
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n4, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
    io<bf16_t>::st4(dst + i * 4, *reinterpret_cast<const float4*>(src + i * 4));
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int64_t i = n4 * 4; i < n; ++i) dst[i] = f2bf(src[i]);
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int64_t n4, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
    *reinterpret_cast<float4*>(dst + i * 4) = io<bf16_t>::ld4(src + i * 4);
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int64_t i = n4 * 4; i < n; ++i) dst[i] = bf2f(src[i]);
}

// 64x64 LDS-tiled transpose
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ src, T* __restrict__ dst, int rows, int cols,
                                                        int64_t src_stride, int64_t dst_stride) {
  __shared__ T tile[64][65];
  src += blockIdx.z * src_stride;          // batch of equally shaped matrices (the per-layer weights of one kind)
  dst += blockIdx.z * dst_stride;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    if (r < rows && c < cols) tile[i][tx] = src[(int64_t)r * cols + c];
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (r < rows && c < cols) dst[(int64_t)c * rows + r] = tile[tx][i];
  }
}

// the same for 2-byte elements with rows, cols multiples of 8 and 16-B aligned matrices (the bf16 W^T shadows of the trunk's Linears,
// rebuilt every optimizer step: 170 MB in, 170 MB out): 16-B global accesses on both sides -- a thread loads 8 consecutive columns of a
// row and stores 8 consecutive rows of a column (gathered from LDS) -- instead of 2-B accesses.
__global__ __launch_bounds__(256) void transpose16_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int rows, int cols,
                                                          int64_t src_stride, int64_t dst_stride) {
  __shared__ __attribute__((aligned(16))) uint16_t tile[64][72];        // 144-B rows: 16-B aligned, rows 8 apart fall on different banks
  src += blockIdx.z * src_stride;
  dst += blockIdx.z * dst_stride;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int idx = threadIdx.x + it * 256, r = idx >> 3, ch = idx & 7;
    if (r0 + r < rows && c0 + ch * 8 < cols)
      *reinterpret_cast<uint4*>(&tile[r][ch * 8]) = *reinterpret_cast<const uint4*>(src + (int64_t)(r0 + r) * cols + c0 + ch * 8);
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int idx = threadIdx.x + it * 256, c = idx >> 3, ch = idx & 7;
    if (c0 + c < cols && r0 + ch * 8 < rows) {
      uint32_t w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) w[j] = (uint32_t)tile[ch * 8 + 2 * j][c] | ((uint32_t)tile[ch * 8 + 2 * j + 1][c] << 16);
      *reinterpret_cast<uint4*>(dst + (int64_t)(c0 + c) * rows + r0 + ch * 8) = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
}

// column sums: grid (ceil(N/64), splits); each block sums a row slab, atomically adds into out.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, float* __restrict__ out, int M, int N, int64_t ldx,
                                                     int rows_per_block) {
  __shared__ float part[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + tx;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  float s = 0.f;
  if (n < N)
    for (int r = r0 + ty; r < r1; r += 4) s += io<T>::ld(X + (int64_t)r * ldx + n);
  part[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && n < N) atomicAdd(&out[n], part[0][tx] + part[1][tx] + part[2][tx] + part[3][tx]);
}

inline unsigned grid_for(int64_t work, int block = 256, int cap = 256 * 8) {
  int64_t g = cdiv64(work, block);
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int demo_im2col(const float* img, void* cols, int32_t Bn, int32_t H, int32_t W, int32_t patch, int32_t stride,
                           int32_t dt, void* stream) {
  DEMO_REQUIRE(patch % 4 == 0 && patch > 0 && stride > 0 && H >= patch && W >= patch, "demo_im2col: bad geometry");
  const int ny = (H - patch) / stride + 1, nx = (W - patch) / stride + 1;
  const int64_t total = (int64_t)Bn * ny * nx * 3 * patch * (patch / 4);
  hipStream_t s = (hipStream_t)stream;
  if (dt == DEMO_BF16)
    hipLaunchKernelGGL(im2col_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, img, (bf16_t*)cols, Bn, H, W, patch, stride, ny, nx);
  else
    hipLaunchKernelGGL(im2col_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, img, (float*)cols, Bn, H, W, patch, stride, ny, nx);
  DEMO_LAUNCH_CHECK("demo_im2col");
  return 0;
}

extern "C" int demo_embed_tokens(const void* patch, int32_t dt_patch, const float* cls, const float* pos,
                                 const float* cam_embed, const int64_t* cam, int32_t n_cam, float cam_scale, int32_t cam_all_tokens,
                                 const float* ln_w, const float* ln_b, float eps, float* x, float* pre, float* mean,
                                 float* rstd, int32_t Bn, int32_t np, int32_t W, int32_t* err, void* stream) {
  DEMO_REQUIRE(W % 4 == 0 && W <= 2048, "demo_embed_tokens: W=%d unsupported", W);
  DEMO_REQUIRE(!(cam_embed && cam) || n_cam > 0, "demo_embed_tokens: n_cam=%d with a camera table", n_cam);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)cdiv64((int64_t)Bn * (np + 1), 4));
  if (dt_patch == DEMO_BF16)
    hipLaunchKernelGGL(embed_tokens_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)patch, cls, pos, cam_embed, cam,
                       n_cam, cam_scale, cam_all_tokens, ln_w, ln_b, eps, x, pre, mean, rstd, Bn, np, W, err);
  else
    hipLaunchKernelGGL(embed_tokens_kernel<float>, grid, dim3(256), 0, s, (const float*)patch, cls, pos, cam_embed, cam,
                       n_cam, cam_scale, cam_all_tokens, ln_w, ln_b, eps, x, pre, mean, rstd, Bn, np, W, err);
  DEMO_LAUNCH_CHECK("demo_embed_tokens");
  return 0;
}

extern "C" int demo_pife_pool_cat(const void* tokens, int32_t dt, float* z, int32_t Bn, int32_t L, int32_t C, void* stream) {
  DEMO_REQUIRE(C % 4 == 0 && L >= 2, "demo_pife_pool_cat: bad shape");
  hipStream_t s = (hipStream_t)stream;
  if (dt == DEMO_BF16) hipLaunchKernelGGL(pife_pool_cat_kernel<bf16_t>, dim3(Bn, (unsigned)cdiv64(C, 256)), dim3(256), 0, s, (const bf16_t*)tokens, z, L, C);
  else hipLaunchKernelGGL(pife_pool_cat_kernel<float>, dim3(Bn, (unsigned)cdiv64(C, 256)), dim3(256), 0, s, (const float*)tokens, z, L, C);
  DEMO_LAUNCH_CHECK("demo_pife_pool_cat");
  return 0;
}

extern "C" int demo_set_token0(void* tokens, int32_t dt, const float* glob, int64_t glob_ld, int32_t Bn, int32_t L, int32_t C, void* stream) {
  DEMO_REQUIRE(C % 4 == 0 && glob_ld % 4 == 0 && ((uintptr_t)glob & 15) == 0, "demo_set_token0: bad shape/alignment");
  hipStream_t s = (hipStream_t)stream;
  if (dt == DEMO_BF16) hipLaunchKernelGGL(set_token0_kernel<bf16_t>, dim3(Bn), dim3(128), 0, s, (bf16_t*)tokens, glob, glob_ld, L, C);
  else hipLaunchKernelGGL(set_token0_kernel<float>, dim3(Bn), dim3(128), 0, s, (float*)tokens, glob, glob_ld, L, C);
  DEMO_LAUNCH_CHECK("demo_set_token0");
  return 0;
}

extern "C" int demo_get_token0(const void* tokens, int32_t dt, float* out, int64_t out_ld, int32_t Bn, int32_t L, int32_t C, void* stream) {
  DEMO_REQUIRE(C % 4 == 0 && out_ld % 4 == 0 && ((uintptr_t)out & 15) == 0, "demo_get_token0: bad shape/alignment");
  hipStream_t s = (hipStream_t)stream;
  if (dt == DEMO_BF16) hipLaunchKernelGGL(get_token0_kernel<bf16_t>, dim3(Bn), dim3(128), 0, s, (const bf16_t*)tokens, out, out_ld, L, C);
  else hipLaunchKernelGGL(get_token0_kernel<float>, dim3(Bn), dim3(128), 0, s, (const float*)tokens, out, out_ld, L, C);
  DEMO_LAUNCH_CHECK("demo_get_token0");
  return 0;
}

extern "C" int demo_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream) {
  DEMO_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 7) == 0, "demo_cast_f32_bf16: misaligned");
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n / 4, n);
  DEMO_LAUNCH_CHECK("demo_cast_f32_bf16");
  return 0;
}
extern "C" int demo_droppath_mask(float* scale, int32_t n, float keep_prob, uint64_t seed, const uint64_t* seed_dev, void* stream) {
  DEMO_REQUIRE(scale && n > 0 && keep_prob > 0.f && keep_prob <= 1.f, "demo_droppath_mask: n=%d keep=%f", n, keep_prob);
  hipLaunchKernelGGL(droppath_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, scale, n, keep_prob, seed, seed_dev);
  DEMO_LAUNCH_CHECK("demo_droppath_mask");
  return 0;
}
extern "C" int demo_rowscale_add(const float* x, const void* t, int32_t dt, const float* scale, float* out, int32_t Bn, int32_t L, int32_t W,
                                 void* stream) {
  DEMO_REQUIRE(x && t && scale && out && Bn > 0 && L > 0 && W > 0 && W % 4 == 0, "demo_rowscale_add: bad operand");
  const int64_t n4 = (int64_t)Bn * L * W / 4;
  if (dt == DEMO_BF16)
    hipLaunchKernelGGL(rowscale_add_kernel<bf16_t>, dim3(grid_for(n4)), dim3(256), 0, (hipStream_t)stream, x, (const bf16_t*)t, scale, out, n4, W / 4, L);
  else
    hipLaunchKernelGGL(rowscale_add_kernel<float>, dim3(grid_for(n4)), dim3(256), 0, (hipStream_t)stream, x, (const float*)t, scale, out, n4, W / 4, L);
  DEMO_LAUNCH_CHECK("demo_rowscale_add");
  return 0;
}
extern "C" int demo_rowscale_cast(const float* x, const float* scale, void* out, int32_t dt, int32_t Bn, int32_t L, int32_t W, void* stream) {
  DEMO_REQUIRE(x && scale && out && Bn > 0 && L > 0 && W > 0 && W % 4 == 0, "demo_rowscale_cast: bad operand");
  const int64_t n4 = (int64_t)Bn * L * W / 4;
  if (dt == DEMO_BF16)
    hipLaunchKernelGGL(rowscale_cast_kernel<bf16_t>, dim3(grid_for(n4)), dim3(256), 0, (hipStream_t)stream, x, scale, (bf16_t*)out, n4, W / 4, L);
  else
    hipLaunchKernelGGL(rowscale_cast_kernel<float>, dim3(grid_for(n4)), dim3(256), 0, (hipStream_t)stream, x, scale, (float*)out, n4, W / 4, L);
  DEMO_LAUNCH_CHECK("demo_rowscale_cast");
  return 0;
}
extern "C" int demo_cast_bf16_f32(const void* src, float* dst, int64_t n, void* stream) {
  DEMO_REQUIRE(((uintptr_t)dst & 15) == 0 && ((uintptr_t)src & 7) == 0, "demo_cast_bf16_f32: misaligned");
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, dst, n / 4, n);
  DEMO_LAUNCH_CHECK("demo_cast_bf16_f32");
  return 0;
}

extern "C" int demo_transpose_batched(const void* src, void* dst, int32_t rows, int32_t cols, int32_t batch, int64_t src_stride,
                                      int64_t dst_stride, int32_t dt, void* stream) {
  DEMO_REQUIRE(rows > 0 && cols > 0 && batch > 0 && batch <= 65535, "demo_transpose: bad shape %d x %d x %d", batch, rows, cols);
  dim3 grid((unsigned)cdiv64(cols, 64), (unsigned)cdiv64(rows, 64), (unsigned)batch);
  hipStream_t s = (hipStream_t)stream;
  const bool wide = dt == DEMO_BF16 && rows % 8 == 0 && cols % 8 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0 &&
                    src_stride % 8 == 0 && dst_stride % 8 == 0;
  if (wide)
    hipLaunchKernelGGL(transpose16_kernel, grid, dim3(256), 0, s, (const uint16_t*)src, (uint16_t*)dst, rows, cols, src_stride, dst_stride);
  else if (dt == DEMO_BF16)
    hipLaunchKernelGGL(transpose_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, rows, cols, src_stride, dst_stride);
  else
    hipLaunchKernelGGL(transpose_kernel<float>, grid, dim3(256), 0, s, (const float*)src, (float*)dst, rows, cols, src_stride, dst_stride);
  DEMO_LAUNCH_CHECK("demo_transpose");
  return 0;
}

extern "C" int demo_transpose(const void* src, void* dst, int32_t rows, int32_t cols, int32_t dt, void* stream) {
  return demo_transpose_batched(src, dst, rows, cols, 1, 0, 0, dt, stream);
}

extern "C" int demo_colsum(const void* X, float* out, int32_t M, int32_t N, int64_t ldx, int32_t dt, int32_t accumulate, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)N, s);
    if (e != hipSuccess) DEMO_FAIL(-2, "demo_colsum: memset failed: %s", hipGetErrorString(e));
  }
  const int rows_per_block = 256;
  dim3 grid((unsigned)cdiv64(N, 64), (unsigned)cdiv64(M, rows_per_block));
  if (dt == DEMO_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)X, out, M, N, ldx, rows_per_block);
  else hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, s, (const float*)X, out, M, N, ldx, rows_per_block);
  DEMO_LAUNCH_CHECK("demo_colsum");
  return 0;
}
