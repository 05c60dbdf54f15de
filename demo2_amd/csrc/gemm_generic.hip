// Generic strided, two-level-batched GEMM with fp32 accumulation on the vector ALUs.
// Serves (a) the fp32 parity mode, (b) every small / oddly shaped product of the path
// (PIFE reduce, HDM folded keys and per-head value projections, ATMoE experts, classifiers),
// (c) any transposition via strides.  The large bf16 products go to gemm_bf16.hip instead.
#include "common.h"

namespace {

// The output tile is GT x GT, GT = 64 (16 x 16 threads, 4 x 4 outputs each) or 32 (2 x 2 outputs each).  The products this kernel serves on
// the path are a few hundred rows at most (the batch, or 3 x the batch): a 64^2 grid is then a few dozen workgroups of ONE wave per SIMD
// each, every K-step an unhidden chain {global loads -> LDS -> 512 dependent-issue FMAs} of ~1.3 us.  With 32^2 tiles the same product
// is four times the workgroups -- more CUs busy, several waves per SIMD -- at a quarter of the FMAs per thread and step; it is taken
// whenever the 64^2 grid would not fill the chip (launch_c).  Same-box A/B of the training step: 35.17 -> 34.90 ms.
// GK: reduction elements per step (32 for both tile forms; 64- and 128-deep steps of the 32^2 form -- fewer, fatter {global load -> LDS ->
// barrier} chains, 16 loads per thread and operand in flight -- measured equal and slower on the training step: registers and LDS per
// workgroup cost the co-resident workgroups that hide the chain).

template <typename T> __device__ __forceinline__ float ldg(const void* p, int64_t i) {
  return io<T>::ld(reinterpret_cast<const T*>(p) + i);
}

struct GArgs {
  demo_gemm_desc d;
  int splitk;   // > 1: blockIdx.z = batch * splitk + slice; slices combine with fp32 atomics (C pre-zeroed or accumulated into)
  int kper;     // reduction elements per slice (multiple of GK)
};

template <typename TA, typename TB, typename TC, int GT, int GK>
__global__ __launch_bounds__(256) void gemm_generic_kernel(const GArgs args) {
  const demo_gemm_desc& d = args.d;
  constexpr int TT = GT / 16;          // outputs per thread and dimension
  __shared__ float As[GK][GT + 4];
  __shared__ float Bs[GK][GT + 4];

  const int z = blockIdx.z / args.splitk, slice = blockIdx.z - z * args.splitk;
  const int kbeg = slice * args.kper, kend = min(d.K, kbeg + args.kper);
  const int z1 = z / d.batch2, z2 = z % d.batch2;
  const TA* A = reinterpret_cast<const TA*>(d.A) + z1 * d.sA1 + z2 * d.sA2;
  const TB* B = reinterpret_cast<const TB*>(d.B) + z1 * d.sB1 + z2 * d.sB2;
  const int64_t coff = z1 * d.sC1 + z2 * d.sC2;
  TC* C = reinterpret_cast<TC*>(d.C) + coff;
  const float* bias = d.bias ? d.bias + z1 * d.sBias1 + z2 * d.sBias2 : nullptr;
  const bool has_res = d.residual != nullptr;

  const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
  const int t = threadIdx.x;
  const int ty = t >> 4, tx = t & 15;   // 16 x 16 threads, TT x TT outputs each

  // load mapping: walk the contiguous direction fastest
  const bool a_kfast = (d.a_cs == 1) || (d.a_rs != 1);
  const bool b_kfast = (d.b_cs == 1) || (d.b_rs != 1);

  float acc[TT][TT];
#pragma unroll
  for (int i = 0; i < TT; ++i)
#pragma unroll
    for (int j = 0; j < TT; ++j) acc[i][j] = 0.f;

  // One K-step ahead in registers: the loads of step s+1 are issued before the FMAs of step s, so a step costs one memory
  // latency OR its arithmetic, not their sum (the skinny products here are a few workgroups walking a long K: the
  // non-prefetched form spent 2.6 us per 16-deep step, 170 us for the PIFE reduce Linear).
  constexpr int LPT = GT * GK / 256;                  // loads per thread per operand per step
  float ra[LPT], rb[LPT];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
      const int idx = t + i * 256;
      int r, k;
      if (a_kfast) { r = idx / GK; k = idx % GK; } else { r = idx % GT; k = idx / GT; }
      const int gm = m0 + r, gk = k0 + k;
      ra[i] = (gm < d.M && gk < kend) ? io<TA>::ld(A + (int64_t)gm * d.a_rs + (int64_t)gk * d.a_cs) : 0.f;
      if (b_kfast) { r = idx / GK; k = idx % GK; } else { r = idx % GT; k = idx / GT; }
      const int gn = n0 + r, gk2 = k0 + k;
      rb[i] = (gn < d.N && gk2 < kend) ? io<TB>::ld(B + (int64_t)gn * d.b_rs + (int64_t)gk2 * d.b_cs) : 0.f;
    }
  };
  if (kbeg < kend) fetch(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += GK) {
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
      const int idx = t + i * 256;
      int r, k;
      if (a_kfast) { r = idx / GK; k = idx % GK; } else { r = idx % GT; k = idx / GT; }
      As[k][r] = ra[i];
      if (b_kfast) { r = idx / GK; k = idx % GK; } else { r = idx % GT; k = idx / GT; }
      Bs[k][r] = rb[i];
    }
    __syncthreads();
    if (k0 + GK < kend) fetch(k0 + GK);
#pragma unroll
    for (int k = 0; k < GK; ++k) {
      float av[TT], bv[TT];
      if constexpr (TT == 4) {
        const float4 a = *reinterpret_cast<const float4*>(&As[k][ty * 4]);
        const float4 b = *reinterpret_cast<const float4*>(&Bs[k][tx * 4]);
        av[0] = a.x; av[1] = a.y; av[2] = a.z; av[3] = a.w;
        bv[0] = b.x; bv[1] = b.y; bv[2] = b.z; bv[3] = b.w;
      } else {
        const float2 a = *reinterpret_cast<const float2*>(&As[k][ty * 2]);
        const float2 b = *reinterpret_cast<const float2*>(&Bs[k][tx * 2]);
        av[0] = a.x; av[1] = a.y;
        bv[0] = b.x; bv[1] = b.y;
      }
#pragma unroll
      for (int i = 0; i < TT; ++i)
#pragma unroll
        for (int j = 0; j < TT; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < TT; ++i) {
    const int gm = m0 + ty * TT + i;
    if (gm >= d.M) continue;
#pragma unroll
    for (int j = 0; j < TT; ++j) {
      const int gn = n0 + tx * TT + j;
      if (gn >= d.N) continue;
      float v = acc[i][j] * d.alpha;
      const int64_t ci = (int64_t)gm * d.c_rs + gn;
      if (args.splitk > 1) {                       // fp32 C only, no non-linear epilogue (checked on the host)
        if (bias && slice == 0) v += bias[gn];
        atomicAdd(reinterpret_cast<float*>(d.C) + coff + ci, v);
        continue;
      }
      if (bias) v += bias[gn];
      if (d.pre_out && d.pre_out_kind >= 1) {
        float dv;
        v = act_both(v, d.act, dv);
        if (d.pre_out_kind == 2) reinterpret_cast<uint8_t*>(d.pre_out)[coff + ci] = d8_encode(dv);      // DEMO_D8 code
        else io<TC>::st(reinterpret_cast<TC*>(d.pre_out) + coff + ci, dv);
      } else {
        if (d.pre_out) io<TC>::st(reinterpret_cast<TC*>(d.pre_out) + coff + ci, v);
        v = act_apply(v, d.act);
      }
      if (d.act_grad_pre) {
        const float pre = d.dtPre == DEMO_D8 ? d8_decode(reinterpret_cast<const uint8_t*>(d.act_grad_pre)[coff + ci])
                        : d.dtPre == DEMO_BF16 ? ldg<bf16_t>(d.act_grad_pre, coff + ci) : ldg<float>(d.act_grad_pre, coff + ci);
        v *= act_grad(pre, d.act_grad_kind);
      }
      if (has_res) v += d.dtRes == DEMO_BF16 ? ldg<bf16_t>(d.residual, coff + ci) : ldg<float>(d.residual, coff + ci);
      if (d.accumulate) v += io<TC>::ld(C + ci);
      io<TC>::st(C + ci, v);
    }
  }
}

__global__ __launch_bounds__(256) void clear_c_kernel(float* __restrict__ C, int M, int N, int64_t c_rs, int batch2, int64_t sC1,
                                                      int64_t sC2) {
  const int z = blockIdx.z;
  float* c = C + (z / batch2) * sC1 + (z % batch2) * sC2;
  const int64_t mn = (int64_t)M * N;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < mn; i += (int64_t)gridDim.x * 256) c[(i / N) * c_rs + (i % N)] = 0.f;
}

#ifndef DEMO_GENERIC_GK32
#define DEMO_GENERIC_GK32 32      // reduction elements per step of the 32^2 form (A/B on the training step: 32 and 64 equal, 128 slower)
#endif
int g_generic_tile = 0;     // tuning / test hook (demo_gemm_generic_tile): 0 = heuristic, 32 / 64 = that tile for every generic launch

template <typename TA, typename TB>
int launch_c(const demo_gemm_desc& d, hipStream_t s) {
  GArgs a;
  a.d = d;
  a.splitk = 1;
  a.kper = d.K;
  const int nb = d.batch1 * d.batch2;
  const int64_t tiles64 = cdiv64(d.N, 64) * cdiv64(d.M, 64) * nb;
  // 32^2 tiles (128-deep steps) when the 64^2 grid would leave most of the 256 CUs idle or at one wave per SIMD
  const bool small = g_generic_tile == 32 || (g_generic_tile == 0 && tiles64 < 384);
  const int GT = small ? 32 : 64, GKs = small ? DEMO_GENERIC_GK32 : 32;
  const int64_t tiles = cdiv64(d.N, GT) * cdiv64(d.M, GT) * nb;
  // skinny products with a long reduction (classifier 128 x 171 x 3584, gate Linear 128 x 512 x 3584 ...): a handful of
  // workgroups would walk K serially at one memory latency per step -> split the reduction across workgroups
  const bool linear_epi = d.act == DEMO_ACT_NONE && !d.pre_out && !d.act_grad_pre && !d.residual && d.dtC == DEMO_F32;
  if (linear_epi && tiles64 < 128 && d.K >= 512) {
    int sk = (int)((small ? 1024 : 256) / tiles);
    const int maxk = d.K / 128;                           // at least 128 reduction elements per slice
    if (sk > maxk) sk = maxk;
    if (sk > 64) sk = 64;
    if (sk > 1) {
      const int ksteps = (int)cdiv64(d.K, GKs);
      const int per = (int)cdiv64(ksteps, sk);
      a.kper = per * GKs;
      a.splitk = (int)cdiv64(ksteps, per);
      if (a.splitk > 1 && !d.accumulate) {                       // slices add into C: clear it first (rows may be strided)
        if (nb == 1) {
          hipError_t e = hipMemset2DAsync(d.C, (size_t)d.c_rs * sizeof(float), 0, (size_t)d.N * sizeof(float), (size_t)d.M, s);
          if (e != hipSuccess) DEMO_FAIL(-2, "demo_gemm(generic): memset failed: %s", hipGetErrorString(e));
        } else {                                                 // one launch for the whole batch, not a memset per matrix
          const int64_t mn = (int64_t)d.M * d.N;
          hipLaunchKernelGGL(clear_c_kernel, dim3((unsigned)(cdiv64(mn, 256) < 64 ? cdiv64(mn, 256) : 64), 1, (unsigned)nb), dim3(256),
                             0, s, reinterpret_cast<float*>(d.C), d.M, d.N, d.c_rs, d.batch2, d.sC1, d.sC2);
          DEMO_LAUNCH_CHECK("demo_gemm(generic clear)");
        }
      }
    }
  }
  dim3 grid((unsigned)cdiv64(d.N, GT), (unsigned)cdiv64(d.M, GT), (unsigned)(nb * a.splitk));
  if (small) {
    if (d.dtC == DEMO_F32)
      hipLaunchKernelGGL((gemm_generic_kernel<TA, TB, float, 32, DEMO_GENERIC_GK32>), grid, dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL((gemm_generic_kernel<TA, TB, bf16_t, 32, DEMO_GENERIC_GK32>), grid, dim3(256), 0, s, a);
  } else {
    if (d.dtC == DEMO_F32)
      hipLaunchKernelGGL((gemm_generic_kernel<TA, TB, float, 64, 32>), grid, dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL((gemm_generic_kernel<TA, TB, bf16_t, 64, 32>), grid, dim3(256), 0, s, a);
  }
  DEMO_LAUNCH_CHECK("demo_gemm(generic)");
  return 0;
}

}  // namespace

extern "C" int demo_gemm_generic_tile(int t) { g_generic_tile = (t == 32 || t == 64) ? t : 0; return 0; }

int demo_gemm_generic(const demo_gemm_desc& d, hipStream_t s) {
  if (d.dtA == DEMO_F32 && d.dtB == DEMO_F32) return launch_c<float, float>(d, s);
  if (d.dtA == DEMO_BF16 && d.dtB == DEMO_BF16) return launch_c<bf16_t, bf16_t>(d, s);
  if (d.dtA == DEMO_F32 && d.dtB == DEMO_BF16) return launch_c<float, bf16_t>(d, s);
  return launch_c<bf16_t, float>(d, s);
}
