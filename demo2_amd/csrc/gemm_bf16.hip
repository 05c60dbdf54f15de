// bf16 x bf16 -> fp32-accumulate MFMA GEMM for gfx950:  C[M,N] = epi(A[M,K] . B[N,K]^T)
// Both operands K-contiguous (the forward layout of every nn.Linear on the path:
// activations (rows, K) times weight (N, K)).
//
// Structure (cdna_hip_programming.md section 5, "minimum 2-phase" form):
//   * 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave =
//     4x4 v_mfma_f32_16x16x32_bf16 tiles), BK = 64;
//   * tiles staged HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip), two LDS
//     buffers, the next tile in flight while the current one is multiplied;
//   * LDS image is lane-linear per wave instruction, so the bank-conflict XOR swizzle
//     (16-B chunk ^= (row>>1)&7) is applied on the per-lane SOURCE address and again on the
//     ds_read_b128 fragment address (rule 21 of the guide);
//   * MFMA operands are passed swapped (B fragment as the A operand) so that every lane ends
//     up with 4 consecutive output columns -> 16-B / 8-B epilogue accesses for bias,
//     residual and the store;
//   * workgroup -> tile map is XCD-aware: consecutive ids of one XCD walk a group of 8
//     row-panels against one weight panel so both stay in that XCD's L2.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;      // 16 KiB per operand tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;    // A + B
constexpr int GROUP_M = 8;

struct FArgs {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  const float* bias;
  const float* residual;
  const void* pre;
  void* pre_out;
  int M, N, K;
  int64_t lda, ldb, ldc;
  int64_t sA, sB, sC, sBias;   // batch strides (single level)
  int dtC, act, dtPre, act_grad_kind, accumulate;
  float alpha;
  int ntm, ntn;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ G, int64_t ld, int row0, int row_max, int k0,
                                           char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int seg = j * 4 + wave;            // 1-KiB segment = 8 tile rows
    const int r = seg * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int gr = row0 + r;
    gr = gr < row_max ? gr : row_max - 1;     // clamp: rows past the edge are loaded but never stored
    const bf16_t* src = G + (int64_t)gr * ld + k0 + c * 8;
    __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(lds_tile + seg * 1024), 16, 0, 0);
  }
}

__global__ __launch_bounds__(256, 2) void gemm_bf16_nt_kernel(const FArgs p) {
  __shared__ __attribute__((aligned(1024))) char smem[2 * BUF_BYTES];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  // ---- XCD-aware tile id ------------------------------------------------------------------
  const int nwg = p.ntm * p.ntn;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
  const int tid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
  const int per_group = GROUP_M * p.ntn;
  const int grp = tid / per_group, in_grp = tid - grp * per_group;
  const int gm0 = grp * GROUP_M;
  const int gsz = (p.ntm - gm0) < GROUP_M ? (p.ntm - gm0) : GROUP_M;
  const int tm = gm0 + in_grp % gsz, tn = in_grp / gsz;
  const int m0 = tm * BM, n0 = tn * BN;

  const int z = blockIdx.y;
  const bf16_t* A = p.A + z * p.sA;
  const bf16_t* B = p.B + z * p.sB;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // per-lane fragment byte offsets inside a tile (row part + swizzled chunk for k-step 0 / 1)
  const int frow = lane & 15;
  const int sw = (lane >> 1) & 7;
  const int foff0 = frow * 128 + (((lane >> 4)) ^ sw) * 16;
  const int foff1 = frow * 128 + ((4 + (lane >> 4)) ^ sw) * 16;

  const int nk = p.K / BK;
  stage_tile(A, p.lda, m0, p.M, 0, smem, wave, lane);
  stage_tile(B, p.ldb, n0, p.N, 0, smem + TILE_BYTES, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int t = 0; t < nk; ++t) {
    char* cur = smem + (t & 1) * BUF_BYTES;
    if (t + 1 < nk) {
      char* nxt = smem + ((t + 1) & 1) * BUF_BYTES;
      stage_tile(A, p.lda, m0, p.M, (t + 1) * BK, nxt, wave, lane);
      stage_tile(B, p.ldb, n0, p.N, (t + 1) * BK, nxt + TILE_BYTES, wave, lane);
    }
    const char* la = cur + wr * (64 * 128);
    const char* lb = cur + TILE_BYTES + wc * (64 * 128);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int fo = kk ? foff1 : foff0;
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(la + i * 2048 + fo);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(lb + j * 2048 + fo);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: lane holds C[m][n .. n+3] -------------------------------------------------
  const int64_t coff = z * p.sC;
  const float* bias = p.bias ? p.bias + z * p.sBias : nullptr;
  const int ng = (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wr * 64 + i * 16 + frow;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wc * 64 + j * 16 + ng;
      float v[4] = {acc[i][j][0] * p.alpha, acc[i][j][1] * p.alpha, acc[i][j][2] * p.alpha, acc[i][j][3] * p.alpha};
      if (bias) {
        const float4 b4 = *reinterpret_cast<const float4*>(bias + n);
        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
      }
      const int64_t ci = coff + (int64_t)m * p.ldc + n;
      if (p.pre_out) {
        if (p.dtC == DEMO_F32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.pre_out) + ci) = make_float4(v[0], v[1], v[2], v[3]);
        else io<bf16_t>::st4(reinterpret_cast<bf16_t*>(p.pre_out) + ci, make_float4(v[0], v[1], v[2], v[3]));
      }
      if (p.act != DEMO_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], p.act);
      }
      if (p.pre) {
        float4 pr;
        if (p.dtPre == DEMO_BF16) pr = io<bf16_t>::ld4(reinterpret_cast<const bf16_t*>(p.pre) + ci);
        else pr = io<float>::ld4(reinterpret_cast<const float*>(p.pre) + ci);
        v[0] *= act_grad(pr.x, p.act_grad_kind); v[1] *= act_grad(pr.y, p.act_grad_kind);
        v[2] *= act_grad(pr.z, p.act_grad_kind); v[3] *= act_grad(pr.w, p.act_grad_kind);
      }
      if (p.residual) {
        const float4 r4 = *reinterpret_cast<const float4*>(p.residual + ci);
        v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
      }
      if (p.dtC == DEMO_F32) {
        float* c = reinterpret_cast<float*>(p.C) + ci;
        if (p.accumulate) {
          const float4 o = *reinterpret_cast<const float4*>(c);
          v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
        }
        *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        io<bf16_t>::st4(reinterpret_cast<bf16_t*>(p.C) + ci, make_float4(v[0], v[1], v[2], v[3]));
      }
    }
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// returns 1 if the descriptor fits the MFMA kernel's contract, 0 otherwise
int demo_gemm_bf16_eligible(const demo_gemm_desc& d) {
  if (d.dtA != DEMO_BF16 || d.dtB != DEMO_BF16) return 0;
  if (d.a_cs != 1 || d.b_cs != 1) return 0;
  if (d.K % BK || d.N % BN || d.M < 1) return 0;
  if (d.a_rs % 8 || d.b_rs % 8 || d.c_rs % 4) return 0;
  if (!al16(d.A) || !al16(d.B) || !al16(d.C)) return 0;
  if (d.bias && !al16(d.bias)) return 0;
  if (d.residual && !al16(d.residual)) return 0;
  if (d.act_grad_pre && !al16(d.act_grad_pre)) return 0;
  if (d.pre_out && !al16(d.pre_out)) return 0;
  if (d.accumulate && d.dtC != DEMO_F32) return 0;
  if (d.batch1 != 1) {
    if (d.batch2 != 1) return 0;               // single-level batching only
  }
  const int64_t sA = d.batch1 != 1 ? d.sA1 : d.sA2, sB = d.batch1 != 1 ? d.sB1 : d.sB2;
  const int64_t sC = d.batch1 != 1 ? d.sC1 : d.sC2, sBi = d.batch1 != 1 ? d.sBias1 : d.sBias2;
  if (d.batch1 * d.batch2 > 1 && (sA % 8 || sB % 8 || sC % 4 || sBi % 4)) return 0;
  if (d.M < 64) return 0;                       // skinny products: the generic kernel wastes less
  return 1;
}

int demo_gemm_bf16(const demo_gemm_desc& d, hipStream_t s) {
  FArgs p;
  p.A = reinterpret_cast<const bf16_t*>(d.A);
  p.B = reinterpret_cast<const bf16_t*>(d.B);
  p.C = d.C;
  p.bias = d.bias;
  p.residual = d.residual;
  p.pre = d.act_grad_pre;
  p.pre_out = d.pre_out;
  p.M = d.M; p.N = d.N; p.K = d.K;
  p.lda = d.a_rs; p.ldb = d.b_rs; p.ldc = d.c_rs;
  const bool lvl1 = d.batch1 != 1;
  p.sA = lvl1 ? d.sA1 : d.sA2; p.sB = lvl1 ? d.sB1 : d.sB2; p.sC = lvl1 ? d.sC1 : d.sC2; p.sBias = lvl1 ? d.sBias1 : d.sBias2;
  p.dtC = d.dtC; p.act = d.act; p.dtPre = d.dtPre; p.act_grad_kind = d.act_grad_kind; p.accumulate = d.accumulate;
  p.alpha = d.alpha;
  p.ntm = (int)cdiv64(d.M, BM);
  p.ntn = d.N / BN;
  dim3 grid((unsigned)(p.ntm * p.ntn), (unsigned)(d.batch1 * d.batch2));
  hipLaunchKernelGGL(gemm_bf16_nt_kernel, grid, dim3(256), 0, s, p);
  DEMO_LAUNCH_CHECK("demo_gemm(bf16 mfma)");
  return 0;
}
