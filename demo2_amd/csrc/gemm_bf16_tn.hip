// bf16 MFMA GEMM with BOTH operands reduction-major:  C[Mo,No] (+)= alpha * sum_r A[r,Mo] * B[r,No]
// This is the weight-gradient product dW = dY^T X of every nn.Linear: A = dY (rows, N), B = X (rows, K),
// both exactly as the forward pass left them in HBM (row-major, rows = tokens) -- no transposed copies of
// activations are ever made.
//
//   * tile 128 (Mo) x 128 (No) x 64 (reduction rows) per 256-thread workgroup; an operand tile is 64 rows of
//     256 B, staged with global_load_lds_dwordx4 (4 rows per wave instruction), double buffered;
//   * MFMA fragments need 8 consecutive REDUCTION indices per lane, i.e. a column of the staged tile:
//     ds_read_b64_tr_b16 (the gfx950 transposing LDS read) delivers them, two reads per fragment;
//   * 256-B rows put a column of 16-B chunks on one bank group: a half-wave's transposed read touches rows
//     {8g+q : g in a pair, q<4} x 2 chunks, so chunk index ^= 2*(row&3) + 8*((row>>3)&1) (applied on the per-lane
//     source address of the LDS-DMA and on the read) spreads them over all 16 chunk positions -> conflict free;
//   * the reduction dimension is the long one here (tokens: 24768 at B=64) while Mo x No is only 36..144 tiles,
//     so the grid splits the reduction (split-K) and combines with fp32 atomics into the gradient buffer.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BK * 256;          // 16 KiB per operand tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

struct TArgs {
  const bf16_t* A;
  const bf16_t* B;
  float* C;
  int Mo, No, Kr;
  int64_t lda, ldb, ldc;
  int splitk, kper;     // kper = reduction rows per split (multiple of BK)
  float alpha;
  int ntm, ntn;
  int part_m;           // 1: XCDs partition the Mo panels, 0: the No panels
  int use_atomics;      // 1: atomicAdd into C; 0: C = (accumulate ? C : 0) + result with plain stores
  int accumulate;
  float* colsum;        // optional: colsum[m] += sum_r A[r, m]  (the bias gradient of the same Linear), atomics
  float* partial;       // 256^2 form, optional: (splitk, Mo, No) workspace -> slices store plain partial tiles, summed by a 2nd pass
};

// Up to four products in ONE launch of the 256^2 form (demo_gemm_tn_bf16_group): the workgroups of product i are blockIdx.x in
// [base[i], base[i+1]) (multiples of 8, so blockIdx & 7 stays the XCD id), longest reduction slices first.
struct TGroup {
  TArgs g[4];
  int base[5];
  int n;
};

// LDS-DMA staging of a full 64-row tile: rows [k0, k0+64) x cols [c0, c0+128)
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ G, int64_t ld, int k0, int c0, char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int seg = j * 4 + wave;               // 1 KiB = 4 rows
    const int r = seg * 4 + (lane >> 4);
    const int c = (lane & 15) ^ (((r & 3) << 1) | (((r >> 3) & 1) << 3));
    const bf16_t* src = G + (int64_t)(k0 + r) * ld + c0 + c * 8;
    __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(lds_tile + seg * 1024), 16, 0, 0);
  }
}
// register-staged variant for the (single) partial tile at the end of the reduction: rows >= k_end read as zero
__device__ __forceinline__ void stage_tile_tail(const bf16_t* __restrict__ G, int64_t ld, int k0, int k_end, int c0, char* lds_tile,
                                                int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int seg = j * 4 + wave;
    const int r = seg * 4 + (lane >> 4);
    const int c = (lane & 15) ^ (((r & 3) << 1) | (((r >> 3) & 1) << 3));
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (k0 + r < k_end) v = *reinterpret_cast<const uint4*>(G + (int64_t)(k0 + r) * ld + c0 + c * 8);
    *reinterpret_cast<uint4*>(lds_tile + seg * 1024 + lane * 16) = v;
  }
}

// K-loop.  CS = this wave also accumulates the bias gradient: its A fragments times a fragment of ones leave sum_r A[r, m]
// in every column of a 16x16 accumulator (4 extra MFMAs per 32-deep step).  CS is a template parameter and the choice is
// made ONCE outside the loop: a wave-uniform `if` around MFMAs inside the loop made hipcc copy all 64 accumulator
// registers every iteration (68 v_mov per K-step).
template <bool CS>
__device__ __forceinline__ void tn_mainloop(const TArgs& p, char* smem, f32x4 (&acc)[4][4], f32x4 (&csum)[4], int m0, int n0, int ks,
                                            int ke, int wave, int lane, int wr, int wc) {
  // transposed-read addressing: lane (g, idx=4q+pp): row 8g+q (+4), 8-B piece at column 4*pp of a 16-column block
  const int g = lane >> 4, idx = lane & 15, q = idx >> 2, pp = idx & 3;
  const int trow = 8 * g + q;                         // row inside a 32-row k-step (first read); second read = +4 rows
  const int sw = (q << 1) | ((g & 1) << 3);           // chunk swizzle of rows 8g+q and 8g+q+4
  const int half = (pp & 1) << 3, csub = pp >> 1;
  int aoff[4], boff[4];                                // per-lane byte offsets inside an operand tile (k-step 0, first read)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    aoff[i] = trow * 256 + (((wr * 8 + i * 2 + csub) ^ sw) << 4) + half;
    boff[i] = trow * 256 + (((wc * 8 + i * 2 + csub) ^ sw) << 4) + half;
  }
  const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};

  const int nk = (ke - ks + BK - 1) / BK;
  auto stage = [&](int t, char* buf) {
    const int k0 = ks + t * BK;
    if (k0 + BK <= ke) {
      stage_tile(p.A, p.lda, k0, m0, buf, wave, lane);
      stage_tile(p.B, p.ldb, k0, n0, buf + TILE_BYTES, wave, lane);
    } else {
      stage_tile_tail(p.A, p.lda, k0, ke, m0, buf, wave, lane);
      stage_tile_tail(p.B, p.ldb, k0, ke, n0, buf + TILE_BYTES, wave, lane);
    }
  };
  stage(0, smem);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int t = 0; t < nk; ++t) {
    char* cur = smem + (t & 1) * BUF_BYTES;
    if (t + 1 < nk) stage(t + 1, smem + ((t + 1) & 1) * BUF_BYTES);
    const char* la = cur;
    const char* lb = cur + TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(la + kk * 8192 + aoff[i]));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(la + kk * 8192 + 1024 + aoff[i]));
        af[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lb + kk * 8192 + boff[j]));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lb + kk * 8192 + 1024 + boff[j]));
        bfr[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      if constexpr (CS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) csum[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], ones, csum[i], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
}

__global__ __launch_bounds__(256, 2) void gemm_bf16_tn_kernel(const TArgs p) {
  __shared__ __attribute__((aligned(1024))) char smem[2 * BUF_BYTES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  // XCD-aware work map.  Both operands are streamed (tokens x features), each column panel is shared by every tile of
  // the other dimension.  Workgroup ids are dealt round-robin to the 8 XCDs (private L2s): XCD x owns the panels
  // x, x+8, ... of the dimension with MORE panels (the bigger operand), so that operand crosses the fabric once
  // instead of 8 times; inside an XCD the order is split-major so co-resident workgroups walk the same token rows.
  const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3;
  const int P = p.part_m ? p.ntm : p.ntn, Q = p.part_m ? p.ntn : p.ntm;
  const int np = (P - xcd + 7) >> 3;                  // panels owned by this XCD
  if (w >= np * Q * p.splitk) return;                 // padding workgroups (P not a multiple of 8)
  const int split = w / (np * Q), rem = w - split * (np * Q);
  const int pidx = xcd + 8 * (rem / Q), qidx = rem % Q;
  const int tm = p.part_m ? pidx : qidx, tn = p.part_m ? qidx : pidx;
  const int m0 = tm * BM, n0 = tn * BN;
  const int ks = split * p.kper;
  const int ke = min(p.Kr, ks + p.kper);
  if (ks >= ke) return;

  f32x4 acc[4][4], csum[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    csum[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // bias gradient: only the waves that own the first column panel (tn == 0, wc == 0) carry it
  const bool do_colsum = p.colsum != nullptr && tn == 0 && wc == 0;          // wave-uniform
  if (do_colsum) tn_mainloop<true>(p, smem, acc, csum, m0, n0, ks, ke, wave, lane, wr, wc);
  else tn_mainloop<false>(p, smem, acc, csum, m0, n0, ks, ke, wave, lane, wr, wc);
  const int g = lane >> 4, idx = lane & 15;

  if (do_colsum && idx == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(p.colsum + m0 + wr * 64 + i * 16 + 4 * g + r, csum[i][r]);
  }
  // D[row = m_local = 4g + reg][col = n_local = lane & 15]: per register 4 rows x 64 contiguous bytes
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wc * 64 + j * 16 + idx;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wr * 64 + i * 16 + 4 * g + r;
        float* c = p.C + (int64_t)m * p.ldc + n;
        const float v = acc[i][j][r] * p.alpha;
        if (p.use_atomics) atomicAdd(c, v);
        else *c = p.accumulate ? *c + v : v;
      }
    }
  }
}

// ---- 256 x 256 ping-pong form ------------------------------------------------------------------------------------------
// Same product, output tile 256 (Mo) x 256 (No) per 512-thread workgroup (8 waves as 2 x 4, 128 x 64 per wave), reduction
// staged 32 rows at a time: a stage is 2 x 32 rows x 512 B = 32 KiB, the whole 160-KiB LDS is a ring of FIVE stages with
// three of them in flight (both operands are activations streaming from HBM, ~96 KiB per CU of loads in the air).  Half the
// L2 -> LDS bytes per FLOP of the 128^2 form, one workgroup per CU, the split-K factor chosen so that the grid is ~one
// round of the 256 CUs.
// A stage is one 32-deep MFMA step, worked in two phases (upper / lower 64 output rows of the wave, 16 MFMAs each):
//     phase = { issue the phase's transposing LDS reads (+ the LDS-DMA of stage t+3 in the first phase) ; s_barrier ;
//               MFMAs ; s_barrier }
// with the two wave halves (wr = 0 / 1, one wave of each per SIMD) running one barrier apart: one half multiplies while
// the other reads LDS (same scheme as the NT ping-pong kernel, gemm_bf16.hip).
// Hazards: stage t+1 is waited for (counted vmcnt: stages t+2, t+3 stay in flight) before the first barrier of phase 2 of
// stage t and first read two barriers later; slot (t+3)%5 is re-filled at phase 1 of stage t, its last reader was phase 2
// of stage t-2.  512-B rows: the bank of a 16-B chunk depends on chunk&15 only, so the swizzle of the 128^2 form carries over.
#ifndef DEMO_PP_PHASES
#define DEMO_PP_PHASES 2      // 32 MFMAs between barriers (see gemm_bf16.hip); 4 = the round-1 form with 16
#endif
constexpr int PBM = 256, PBN = 256, PBK = 32, PSTAGES = 5;
constexpr int POP_BYTES = PBK * 512;          // 16 KiB per operand per stage
constexpr int PSTAGE_BYTES = 2 * POP_BYTES;

// LDS-DMA of 32 rows x 256 columns (512 B per row): 2 rows per wave instruction, 2 instructions per wave
__device__ __forceinline__ void pp_stage_op(const bf16_t* __restrict__ G, int64_t ld, int k0, int c0, char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int seg = j * 8 + wave;               // 1 KiB = 2 rows
    const int r = seg * 2 + (lane >> 5);
    const int c = (lane & 31) ^ (((r & 3) << 1) | (((r >> 3) & 1) << 3));
    const bf16_t* src = G + (int64_t)(k0 + r) * ld + c0 + c * 8;
    __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(lds_tile + seg * 1024), 16, 0, 0);
  }
}
// the partial stage at the end of the reduction: through registers, rows >= k_end read as zero (waits for its own loads)
__device__ __forceinline__ void pp_stage_op_tail(const bf16_t* __restrict__ G, int64_t ld, int k0, int k_end, int c0, char* lds_tile,
                                                 int wave, int lane) {
  uint4 v[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int seg = j * 8 + wave;
    const int r = seg * 2 + (lane >> 5);
    const int c = (lane & 31) ^ (((r & 3) << 1) | (((r >> 3) & 1) << 3));
    v[j] = make_uint4(0u, 0u, 0u, 0u);
    if (k0 + r < k_end) v[j] = *reinterpret_cast<const uint4*>(G + (int64_t)(k0 + r) * ld + c0 + c * 8);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) *reinterpret_cast<uint4*>(lds_tile + (j * 8 + wave) * 1024 + lane * 16) = v[j];
}

// The two transposing reads of one fragment as INLINE ASM, not waited for here.  With the builtin form hipcc sees an LDS load
// that may alias the LDS-DMA writes (global_load_lds) and puts s_waitcnt vmcnt(0) in front of every group of reads: that
// drains the three stages kept in flight and the kernel runs at memory-latency speed (measured: 750 TFLOP/s with the
// builtin).  The caller waits lgkmcnt(0) after the barrier and pins the order with sched_barrier, as in the NT kernel.
__device__ __forceinline__ void pp_tr_read(s16x4 (&f)[2], unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\t"
               "ds_read_b64_tr_b16 %1, %2 offset:2048"
               : "=&v"(f[0]), "=&v"(f[1])
               : "v"(addr)
               : "memory");
}
__device__ __forceinline__ bf16x8 pp_join(const s16x4 (&f)[2]) { return __builtin_shufflevector(f[0], f[1], 0, 1, 2, 3, 4, 5, 6, 7); }
__device__ __forceinline__ unsigned pp_lds_off(const char* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}

__device__ __forceinline__ bf16x8 pp_tr_frag(const char* base, int off) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + off));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + off + 4 * 512));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <bool CS>
__global__ __launch_bounds__(512, 1) void gemm_bf16_tn_pp_kernel(const TGroup grp) {
  __shared__ __attribute__((aligned(1024))) char smem[PSTAGES * PSTAGE_BYTES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  // which product of the launch this workgroup belongs to (wave-uniform; a single product is n = 1)
  int gi = 0;
#pragma unroll
  for (int k = 1; k < 4; ++k)
    if (k < grp.n && (int)blockIdx.x >= grp.base[k]) gi = k;
  const TArgs& p = grp.g[gi];
  const unsigned bid = blockIdx.x - (unsigned)grp.base[gi];
  // work map: items = split-major list of (split, tile); XCD x (= blockIdx & 7) takes the x-th run of consecutive items, i.e.
  // (mostly) one split: its ~32 co-resident workgroups read the same token rows of both operands through one L2.
  const int tiles = p.ntm * p.ntn, items = tiles * p.splitk;
  const int per = (items + 7) >> 3;
  const int item = (bid & 7) * per + (bid >> 3);
  if ((int)(bid >> 3) >= per || item >= items) return;
  const int split = item / tiles, tile = item - split * tiles;
  const int tm = tile / p.ntn, tn = tile - tm * p.ntn;
  const int m0 = tm * PBM, n0 = tn * PBN;
  const int ks = split * p.kper;
  const int ke = min(p.Kr, ks + p.kper);
  if (ks >= ke) return;
  const int nk = (ke - ks + PBK - 1) / PBK;

  // transposed-read addressing: lane (g, idx = 4q+pp): row 8g+q (second read +4 rows), 8-B piece pp of a 16-column block
  const int g = lane >> 4, idx = lane & 15, q = idx >> 2, pp = idx & 3;
  const int sw = (q << 1) | ((g & 1) << 3);
  const int lbase = (8 * g + q) * 512 + ((pp & 1) << 3), csub = pp >> 1;
  // Row tiles are taken in an order rotated by wc: accumulator slot i of a phase holds row tile (i + wc) & 3 of that phase.
  // The four waves of a half read the same A fragments anyway; with the rotation "the row tile whose bias gradient this
  // wave sums" is always fragment 0 -- a compile-time register, no per-phase select chain (which cost 12 % with colsum on).
  int aoff[8], boff[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) aoff[i] = lbase + ((((wr * 8 + (i & 4) + ((i + wc) & 3)) * 2 + csub) ^ sw) << 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) boff[j] = POP_BYTES + lbase + ((((wc * 4 + j) * 2 + csub) ^ sw) << 4);

  f32x4 acc[8][4], csum[2];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  csum[0] = csum[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // bias gradient: only the workgroups of the first column panel contribute (the others multiply by zeros)
  const short one_or_zero = tn == 0 ? (short)0x3F80 : (short)0;
  const bf16x8 ones = {one_or_zero, one_or_zero, one_or_zero, one_or_zero, one_or_zero, one_or_zero, one_or_zero, one_or_zero};

  auto stage = [&](int t) {
    char* buf = smem + (t % PSTAGES) * PSTAGE_BYTES;
    const int k0 = ks + t * PBK;
    if (k0 + PBK <= ke) {
      pp_stage_op(p.A, p.lda, k0, m0, buf, wave, lane);
      pp_stage_op(p.B, p.ldb, k0, n0, buf + POP_BYTES, wave, lane);
    } else {
      pp_stage_op_tail(p.A, p.lda, k0, ke, m0, buf, wave, lane);
      pp_stage_op_tail(p.B, p.ldb, k0, ke, n0, buf + POP_BYTES, wave, lane);
    }
  };
  stage(0);
  if (nk > 1) stage(1);
  if (nk > 2) stage(2);
  if (nk > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();        // this half runs one barrier behind from here on

  int slot = 0;
  const unsigned lds_base = pp_lds_off(smem);
  for (int t = 0; t < nk; ++t) {
    const unsigned cur = lds_base + slot * PSTAGE_BYTES;
    slot = slot + 1 == PSTAGES ? 0 : slot + 1;
#if DEMO_PP_PHASES == 2
    // ONE phase per stage: all 8 x 4 MFMA tiles of the wave (32 MFMAs) between two barriers -- 48 fragment registers instead of 32,
    // half the barriers and half the hand-overs of the MFMA pipe between the wave halves.  Stage t+1 must have landed before GLOBAL
    // barrier 2t+1 (its first reader, the leading half, is on the far side): the leading half waits for its pieces after its MFMAs,
    // before the closing barrier; the trailing half, one barrier behind, before its first barrier -- the same global barrier.
    s16x4 araw[8][2], braw[4][2];
    bf16x8 af[8], bfr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) pp_tr_read(braw[j], cur + boff[j]);
#pragma unroll
    for (int i = 0; i < 8; ++i) pp_tr_read(araw[i], cur + aoff[i]);
    if (t + 3 < nk) stage(t + 3);
    if (wr == 1) {
      if (t + 3 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) bfr[j] = pp_join(braw[j]);
#pragma unroll
    for (int i = 0; i < 8; ++i) af[i] = pp_join(araw[i]);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    if constexpr (CS) {
      // bias gradient, branch-free: wave wc sums output row tile wc of each 64-row half = fragments 0 and 4 (rotated order)
      csum[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], ones, csum[0], 0, 0, 0);
      csum[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[4], ones, csum[1], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 0) {
      if (t + 3 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
#else
    s16x4 araw[4][2], braw[4][2];
    bf16x8 af[4], bfr[4];
    // ---- phase 1: output rows 0..63 of the wave
#pragma unroll
    for (int j = 0; j < 4; ++j) pp_tr_read(braw[j], cur + boff[j]);
#pragma unroll
    for (int i = 0; i < 4; ++i) pp_tr_read(araw[i], cur + aoff[i]);
    if (t + 3 < nk) stage(t + 3);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) bfr[j] = pp_join(braw[j]);
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = pp_join(araw[i]);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    if constexpr (CS) {
      // bias gradient, branch-free (a wave-uniform `if` around an MFMA makes hipcc copy accumulators): wave wc sums output
      // row tile wc of this phase = its fragment 0 (rotated order) times ones: one extra MFMA per 16
      csum[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], ones, csum[0], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 2: output rows 64..127 of the wave; stage t+1 must have landed before the first barrier
#pragma unroll
    for (int i = 0; i < 4; ++i) pp_tr_read(araw[i], cur + aoff[4 + i]);
    if (t + 3 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = pp_join(araw[i]);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[4 + i][j], 0, 0, 0);
    if constexpr (CS) {
      csum[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], ones, csum[1], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  #endif
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();        // re-join the halves

  if constexpr (CS) {
    // csum[ii]: D[row = m_local = 4g + reg][any column]: output row tile ii*4 + wc of this wave's half
    if (idx == 0) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(p.colsum + m0 + wr * 128 + (ii * 4 + wc) * 16 + 4 * g + r, csum[ii][r]);
    }
  }
  if (p.partial) {
    // Two-pass split-K: the slice's partial tile goes to the workspace IN ACCUMULATOR ORDER -- one float4 per lane and MFMA tile, 1 KiB
    // contiguous per wave instruction, [item][wave][i][j][lane] -- instead of the output's row-major order, where a register's four values
    // are four rows x 64-byte runs (4-B stores, 1024 wave instructions per tile against 256 here).  Only tn_reduce_kernel reads the
    // workspace; it works in the same order and scatters the ONE final image into C.
    float4* dst = reinterpret_cast<float4*>(p.partial) + (int64_t)item * (PBM * PBN / 4) + wave * (32 * 64) + lane;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        dst[(i * 4 + j) * 64] = make_float4(acc[i][j][0] * p.alpha, acc[i][j][1] * p.alpha, acc[i][j][2] * p.alpha, acc[i][j][3] * p.alpha);
    return;
  }
  // D[row = m_local = 4g + reg][col = n_local = lane & 15]: per register 4 rows x 64 contiguous bytes
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wc * 64 + j * 16 + idx;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wr * 128 + ((i & 4) + ((i + wc) & 3)) * 16 + 4 * g + r;      // rotated row-tile order
        const float v = acc[i][j][r] * p.alpha;
        float* c = p.C + (int64_t)m * p.ldc + n;
        if (p.use_atomics) atomicAdd(c, v);
        else *c = p.accumulate ? *c + v : v;
      }
    }
  }
}

// second pass of the two-pass split-K form: C (+)= sum over slices of the partial tiles.  The workspace holds every (slice, tile) item in
// the first pass's accumulator order (see there); a thread owns one float4 of that order -- four consecutive output ROWS of one column,
// lane-adjacent threads adjacent columns -- sums it over the slices with 16-B loads and scatters the four values into C.
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ partial, int splitk, int ntm, int ntn, float* __restrict__ C,
                                                        int64_t ldc, int accumulate) {
  const int tiles = ntm * ntn;
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;          // float4 index: tile * 16384 + position inside the tile
  if (q >= (int64_t)tiles * (PBM * PBN / 4)) return;
  const int tile = (int)(q >> 14), e = (int)(q & 16383);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4* p4 = reinterpret_cast<const float4*>(partial) + q;
  const int64_t slice4 = (int64_t)tiles * (PBM * PBN / 4);
#pragma unroll 4
  for (int k = 0; k < splitk; ++k) {
    const float4 v = p4[k * slice4];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const int wave = e >> 11, i = (e >> 8) & 7, j = (e >> 6) & 3, lane = e & 63;
  const int wr = wave >> 2, wc = wave & 3, g = lane >> 4, idx = lane & 15;
  const int tm = tile / ntn, tn = tile - tm * ntn;
  const int m = tm * PBM + wr * 128 + ((i & 4) + ((i + wc) & 3)) * 16 + 4 * g;        // (rotated row-tile order of the first pass)
  const int n = tn * PBN + wc * 64 + j * 16 + idx;
  float* c = C + (int64_t)m * ldc + n;
  const float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
  for (int r = 0; r < 4; ++r) c[r * ldc] = accumulate ? c[r * ldc] + v[r] : v[r];
}

// the same for the products of a grouped launch: blocks [base[i], base[i+1]) reduce product i
struct RGroup {
  const float* partial[4];
  float* C[4];
  int64_t ldc[4];
  int splitk[4], ntm[4], ntn[4];
  int base[5];
  int n, accumulate;
};
__global__ __launch_bounds__(256) void tn_reduce_group_kernel(const RGroup r) {
  int gi = 0;
#pragma unroll
  for (int k = 1; k < 4; ++k)
    if (k < r.n && (int)blockIdx.x >= r.base[k]) gi = k;
  const int tiles = r.ntm[gi] * r.ntn[gi];
  const int64_t q = (int64_t)(blockIdx.x - r.base[gi]) * 256 + threadIdx.x;
  if (q >= (int64_t)tiles * (PBM * PBN / 4)) return;
  const int tile = (int)(q >> 14), e = (int)(q & 16383);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4* p4 = reinterpret_cast<const float4*>(r.partial[gi]) + q;
  const int64_t slice4 = (int64_t)tiles * (PBM * PBN / 4);
  const int splitk = r.splitk[gi];
#pragma unroll 4
  for (int k = 0; k < splitk; ++k) {
    const float4 v = p4[k * slice4];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const int wave = e >> 11, i = (e >> 8) & 7, j = (e >> 6) & 3, lane = e & 63;
  const int wr = wave >> 2, wc = wave & 3, g = lane >> 4, idx = lane & 15;
  const int ntn = r.ntn[gi], tm = tile / ntn, tn = tile - tm * ntn;
  const int m = tm * PBM + wr * 128 + ((i & 4) + ((i + wc) & 3)) * 16 + 4 * g;
  const int n = tn * PBN + wc * 64 + j * 16 + idx;
  const int64_t ldc = r.ldc[gi];
  float* c = r.C[gi] + (int64_t)m * ldc + n;
  const float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) c[rr * ldc] = r.accumulate ? c[rr * ldc] + v[rr] : v[rr];
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// split-K plan of the 256^2 form for one product: slices so that the grid is about one round of the 256 CUs, at least 16 stages each
inline void pp_plan(int Mo, int No, int Kr, int& ntm, int& ntn, int& splitk, int& kper) {
  ntm = Mo / PBM; ntn = No / PBN;
  const int tiles = ntm * ntn;
  const int kstages = (Kr + PBK - 1) / PBK;
  int sk = tiles >= 256 ? 1 : 256 / tiles;
  if (sk > kstages / 16) sk = kstages / 16 > 0 ? kstages / 16 : 1;
  const int per = (kstages + sk - 1) / sk;
  kper = per * PBK;
  splitk = (kstages + per - 1) / per;
}

// Split-K plan of a GROUP: ONE split count for every product (they share the reduction length, so all workgroups of the launch walk slices of
// equal length and the products' grids tile the CUs without a ragged tail), chosen so that tiles x splits fills whole rounds of the 256 CUs with
// the FEWEST slices: the four products of a trunk block (108 tiles) take 7 slices each = 756 workgroups = 2.95 rounds, against 252 + 252 + 243 +
// 252 = 999 workgroups of four different lengths under the per-product plans -- a quarter fewer partial tiles to store and to sum.
inline int group_split(int total_tiles, int Kr) {
  const int kstages = (Kr + PBK - 1) / PBK;
  int best = 1;
  double best_eff = 0.0;
  for (int R = 1; R <= 8; ++R) {
    int sk = 256 * R / total_tiles;
    if (sk < 1) continue;
    if (sk > kstages / 16) sk = kstages / 16 > 0 ? kstages / 16 : 1;      // at least 16 stages per workgroup
    const double wgs = (double)total_tiles * sk;
    const double eff = wgs / (256.0 * (int)((wgs + 255.0) / 256.0));
    if (eff > best_eff + 0.02) { best_eff = eff; best = sk; }              // a later (larger) plan must be clearly fuller to be taken
    if (best_eff >= 0.95) break;
  }
  return best;
}
inline void pp_plan_fixed(int Mo, int No, int Kr, int sk, int& ntm, int& ntn, int& splitk, int& kper) {
  ntm = Mo / PBM; ntn = No / PBN;
  const int kstages = (Kr + PBK - 1) / PBK;
  const int per = (kstages + sk - 1) / sk;
  kper = per * PBK;
  splitk = (kstages + per - 1) / per;
}

}  // namespace

static int g_tn_force = 0;     // bench hook (shared with demo_gemm_force_tile): 128 forces the 128^2 form
void demo_gemm_tn_force(int t) { g_tn_force = t; }

// floats of workspace with which demo_gemm_tn_bf16 takes the two-pass split-K form for this shape (0: it would not split)
extern "C" int64_t demo_gemm_tn_workspace(int32_t Mo, int32_t No, int32_t Kr) {
  if (Mo <= 0 || No <= 0 || Kr < 2048 || Mo % PBM || No % PBN) return 0;
  const int tiles = (Mo / PBM) * (No / PBN);
  const int kstages = (Kr + PBK - 1) / PBK;
  int splitk = tiles >= 256 ? 1 : 256 / tiles;
  if (splitk > kstages / 16) splitk = kstages / 16 > 0 ? kstages / 16 : 1;
  const int per = (kstages + splitk - 1) / splitk;
  splitk = (kstages + per - 1) / per;
  return splitk > 1 ? (int64_t)splitk * Mo * No : 0;
}

extern "C" int demo_gemm_tn_bf16(const void* A, const void* B, float* C, int32_t Mo, int32_t No, int32_t Kr, int64_t lda, int64_t ldb,
                                 int64_t ldc, float alpha, int32_t accumulate, float* colsum, float* workspace, int64_t workspace_elems,
                                 void* stream) {
  DEMO_REQUIRE(A && B && C, "demo_gemm_tn_bf16: null operand");
  DEMO_REQUIRE(Mo > 0 && No > 0 && Kr > 0 && Mo % BM == 0 && No % BN == 0, "demo_gemm_tn_bf16: Mo=%d No=%d must be multiples of 128", Mo, No);
  DEMO_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && al16(A) && al16(B), "demo_gemm_tn_bf16: operands must be 16-B aligned with ld %% 8 == 0");
  hipStream_t s = (hipStream_t)stream;
  TArgs p;
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C;
  p.Mo = Mo; p.No = No; p.Kr = Kr; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.alpha = alpha;
  p.partial = nullptr;
  if (g_tn_force != 128 && Mo % PBM == 0 && No % PBN == 0 && Kr >= 2048) {
    // 256^2 ping-pong form: split the reduction so that the grid is about one round of the 256 CUs
    p.ntm = Mo / PBM; p.ntn = No / PBN;
    const int tiles = p.ntm * p.ntn;
    const int kstages = (Kr + PBK - 1) / PBK;
    int splitk = tiles >= 256 ? 1 : 256 / tiles;
    if (splitk > kstages / 16) splitk = kstages / 16 > 0 ? kstages / 16 : 1;      // at least 16 stages per workgroup
    // (capping the slices per tile to cut the fp32-atomic traffic was measured: 42 -> 16 slices on a 512 x 768 output costs
    // 103 -> 119 us, 28 -> 16 on 768 x 768 costs 117 -> 128 us: filling the CUs wins)
    const int per = (kstages + splitk - 1) / splitk;
    p.kper = per * PBK;
    p.splitk = (kstages + per - 1) / per;
    p.use_atomics = p.splitk > 1;
    p.accumulate = accumulate;
    p.colsum = colsum;
    p.part_m = 0;
    // Two-pass split-K when the caller lends a workspace: every slice stores its 256x256 partial tile with plain stores and a
    // second small kernel sums the slices into C.  With fp32 atomics straight into C the ~16.5 M element-atomics of a launch
    // (252 workgroups x 65536) run at the L2 atomic units' rate, ~60 us per launch whatever the split count; the partial
    // image (<= 66 MB) is written and re-read at HBM speed instead.
    const int64_t need = (int64_t)p.splitk * Mo * No;
    const bool two_pass = p.splitk > 1 && workspace && workspace_elems >= need && (ldc % 4 == 0) && al16(C) && g_tn_force != 129;
    if (two_pass) p.partial = workspace;
    if (p.use_atomics && !accumulate && !two_pass) {
      hipError_t e = hipMemset2DAsync(C, (size_t)ldc * sizeof(float), 0, (size_t)No * sizeof(float), (size_t)Mo, s);
      if (e != hipSuccess) DEMO_FAIL(-2, "demo_gemm_tn_bf16: memset failed: %s", hipGetErrorString(e));
    }
    const unsigned nwg = 8u * (unsigned)((tiles * p.splitk + 7) / 8);
    TGroup grp;
    grp.g[0] = p;
    grp.n = 1;
    grp.base[0] = 0;
    grp.base[1] = grp.base[2] = grp.base[3] = grp.base[4] = (int)nwg;
    if (colsum) hipLaunchKernelGGL(gemm_bf16_tn_pp_kernel<true>, dim3(nwg), dim3(512), 0, s, grp);
    else hipLaunchKernelGGL(gemm_bf16_tn_pp_kernel<false>, dim3(nwg), dim3(512), 0, s, grp);
    DEMO_LAUNCH_CHECK("demo_gemm_tn_bf16(256)");
    if (two_pass) {
      hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)cdiv64((int64_t)Mo * No / 4, 256)), dim3(256), 0, s, workspace, p.splitk, p.ntm, p.ntn, C,
                         ldc, accumulate);
      DEMO_LAUNCH_CHECK("demo_gemm_tn_bf16(reduce)");
    }
    return 0;
  }
  p.ntm = Mo / BM; p.ntn = No / BN;
  const int tiles = p.ntm * p.ntn;
  const int ktiles = (Kr + BK - 1) / BK;
  int splitk = 1;
  if (tiles < 512) splitk = (512 + tiles - 1) / tiles;     // aim at ~2 workgroups per CU
  if (splitk > ktiles) splitk = ktiles;
  if (splitk > 32) splitk = 32;
  const int kt_per = (ktiles + splitk - 1) / splitk;
  p.kper = kt_per * BK;
  p.splitk = (ktiles + kt_per - 1) / kt_per;
  p.use_atomics = p.splitk > 1;
  p.accumulate = accumulate;
  p.colsum = colsum;
  if (p.use_atomics && !accumulate) {
    hipError_t e = hipMemset2DAsync(C, (size_t)ldc * sizeof(float), 0, (size_t)No * sizeof(float), (size_t)Mo, s);
    if (e != hipSuccess) DEMO_FAIL(-2, "demo_gemm_tn_bf16: memset failed: %s", hipGetErrorString(e));
  }
  p.part_m = p.ntm >= p.ntn;
  const int Pp = p.part_m ? p.ntm : p.ntn, Qq = p.part_m ? p.ntn : p.ntm;
  const unsigned nwg = 8u * (unsigned)(((Pp + 7) / 8) * Qq * p.splitk);
  hipLaunchKernelGGL(gemm_bf16_tn_kernel, dim3(nwg), dim3(256), 0, s, p);
  DEMO_LAUNCH_CHECK("demo_gemm_tn_bf16");
  return 0;
}

// ---- grouped launch (round 3) ----------------------------------------------------------------------------------------------------
// The weight gradients of ONE transformer block -- dW of fc2, fc1, qkv and the output projection: the same 49536 reduction rows, four
// operand pairs -- as ONE launch of the 256^2 form and ONE reduce launch.  Each product keeps its own split-K plan (so its workgroups
// read the same token rows through one L2, as in the single launch); what changes is that the four grids are one grid, longest
// slices first: the hardware starts a workgroup of the next product wherever one of the previous product retires, so (a) the tail of
// every product (a last wave of workgroups on a few CUs), (b) its ~10 us burst of partial-tile stores -- a retiring wave does not wait
// for its stores -- and (c) three launch ramps and three reduce launches disappear under the next product's main loop.
// items[i]: A (Kr, Mo) / B (Kr, No) bf16 row-major with leading dimensions lda / ldb, C fp32 (Mo, No) with ldc, colsum (Mo) or NULL --
// either every item has one or none.  Requires Mo, No multiples of 256 and Kr >= 2048 for every item (else -1: use demo_gemm_tn_bf16).
extern "C" int64_t demo_gemm_tn_group_workspace(const demo_tn_item* items, int32_t n, int32_t Kr) {
  if (!items || n < 1 || n > 4) return 0;
  int64_t need = 0;
  int total_tiles = 0;
  for (int i = 0; i < n; ++i) {
    if (items[i].Mo <= 0 || items[i].No <= 0 || items[i].Mo % PBM || items[i].No % PBN || Kr < 2048) return 0;
    total_tiles += (items[i].Mo / PBM) * (items[i].No / PBN);
  }
  const int gsk = n > 1 ? group_split(total_tiles, Kr) : 0;
  for (int i = 0; i < n; ++i) {
    int ntm, ntn, sk, kper;
    if (n > 1) pp_plan_fixed(items[i].Mo, items[i].No, Kr, gsk, ntm, ntn, sk, kper);
    else pp_plan(items[i].Mo, items[i].No, Kr, ntm, ntn, sk, kper);
    need += (int64_t)sk * items[i].Mo * items[i].No;
  }
  return need;
}

extern "C" int demo_gemm_tn_bf16_group(const demo_tn_item* items, int32_t n, int32_t Kr, float alpha, int32_t accumulate, float* workspace,
                                       int64_t workspace_elems, void* stream) {
  DEMO_REQUIRE(items && n >= 1 && n <= 4, "demo_gemm_tn_bf16_group: 1..4 items");
  const int64_t need = demo_gemm_tn_group_workspace(items, n, Kr);
  DEMO_REQUIRE(need > 0, "demo_gemm_tn_bf16_group: every item needs Mo, No multiples of 256 and Kr >= 2048");
  DEMO_REQUIRE(workspace && workspace_elems >= need, "demo_gemm_tn_bf16_group: workspace of %lld floats required", (long long)need);
  bool any_cs = false, all_cs = true;
  for (int i = 0; i < n; ++i) {
    DEMO_REQUIRE(items[i].A && items[i].B && items[i].C, "demo_gemm_tn_bf16_group: null operand in item %d", i);
    DEMO_REQUIRE(items[i].lda % 8 == 0 && items[i].ldb % 8 == 0 && al16(items[i].A) && al16(items[i].B),
                 "demo_gemm_tn_bf16_group: operands must be 16-B aligned with ld %% 8 == 0 (item %d)", i);
    any_cs = any_cs || items[i].colsum != nullptr;
    all_cs = all_cs && items[i].colsum != nullptr;
  }
  DEMO_REQUIRE(!any_cs || all_cs, "demo_gemm_tn_bf16_group: either every item has a colsum buffer or none");
  hipStream_t s = (hipStream_t)stream;
  // order: longest slices first
  int order[4] = {0, 1, 2, 3}, stages[4] = {0, 0, 0, 0};
  TArgs a[4];
  int64_t off = 0;
  int total_tiles = 0;
  for (int i = 0; i < n; ++i) total_tiles += (items[i].Mo / PBM) * (items[i].No / PBN);
  const int gsk = n > 1 ? group_split(total_tiles, Kr) : 0;
  for (int i = 0; i < n; ++i) {
    TArgs& p = a[i];
    p.A = (const bf16_t*)items[i].A; p.B = (const bf16_t*)items[i].B; p.C = items[i].C;
    p.Mo = items[i].Mo; p.No = items[i].No; p.Kr = Kr; p.lda = items[i].lda; p.ldb = items[i].ldb; p.ldc = items[i].ldc; p.alpha = alpha;
    if (n > 1) pp_plan_fixed(p.Mo, p.No, Kr, gsk, p.ntm, p.ntn, p.splitk, p.kper);
    else pp_plan(p.Mo, p.No, Kr, p.ntm, p.ntn, p.splitk, p.kper);
    p.part_m = 0; p.use_atomics = 0; p.accumulate = accumulate; p.colsum = items[i].colsum;
    p.partial = workspace + off;                            // (a product that is not split still goes through its one-slice image)
    off += (int64_t)p.splitk * p.Mo * p.No;
    stages[i] = p.kper;
  }
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j)
      if (stages[order[j]] > stages[order[i]]) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
  TGroup grp;
  RGroup red;
  grp.n = red.n = n;
  red.accumulate = accumulate;
  int base = 0, rbase = 0;
  for (int k = 0; k < 4; ++k) {
    grp.base[k] = base;
    red.base[k] = rbase;
    if (k < n) {
      const TArgs& p = a[order[k]];
      grp.g[k] = p;
      base += 8 * ((p.ntm * p.ntn * p.splitk + 7) / 8);
      red.partial[k] = p.partial; red.C[k] = p.C; red.ldc[k] = p.ldc; red.splitk[k] = p.splitk; red.ntm[k] = p.ntm; red.ntn[k] = p.ntn;
      rbase += (int)cdiv64((int64_t)p.Mo * p.No / 4, 256);
    } else {
      grp.g[k] = a[order[0]];
      red.partial[k] = nullptr; red.C[k] = nullptr; red.ldc[k] = 0; red.splitk[k] = 0; red.ntm[k] = red.ntn[k] = 0;
    }
  }
  grp.base[4] = base;
  red.base[4] = rbase;
  if (all_cs) hipLaunchKernelGGL(gemm_bf16_tn_pp_kernel<true>, dim3((unsigned)base), dim3(512), 0, s, grp);
  else hipLaunchKernelGGL(gemm_bf16_tn_pp_kernel<false>, dim3((unsigned)base), dim3(512), 0, s, grp);
  DEMO_LAUNCH_CHECK("demo_gemm_tn_bf16_group");
  hipLaunchKernelGGL(tn_reduce_group_kernel, dim3((unsigned)rbase), dim3(256), 0, s, red);
  DEMO_LAUNCH_CHECK("demo_gemm_tn_bf16_group(reduce)");
  return 0;
}
