// bf16 x bf16 -> fp32-accumulate MFMA GEMM for gfx950:  C[M,N] = epi(A[M,K] . B[N,K]^T)
// Both operands K-contiguous (the forward layout of every nn.Linear on the path: activations (rows, K) times
// weight (N, K); the backward dX = dY W uses a transposed bf16 weight copy so it takes the same kernel).
//
// Structure (cdna_hip_programming.md section 5):
//   * BM x BN output tile per workgroup, BK = 64; two configurations:
//       256 x 256, 8 waves as 2(M) x 4(N), 128 x 64 per wave (8 x 4 MFMA tiles)  -- the large products; halves the
//                 L2 -> LDS bytes per FLOP of the 128^2 tile (at full MFMA rate a 128^2 tile needs the whole L2 bandwidth);
//       128 x 128, 4 waves as 2 x 2, 64 x 64 per wave                              -- N not a multiple of 256 / small M;
//   * tiles staged HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR round trip), two LDS buffers, the next
//     K-tile in flight while the current one is multiplied;
//   * LDS image is lane-linear per wave instruction, so the bank-conflict XOR swizzle (16-B chunk ^= (row>>1)&7)
//     is applied on the per-lane SOURCE address and again on the ds_read_b128 fragment address (rule 21);
//   * MFMA operands are passed swapped (B fragment as the A operand) so that every lane ends up with 4 consecutive
//     output columns -> 16-B / 8-B epilogue accesses for bias, residual, act'(pre) and the store;
//   * workgroup -> tile map is XCD-aware: consecutive ids of one XCD walk a group of row-panels against one weight
//     panel so both stay in that XCD's L2.
#include "common.h"

namespace {

constexpr int BK = 64;
constexpr int GROUP_M = 8;

struct FArgs {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  const float* bias;
  const void* residual;        // fp32 (dtRes 0) or bf16 (dtRes DEMO_BF16), indexed like C
  const void* pre;
  void* pre_out;
  int pre_out_kind;            // 0: pre-activation, 1: act'(pre-activation), 2: act'(pre-activation) as DEMO_D8 codes (pre_out is a byte buffer)
  int dtRes;
  int M, N, K;
  int64_t lda, ldb, ldc;
  int64_t sA, sB, sC, sBias;   // batch strides (single level)
  int dtC, act, dtPre, act_grad_kind, accumulate;
  float alpha;
  int ntm, ntn;
  int group_m;                 // 256^2 kernel: row panels per tile-order group
  int dephase;                 // 256^2 kernel, tuning hook: first-round workgroups start ((id >> 3) & 3) * dephase * 64 clocks late
  int epi_stage;               // 256^2 kernel, register epilogue: bf16 outputs leave through an LDS image of the tile (whole-row stores)
  int colslice;                // 256^2 kernel: > 1 = every XCD keeps to ntn / colslice weight panels for the whole launch (see the kernel)
  int rev_m;                   // 256^2 kernel, tuning hook: walk the row panels last to first (the rows the producer kernel wrote last come first)
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// stage ROWS x 64 bf16 (ROWS*128 B) with NW waves: 1 KiB (8 rows) per wave instruction
// PERM (weight operand of the 256^2 kernel's register epilogue): LDS row l of every 64-row panel receives global row
// perm(l) = (j>>1)*32 + (x>>2)*8 + (j&1)*4 + (x&3) with j = l>>4 (the MFMA column tile) and x = l&15 (the row inside its fragment),
// so that the lane holding output columns {4*(lane>>4) + r} of the four column tiles owns 2 x 8 CONSECUTIVE columns of C.
__device__ __forceinline__ int perm64(int l) {
  const int j = l >> 4, x = l & 15;
  return ((j >> 1) << 5) | ((x >> 2) << 3) | ((j & 1) << 2) | (x & 3);
}
// AUX: cache-policy bits of the LDS-DMA (0 default, 2 = nt: streamed lines are the first to leave L2)
#ifndef DEMO_A_AUX
#define DEMO_A_AUX 0      // 256^2 kernel, ACTIVATION operand (read once per row panel; -DDEMO_A_AUX=2 is an A/B variant of the library)
#endif
template <int ROWS, int NW, bool PERM = false, int AUX = 0>
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ G, int64_t ld, int row0, int row_max, int k0,
                                           char* lds_tile, int wave, int lane) {
  constexpr int NSEG = ROWS / 8;
#pragma unroll
  for (int j = 0; j < NSEG / NW; ++j) {
    const int seg = j * NW + wave;
    const int r = seg * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int gr = row0 + (PERM ? (r & ~63) + perm64(r & 63) : r);
    gr = gr < row_max ? gr : row_max - 1;     // clamp: rows past the edge are loaded but never stored
    const bf16_t* src = G + (int64_t)gr * ld + k0 + c * 8;
    __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(lds_tile + seg * 1024), 16, 0, AUX);
  }
}

// register staging (T14): global -> VGPR now, VGPR -> swizzled LDS later
template <int ROWS, int NW>
__device__ __forceinline__ void load_tile_regs(uint4 (&r)[ROWS / 8 / NW], const bf16_t* __restrict__ G, int64_t ld, int row0,
                                               int row_max, int k0, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < ROWS / 8 / NW; ++j) {
    int gr = row0 + (j * NW + wave) * 8 + (lane >> 3);
    gr = gr < row_max ? gr : row_max - 1;
    r[j] = *reinterpret_cast<const uint4*>(G + (int64_t)gr * ld + k0 + (lane & 7) * 8);
  }
}
template <int ROWS, int NW>
__device__ __forceinline__ void write_tile_regs(const uint4 (&r)[ROWS / 8 / NW], char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < ROWS / 8 / NW; ++j) {
    const int row = (j * NW + wave) * 8 + (lane >> 3);
    *reinterpret_cast<uint4*>(lds_tile + row * 128 + (((lane & 7) ^ ((row >> 1) & 7)) << 4)) = r[j];
  }
}

// bias -> (save pre-activation) -> activation -> * act'(pre) -> + residual -> (+= C) -> store, for 4 consecutive columns
// (the bias quad is passed by value: in the LDS-staged epilogues a thread owns the same 8 columns for every row it stores, so
// the caller loads it once per output tile instead of once per row, a dependent L2 round trip each time)
__device__ __forceinline__ void epi4(const FArgs& p, const float4 b4, int64_t coff, int m, int n, float (&v)[4]) {
  v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
  const int64_t ci = coff + (int64_t)m * p.ldc + n;
  if (p.pre_out) {
    float s[4] = {v[0], v[1], v[2], v[3]};
    if (p.pre_out_kind >= 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = act_both(v[e], p.act, s[e]);
    }
    if (p.pre_out_kind == 2) *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(p.pre_out) + ci) = d8_pack4(s[0], s[1], s[2], s[3]);
    else if (p.dtC == DEMO_F32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.pre_out) + ci) = make_float4(s[0], s[1], s[2], s[3]);
    else io<bf16_t>::st4(reinterpret_cast<bf16_t*>(p.pre_out) + ci, make_float4(s[0], s[1], s[2], s[3]));
  }
  if (p.act != DEMO_ACT_NONE && !(p.pre_out && p.pre_out_kind >= 1)) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], p.act);
  }
  if (p.pre) {
    float4 pr;
    if (p.dtPre == DEMO_D8) pr = d8_unpack4(*reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(p.pre) + ci));
    else if (p.dtPre == DEMO_BF16) pr = io<bf16_t>::ld4(reinterpret_cast<const bf16_t*>(p.pre) + ci);
    else pr = io<float>::ld4(reinterpret_cast<const float*>(p.pre) + ci);
    v[0] *= act_grad(pr.x, p.act_grad_kind); v[1] *= act_grad(pr.y, p.act_grad_kind);
    v[2] *= act_grad(pr.z, p.act_grad_kind); v[3] *= act_grad(pr.w, p.act_grad_kind);
  }
  if (p.residual) {
    const float4 r4 = p.dtRes == DEMO_BF16 ? io<bf16_t>::ld4(reinterpret_cast<const bf16_t*>(p.residual) + ci)
                                           : *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.residual) + ci);
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

// ---- row-major epilogue of the LDS-staged forms ------------------------------------------------------------------------
// A thread owns one 8-column chunk (column n) and NIT rows r0, r0+RSTEP, ... of the fp32 tile image `ct` (row stride BN_
// floats, 16-B slots XOR-swizzled with row&7).  Rows are handled G at a time (4; 2 in the 256^2 kernel, whose accumulators are still live): first every global operand
// the epilogue consumes for the G rows is requested (act'(pre) source, residual, old C), then the rows are finished one by one.  The
// per-row form (load inside the row's arithmetic) paid one dependent HBM/L2 latency per row: the dY.W^T * act'(pre) product
// of the MLP backward ran at 456 TF against 870 for the same shape with a plain epilogue.  Stores are 16 B per thread and
// row for bf16 outputs (8 columns), 2 x 16 B for fp32.
struct EpiOps {
  float4 pre0, pre1, res0, res1, old0, old1;
};
__device__ __forceinline__ void bf16x8_to_f32(const uint4 r, float4& lo, float4& hi) {
  lo = make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                   __uint_as_float(r.y & 0xffff0000u));
  hi = make_float4(__uint_as_float(r.z << 16), __uint_as_float(r.z & 0xffff0000u), __uint_as_float(r.w << 16),
                   __uint_as_float(r.w & 0xffff0000u));
}
// (one v_cvt_pk_bf16_f32 per PAIR: the `f2bf(a) | f2bf(b) << 16` spelling compiled to two conversions, a shift and an or)
__device__ __forceinline__ uint4 f32x8_to_bf16(const v2f (&v)[4]) {
  return make_uint4(pack_bf16x2(v[0].x, v[0].y), pack_bf16x2(v[1].x, v[1].y), pack_bf16x2(v[2].x, v[2].y), pack_bf16x2(v[3].x, v[3].y));
}
__device__ __forceinline__ v2f lo2(const float4 a) { return (v2f){a.x, a.y}; }
__device__ __forceinline__ v2f hi2(const float4 a) { return (v2f){a.z, a.w}; }
// ACT / GK: activation and activation-gradient kinds as COMPILE-TIME constants (-1 = read them from p at run time).  With both
// as run-time switches every row instance carried QuickGELU and erf-GELU, forward and derivative: 157 KB of code for the
// 256^2 kernel against a 64-KB instruction cache, and its epilogue ran at instruction-fetch speed.
// 16-B output store with a compile-time cache policy (-DDEMO_STORE_POL=0 default / 1 nt / 2 sc1 = write-through, line dropped from L2).
// The output tile is written once and not read again by this kernel, while the weight panel is re-read from L2 by every row panel:
// plain stores leave 128 KB of dirty lines per tile in the XCD's 4-MB L2.  (A run-time switch put the epilogue's values in scratch.)
#ifndef DEMO_STORE_POL
#define DEMO_STORE_POL 0
#endif
// base: wave-uniform buffer base (p.C / p.pre_out), off: this lane's BYTE offset into it (< 2^31: the largest output is 304 MB)
// DEMO_EPI_ABLATE (measurement builds only, never the product): 1 = the epilogue computes everything but stores nothing, 2 = no epilogue
// at all (the accumulators are kept live so that the main loop is not optimised away): cost-structure ablations, guide rule 17
#ifndef DEMO_EPI_ABLATE
#define DEMO_EPI_ABLATE 0
#endif
// phases per K-tile of the 256^2 kernels' main loops: 2 (32 MFMAs between barriers) or 4 (16, the round-1 form; A/B variant)
// DEMO_FEED_ABLATE (measurement builds only, never the product; results are wrong by construction): 1 = the 256^2 NT main loop issues
// no LDS-DMA after its prologue (what the loop costs with NO operand feed), 2 = every workgroup stages the operands of tile (0, 0)
// (the same instructions, every load an L2 hit): separates the cost of issuing the feed from the cost of its misses
#ifndef DEMO_FEED_ABLATE
#define DEMO_FEED_ABLATE 0
#endif
#ifndef DEMO_PP_PHASES
#define DEMO_PP_PHASES 2
#endif
__device__ __forceinline__ void st16(void* base, int64_t off, const u32x4 v) {
#if DEMO_EPI_ABLATE == 1
  asm volatile("" ::"v"(v), "v"(off));
#elif DEMO_STORE_POL == 1
  __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(reinterpret_cast<char*>(base) + off));
#elif DEMO_STORE_POL == 2
  // buffer store with aux = 16 (sc1): the compiler counts it like any other store (an inline-asm global_store put the values in scratch)
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)off, 0, 16);
#else
  *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(base) + off) = v;
#endif
}
// 8-B store of eight DEMO_D8 codes (the second output of an fc1 forward that saves act'(x) in 8 bits); off: BYTE offset
__device__ __forceinline__ void st8(void* base, int64_t off, const uint2 v) {
#if DEMO_EPI_ABLATE == 1
  asm volatile("" ::"v"(v.x), "v"(v.y), "v"(off));
#else
  *reinterpret_cast<uint2*>(reinterpret_cast<char*>(base) + off) = v;
#endif
}
__device__ __forceinline__ uint2 d8_pack8(const v2f (&s)[4]) { return make_uint2(d8_pack4(s[0], s[1]), d8_pack4(s[2], s[3])); }
__device__ __forceinline__ void d8_unpack8(const uint2 r, float4& lo, float4& hi) { lo = d8_unpack4(r.x); hi = d8_unpack4(r.y); }
__device__ __forceinline__ u32x4 as_u32x4(const uint4 r) { return (u32x4){r.x, r.y, r.z, r.w}; }
__device__ __forceinline__ u32x4 as_u32x4(float a, float b, float c, float d) {
  return (u32x4){__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(d)};
}

// v: in = alpha * accumulator (+ bias when ADD_BIAS is false: the caller folded it into one fma), out = the 8 values of C as four PAIRS;
// s: the 8 values of the second output (pre-activation or act'), valid iff p.pre_out.  Everything is two-wide (v_pk_*_f32: two floats per
// lane and issue slot): with all eight waves of a tile in their epilogue at once and no MFMA to hide behind, the epilogue is VALU time.
// LOADS / SECOND: what the caller knows at compile time about the optional operands (LOADS: pp_epi_k's mask, -1 = look at p) and about a
// second output (0 none, 1 yes, -1 = look at p): a run-time `if (p.residual) v += r` compiles to the add AND a v_cndmask per element.
template <int ACT, int GK, bool ADD_BIAS = true, int LOADS = -1, int SECOND = -1>
__device__ __forceinline__ void epi8_compute(const FArgs& p, const float4 bz0, const float4 bz1, const EpiOps& x, v2f (&v)[4], v2f (&s)[4]) {
  const int act = ACT >= 0 ? ACT : p.act;
  const int gk = GK >= 0 ? GK : p.act_grad_kind;
  const bool has_pre = LOADS >= 0 ? (LOADS & 2) != 0 : (GK >= 0 ? GK != DEMO_ACT_NONE : p.pre != nullptr);
  const bool has_res = LOADS >= 0 ? (LOADS & 1) != 0 : p.residual != nullptr;
  const bool has_old = LOADS >= 0 ? false : (p.dtC == DEMO_F32 && p.accumulate);
  const bool second = SECOND >= 0 ? SECOND != 0 : p.pre_out != nullptr;
  if (ADD_BIAS) { v[0] += lo2(bz0); v[1] += hi2(bz0); v[2] += lo2(bz1); v[3] += hi2(bz1); }
  const bool save_deriv = second && p.pre_out_kind >= 1;
  if (second) {
    if (save_deriv) {                                      // activation and its derivative from one exp / rcp (or erf) per element
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = act_both2(v[e], act, s[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] = v[e];
    }
  }
  if (act != DEMO_ACT_NONE && !save_deriv) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = act_apply2(v[e], act);
  }
  if (has_pre) {
    const v2f pr[4] = {lo2(x.pre0), hi2(x.pre0), lo2(x.pre1), hi2(x.pre1)};
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= act_grad2(pr[e], gk);
  }
  if (has_res) { v[0] += lo2(x.res0); v[1] += hi2(x.res0); v[2] += lo2(x.res1); v[3] += hi2(x.res1); }
  if (has_old) { v[0] += lo2(x.old0); v[1] += hi2(x.old0); v[2] += lo2(x.old1); v[3] += hi2(x.old1); }
}
template <int ACT, int GK, bool ADD_BIAS = true, int LOADS = -1>
__device__ __forceinline__ void epi8_finish(const FArgs& p, const float4 bz0, const float4 bz1, const EpiOps& x, int64_t ci, v2f (&v)[4]) {
  v2f s[4];
  epi8_compute<ACT, GK, ADD_BIAS, LOADS>(p, bz0, bz1, x, v, s);
  if (p.pre_out) {
    if (p.pre_out_kind == 2) {
      st8(p.pre_out, ci, d8_pack8(s));
    } else if (p.dtC == DEMO_F32) {
      st16(p.pre_out, ci * 4, as_u32x4(s[0].x, s[0].y, s[1].x, s[1].y));
      st16(p.pre_out, ci * 4 + 16, as_u32x4(s[2].x, s[2].y, s[3].x, s[3].y));
    } else {
      st16(p.pre_out, ci * 2, as_u32x4(f32x8_to_bf16(s)));
    }
  }
  if (p.dtC == DEMO_F32) {
    st16(p.C, ci * 4, as_u32x4(v[0].x, v[0].y, v[1].x, v[1].y));
    st16(p.C, ci * 4 + 16, as_u32x4(v[2].x, v[2].y, v[3].x, v[3].y));
  } else {
    st16(p.C, ci * 2, as_u32x4(f32x8_to_bf16(v)));
  }
}
// The act'(pre) source (bf16) of ALL of a thread's rows can be requested before the tile is even staged through LDS (4
// registers per row): its HBM latency then hides under the staging instead of being paid once per row group.
template <int NIT, int RSTEP>
__device__ __forceinline__ bool epi_issue_pre(const FArgs& p, int r0, int mrow0, int n, int64_t coff, uint4 (&raw)[NIT]) {
  if (!(p.pre && p.dtPre == DEMO_BF16)) return false;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    int m = mrow0 + r0 + it * RSTEP;
    m = m < p.M ? m : p.M - 1;
    raw[it] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.pre) + coff + (int64_t)m * p.ldc + n);
  }
  return true;
}
template <int NIT, int RSTEP, int BN_, int G, int ACT, int GK>
__device__ __forceinline__ void epi_rows_k(const FArgs& p, const float* __restrict__ ct, int r0, int mrow0, int n, int c, const float4 bz0,
                                           const float4 bz1, int64_t coff, const uint4 (&epre)[NIT], bool have_epre) {
  const bool has_pre = GK >= 0 ? GK != DEMO_ACT_NONE : p.pre != nullptr;
  static_assert(NIT % G == 0, "rows are finished G at a time");
  static_assert(RSTEP % 8 == 0, "the LDS slot swizzle (row & 7) must be the same for every row of a thread");
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // per-row index arithmetic hoisted: a thread's rows are RSTEP apart, so the swizzled LDS slots are fixed and both the LDS and
  // the global address advance by constants (the epilogue is VALU-sensitive: ~50 instructions per row before this)
  const float* ctl = ct + r0 * BN_ + (((2 * c) ^ (r0 & 7)) << 2);
  const float* cth = ct + r0 * BN_ + (((2 * c + 1) ^ (r0 & 7)) << 2);
  const int mfirst = mrow0 + r0;
  const int64_t ci0 = coff + (int64_t)(mfirst < p.M ? mfirst : p.M - 1) * p.ldc + n;      // (clamped: edge tiles)
  const int64_t cstep = (int64_t)RSTEP * p.ldc;
  const int rows_ok = mfirst < p.M ? (p.M - 1 - mfirst) / RSTEP : -1;                     // rows 0..rows_ok of this thread exist
#pragma unroll
  for (int g0 = 0; g0 < NIT; g0 += G) {
    EpiOps x[G];
    int64_t ci[G];
#pragma unroll
    for (int u = 0; u < G; ++u) {
      ci[u] = (g0 + u) <= rows_ok ? ci0 + (g0 + u) * cstep : ci0;      // clamped for the loads; the store below is guarded
      x[u].pre0 = x[u].pre1 = x[u].res0 = x[u].res1 = x[u].old0 = x[u].old1 = z4;
    }
    if (has_pre) {
      if (have_epre) {
#pragma unroll
        for (int u = 0; u < G; ++u) bf16x8_to_f32(epre[g0 + u], x[u].pre0, x[u].pre1);
      } else if (p.dtPre == DEMO_BF16) {
        uint4 raw[G];
#pragma unroll
        for (int u = 0; u < G; ++u) raw[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.pre) + ci[u]);
#pragma unroll
        for (int u = 0; u < G; ++u) bf16x8_to_f32(raw[u], x[u].pre0, x[u].pre1);
      } else if (p.dtPre == DEMO_D8) {
        uint2 raw8[G];
#pragma unroll
        for (int u = 0; u < G; ++u) raw8[u] = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(p.pre) + ci[u]);
#pragma unroll
        for (int u = 0; u < G; ++u) d8_unpack8(raw8[u], x[u].pre0, x[u].pre1);
      } else {
#pragma unroll
        for (int u = 0; u < G; ++u) {
          x[u].pre0 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.pre) + ci[u]);
          x[u].pre1 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.pre) + ci[u] + 4);
        }
      }
    }
    if (p.residual) {
      if (p.dtRes == DEMO_BF16) {
        uint4 rraw[G];
#pragma unroll
        for (int u = 0; u < G; ++u) rraw[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.residual) + ci[u]);
#pragma unroll
        for (int u = 0; u < G; ++u) bf16x8_to_f32(rraw[u], x[u].res0, x[u].res1);
      } else {
#pragma unroll
        for (int u = 0; u < G; ++u) {
          x[u].res0 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.residual) + ci[u]);
          x[u].res1 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.residual) + ci[u] + 4);
        }
      }
    }
    if (p.accumulate && p.dtC == DEMO_F32) {
#pragma unroll
      for (int u = 0; u < G; ++u) {
        x[u].old0 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.C) + ci[u]);
        x[u].old1 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.C) + ci[u] + 4);
      }
    }
#pragma unroll
    for (int u = 0; u < G; ++u) {
      if ((g0 + u) > rows_ok) continue;
      const float4 lo = *reinterpret_cast<const float4*>(ctl + (g0 + u) * (RSTEP * BN_));
      const float4 hi = *reinterpret_cast<const float4*>(cth + (g0 + u) * (RSTEP * BN_));
      v2f v[4] = {lo2(lo), hi2(lo), lo2(hi), hi2(hi)};
      epi8_finish<ACT, GK>(p, bz0, bz1, x[u], ci[u], v);
    }
  }
}
// one compact copy per (activation, activation-gradient) pair the path uses; anything else takes the run-time form
template <int NIT, int RSTEP, int BN_, int G>
__device__ __forceinline__ void epi_rows(const FArgs& p, const float* __restrict__ ct, int r0, int mrow0, int n, int c, const float4 bz0,
                                         const float4 bz1, int64_t coff, const uint4 (&epre)[NIT], bool have_epre) {
  const int gk = p.pre ? p.act_grad_kind : DEMO_ACT_NONE;
  if (p.act == DEMO_ACT_NONE && gk == DEMO_ACT_NONE)
    epi_rows_k<NIT, RSTEP, BN_, G, DEMO_ACT_NONE, DEMO_ACT_NONE>(p, ct, r0, mrow0, n, c, bz0, bz1, coff, epre, have_epre);
  else if (p.act == DEMO_ACT_QUICKGELU && gk == DEMO_ACT_NONE)
    epi_rows_k<NIT, RSTEP, BN_, G, DEMO_ACT_QUICKGELU, DEMO_ACT_NONE>(p, ct, r0, mrow0, n, c, bz0, bz1, coff, epre, have_epre);
  else if (p.act == DEMO_ACT_NONE && gk == DEMO_ACT_QUICKGELU)
    epi_rows_k<NIT, RSTEP, BN_, G, DEMO_ACT_NONE, DEMO_ACT_QUICKGELU>(p, ct, r0, mrow0, n, c, bz0, bz1, coff, epre, have_epre);
  else if (p.act == DEMO_ACT_GELU && gk == DEMO_ACT_NONE)
    epi_rows_k<NIT, RSTEP, BN_, G, DEMO_ACT_GELU, DEMO_ACT_NONE>(p, ct, r0, mrow0, n, c, bz0, bz1, coff, epre, have_epre);
  else if (p.act == DEMO_ACT_NONE && gk == DEMO_ACT_GELU)
    epi_rows_k<NIT, RSTEP, BN_, G, DEMO_ACT_NONE, DEMO_ACT_GELU>(p, ct, r0, mrow0, n, c, bz0, bz1, coff, epre, have_epre);
  else if (p.act == DEMO_ACT_NONE && gk == DEMO_ACT_SAVED)
    epi_rows_k<NIT, RSTEP, BN_, G, DEMO_ACT_NONE, DEMO_ACT_SAVED>(p, ct, r0, mrow0, n, c, bz0, bz1, coff, epre, have_epre);
  else
    epi_rows_k<NIT, RSTEP, BN_, G, -1, -1>(p, ct, r0, mrow0, n, c, bz0, bz1, coff, epre, have_epre);
}

// ---- register epilogue of the 256^2 kernel --------------------------------------------------------------------------------------
// With the weight rows staged in perm64 order a lane (frow = lane & 15, g4 = lane >> 4) of wave (wr, wc) holds, for each of its 8
// row tiles i, C[m0 + wr*128 + i*16 + frow][n0 + wc*64 + h*32 + g4*8 + 0..7] for h = 0, 1 in acc[i][2h], acc[i][2h+1]: two 8-column
// chunks, each one 16-B bf16 store (or 2 x 16 B of fp32).  No LDS image of the tile and no barrier: the eight waves finish their rows
// independently (the LDS-staged form wrote and re-read 256 KB of LDS per tile behind four workgroup barriers), and four lanes of
// a store instruction cover 64 contiguous bytes of a row.  Rows are finished G row tiles at a time, every global operand of the
// group requested first.
// LOADS: which global operands besides the bias the epilogue reads, as a compile-time mask (1 fp32 residual, 5 = 1 | 4 bf16 residual, 2 act'(pre) source in bf16,
// 10 = 2 | 8 saved derivative as DEMO_D8 codes, 0 none) or -1 = decided at run time (any combination, G = 1: every unit then budgets all 24 operand registers).
// SINK 1 (bf16 outputs only): the finished values do not leave in the accumulator layout (16 rows x 64 B per store instruction: a
// CU's store path moves ~10 B/clk that way, tools/probes/store_probe.hip) but through an LDS image of the output tile -- 256 rows x
// 512 B over the main loop's ring, 16-B chunk index XOR (row & 31) -- from which every wave drains 32 whole rows, two per store
// instruction (~20 B/clk per CU).  SINK 2 (two outputs: fc1 forward) works the tile in two 128-row halves, each with a 64-KB image per output.
// SINK 3: the same with the second output as DEMO_D8 codes -- its half image is 128 rows x 256 B (8 B per lane and 8-column chunk; the 16-B
// slot index XOR (row & 15) keeps both the 8-B writes and the 16-B drain reads on distinct banks), drained four whole rows per instruction.
__device__ __forceinline__ unsigned pp_tile_slot(int row, int chunk) { return (unsigned)(row * 512 + ((chunk ^ (row & 31)) << 4)); }
__device__ __forceinline__ unsigned pp_d8_slot(int row, int chunk) { return (unsigned)(row * 256 + ((((chunk >> 1) ^ row) & 15) << 4) + ((chunk & 1) << 3)); }
__device__ __forceinline__ void pp_drain_half_d8(const FArgs& p, void* out, const char* img, int m0, int n0, int64_t coff, int half, int wave,
                                                 int lane) {
  u32x4 r[4];
  const int rsub = lane >> 4, s16 = lane & 15;
  // (hr & 15 = it * 4 + rsub: the slot is a per-lane base XOR a constant, the row a constant offset -- one VALU op per read)
  const unsigned rd0 = (unsigned)((wave * 16 + rsub) * 256 + ((s16 ^ rsub) << 4));
#pragma unroll
  for (int it = 0; it < 4; ++it) r[it] = *reinterpret_cast<const u32x4*>(img + (rd0 ^ (unsigned)(it * 64)) + it * 1024);
  // image row hr = wave * 16 + it * 4 + rsub is tile row (hr >> 6) * 128 + half * 64 + (hr & 63): linear in `it`, so the row offset advances
  // by a constant (a 64-bit multiply per row cost more VALU time than the epilogue's arithmetic)
  const int mf = m0 + (wave >> 2) * 128 + half * 64 + (wave & 3) * 16 + rsub;
  const int64_t off0 = coff + (int64_t)mf * p.ldc + n0 + s16 * 16, step = 4 * p.ldc;        // (byte offsets: one byte per element)
  if (m0 + 256 <= p.M) {                            // whole tile inside the matrix (all but the last row panel): no per-row guard
#pragma unroll
    for (int it = 0; it < 4; ++it) st16(out, off0 + it * step, r[it]);
  } else {
#pragma unroll
    for (int it = 0; it < 4; ++it)
      if (mf + it * 4 < p.M) st16(out, off0 + it * step, r[it]);
  }
}
__device__ __forceinline__ void pp_drain(const FArgs& p, void* out, const char* lds, int m0, int n0, int64_t coff, int wave, int lane) {
  u32x4 r[16];
  const int rsub = lane >> 5, chunk = lane & 31;
  // pp_tile_slot(wave * 32 + it * 2 + rsub, chunk): row & 31 = it * 2 + rsub, so the slot is a per-lane base XOR (it * 2) << 4 and the row a
  // constant offset (the spelled-out form cost five VALU ops per read)
  const unsigned rd0 = (unsigned)((wave * 32 + rsub) * 512 + ((chunk ^ rsub) << 4));
#pragma unroll
  for (int it = 0; it < 16; ++it) r[it] = *reinterpret_cast<const u32x4*>(lds + (rd0 ^ (unsigned)(it * 32)) + it * 1024);
  const int mf = m0 + wave * 32 + rsub;
  const int64_t off0 = (coff + (int64_t)mf * p.ldc + n0 + chunk * 8) * 2, step = 4 * p.ldc;       // two rows on, in bytes
  if (m0 + 256 <= p.M) {
#pragma unroll
    for (int it = 0; it < 16; ++it) st16(out, off0 + it * step, r[it]);
  } else {
#pragma unroll
    for (int it = 0; it < 16; ++it)
      if (mf + it * 2 < p.M) st16(out, off0 + it * step, r[it]);
  }
}
// half images of SINK 2: 128 image rows = the 64-row half `half` of both wave rows; 16 image rows per wave
__device__ __forceinline__ void pp_drain_half(const FArgs& p, void* out, const char* img, int m0, int n0, int64_t coff, int half, int wave,
                                              int lane) {
  u32x4 r[8];
  const int rsub = lane >> 5, chunk = lane & 31;
  // image row wave * 16 + it * 2 + rsub: row & 31 = (wave & 1) * 16 + it * 2 + rsub
  const unsigned rd0 = (unsigned)((wave * 16 + rsub) * 512 + ((chunk ^ (rsub + (wave & 1) * 16)) << 4));
#pragma unroll
  for (int it = 0; it < 8; ++it) r[it] = *reinterpret_cast<const u32x4*>(img + (rd0 ^ (unsigned)(it * 32)) + it * 1024);
  const int mf = m0 + (wave >> 2) * 128 + half * 64 + (wave & 3) * 16 + rsub;          // image row wave * 16 + it * 2 + rsub, as above
  const int64_t off0 = (coff + (int64_t)mf * p.ldc + n0 + chunk * 8) * 2, step = 4 * p.ldc;
  if (m0 + 256 <= p.M) {
#pragma unroll
    for (int it = 0; it < 8; ++it) st16(out, off0 + it * step, r[it]);
  } else {
#pragma unroll
    for (int it = 0; it < 8; ++it)
      if (mf + it * 2 < p.M) st16(out, off0 + it * step, r[it]);
  }
}
template <int G, int ACT, int GK, int LOADS, int SINK = 0>
__device__ __forceinline__ void pp_epi_k(const FArgs& p, const f32x4 (&acc)[8][4], int mrow, int ncol, int64_t coff, const float* bias,
                                         char* lds = nullptr, int m0 = 0, int n0 = 0) {
  const bool has_pre = LOADS >= 0 ? (LOADS & 2) != 0 : (GK >= 0 ? GK != DEMO_ACT_NONE : p.pre != nullptr);
  const bool has_res = LOADS >= 0 ? (LOADS & 1) != 0 : p.residual != nullptr;
  const bool has_old = LOADS >= 0 ? false : (p.accumulate && p.dtC == DEMO_F32);
  const bool pre_d8 = LOADS >= 0 ? (LOADS & 8) != 0 : p.dtPre == DEMO_D8;
  const bool pre_bf16 = LOADS >= 0 ? !pre_d8 : p.dtPre == DEMO_BF16;
  const bool res_bf16 = LOADS >= 0 ? (LOADS & 4) != 0 : p.dtRes == DEMO_BF16;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 bz[2][2] = {{z4, z4}, {z4, z4}};
  if (bias) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bz[h][0] = *reinterpret_cast<const float4*>(bias + ncol + h * 32);
      bz[h][1] = *reinterpret_cast<const float4*>(bias + ncol + h * 32 + 4);
    }
  }
  // LDS image slots (SINK >= 1): the lane's tile row is r0 + i * 16 with r0 = wr * 128 + frow (frow < 16) and its 8-column chunk c0 | h << 2 with
  // c0 = wc * 8 + g4 (g4 < 4), so pp_tile_slot / pp_d8_slot reduce to a per-lane base XOR a compile-time constant plus a constant row offset:
  // one VALU op per write (the spelled-out forms re-derive row & 31 and the XOR every time)
  const int r0 = mrow - m0, c0 = (ncol - n0) >> 3;
  const unsigned wbase_full = (unsigned)(r0 * 512 + ((c0 ^ (r0 & 15)) << 4));                                  // SINK 1: 256-row image
  const int hr0 = (r0 & 15) + (r0 >> 7) * 64;                                                                  // SINK 2 / 3: image row of row tile 0 / 4
  const unsigned wbase_half = (unsigned)(hr0 * 512 + ((c0 ^ (r0 & 15)) << 4));
  const unsigned wbase_d8 = (unsigned)(hr0 * 256 + ((((c0 >> 1) ^ r0) & 15) << 4) + ((c0 & 1) << 3));
  // a lane's rows are 16 apart: the element offset of row tile i is rb_first + i * 16 ldc (one 64-bit multiply per lane, not per row)
  const int64_t rb_first = coff + (int64_t)mrow * p.ldc + ncol, rb_last = coff + (int64_t)(p.M - 1) * p.ldc + ncol, step16 = 16 * p.ldc;
#pragma unroll
  for (int i0 = 0; i0 < 8; i0 += G) {
    EpiOps x[G][2];
    int64_t ci[G][2];
    bool ok[G];
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int m = mrow + (i0 + u) * 16;
      ok[u] = m < p.M;
      const int64_t rb = ok[u] ? rb_first + (i0 + u) * step16 : rb_last;          // clamped for the loads; stores are guarded
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        ci[u][h] = rb + h * 32;
        x[u][h].pre0 = x[u][h].pre1 = x[u][h].res0 = x[u][h].res1 = x[u][h].old0 = x[u][h].old1 = z4;
      }
    }
    if (has_pre) {
      if (pre_d8) {
        uint2 raw8[G][2];
#pragma unroll
        for (int u = 0; u < G; ++u)
#pragma unroll
          for (int h = 0; h < 2; ++h) raw8[u][h] = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(p.pre) + ci[u][h]);
#pragma unroll
        for (int u = 0; u < G; ++u)
#pragma unroll
          for (int h = 0; h < 2; ++h) d8_unpack8(raw8[u][h], x[u][h].pre0, x[u][h].pre1);
      } else if (pre_bf16) {
        uint4 raw[G][2];
#pragma unroll
        for (int u = 0; u < G; ++u)
#pragma unroll
          for (int h = 0; h < 2; ++h) raw[u][h] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.pre) + ci[u][h]);
#pragma unroll
        for (int u = 0; u < G; ++u)
#pragma unroll
          for (int h = 0; h < 2; ++h) bf16x8_to_f32(raw[u][h], x[u][h].pre0, x[u][h].pre1);
      } else {
#pragma unroll
        for (int u = 0; u < G; ++u)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            x[u][h].pre0 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.pre) + ci[u][h]);
            x[u][h].pre1 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.pre) + ci[u][h] + 4);
          }
      }
    }
    if (has_res) {
      if (res_bf16) {
        uint4 rraw[G][2];
#pragma unroll
        for (int u = 0; u < G; ++u)
#pragma unroll
          for (int h = 0; h < 2; ++h) rraw[u][h] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.residual) + ci[u][h]);
#pragma unroll
        for (int u = 0; u < G; ++u)
#pragma unroll
          for (int h = 0; h < 2; ++h) bf16x8_to_f32(rraw[u][h], x[u][h].res0, x[u][h].res1);
      } else {
#pragma unroll
        for (int u = 0; u < G; ++u)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            x[u][h].res0 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.residual) + ci[u][h]);
            x[u][h].res1 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.residual) + ci[u][h] + 4);
          }
      }
    }
    if (has_old) {
#pragma unroll
      for (int u = 0; u < G; ++u)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          x[u][h].old0 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.C) + ci[u][h]);
          x[u][h].old1 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.C) + ci[u][h] + 4);
        }
    }
#pragma unroll
    for (int u = 0; u < G; ++u) {
      if (SINK == 0 && !ok[u]) continue;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4 a = acc[i0 + u][2 * h], b = acc[i0 + u][2 * h + 1];
        // alpha * accumulator + bias, one v_pk_fma_f32 per pair (bias zero when it was the accumulators' initial value)
        v2f v[4] = {pk_fma((v2f){a[0], a[1]}, p.alpha, lo2(bz[h][0])), pk_fma((v2f){a[2], a[3]}, p.alpha, hi2(bz[h][0])),
                    pk_fma((v2f){b[0], b[1]}, p.alpha, lo2(bz[h][1])), pk_fma((v2f){b[2], b[3]}, p.alpha, hi2(bz[h][1]))};
        if constexpr (SINK == 0) {
          epi8_finish<ACT, GK, false, LOADS>(p, bz[h][0], bz[h][1], x[u][h], ci[u][h], v);
        } else {
          if constexpr (SINK == 3) {                   // activation + DEMO_D8 code of its derivative, each code packed as soon as it exists
            static_assert(ACT > 0 && LOADS == 0, "SINK 3 is the fc1 forward: bias, activation, two outputs");
            uint32_t w[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              v2f d0, d1;
              v[2 * e] = act_both2(v[2 * e], ACT, d0);
              v[2 * e + 1] = act_both2(v[2 * e + 1], ACT, d1);
              w[e] = d8_pack4(d0, d1);
              if (e == 0) __builtin_amdgcn_sched_barrier(0);     // four elements' transcendental temporaries at a time: the kernel sits at 256 VGPRs
            }
            const int iq = (i0 + u) & 3;
            *reinterpret_cast<uint2*>(lds + 65536 + (wbase_d8 ^ (unsigned)(h << 5)) + iq * 4096) = make_uint2(w[0], w[1]);
            *reinterpret_cast<u32x4*>(lds + (wbase_half ^ (unsigned)((h << 6) ^ ((iq & 1) << 8))) + iq * 8192) = as_u32x4(f32x8_to_bf16(v));
            continue;
          }
          v2f sv[4];
          epi8_compute<ACT, GK, false, LOADS, SINK == 1 ? 0 : 1>(p, bz[h][0], bz[h][1], x[u][h], v, sv);
          if constexpr (SINK == 1) {
            const int i = i0 + u;
            *reinterpret_cast<u32x4*>(lds + (wbase_full ^ (unsigned)((h << 6) ^ ((i & 1) << 8))) + i * 8192) = as_u32x4(f32x8_to_bf16(v));
          } else {                                   // two 128-row half images (C, second output): image row = wr * 64 + (i & 3) * 16 + frow
            const int iq = (i0 + u) & 3;
            const unsigned slot = (wbase_half ^ (unsigned)((h << 6) ^ ((iq & 1) << 8))) + iq * 8192;
            *reinterpret_cast<u32x4*>(lds + slot) = as_u32x4(f32x8_to_bf16(v));
            *reinterpret_cast<u32x4*>(lds + 65536 + slot) = as_u32x4(f32x8_to_bf16(sv));
          }
        }
      }
    }
    if constexpr (SINK == 2 || SINK == 3) {          // (G = 1) the wave's row tiles 0..3, then 4..7: drain both half images
      if (i0 + G == 4 || i0 + G == 8) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        __syncthreads();
        pp_drain_half(p, p.C, lds, m0, n0, coff, (i0 + G) / 4 - 1, wave, lane);
        if constexpr (SINK == 2) pp_drain_half(p, p.pre_out, lds + 65536, m0, n0, coff, (i0 + G) / 4 - 1, wave, lane);
        else pp_drain_half_d8(p, p.pre_out, lds + 65536, m0, n0, coff, (i0 + G) / 4 - 1, wave, lane);
        if (i0 + G == 4) __syncthreads();
      }
    }
  }
  if constexpr (SINK == 1) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    pp_drain(p, p.C, lds, m0, n0, coff, wave, lane);
  }
}
// FAM: the kernel is instantiated once per epilogue FAMILY so that neither instance carries (and allocates registers for) the other's
// variants -- the kernel sits at 256 VGPRs, and the DEMO_D8 variants added to a single instance pushed two per-lane constants into scratch:
//   0: descriptors WITHOUT a second output and WITHOUT an act' operand (bias / activation / residual / accumulate: qkv, proj, fc2, dX products)
//   1: the MLP's special epilogues -- pre_out (fc1 forward saving the pre-activation, act', or its DEMO_D8 codes) or act_grad_pre (dY W2^T * act')
template <int FAM>
__device__ __forceinline__ void pp_epi(const FArgs& p, const f32x4 (&acc)[8][4], int mrow, int ncol, int64_t coff, const float* bias,
                                       char* lds = nullptr, int m0 = 0, int n0 = 0) {
  const int gk = p.pre ? p.act_grad_kind : DEMO_ACT_NONE;
  const bool old = p.accumulate && p.dtC == DEMO_F32;
  const bool image = lds && p.epi_stage && p.dtC == DEMO_BF16;      // whole-row stores through an LDS image of the tile (bf16 outputs)
  if constexpr (FAM == 0) {
    if (image && !p.residual) {
      if (p.act == DEMO_ACT_NONE) { pp_epi_k<1, DEMO_ACT_NONE, DEMO_ACT_NONE, 0, 1>(p, acc, mrow, ncol, coff, bias, lds, m0, n0); return; }
      if (p.act == DEMO_ACT_QUICKGELU) { pp_epi_k<1, DEMO_ACT_QUICKGELU, DEMO_ACT_NONE, 0, 1>(p, acc, mrow, ncol, coff, bias, lds, m0, n0); return; }
    } else if (image && p.dtRes == DEMO_BF16 && p.act == DEMO_ACT_NONE) {
      pp_epi_k<4, DEMO_ACT_NONE, DEMO_ACT_NONE, 5, 1>(p, acc, mrow, ncol, coff, bias, lds, m0, n0);
      return;
    }
    if (!p.residual && !old) {                                 // nothing to read but the bias: trunk qkv forward, plain dX products
      if (p.act == DEMO_ACT_NONE) pp_epi_k<1, DEMO_ACT_NONE, DEMO_ACT_NONE, 0>(p, acc, mrow, ncol, coff, bias);
      else if (p.act == DEMO_ACT_QUICKGELU) pp_epi_k<1, DEMO_ACT_QUICKGELU, DEMO_ACT_NONE, 0>(p, acc, mrow, ncol, coff, bias);
      else if (p.act == DEMO_ACT_GELU) pp_epi_k<1, DEMO_ACT_GELU, DEMO_ACT_NONE, 0>(p, acc, mrow, ncol, coff, bias);
      else pp_epi_k<1, -1, -1, -1>(p, acc, mrow, ncol, coff, bias);
    } else if (p.residual && !old && p.act == DEMO_ACT_NONE) {       // proj / fc2 forward: + residual (fp32, or the bf16 stream)
      if (p.dtRes == DEMO_BF16) pp_epi_k<4, DEMO_ACT_NONE, DEMO_ACT_NONE, 5>(p, acc, mrow, ncol, coff, bias);
      else pp_epi_k<4, DEMO_ACT_NONE, DEMO_ACT_NONE, 1>(p, acc, mrow, ncol, coff, bias);
    } else {
      pp_epi_k<1, -1, -1, -1>(p, acc, mrow, ncol, coff, bias);
    }
  } else {
    if (image && !p.residual) {
      if (!p.pre && p.pre_out && p.pre_out_kind == 2) {        // fc1 forward: two outputs, the second as DEMO_D8 codes of act'
        if (p.act == DEMO_ACT_QUICKGELU) { pp_epi_k<1, DEMO_ACT_QUICKGELU, DEMO_ACT_NONE, 0, 3>(p, acc, mrow, ncol, coff, bias, lds, m0, n0); return; }
        if (p.act == DEMO_ACT_GELU) { pp_epi_k<1, DEMO_ACT_GELU, DEMO_ACT_NONE, 0, 3>(p, acc, mrow, ncol, coff, bias, lds, m0, n0); return; }
      } else if (!p.pre && p.pre_out) {                        // fc1 forward: two outputs (pre-activation or act')
        if (p.act == DEMO_ACT_QUICKGELU) { pp_epi_k<1, DEMO_ACT_QUICKGELU, DEMO_ACT_NONE, 0, 2>(p, acc, mrow, ncol, coff, bias, lds, m0, n0); return; }
        if (p.act == DEMO_ACT_GELU) { pp_epi_k<1, DEMO_ACT_GELU, DEMO_ACT_NONE, 0, 2>(p, acc, mrow, ncol, coff, bias, lds, m0, n0); return; }
      } else if (p.pre && !p.pre_out && p.act == DEMO_ACT_NONE) {
        if (p.dtPre == DEMO_D8 && gk == DEMO_ACT_SAVED) { pp_epi_k<4, DEMO_ACT_NONE, DEMO_ACT_SAVED, 10, 1>(p, acc, mrow, ncol, coff, bias, lds, m0, n0); return; }   // MLP backward on an 8-bit saved derivative
        if (p.dtPre == DEMO_BF16 && gk == DEMO_ACT_SAVED) { pp_epi_k<4, DEMO_ACT_NONE, DEMO_ACT_SAVED, 2, 1>(p, acc, mrow, ncol, coff, bias, lds, m0, n0); return; }
        if (p.dtPre == DEMO_BF16 && gk == DEMO_ACT_QUICKGELU) { pp_epi_k<4, DEMO_ACT_NONE, DEMO_ACT_QUICKGELU, 2, 1>(p, acc, mrow, ncol, coff, bias, lds, m0, n0); return; }
      }
    }
    const bool plain_pre = p.pre && !p.pre_out && !p.residual && !old && p.act == DEMO_ACT_NONE;
    if (plain_pre && p.dtPre == DEMO_BF16 && gk == DEMO_ACT_QUICKGELU) {
      pp_epi_k<4, DEMO_ACT_NONE, DEMO_ACT_QUICKGELU, 2>(p, acc, mrow, ncol, coff, bias);   // MLP backward: (dY W2^T) * act'(pre)
    } else if (plain_pre && p.dtPre == DEMO_BF16 && gk == DEMO_ACT_GELU) {
      pp_epi_k<4, DEMO_ACT_NONE, DEMO_ACT_GELU, 2>(p, acc, mrow, ncol, coff, bias);
    } else if (plain_pre && p.dtPre == DEMO_BF16 && gk == DEMO_ACT_SAVED) {
      pp_epi_k<4, DEMO_ACT_NONE, DEMO_ACT_SAVED, 2>(p, acc, mrow, ncol, coff, bias);       // MLP backward on a saved derivative: one multiply
    } else if (plain_pre && p.dtPre == DEMO_D8 && gk == DEMO_ACT_SAVED) {
      pp_epi_k<4, DEMO_ACT_NONE, DEMO_ACT_SAVED, 10>(p, acc, mrow, ncol, coff, bias);
    } else if (!p.pre && !p.residual && !old && p.act == DEMO_ACT_QUICKGELU) {             // fc1 forward without the image (epi_stage off, fp32 C)
      pp_epi_k<1, DEMO_ACT_QUICKGELU, DEMO_ACT_NONE, 0>(p, acc, mrow, ncol, coff, bias);
    } else if (!p.pre && !p.residual && !old && p.act == DEMO_ACT_GELU) {
      pp_epi_k<1, DEMO_ACT_GELU, DEMO_ACT_NONE, 0>(p, acc, mrow, ncol, coff, bias);
    } else {
      pp_epi_k<1, -1, -1, -1>(p, acc, mrow, ncol, coff, bias);
    }
  }
}

__device__ __forceinline__ unsigned lds_off(const char* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
// four ds_read_b128 of tile rows 0 / 16 / 32 / 48 (2048-B apart) from one base; NOT waited for here
__device__ __forceinline__ void lds_read4(bf16x8 (&f)[4], unsigned base) {
  asm volatile("ds_read_b128 %0, %4\n\t"
               "ds_read_b128 %1, %4 offset:2048\n\t"
               "ds_read_b128 %2, %4 offset:4096\n\t"
               "ds_read_b128 %3, %4 offset:6144"
               : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3])
               : "v"(base)
               : "memory");
}

// two ds_read_b128 of tile rows 0 / 16 from one base; NOT waited for here
__device__ __forceinline__ void lds_read2(bf16x8 (&f)[2], unsigned base) {
  asm volatile("ds_read_b128 %0, %2\n\t"
               "ds_read_b128 %1, %2 offset:2048"
               : "=&v"(f[0]), "=&v"(f[1])
               : "v"(base)
               : "memory");
}

// one ds_read_b128 at a compile-time byte offset from a base; NOT waited for here
template <int OFF> __device__ __forceinline__ void lds_read1(bf16x8& f, unsigned base) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(f) : "v"(base), "n"(OFF) : "memory");
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else static_assert(N == 0, "add the literal");
}

// NSTAGE LDS buffers; tiles t+1 .. t+NSTAGE-1 are in flight (LDS-DMA) while tile t is multiplied.  One raw s_barrier per
// K-step; the wait before it is a COUNTED vmcnt that leaves the younger tiles' loads outstanding (a __syncthreads()
// would drain them: cdna_hip_programming.md "Pipelining across barriers").
template <int BM, int BN, int WM, int WN, int NSTAGE, bool RS = false>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN * 64 * ((BM + BN) * BK * 2 * NSTAGE > 80 * 1024 ? 1 : 2)) / 256)
void gemm_bf16_nt_kernel(const FArgs p) {
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;       // MFMA tiles per wave
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, BUF_BYTES = A_BYTES + B_BYTES;
  constexpr int LPT = (BM + BN) / 8 / NW;                   // LDS-DMA instructions per wave per K-tile
  __shared__ __attribute__((aligned(1024))) char smem[NSTAGE * BUF_BYTES];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave / WN, wc = wave % WN;

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

  // The bias is the accumulators' initial value (alpha == 1), as in the 256^2 kernel (see there): one summation order for every tile
  // configuration -- the dispatcher's choice must not change a result -- and no bias load or add in the epilogue.  The lane's acc[i][j]
  // holds columns n0 + wc*TN*16 + j*16 + (lane >> 4)*4 + 0..3 of its rows.
#ifndef DEMO_BIAS_INIT
#define DEMO_BIAS_INIT 1
#endif
  const bool bias_in_acc = DEMO_BIAS_INIT && p.bias != nullptr && p.alpha == 1.0f;
  f32x4 acc[TM][TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (bias_in_acc) b4 = *reinterpret_cast<const f32x4*>(p.bias + z * p.sBias + n0 + wc * (TN * 16) + j * 16 + (lane >> 4) * 4);
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[i][j] = b4;
  }

  // per-lane fragment byte offsets inside a tile (row part + swizzled chunk for k-step 0 / 1)
  const int frow = lane & 15;
  const int sw = (lane >> 1) & 7;
  const int foff0 = frow * 128 + (((lane >> 4)) ^ sw) * 16;
  const int foff1 = frow * 128 + ((4 + (lane >> 4)) ^ sw) * 16;

  const int nk = p.K / BK;
  uint4 ra[RS ? BM / 8 / NW : 1], rb[RS ? BN / 8 / NW : 1];      // register staging (RS): one K-tile in VGPRs
  if constexpr (RS) {
    static_assert(NSTAGE == 2, "register staging uses two LDS buffers");
    load_tile_regs<BM, NW>(ra, A, p.lda, m0, p.M, 0, wave, lane);
    load_tile_regs<BN, NW>(rb, B, p.ldb, n0, p.N, 0, wave, lane);
    write_tile_regs<BM, NW>(ra, smem, wave, lane);
    write_tile_regs<BN, NW>(rb, smem + A_BYTES, wave, lane);
    __syncthreads();
  } else {
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) {
      if (s < nk) {
        stage_tile<BM, NW>(A, p.lda, m0, p.M, s * BK, smem + s * BUF_BYTES, wave, lane);
        stage_tile<BN, NW>(B, p.ldb, n0, p.N, s * BK, smem + s * BUF_BYTES + A_BYTES, wave, lane);
      }
    }
  }

  int slot = 0;                                   // LDS buffer of tile t
  for (int t = 0; t < nk; ++t) {
    if constexpr (RS) {
      // T14 async-stage split: issue tile t+1's global loads now (they fly under this tile's MFMAs) and write them
      // to the other LDS buffer after the multiply; one barrier per K-step.
      if (t + 1 < nk) {
        load_tile_regs<BM, NW>(ra, A, p.lda, m0, p.M, (t + 1) * BK, wave, lane);
        load_tile_regs<BN, NW>(rb, B, p.ldb, n0, p.N, (t + 1) * BK, wave, lane);
      }
    } else {
      // tile t has landed once at most `younger` tiles' loads of this wave are still outstanding
      const int younger = min(NSTAGE - 2, nk - 1 - t);
      if (NSTAGE >= 4 && younger >= 2) wait_vmcnt<2 * LPT>();
      else if (NSTAGE >= 3 && younger >= 1) wait_vmcnt<LPT>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();               // every wave's part of tile t landed; every wave is done with tile t-1
      if (t + NSTAGE - 1 < nk) {
        const int ns = slot == 0 ? NSTAGE - 1 : slot - 1;        // buffer of tile t-1 == buffer of tile t+NSTAGE-1
        char* nxt = smem + ns * BUF_BYTES;
        stage_tile<BM, NW>(A, p.lda, m0, p.M, (t + NSTAGE - 1) * BK, nxt, wave, lane);
        stage_tile<BN, NW>(B, p.ldb, n0, p.N, (t + NSTAGE - 1) * BK, nxt + A_BYTES, wave, lane);
      }
    }
    char* cur = smem + slot * BUF_BYTES;
    slot = slot + 1 == NSTAGE ? 0 : slot + 1;
    const char* la = cur + wr * (TM * 16 * 128);
    const char* lb = cur + A_BYTES + wc * (TN * 16 * 128);
    if constexpr (TM == 4 && TN == 4) {
      // all 2*(TM+TN) fragment reads of the K-step are issued before the first MFMA: one exposed LDS latency per
      // K-step instead of one per 8-MFMA group (the compiler then waits with counted lgkmcnt as data arrives)
      // The reads are inline asm so that the waits can be COUNTED by hand (hipcc waits lgkmcnt(0) for its own loads
      // here): k-half 0 is multiplied while k-half 1 is still arriving.  Order is pinned by sched_barrier(0) after
      // each wait (cdna_hip_programming.md 5.7 form (iii), rule 18).
      static_assert(TM == 4 && TN == 4, "hand-scheduled K-step is written for 4x4 MFMA tiles per wave");
      bf16x8 af[2][4], bfr[2][4];
      const unsigned ab0 = lds_off(la + foff0), ab1 = lds_off(la + foff1);
      const unsigned bb0 = lds_off(lb + foff0), bb1 = lds_off(lb + foff1);
      lds_read4(bfr[0], bb0);
      lds_read4(af[0], ab0);
      lds_read4(bfr[1], bb1);
      lds_read4(af[1], ab1);
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[0][j], af[0][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);          // keep k-half 0's MFMAs above the wait for k-half 1
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[1][j], af[1][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    } else if constexpr (TM == 5 && TN == 4) {
      // 160 x 128 tile (the row-split remainders: 3 LDS buffers, two K-tiles in flight): the 18 fragment reads of the K-step as inline asm --
      // a compiler-visible LDS load is ordered behind the pending LDS-DMA writes with vmcnt(0), which would drain the tiles in flight
      bf16x8 af[2][4], a5[2], bfr[2][4];              // row tiles 0..3 and row tile 4 of the wave, both k-halves
      const unsigned ab0 = lds_off(la + foff0), ab1 = lds_off(la + foff1);
      const unsigned bb0 = lds_off(lb + foff0), bb1 = lds_off(lb + foff1);
      lds_read4(bfr[0], bb0);
      lds_read4(af[0], ab0);
      lds_read1<8192>(a5[0], ab0);
      lds_read4(bfr[1], bb1);
      lds_read4(af[1], ab1);
      lds_read1<8192>(a5[1], ab1);
      asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[4][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kk][j], a5[kk], acc[4][j], 0, 0, 0);
        if (kk == 0) {
          __builtin_amdgcn_sched_barrier(0);          // keep k-half 0's MFMAs above the wait for k-half 1
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    } else if constexpr (TM == 2 && TN == 2) {
      // 64^2 tile: the eight fragment reads as inline asm (a compiler-visible LDS load is ordered behind the LDS-DMA writes with
      // vmcnt(0), which would drain the three K-tiles kept in flight)
      bf16x8 af[2][2], bfr[2][2];
      lds_read2(bfr[0], lds_off(lb + foff0));
      lds_read2(af[0], lds_off(la + foff0));
      lds_read2(bfr[1], lds_off(lb + foff1));
      lds_read2(af[1], lds_off(la + foff1));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int fo = kk ? foff1 : foff0;
        bf16x8 af[TM], bfr[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(lb + j * 2048 + fo);
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(la + i * 2048 + fo);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      }
    }
    if constexpr (RS) {
      if (t + 1 < nk) {
        char* nxt = smem + slot * BUF_BYTES;      // `slot` already points at tile t+1's buffer
        write_tile_regs<BM, NW>(ra, nxt, wave, lane);
        write_tile_regs<BN, NW>(rb, nxt + A_BYTES, wave, lane);
      }
      __syncthreads();
    }
  }

  // ---- epilogue ---------------------------------------------------------------------------------
  const int64_t coff = z * p.sC;
  const float* bias = (p.bias && !bias_in_acc) ? p.bias + z * p.sBias : nullptr;
  if constexpr (BM * BN * 4 <= NSTAGE * BUF_BYTES) {
    // Stage the fp32 tile through LDS (the K-loop buffers are free now) and write whole rows: every wave instruction
    // then covers 1-2 full tile rows with 16-B accesses per lane for bias / act'(pre) / residual / C.  The direct form
    // (16 narrow 8-B stores per lane to 16 different rows) is store-ISSUE bound: ~4 us per 128^2 tile (guide, T21).
    constexpr int CPR = BN / 8;                           // 8-float chunks per row
    constexpr int NIT = (BM * CPR) / (NW * 64), RSTEP = NW * 64 / CPR;
    static_assert((NW * 64) % CPR == 0, "a thread must own the same column chunk in every pass");
    const int c = threadIdx.x % CPR;
    // requested before the staging: bias for this thread's columns, the act'(pre) source for all of its rows
    float4 bz0 = make_float4(0.f, 0.f, 0.f, 0.f), bz1 = bz0;
    if (bias) {
      bz0 = *reinterpret_cast<const float4*>(bias + n0 + c * 8);
      bz1 = *reinterpret_cast<const float4*>(bias + n0 + c * 8 + 4);
    }
    uint4 epre[NIT];
    const bool have_epre = epi_issue_pre<NIT, RSTEP>(p, threadIdx.x / CPR, m0, n0 + c * 8, coff, epre);
    __syncthreads();                                      // all waves are done reading the last K-tile
    float* ct = reinterpret_cast<float*>(smem);
    const int g4 = lane >> 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int r = wr * (TM * 16) + i * 16 + frow;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int slot = (wc * TN * 4 + j * 4 + g4) ^ (frow & 7);
        *reinterpret_cast<float4*>(ct + r * BN + slot * 4) =
            make_float4(acc[i][j][0] * p.alpha, acc[i][j][1] * p.alpha, acc[i][j][2] * p.alpha, acc[i][j][3] * p.alpha);
      }
    }
    __syncthreads();
    epi_rows<NIT, RSTEP, BN, (NIT < 4 ? NIT : (NIT % 4 == 0 ? 4 : 2))>(p, ct, threadIdx.x / CPR, m0, n0 + c * 8, c, bz0, bz1, coff, epre, have_epre);
  } else {
    // direct form: lane holds C[m][n .. n+3] for each of its MFMA tiles
    const int ng = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wr * (TM * 16) + i * 16 + frow;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float v[4] = {acc[i][j][0] * p.alpha, acc[i][j][1] * p.alpha, acc[i][j][2] * p.alpha, acc[i][j][3] * p.alpha};
        const int n = n0 + wc * (TN * 16) + j * 16 + ng;
        epi4(p, bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f), coff, m, n, v);
      }
    }
  }
}

// ---- 256 x 256 ping-pong kernel ----------------------------------------------------------------------------------------
// 8 waves as 2 (M) x 4 (N), 128 x 64 per wave (8 x 4 MFMA tiles, 128 accumulator registers), BK = 64, two 64-KiB LDS
// buffers, one workgroup per CU.  Half the L2 -> LDS bytes per FLOP of the 128^2 tile, which is what bounds that kernel
// on the path's shapes (a streaming 50k-row activation operand: ~1.6 us per K-tile of load latency).
//
// A K-tile is worked in four phases, one C quadrant (64 x 32 per wave, 16 MFMAs over K = 64) each:
//     phase = { issue this quadrant's ds_reads (+ a share of the NEXT K-tile's LDS-DMA) ; s_barrier ;
//               lgkmcnt(0) ; 16 MFMAs ; s_barrier }
// and the two wave halves (wr = 0 / 1: one wave of each on every SIMD) run ONE BARRIER APART, so while one half
// multiplies the other half reads LDS -- the MFMA pipe and the LDS pipe of a SIMD are both busy without any
// intra-wave software pipelining (cdna_hip_programming.md section 5, "8-phase template": same phase body; here the whole
// next K-tile is requested at the top of the current one, the largest lead a 2-buffer scheme allows, because the
// activation operand misses L2).
//   P1: read a0 (4 row tiles x 2 k-halves), b0 (2 col tiles x 2)      stage A(t+1)     MFMA acc[0..3][0..1]
//   P2: read b1                                                       stage B(t+1)     MFMA acc[0..3][2..3]
//   P3: read a1 (over a0's registers)                                                   MFMA acc[4..7][2..3]
//   P4: (b0 still in registers)  vmcnt(0) before the first barrier                      MFMA acc[4..7][0..1]
// Hazards: tile t+1 is read (P1 of t+1) two barriers after every wave's own vmcnt(0); buffer (t+1)&1 is re-staged at P1/P2 of
// tile t, four barriers after its last ds_read (P3 of tile t-1) was waited for by the later half.
template <bool REG_EPI, int FAM>
__global__ __launch_bounds__(512, 1) void gemm_bf16_nt_pp_kernel(const FArgs p) {
  constexpr int BM = 256, BN = 256, NW = 8;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  // the whole 160-KiB LDS: a ring of THREE A tiles (the activation operand streams from HBM: two K-tiles of lead) and TWO B
  // tiles (the weight panel is L2-resident: one K-tile of lead)
  __shared__ __attribute__((aligned(1024))) char smem[3 * A_BYTES + 2 * B_BYTES];
  char* const smem_b = smem + 3 * A_BYTES;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // XCD-aware tile id (as in the 128^2 kernel).  One output tile per workgroup: a persistent variant (256 workgroups walking
  // their XCD's tile list, the next tile's first K-tiles requested before the epilogue) was measured twice -- with the LDS-staged
  // epilogue (no faster, +30 VGPRs) and with the register epilogue, where the LDS is free during the epilogue (qkv forward 179 ->
  // 211 us, fc1 forward 295 -> 334: the loop-carried addresses spill, the static tile assignment loses the dispatcher's dynamic
  // balancing, and the wait for the next tile's loads also waits for the epilogue's stores).
  // Round 2 built it a third time (commit 3a7f73c and its follow-up experiment): 256 workgroups walking tiles id, id + 256, ..., every
  // per-lane constant derived from a lane id recomputed per tile (mbcnt) so that nothing is hoisted and spilled, the epilogue's LDS
  // traffic as inline asm (a compiler-visible LDS access waits vmcnt(0) for the pending LDS-DMA prefetch), and (a) the next tile's A(0)
  // or (b) its whole first K-tile (A slots 0 / 1, B slot 1; image in the 64 KB of A slot 2 + B slot 0, two steps; entry wait
  // vmcnt(#stores)) requested before the epilogue.  Same-box three-way A/B against this kernel: (a) qkv forward 204.3 -> 198.5 us, fc1
  // forward 331 -> 323, step 36.26 -> 36.23 ms; (b) qkv 202.5 -> 198.1, fc1 334.6 -> 324.0, but the load-carrying epilogues lose their
  // early operand requests to the stepwise image (act' product 285 -> 307 us) and the step reads 37.1 -> 37.5 ms.  The walk itself is worth
  // ~5 % on K = 768 (the same code with one tile per workgroup is that much slower), the scaffolding costs 3 %: not kept.
  const int nwg = p.ntm * p.ntn;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
  const int tid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
  // group of row panels walked against all weight panels: sized so that the ~32 workgroups an XCD runs at a time cover ALL
  // ntn tiles of each of their row panels -- the cold operand (activations, 76..304 MB) is then fetched once, and it is the
  // hot one (the weight matrix, <= 5 MB, L2 / Infinity-Cache resident) that gets re-read
  const int group_m = p.group_m;
  int m0, n0;
  if (p.colslice > 1) {
    // Column-sliced order: XCD x keeps to weight panels [slice * c, (slice + 1) * c) (slice = x % S, c = ntn / S) for the whole launch
    // and walks its share (one of 8 / S) of the row panels against them.  The slice (c * 393 KB at K = 768) then stays in the XCD's
    // 4-MB L2 from one round of tiles to the next, where the whole matrix (3.5 .. 4.7 MB) plus the streaming activations does not.
    const int S = p.colslice, R = 8 / S, c = p.ntn / S;
    const int j = bid >> 3, slice = xcd % S, rs = xcd / S;
    const int rbase = p.ntm / R, rrem = p.ntm - rbase * R;
    const int rows = rbase + (rs < rrem ? 1 : 0), row0 = rs * rbase + (rs < rrem ? rs : rrem);
    if (j >= rows * c) return;                       // (the grid is 8 x the largest share)
    const int per_group = group_m * c;
    const int grp = j / per_group, in_grp = j - grp * per_group;
    const int gm0 = grp * group_m;
    const int gsz = (rows - gm0) < group_m ? (rows - gm0) : group_m;
    m0 = (row0 + gm0 + in_grp % gsz) * BM;
    n0 = (slice * c + in_grp / gsz) * BN;
  } else {
    const int per_group = group_m * p.ntn;
    const int grp = tid / per_group, in_grp = tid - grp * per_group;
    const int gm0 = grp * group_m;
    const int gsz = (p.ntm - gm0) < group_m ? (p.ntm - gm0) : group_m;
    m0 = (gm0 + in_grp % gsz) * BM;
    n0 = (in_grp / gsz) * BN;
  }
  if (p.rev_m) m0 = (p.ntm - 1) * BM - m0;
  const int z = blockIdx.y;
  const bf16_t* A = p.A + z * p.sA;
  const bf16_t* B = p.B + z * p.sB;

  const int frow = lane & 15;
  const int sw = (lane >> 1) & 7;
  const int foff0 = frow * 128 + (((lane >> 4)) ^ sw) * 16;
  const int foff1 = frow * 128 + ((4 + (lane >> 4)) ^ sw) * 16;
  const int nk = p.K / BK;

  // The bias is the accumulators' INITIAL VALUE (register epilogue, alpha == 1): requested before the first operand tile, its latency
  // overlaps the prologue's loads, and the epilogue neither loads nor adds it.  Requested at the top of the epilogue (the earlier form) the
  // latency -- an L2 round trip under load -- was paid once per output tile with nothing to hide it, 7 to 9 times per launch on the K = 768
  // products.  The lane's accumulator acc[i][2h], acc[i][2h+1] holds columns ncol + 32h + 0..3 / 4..7 (pp_epi_k), the same for every row
  // tile i.  -DDEMO_BIAS_INIT=0 restores the late request (A/B variant of the library).
#ifndef DEMO_BIAS_INIT
#define DEMO_BIAS_INIT 1
#endif
  const bool bias_in_acc = DEMO_BIAS_INIT && REG_EPI && p.bias != nullptr && p.alpha == 1.0f;
  f32x4 binit[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) binit[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (bias_in_acc) {
    const float* bp = p.bias + z * p.sBias + n0 + wc * 64 + (lane >> 4) * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) binit[j] = *reinterpret_cast<const f32x4*>(bp + (j >> 1) * 32 + (j & 1) * 4);
  }

  // (De-phasing the first round of workgroups with s_sleep, so that CUs do not reach their store-heavy epilogues together,
  // was tried in round 1 -- 2/4/8 phase groups, 7..17 us apart -- and measured 2..6 % slower on the training step.  Round 2
  // re-measured it with small offsets behind a tuning hook (demo_gemm_set_dephase), see DESIGN.md.)
  if (p.dephase > 0 && bid < 256) {
    const int n = ((bid >> 3) & 3) * p.dephase;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
  }
  // first K-tiles: A(0) -> A slot 0, B(0) -> B slot 1, A(1) -> A slot 1
  stage_tile<BM, NW, false, DEMO_A_AUX>(A, p.lda, m0, p.M, 0, smem, wave, lane);
  stage_tile<BN, NW, REG_EPI>(B, p.ldb, n0, p.N, 0, smem_b + B_BYTES, wave, lane);
  if (nk > 1) {
    stage_tile<BM, NW, false, DEMO_A_AUX>(A, p.lda, m0, p.M, BK, smem + A_BYTES, wave, lane);
    wait_vmcnt<4>();                                // K-tile 0 landed; A(1), the 4 youngest loads, may still fly
  } else {
    wait_vmcnt<0>();
  }
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = binit[j];
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();      // this half runs one barrier behind from here on

#define PP_MFMA(AF, BF, I0, J0)                                                                                  \
  do {                                                                                                           \
    __builtin_amdgcn_s_barrier();                                                                                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    __builtin_amdgcn_s_setprio(1);                                                                               \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                             \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                              \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                            \
          acc[I0 + i][J0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[kk][j], AF[kk][i], acc[I0 + i][J0 + j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    __builtin_amdgcn_s_barrier();                                                                                \
  } while (0)

#if DEMO_PP_PHASES == 2
  // TWO phases per K-tile (32 MFMAs each: one 64-row half of the wave's C against all four column tiles) instead of four of 16:
  // the same fragment registers (a: 32, b: 32), half the barriers, and a hand-over of the MFMA pipe between the two wave halves
  // every 512 instead of every 256 MFMA clocks.
  //   P1: read b0, b1 (4 col tiles x 2 k-halves), a0 (4 row tiles x 2)   stage B(t+1)    MFMA acc[0..3][0..3]
  //   P2: read a1 (over a0's registers)                                   stage A(t+2)    MFMA acc[4..7][0..3]
  // K-tile t+1 must have landed before GLOBAL barrier 4t+3 (the first barrier whose far side reads it: P1(t+1) of the leading
  // half): the leading half waits for its own pieces after the MFMAs of P2, before its closing barrier; the trailing half, one
  // barrier behind, before the FIRST barrier of its P2 -- the same global barrier.  B(t+1) has had a whole K-tile to land
  // (three quarters of one for the trailing half).  Re-staging: B slot t&1 was last read at P1(t-1) (waited for before global
  // barrier 4t-2), A slot of t-1 at P2(t-1) (before global barrier 4t); the stagings follow global barriers 4t-1 and 4t+1.
#define PP_MFMA32(I0)                                                                                            \
  do {                                                                                                           \
    __builtin_amdgcn_s_barrier();                                                                                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    __builtin_amdgcn_s_setprio(1);                                                                               \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                             \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                              \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                          \
          acc[I0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0f[kk][j], af[kk][i], acc[I0 + i][j], 0, 0, 0); \
          acc[I0 + i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1f[kk][j], af[kk][i], acc[I0 + i][2 + j], 0, 0, 0); \
        }                                                                                                        \
    __builtin_amdgcn_s_setprio(0);                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
  } while (0)
  int sa = 0;                                       // A ring slot of tile t
  for (int t = 0; t < nk; ++t) {
    const bool has1 = t + 1 < nk, has2 = t + 2 < nk;
    const int sa2 = sa == 0 ? 2 : sa - 1;           // slot of tile t+2 == slot of tile t-1
    const unsigned la = lds_off(smem + sa * A_BYTES + wr * (128 * 128));
    const unsigned lb = lds_off(smem_b + ((t + 1) & 1) * B_BYTES + wc * (64 * 128));      // B(t) sits in slot (t+1)&1
    bf16x8 af[2][4], b0f[2][2], b1f[2][2];
    // ---- P1
    lds_read2(b0f[0], lb + foff0);
    lds_read2(b0f[1], lb + foff1);
    lds_read2(b1f[0], lb + 4096 + foff0);
    lds_read2(b1f[1], lb + 4096 + foff1);
    lds_read4(af[0], la + foff0);
    lds_read4(af[1], la + foff1);
#if DEMO_FEED_ABLATE == 1
    if (has1 && p.K < 0) stage_tile<BN, NW, REG_EPI>(B, p.ldb, n0, p.N, (t + 1) * BK, smem_b + (t & 1) * B_BYTES, wave, lane);
#elif DEMO_FEED_ABLATE == 2
    if (has1) stage_tile<BN, NW, REG_EPI>(B, p.ldb, 0, p.N, (t + 1) * BK, smem_b + (t & 1) * B_BYTES, wave, lane);
#else
    if (has1) stage_tile<BN, NW, REG_EPI>(B, p.ldb, n0, p.N, (t + 1) * BK, smem_b + (t & 1) * B_BYTES, wave, lane);
#endif
    PP_MFMA32(0);
    __builtin_amdgcn_s_barrier();
    // ---- P2
    lds_read4(af[0], la + 8192 + foff0);
    lds_read4(af[1], la + 8192 + foff1);
#if DEMO_FEED_ABLATE == 1
    if (has2 && p.K < 0) stage_tile<BM, NW, false, DEMO_A_AUX>(A, p.lda, m0, p.M, (t + 2) * BK, smem + sa2 * A_BYTES, wave, lane);
#elif DEMO_FEED_ABLATE == 2
    if (has2) stage_tile<BM, NW, false, DEMO_A_AUX>(A, p.lda, 0, p.M, (t + 2) * BK, smem + sa2 * A_BYTES, wave, lane);
#else
    if (has2) stage_tile<BM, NW, false, DEMO_A_AUX>(A, p.lda, m0, p.M, (t + 2) * BK, smem + sa2 * A_BYTES, wave, lane);
#endif
    if (wr == 1) {
      if (has2) wait_vmcnt<4>();
      else if (has1) wait_vmcnt<0>();
    }
    PP_MFMA32(4);
    if (wr == 0) {
      if (has2) wait_vmcnt<4>();
      else if (has1) wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    sa = sa == 2 ? 0 : sa + 1;
  }
#undef PP_MFMA32
#else
  int sa = 0;                                       // A ring slot of tile t
  for (int t = 0; t < nk; ++t) {
    const bool has1 = t + 1 < nk, has2 = t + 2 < nk;
    const int sa2 = sa == 0 ? 2 : sa - 1;           // slot of tile t+2 == slot of tile t-1
    const unsigned la = lds_off(smem + sa * A_BYTES + wr * (128 * 128));
    const unsigned lb = lds_off(smem_b + ((t + 1) & 1) * B_BYTES + wc * (64 * 128));      // B(t) sits in slot (t+1)&1
    bf16x8 af[2][4], b0f[2][2], b1f[2][2];
    // ---- P1   (B's slot t&1 was last read at P2 of tile t-1; A's slot sa2 at P3 of tile t-1)
    lds_read2(b0f[0], lb + foff0);
    lds_read2(b0f[1], lb + foff1);
    lds_read4(af[0], la + foff0);
    lds_read4(af[1], la + foff1);
    if (has1) stage_tile<BN, NW, REG_EPI>(B, p.ldb, n0, p.N, (t + 1) * BK, smem_b + (t & 1) * B_BYTES, wave, lane);
    PP_MFMA(af, b0f, 0, 0);
    // ---- P2
    lds_read2(b1f[0], lb + 4096 + foff0);
    lds_read2(b1f[1], lb + 4096 + foff1);
    if (has2) stage_tile<BM, NW, false, DEMO_A_AUX>(A, p.lda, m0, p.M, (t + 2) * BK, smem + sa2 * A_BYTES, wave, lane);
    PP_MFMA(af, b1f, 0, 2);
    // ---- P3
    lds_read4(af[0], la + 8192 + foff0);
    lds_read4(af[1], la + 8192 + foff1);
    PP_MFMA(af, b1f, 4, 2);
    // ---- P4 (b0 still in registers): K-tile t+1 (A issued a K-tile ago, B at P1) must have landed; A(t+2), the 4 youngest
    // loads, stays in flight
    if (has2) wait_vmcnt<4>();
    else if (has1) wait_vmcnt<0>();
    PP_MFMA(af, b0f, 4, 0);
    sa = sa == 2 ? 0 : sa + 1;
  }
#endif
#undef PP_MFMA
  if (wr == 0) __builtin_amdgcn_s_barrier();      // re-join the halves: every ds_read of this output tile is done

  // ---- epilogue: the fp32 tile goes through LDS one 128-row half at a time (whole-row 16-B accesses for bias / act'(pre) /
  // residual / C, see the 128^2 kernel).  Code size matters here: epi4 carries every epilogue variant, and a fully unrolled
  // four-quarter form of this loop (3x the instructions) measured 25 % slower end to end on K = 768 -- instruction fetch.
  const int64_t coff = z * p.sC;
  const float* bias = p.bias ? p.bias + z * p.sBias : nullptr;
  const int g4 = lane >> 4;
  if constexpr (REG_EPI) {
    // register epilogue (see pp_epi_k): no LDS image, no barrier
#if DEMO_EPI_ABLATE == 2
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
#else
    pp_epi<FAM>(p, acc, m0 + wr * 128 + frow, n0 + wc * 64 + g4 * 8, coff, bias_in_acc ? nullptr : bias, smem, m0, n0);
#endif
    return;
  }
  float* ct = reinterpret_cast<float*>(smem);
  const int c = threadIdx.x & 31;                   // this thread's 8-column chunk, the same in every pass
  float4 bz0 = make_float4(0.f, 0.f, 0.f, 0.f), bz1 = bz0;
  if (bias) {
    bz0 = *reinterpret_cast<const float4*>(bias + n0 + c * 8);
    bz1 = *reinterpret_cast<const float4*>(bias + n0 + c * 8 + 4);
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {            // unrolled: acc must be indexed statically (else it lives in scratch)
    // (no early request of the act'(pre) source here: with the other half's 128 accumulators still live, its 32 registers
    // spill -- the dispatcher sends act'(pre) products to the 128^2 kernel, which has them to spare)
    const uint4 epre[8] = {};
    const bool have_epre = false;
    __syncthreads();
    if (wr == half) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = i * 16 + frow;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int slot = (wc * 16 + j * 4 + g4) ^ (frow & 7);
          *reinterpret_cast<float4*>(ct + r * BN + slot * 4) =
              make_float4(acc[i][j][0] * p.alpha, acc[i][j][1] * p.alpha, acc[i][j][2] * p.alpha, acc[i][j][3] * p.alpha);
        }
      }
    }
    __syncthreads();
    epi_rows<8, 16, BN, 2>(p, ct, threadIdx.x >> 5, m0 + half * 128, n0 + c * 8, c, bz0, bz1, coff, epre, have_epre);   // 128 rows x 32 chunks
  }
}

// ---- 256 x 128 "twin" kernel: TWO workgroups per CU -----------------------------------------------------------------------------
// Same wave tile as the 256^2 kernel (128 x 64 per wave: 8 x 4 MFMA tiles, 128 accumulator registers, the register epilogue pp_epi
// with its permuted weight rows), but a workgroup is 4 waves (2 x 2) on a 256 x 128 output tile with a 32-deep K-step, so its LDS
// ring is 3 stages x 24 KB = 72 KB and TWO workgroups share a CU (one wave of each per SIMD, 256 VGPRs apiece).  What that buys: the
// 256^2 kernel's epilogue -- 128..256 KB of output stores per tile, ~5.5 us per round of tiles at HBM write speed, 12 us with
// QuickGELU and two outputs -- overlaps with nothing at one workgroup per CU (tools/gemm_cost.py: N = 3072, K = 768 runs at 1195
// TFLOP/s without its epilogue, 932 with a bias-only one, 725 with the fc1 one); here one workgroup's epilogue and prologue run
// under the other's main loop.  What it costs: 85 instead of 128 flop per byte moved L2 -> LDS, and a barrier every 32 MFMAs.
//   LDS rows are 64 B (32 bf16): a 16-row fragment tile is 1 KB = ONE LDS-DMA piece.  Conflict-free ds_read_b128 needs the 16-B chunk
//   index XOR-ed with S[(row >> 2) & 3], S = {0, 2, 3, 1} (the four 16-lane groups of a b128 read then cover 16 distinct 16-B slots
//   of the 256-B bank row); as everywhere the swizzle is applied to the per-lane SOURCE address of the DMA and to the read address.
constexpr int TW_BK = 32, TW_BM = 256, TW_BN = 128, TW_STAGE = (TW_BM + TW_BN) * TW_BK * 2, TW_NS = 3;
__device__ __forceinline__ int tw_swz(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }      // S = {0, 2, 3, 1}

template <int ROWS, bool PERM>
__device__ __forceinline__ void tw_stage(const bf16_t* __restrict__ G, int64_t ld, int row0, int row_max, int k0, char* lds_tile, int wave,
                                         int lane) {
  constexpr int NP = ROWS / 16;                    // 1-KB pieces (16 rows x 64 B), dealt to the 4 waves
#pragma unroll
  for (int j = 0; j < NP / 4; ++j) {
    const int piece = j * 4 + wave;
    const int r = piece * 16 + (lane >> 2);
    const int c = (lane & 3) ^ tw_swz(r);
    int gr = row0 + (PERM ? (r & ~63) + perm64(r & 63) : r);
    gr = gr < row_max ? gr : row_max - 1;
    const bf16_t* src = G + (int64_t)gr * ld + k0 + c * 8;
    __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(lds_tile + piece * 1024), 16, 0, 0);
  }
}
// four ds_read_b128 of fragment tiles 0..3 (1024 B apart) from one base; NOT waited for here
__device__ __forceinline__ void tw_read4(bf16x8 (&f)[4], unsigned base) {
  asm volatile("ds_read_b128 %0, %4\n\t"
               "ds_read_b128 %1, %4 offset:1024\n\t"
               "ds_read_b128 %2, %4 offset:2048\n\t"
               "ds_read_b128 %3, %4 offset:3072"
               : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3])
               : "v"(base)
               : "memory");
}

__global__ __launch_bounds__(256, 2) void gemm_bf16_nt_twin_kernel(const FArgs p) {
  __shared__ __attribute__((aligned(1024))) char smem[TW_NS * TW_STAGE];
  constexpr int A_BYTES = TW_BM * TW_BK * 2;                       // 16 KB; B follows A inside a stage
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  // XCD-aware tile id, groups of row panels walked against all weight panels (as in the 256^2 kernel)
  const int nwg = p.ntm * p.ntn;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
  const int tid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
  const int group_m = p.group_m;
  const int per_group = group_m * p.ntn;
  const int grp = tid / per_group, in_grp = tid - grp * per_group;
  const int gm0 = grp * group_m;
  const int gsz = (p.ntm - gm0) < group_m ? (p.ntm - gm0) : group_m;
  const int m0 = (gm0 + in_grp % gsz) * TW_BM, n0 = (in_grp / gsz) * TW_BN;
  const int z = blockIdx.y;
  const bf16_t* A = p.A + z * p.sA;
  const bf16_t* B = p.B + z * p.sB;

  const int frow = lane & 15, g4 = lane >> 4;
  const int foff = frow * 64 + ((g4 ^ tw_swz(frow)) << 4);        // this lane's 16 B inside a 16-row fragment tile
  const int nk = p.K / TW_BK;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // stages 0 and 1 (6 LDS-DMA pieces per wave and stage: 4 of A, 2 of B)
  tw_stage<TW_BM, false>(A, p.lda, m0, p.M, 0, smem, wave, lane);
  tw_stage<TW_BN, true>(B, p.ldb, n0, p.N, 0, smem + A_BYTES, wave, lane);
  if (nk > 1) {
    tw_stage<TW_BM, false>(A, p.lda, m0, p.M, TW_BK, smem + TW_STAGE, wave, lane);
    tw_stage<TW_BN, true>(B, p.ldb, n0, p.N, TW_BK, smem + TW_STAGE + A_BYTES, wave, lane);
  }
  int slot = 0;
  for (int t = 0; t < nk; ++t) {
    if (t + 1 < nk) wait_vmcnt<6>();               // stage t landed; stage t+1 (the 6 youngest) may still fly
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                  // every wave's part of stage t is in LDS; everyone is done reading stage t-1
    if (t + 2 < nk) {
      char* nxt = smem + (slot == 0 ? 2 : slot - 1) * TW_STAGE;     // slot of stage t-1 == slot of stage t+2
      tw_stage<TW_BM, false>(A, p.lda, m0, p.M, (t + 2) * TW_BK, nxt, wave, lane);
      tw_stage<TW_BN, true>(B, p.ldb, n0, p.N, (t + 2) * TW_BK, nxt + A_BYTES, wave, lane);
    }
    const unsigned la = lds_off(smem + slot * TW_STAGE + wr * (128 * 64) + foff);
    const unsigned lb = lds_off(smem + slot * TW_STAGE + A_BYTES + wc * (64 * 64) + foff);
    slot = slot == 2 ? 0 : slot + 1;
    bf16x8 bf[4], a0[4], a1[4];
    tw_read4(bf, lb);
    tw_read4(a0, la);
    tw_read4(a1, la + 4096);
    asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], a0[i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], a1[i], acc[4 + i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  }
  const int64_t coff = z * p.sC;
  const float* bias = p.bias ? p.bias + z * p.sBias : nullptr;
  if (p.pre || p.pre_out) pp_epi<1>(p, acc, m0 + wr * 128 + frow, n0 + wc * 64 + g4 * 8, coff, bias);
  else pp_epi<0>(p, acc, m0 + wr * 128 + frow, n0 + wc * 64 + g4 * 8, coff, bias);
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// returns 1 if the descriptor fits the MFMA kernel's contract, 0 otherwise
int demo_gemm_bf16_eligible(const demo_gemm_desc& d) {
  if (d.dtA != DEMO_BF16 || d.dtB != DEMO_BF16) return 0;
  if (d.a_cs != 1 || d.b_cs != 1) return 0;
  if (d.K % BK || d.N % 128 || d.M < 1) return 0;
  if (d.a_rs % 8 || d.b_rs % 8 || d.c_rs % 4) return 0;
  if (!al16(d.A) || !al16(d.B) || !al16(d.C)) return 0;
  if (d.bias && !al16(d.bias)) return 0;
  if (d.residual && !al16(d.residual)) return 0;
  if (d.residual && d.dtRes == DEMO_BF16 && d.c_rs % 8) return 0;
  if (d.act_grad_pre && !al16(d.act_grad_pre)) return 0;
  if (d.pre_out && !al16(d.pre_out)) return 0;
  if (d.accumulate && d.dtC != DEMO_F32) return 0;
  // the row-major epilogue moves 8 columns (16 B of bf16) per access
  if ((d.dtC == DEMO_BF16 || (d.act_grad_pre && d.dtPre != DEMO_F32) || (d.pre_out && d.pre_out_kind == 2)) && d.c_rs % 8) return 0;
  if (d.batch1 != 1) {
    if (d.batch2 != 1) return 0;               // single-level batching only
  }
  const int64_t sA = d.batch1 != 1 ? d.sA1 : d.sA2, sB = d.batch1 != 1 ? d.sB1 : d.sB2;
  const int64_t sC = d.batch1 != 1 ? d.sC1 : d.sC2, sBi = d.batch1 != 1 ? d.sBias1 : d.sBias2;
  if (d.batch1 * d.batch2 > 1 && (sA % 8 || sB % 8 || sC % 4 || sBi % 4)) return 0;
  if (d.M < 64) return 0;                       // skinny products: the generic kernel wastes less
  return 1;
}

static int g_force_tile = 0;   // test/bench hook: 128 or 256 forces a configuration, 0 = heuristic
static int g_pp_reg_epi = 1;   // 256^2 kernel: 1 = register epilogue over permuted weight rows, 0 = LDS-staged row-major epilogue
static int g_pp_min_eff_pct = 70;
static int g_rev_m_min_k = 0;  // 256^2 kernel, tuning hook: products with K >= this walk their row panels in reverse (0 = never)
extern "C" int demo_gemm_set_reverse_m(int min_k) { g_rev_m_min_k = min_k < 0 ? 0 : min_k; return 0; }
static int g_group_m = 0;      // 256^2 kernel, tuning hook: row panels per tile-order group (0 = heuristic)
extern "C" int demo_gemm_set_group_m(int g) { g_group_m = g < 0 ? 0 : g; return 0; }
static int g_dephase = 0;
extern "C" int demo_gemm_set_dephase(int units) { g_dephase = units < 0 ? 0 : units; return 0; }
static int g_epi_stage = 1;
extern "C" int demo_gemm_set_epi_stage(int on) { g_epi_stage = on ? 1 : 0; return 0; }
static int g_colslice = 0;     // 0 = heuristic, 1 = off, 2 / 4 / 8 = forced where ntn divides
extern "C" int demo_gemm_set_colslice(int s) { g_colslice = s; return 0; }
extern "C" int demo_gemm_store_policy(void) { return DEMO_STORE_POL; }
extern "C" int demo_gemm_set_pp_min_fill(int pct) { g_pp_min_eff_pct = pct; return 0; }
void demo_gemm_tn_force(int t);
// 1280: only pins the weight-gradient (TN) product to its 128^2 form, 1290: to the 256^2 form with fp32 atomics (no two-pass
// split-K); 0 restores the heuristics of both
extern "C" int demo_gemm_force_tile(int t) {
  g_force_tile = (t == 1280 || t == 1290) ? 0 : t;
  demo_gemm_tn_force(t == 1280 ? 128 : t == 1290 ? 129 : 0);
  return 0;
}

// kernel configuration the dispatcher will use for an eligible descriptor (also reported through demo_gemm_path)
int demo_gemm_bf16_config(const demo_gemm_desc& d) {
  const int nb = d.batch1 * d.batch2;
  int cfg = g_force_tile;
  if (cfg == 0) {
    // 256^2 ping-pong kernel when its coarser tiles still fill the 256 CUs evenly (or K is long enough that the main loop,
    // ~1.4 PF/s marginal against ~1.1 for the 128^2 kernel, outweighs the last partial round); measured crossover on
    // M = 49536: N = 2304 / 3072 always, N = 768 from K = 2048
    cfg = 1282;
    // act'(pre) epilogue at N=3072, K=768: 128^2 344 us, 256^2 with the LDS-staged epilogue 382 us (its operand registers spill), 256^2
    // with the register epilogue 319 us
    if (d.N % 256 == 0 && d.M >= 1024 && (!d.act_grad_pre || (g_pp_reg_epi && d.dtPre != DEMO_F32))) {
      const double rounds = (double)cdiv64(d.M, 256) * (d.N / 256) * nb / 256.0;
      const double eff = rounds / (double)(int64_t)(rounds + 0.999999);
      // fill threshold of the last round: 0.85 was the measured crossover with the LDS-staged epilogue; with the register epilogue
      // (and the row split taking care of small leftovers) the 256^2 kernel also wins at 2.27 and 1.52 rounds with K = 768
      // (dY.W_proj 74.8 -> 71.6 us, CLIP projection 57.2 -> 50.7)
      const double min_eff = g_pp_reg_epi ? g_pp_min_eff_pct * 0.01 : 0.85;
      if (eff >= min_eff || d.K >= 2048) cfg = 2566;
    }
  }
  if ((cfg == 2562 || cfg == 2566) && d.N % 256) cfg = 1282;
  // few rows (inference at small batches: 387 rows at B = 1): 128^2 tiles leave most CUs idle and every workgroup walks K at one exposed
  // load latency per step (19 us per trunk product at B = 1); 64^2 tiles with three K-tiles in flight
  if (g_force_tile == 0 && cfg == 1282 && cdiv64(d.M, 128) * cdiv64(d.N, 128) * nb < 128) cfg = 644;
  return cfg;
}

// Row split of a 256^2 launch whose tile count leaves a small last round (582 tiles on 256 CUs = 2.27 rounds at N = 768,
// M = 49536: a third round running 27 % full).  Two plans:
//   (a) main + remainder: the rows that fill whole rounds go to the 256^2 kernel, the rest to the 128^2 kernel, whose 4x smaller
//       tiles at 2 workgroups/CU cover them in one short round (round 1);
//   (b) EQUAL PARTS (round 3): full + 1 launches of the 256^2 kernel with ~tiles / (full + 1) tiles each, every CU at most one tile
//       per launch -- no 128^2 launch (half the flops per byte fed to LDS) and no two-round lock-step.  Measured inside the training step,
//       same box, interleaved runs: taken for the K = 768 products of N = 768 (attention output projection and its dX) 35.98 -> ~35.75 ms;
//       taken for the long-K products as well (K = 2304 / 3072) 35.89, for the long-K products alone 36.14 -- there plan (a) stays.
//       (An isolated microbenchmark, tools/bench_cu_fill.py, shows a 256^2 tile at 66 us with 195 tiles in flight against 78 us with 255
//       at K = 3072 -- the L2 -> LDS feed is shared chip-wide -- and so predicted the opposite for long K; it re-reads ONE part's rows
//       from a warm Infinity Cache, which the step's three parts do not.)  g_split_eq_max_k: largest K that takes plan (b); 0 = never,
//       the default: 0.12 ms of 35.7 (three runs per arm) is inside the box-to-box spread, so the plan stays an opt-in hook
//       (demo_gemm_set_split_equal(1024), DEMO_GEMM_SPLIT_EQUAL=1024).
// demo_gemm_bf16_split_plan: rows[i] of each part (multiples of 256 but the last) and its kernel configuration; returns the number of
// parts (1 = one launch).  g_split_left: largest leftover (in 256^2 tiles) that is split off; 0 disables both plans.
extern "C" int demo_gemm_set_pp_epilogue(int reg) { g_pp_reg_epi = reg != 0; return 0; }
static int g_rem160 = 1;
extern "C" int demo_gemm_set_rem160(int on) { g_rem160 = on != 0; return 0; }
static int g_split_left = 128;
static int g_split_eq_max_k = 0;       // OFF by default: -0.12 ms of a 35.7 ms step at 1024, see above
extern "C" int demo_gemm_set_split(int max_left_tiles) { g_split_left = max_left_tiles; return 0; }
extern "C" int demo_gemm_set_split_equal(int max_k) { g_split_eq_max_k = max_k; return 0; }      // 0 = never (plan (a) only)
int demo_gemm_bf16_split_plan(const demo_gemm_desc& d, int32_t* rows, int32_t* cfg, int max_parts) {
  rows[0] = d.M;
  cfg[0] = 0;
  if (max_parts < 2 || g_force_tile != 0 || g_split_left <= 0 || d.batch1 * d.batch2 != 1) return 1;
  if (demo_gemm_bf16_config(d) != 2566) return 1;
  const int64_t ntm = cdiv64(d.M, 256), ntn = d.N / 256, total = ntm * ntn;
  const int64_t full = total / 256, left = total - full * 256;
  if (full < 1 || left < 32 || left > g_split_left) return 1;   // (a last round under ~1/8 full finishes fast enough on its own: fc1 forward, 24 tiles, measured)
  if (d.K <= g_split_eq_max_k && full + 1 <= max_parts && full <= 3) {
    const int parts = (int)full + 1;
    const int64_t per = cdiv64(ntm, parts);                     // row panels per part
    if (per * ntn <= 256 && per * (parts - 1) < ntm) {
      int64_t done = 0;
      for (int i = 0; i < parts; ++i) {
        const int64_t pm = i + 1 < parts ? per * 256 : d.M - done;
        rows[i] = (int32_t)pm;
        cfg[i] = 2566;
        done += pm;
      }
      return parts;
    }
  }
  const int64_t tm1 = full * 256 / ntn;
  if (tm1 < 1 || tm1 * 256 >= d.M) return 1;
  rows[0] = (int32_t)(tm1 * 256); cfg[0] = 2566;
  rows[1] = (int32_t)(d.M - tm1 * 256);
  // The remainder on 128^2 tiles at two workgroups per CU keeps ONE K-tile (32 KB) in flight per workgroup and walks K at one exposed load
  // latency per step (~1.1 us: 55 us for K = 3072).  When it fits one round of 160 x 128 tiles it takes that configuration instead: one
  // workgroup per CU, three LDS buffers, two K-tiles in flight (g_rem160, demo_gemm_set_rem160).
  cfg[1] = (g_rem160 && cdiv64(rows[1], 160) * (d.N / 128) <= 256) ? 1603 : 1282;
  return 2;
}
// (kept for callers of the round-1 ABI: rows of the first part when the product is split, 0 = one launch)
int demo_gemm_bf16_split_rows(const demo_gemm_desc& d) {
  int32_t rows[4], cfg[4];
  return demo_gemm_bf16_split_plan(d, rows, cfg, 4) > 1 ? rows[0] : 0;
}

// Side stream of the row split (one per host thread and device: the library is entered from the caller's thread and from autograd's).
struct SideStream {
  hipStream_t st = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  int dev = -1;
};
static int g_split_overlap = 0;      // opt-in: measured SLOWER (37.5 vs 36.8 ms per step): the two cross-stream event waits per product cost more on
                                     // the device than the idle CUs they fill; hipExtAnyOrderLaunch (no barrier bit, same stream) is not supported on gfx9
extern "C" int demo_gemm_set_split_overlap(int on) { g_split_overlap = on ? 1 : 0; return 0; }
static SideStream* side_stream() {
  static thread_local SideStream x;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  if (x.st && x.dev == dev) return &x;
  if (x.st) {                                        // the thread moved to another device: start over there
    (void)hipStreamDestroy(x.st); (void)hipEventDestroy(x.fork); (void)hipEventDestroy(x.join);
    x = SideStream();
  }
  if (hipStreamCreateWithFlags(&x.st, hipStreamNonBlocking) != hipSuccess) { x = SideStream(); return nullptr; }
  if (hipEventCreateWithFlags(&x.fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&x.join, hipEventDisableTiming) != hipSuccess) {
    (void)hipStreamDestroy(x.st);
    x = SideStream();
    return nullptr;
  }
  x.dev = dev;
  return &x;
}

// cfg_override: 0 = choose by the heuristic (and split by rows where that pays), otherwise the configuration code to launch.  The two
// parts of a split product are issued through the override, not through the g_force_tile test hook: the library is entered from two
// threads (forward on the caller's, backward on autograd's) and must not pass state between calls through a global.
static int demo_gemm_bf16_launch(const demo_gemm_desc& d, hipStream_t s, int cfg_override);
int demo_gemm_bf16(const demo_gemm_desc& d, hipStream_t s) { return demo_gemm_bf16_launch(d, s, 0); }

static int demo_gemm_bf16_launch(const demo_gemm_desc& d, hipStream_t s, int cfg_override) {
  int32_t prow[4], pcfg[4];
  const int nparts = cfg_override ? 1 : demo_gemm_bf16_split_plan(d, prow, pcfg, 4);
  if (nparts > 1) {
    const int64_t szC = d.dtC == DEMO_F32 ? 4 : 2, szP = d.dtPre == DEMO_F32 ? 4 : d.dtPre == DEMO_D8 ? 1 : 2;
    demo_gemm_desc part[4];
    int64_t m0 = 0;
    for (int i = 0; i < nparts; ++i) {                    // the parts touch disjoint rows
      demo_gemm_desc& b = part[i];
      b = d;
      b.M = prow[i];
      b.A = (const char*)d.A + m0 * d.a_rs * 2;
      b.C = (char*)d.C + m0 * d.c_rs * szC;
      if (d.residual) b.residual = (const char*)d.residual + m0 * d.c_rs * (d.dtRes == DEMO_BF16 ? 2 : 4);
      if (d.act_grad_pre) b.act_grad_pre = (const char*)d.act_grad_pre + m0 * d.c_rs * szP;
      if (d.pre_out) b.pre_out = (char*)d.pre_out + m0 * d.c_rs * (d.pre_out_kind == 2 ? 1 : szC);
      m0 += prow[i];
    }
    // Issued back to back on the caller's stream.  Opt-in experiment for plan (a) (demo_gemm_set_split_overlap): the remainder forked onto
    // a side stream, ready at the same moment as the main part; the caller's stream joins both before anything else runs.
    SideStream* side = (g_split_overlap && nparts == 2 && pcfg[1] != 2566) ? side_stream() : nullptr;
    if (side && hipEventRecord(side->fork, s) == hipSuccess && hipStreamWaitEvent(side->st, side->fork, 0) == hipSuccess) {
      int rc = demo_gemm_bf16_launch(part[1], side->st, pcfg[1]);
      const hipError_t e1 = hipEventRecord(side->join, side->st);
      if (rc == 0) rc = demo_gemm_bf16_launch(part[0], s, 2566);
      const hipError_t e2 = hipStreamWaitEvent(s, side->join, 0);
      if (rc == 0 && (e1 != hipSuccess || e2 != hipSuccess)) DEMO_FAIL(-2, "demo_gemm: joining the side stream failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
      return rc;
    }
    int rc = 0;
    const bool last_first = g_rev_m_min_k > 0 && d.K >= g_rev_m_min_k;     // (reverse-walk hook: the rows the producer wrote last go first)
    for (int i = 0; i < nparts && rc == 0; ++i) {
      const int j = last_first ? nparts - 1 - i : i;
      rc = demo_gemm_bf16_launch(part[j], s, pcfg[j]);
    }
    return rc;
  }
  FArgs p;
  p.A = reinterpret_cast<const bf16_t*>(d.A);
  p.B = reinterpret_cast<const bf16_t*>(d.B);
  p.C = d.C;
  p.bias = d.bias;
  p.residual = d.residual;
  p.pre = d.act_grad_pre;
  p.pre_out = d.pre_out;
  p.pre_out_kind = d.pre_out ? d.pre_out_kind : 0;
  p.dtRes = d.residual ? d.dtRes : 0;
  p.dephase = g_dephase;
  p.rev_m = g_rev_m_min_k > 0 && d.K >= g_rev_m_min_k;
  p.colslice = 0;
  p.epi_stage = g_epi_stage;
  p.M = d.M; p.N = d.N; p.K = d.K;
  p.lda = d.a_rs; p.ldb = d.b_rs; p.ldc = d.c_rs;
  const bool lvl1 = d.batch1 != 1;
  p.sA = lvl1 ? d.sA1 : d.sA2; p.sB = lvl1 ? d.sB1 : d.sB2; p.sC = lvl1 ? d.sC1 : d.sC2; p.sBias = lvl1 ? d.sBias1 : d.sBias2;
  p.dtC = d.dtC; p.act = d.act; p.dtPre = d.dtPre; p.act_grad_kind = d.act_grad_kind; p.accumulate = d.accumulate;
  p.alpha = d.alpha;
  const int nb = d.batch1 * d.batch2;
  // configuration code = tile*10 + stages: 1282 (128^2, 2 buffers, 2 workgroups/CU), 1283, 1284 (1 workgroup/CU),
  // 2562 (256^2, 8 waves), 2563 = 256x128 with 3 buffers
  const int cfg = cfg_override ? cfg_override : demo_gemm_bf16_config(d);
#define LAUNCH(BM_, BN_, WM_, WN_, NS_)                                                                                  \
  do {                                                                                                                   \
    p.ntm = (int)cdiv64(d.M, BM_);                                                                                       \
    p.ntn = d.N / BN_;                                                                                                   \
    hipLaunchKernelGGL((gemm_bf16_nt_kernel<BM_, BN_, WM_, WN_, NS_>), dim3((unsigned)(p.ntm * p.ntn), (unsigned)nb),    \
                       dim3(WM_ * WN_ * 64), 0, s, p);                                                                   \
  } while (0)
#define LAUNCH_RS(BM_, BN_, WM_, WN_)                                                                                    \
  do {                                                                                                                   \
    p.ntm = (int)cdiv64(d.M, BM_);                                                                                       \
    p.ntn = d.N / BN_;                                                                                                   \
    hipLaunchKernelGGL((gemm_bf16_nt_kernel<BM_, BN_, WM_, WN_, 2, true>), dim3((unsigned)(p.ntm * p.ntn), (unsigned)nb), \
                       dim3(WM_ * WN_ * 64), 0, s, p);                                                                   \
  } while (0)
  switch (cfg) {
    case 2562: LAUNCH(256, 256, 2, 4, 2); break;
    case 2563: LAUNCH(256, 128, 4, 2, 3); break;
    case 644: LAUNCH(64, 64, 2, 2, 4); break;        // small-M products (eval at small batches): 4x the workgroups, 3 K-tiles in flight
    case 1283: LAUNCH(128, 128, 2, 2, 3); break;
    case 1603: LAUNCH(160, 128, 2, 2, 3); break;     // row-split remainders that fit one round: one workgroup per CU, two K-tiles in flight
    case 1284: LAUNCH(128, 128, 2, 2, 4); break;
    case 2566:
      p.ntm = (int)cdiv64(d.M, 256);
      p.ntn = d.N / 256;
      // Row panels per group of the tile order.  An XCD's ~32 concurrent tiles cover gm row panels x 32/gm weight panels and fetch gm + 32/gm
      // operand panels per wave of tiles: 18 at gm = 2 (round 2's 32 / ntn for N = 3072), 11.3 at gm = 6.  Round 2 measured 2..8 within 1 % per
      // product in isolation; inside the training step (interleaved runs, two boxes) gm = 6 for every product is 0.33 ms faster than
      // 32 / ntn (34.86 -> 34.53 ms; 5 / 7 / 12: 34.70 / 34.73 / 35.02) and the kernel's L2 fills drop 11 % (family 1) / 5 % (family 0).
      p.group_m = p.ntn >= 3 ? 6 : (32 / p.ntn > 8 ? 8 : 32 / p.ntn);
      if (g_group_m > 0) p.group_m = g_group_m;                                  // tuning hook (demo_gemm_set_group_m, DEMO_GEMM_GROUP_M)
      {
        unsigned grid = (unsigned)(p.ntm * p.ntn);
        p.colslice = 0;
        const int S = g_colslice;
        if (S > 1 && 8 % S == 0 && p.ntn % S == 0 && nb == 1) {
          const int c = p.ntn / S, R = 8 / S;
          p.colslice = S;
          p.group_m = 32 / c < 1 ? 1 : (32 / c > 8 ? 8 : 32 / c);
          grid = 8u * (unsigned)(cdiv64(p.ntm, R) * c);
        }
        if (g_pp_reg_epi && (d.pre_out || d.act_grad_pre))           // epilogue family 1: the MLP's second output / act' operand (see pp_epi)
          hipLaunchKernelGGL((gemm_bf16_nt_pp_kernel<true, 1>), dim3(grid, (unsigned)nb), dim3(512), 0, s, p);
        else if (g_pp_reg_epi)
          hipLaunchKernelGGL((gemm_bf16_nt_pp_kernel<true, 0>), dim3(grid, (unsigned)nb), dim3(512), 0, s, p);
        else
          hipLaunchKernelGGL((gemm_bf16_nt_pp_kernel<false, 0>), dim3(grid, (unsigned)nb), dim3(512), 0, s, p);
      }
      break;
    case 2576:                                       // 256 x 128 twin kernel (two workgroups per CU)
      p.ntm = (int)cdiv64(d.M, 256);
      p.ntn = d.N / 128;
      p.group_m = 64 / p.ntn < 1 ? 1 : (64 / p.ntn > 8 ? 8 : 64 / p.ntn);
      hipLaunchKernelGGL(gemm_bf16_nt_twin_kernel, dim3((unsigned)(p.ntm * p.ntn), (unsigned)nb), dim3(256), 0, s, p);
      break;
    case 1292: LAUNCH_RS(128, 128, 2, 2); break;
    case 2592: LAUNCH_RS(256, 128, 4, 2); break;
    default: LAUNCH(128, 128, 2, 2, 2); break;
  }
#undef LAUNCH
#undef LAUNCH_RS
  DEMO_LAUNCH_CHECK("demo_gemm(bf16 mfma)");
  return 0;
}
