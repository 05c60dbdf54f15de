// Shared device/host helpers for libdemo_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/demo_hip.h"

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define DEMO_WAVE 64

// ---- error plumbing ---------------------------------------------------------------------------
extern thread_local char g_demo_err[512];
#define DEMO_FAIL(code, ...)                                   \
  do {                                                         \
    snprintf(g_demo_err, sizeof(g_demo_err), __VA_ARGS__);     \
    return (code);                                             \
  } while (0)
#define DEMO_REQUIRE(cond, ...)                \
  do {                                         \
    if (!(cond)) DEMO_FAIL(-1, __VA_ARGS__);   \
  } while (0)
#define DEMO_LAUNCH_CHECK(name)                                                        \
  do {                                                                                 \
    hipError_t e__ = hipGetLastError();                                                \
    if (e__ != hipSuccess) DEMO_FAIL(-2, "%s: launch failed: %s", name, hipGetErrorString(e__)); \
  } while (0)

// ---- bf16 <-> f32 -----------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// plain cast semantics (round-to-nearest-even, NaN stays NaN): hipcc emits v_cvt_pk_bf16_f32 at -O3
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(bf16_t, h);
}

// two floats -> one 32-bit word of two bf16 (low half = first): a single v_cvt_pk_bf16_f32, no shift/and/or repacking
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
  const bf2_t v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, v);
}

template <typename T> struct io;
template <> struct io<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  static __device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
  // raw4: the 4 elements as loaded (conversion deferred, so a prefetched value can sit in registers untouched)
  typedef float4 raw4;
  static __device__ __forceinline__ raw4 ld4_raw(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ float4 cvt4(raw4 r) { return r; }
};
template <> struct io<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
  static __device__ __forceinline__ float4 ld4(const bf16_t* p) {
    uint2 r = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                       __uint_as_float(r.y & 0xffff0000u));
  }
  typedef uint2 raw4;
  static __device__ __forceinline__ raw4 ld4_raw(const bf16_t* p) { return *reinterpret_cast<const uint2*>(p); }
  static __device__ __forceinline__ float4 cvt4(raw4 r) {
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                       __uint_as_float(r.y & 0xffff0000u));
  }
  static __device__ __forceinline__ void st4(bf16_t* p, float4 v) {
    uint2 r;
    r.x = pack_bf16x2(v.x, v.y);
    r.y = pack_bf16x2(v.z, v.w);
    *reinterpret_cast<uint2*>(p) = r;
  }
};

// ---- wave reductions (64 lanes) -----------------------------------------------------------------
// Wave-wide reductions on the VALU's DPP data path (no LDS round trips as with ds_bpermute shuffles): quad swap,
// quad-pair swap, half-row mirror, row mirror leave every lane of a 16-lane row with the row's value; the four rows are
// combined through v_readlane, so the result is wave-uniform.  All 64 lanes must be active at the call.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_value(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov<0xB1>(v);     // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);     // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);    // row_half_mirror
  v += dpp_mov<0x140>(v);    // row_mirror
  return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  return fmaxf(fmaxf(lane_value(v, 0), lane_value(v, 16)), fmaxf(lane_value(v, 32), lane_value(v, 48)));
}

// ---- activations ---------------------------------------------------------------------------------
// QuickGELU: x * sigmoid(1.702 x) with sigmoid(1.702 x) = 1 / (1 + 2^(x * QG_K2)), QG_K2 = -1.702 * log2(e): ONE multiply in front of
// v_exp_f32 (2^x), v_rcp_f32 (1 ulp) for the division.  Every form below -- scalar, and the two-wide forms the GEMM epilogues use
// (v_pk_mul/add/fma_f32: two floats per lane and issue slot) -- evaluates the same expression in the same order, so a value does not
// depend on which kernel or tile configuration produced it.
typedef float v2f __attribute__((ext_vector_type(2)));
#define DEMO_QG_K2 (-2.45546696f)
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }          // v_pk_fma_f32
__device__ __forceinline__ v2f pk_fma(v2f a, float b, float c) { return __builtin_elementwise_fma(a, (v2f){b, b}, (v2f){c, c}); }
__device__ __forceinline__ v2f pk_fma(v2f a, float b, v2f c) { return __builtin_elementwise_fma(a, (v2f){b, b}, c); }
__device__ __forceinline__ float qg_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * DEMO_QG_K2)); }
__device__ __forceinline__ v2f qg_sigmoid2(v2f x) {
  const v2f z = x * DEMO_QG_K2;
  v2f e;
  e.x = __builtin_amdgcn_exp2f(z.x); e.y = __builtin_amdgcn_exp2f(z.y);
  const v2f den = e + 1.0f;
  v2f s;
  s.x = __builtin_amdgcn_rcpf(den.x); s.y = __builtin_amdgcn_rcpf(den.y);
  return s;
}
__device__ __forceinline__ float act_apply(float x, int act) {
  if (act == DEMO_ACT_QUICKGELU) return x * qg_sigmoid(x);
  if (act == DEMO_ACT_GELU) return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
  return x;
}
__device__ __forceinline__ float act_grad(float x, int act) {
  if (act == DEMO_ACT_QUICKGELU) {
    const float s = qg_sigmoid(x);
    return fmaf(1.702f * (x * s), 1.0f - s, s);          // s (1 + 1.702 x (1 - s))
  }
  if (act == DEMO_ACT_GELU) {
    float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    return cdf + x * 0.3989422804014327f * __expf(-0.5f * x * x);
  }
  if (act == DEMO_ACT_SAVED) return x;          // the operand already is the derivative (saved by the forward, pre_out_kind 1)
  return 1.0f;
}

// activation and its derivative from ONE evaluation of the shared transcendental part (forward epilogues that save act'(x))
__device__ __forceinline__ float act_both(float x, int act, float& d) {
  if (act == DEMO_ACT_QUICKGELU) {
    const float s = qg_sigmoid(x);
    const float u = x * s;
    d = fmaf(1.702f * u, 1.0f - s, s);          // s (1 + 1.702 x (1 - s))
    return u;
  }
  if (act == DEMO_ACT_GELU) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    d = fmaf(x * 0.3989422804014327f, __expf(-0.5f * x * x), cdf);
    return x * cdf;
  }
  d = 1.0f;
  return x;
}
// two-wide forms (erf-GELU has no packed transcendental path: per component)
__device__ __forceinline__ v2f act_apply2(v2f x, int act) {
  if (act == DEMO_ACT_QUICKGELU) return x * qg_sigmoid2(x);
  if (act == DEMO_ACT_NONE) return x;
  return (v2f){act_apply(x.x, act), act_apply(x.y, act)};
}
__device__ __forceinline__ v2f act_grad2(v2f x, int act) {
  if (act == DEMO_ACT_QUICKGELU) {
    const v2f s = qg_sigmoid2(x);
    return pk_fma(1.702f * (x * s), 1.0f - s, s);
  }
  if (act == DEMO_ACT_SAVED) return x;
  return (v2f){act_grad(x.x, act), act_grad(x.y, act)};
}
__device__ __forceinline__ v2f act_both2(v2f x, int act, v2f& d) {
  if (act == DEMO_ACT_QUICKGELU) {
    const v2f s = qg_sigmoid2(x);
    const v2f u = x * s;
    d = pk_fma(1.702f * u, 1.0f - s, s);
    return u;
  }
  float d0, d1;
  const float r0 = act_both(x.x, act, d0), r1 = act_both(x.y, act, d1);
  d = (v2f){d0, d1};
  return (v2f){r0, r1};
}

// ---- DEMO_D8: 8-bit code of a saved activation derivative (demo_hip.h) ----------------------------------------------------------
// q = rint((a + 0.13) * 200): v_cvt_pk_u8_f32 rounds to nearest even by itself (under the kernels' default fp32 rounding mode), saturates to
// [0, 255] and inserts the byte into a dword -- tools/probes/cvt_u8_probe.hip compares it with an explicit v_rndne_f32 in front (ties, both
// ends of the range, negatives: identical), so the encoder is one fma (two-wide) and one conversion per element.
__device__ __forceinline__ uint32_t d8_pack4(v2f a, v2f b) {
  const v2f qa = pk_fma(a, 200.0f, 26.0f), qb = pk_fma(b, 200.0f, 26.0f);
  uint32_t w = 0;
  w = __builtin_amdgcn_cvt_pk_u8_f32(qa.x, 0, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(qa.y, 1, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(qb.x, 2, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(qb.y, 3, w);
  return w;
}
__device__ __forceinline__ uint32_t d8_pack4(float a0, float a1, float a2, float a3) { return d8_pack4((v2f){a0, a1}, (v2f){a2, a3}); }
__device__ __forceinline__ uint8_t d8_encode(float a) { return (uint8_t)(d8_pack4(a, 0.f, 0.f, 0.f) & 0xffu); }
__device__ __forceinline__ float d8_decode(uint32_t q) { return fmaf((float)q, 0.005f, -0.13f); }
// the four bytes of a dword (v_cvt_f32_ubyte0..3, then two v_pk_fma_f32)
__device__ __forceinline__ void d8_unpack4(uint32_t w, v2f& lo, v2f& hi) {
  lo = pk_fma((v2f){(float)(w & 0xffu), (float)((w >> 8) & 0xffu)}, 0.005f, -0.13f);
  hi = pk_fma((v2f){(float)((w >> 16) & 0xffu), (float)(w >> 24)}, 0.005f, -0.13f);
}
__device__ __forceinline__ float4 d8_unpack4(uint32_t w) {
  v2f lo, hi;
  d8_unpack4(w, lo, hi);
  return make_float4(lo.x, lo.y, hi.x, hi.y);
}

// ---- attention kernels: 16-B output rows ----------------------------------------------------------------------------------------
// Output rows: an accumulator tile gives a lane 4 consecutive d (8 B of bf16) of its key / query row, 16 rows x 32 B per store
// instruction.  v_permlane16_swap exchanges the odd 16-lane rows of one register with the even rows of another: applied to the packed
// halves of the column tiles 2p and 2p+1 it leaves every lane with 8 consecutive d (16 B) -- even lane groups g own tile 2p at
// d = 4g .. 4g+7, odd groups tile 2p+1 at d = 4(g-1) .. 4(g-1)+7 -- so a row leaves in half as many store instructions, 16 B each.
// (the swaps run with every lane of the wave active; `ok` guards only the store)
__device__ __forceinline__ void store_rows16(bf16_t* row, int g, const f32x4 (&t)[4], float scale, bool ok) {
  uint32_t lo[4], hi[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    lo[dt] = pack_bf16x2(t[dt][0] * scale, t[dt][1] * scale);
    hi[dt] = pack_bf16x2(t[dt][2] * scale, t[dt][3] * scale);
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const auto a = __builtin_amdgcn_permlane16_swap(lo[2 * p], lo[2 * p + 1], false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi[2 * p], hi[2 * p + 1], false, false);
    // even g: {own tile 2p (d 4g..4g+3), tile 2p of group g+1 (d 4g+4..4g+7)}; odd g: {tile 2p+1 of group g-1, own tile 2p+1}
    const int d0 = (g & 1) ? (2 * p + 1) * 16 + 4 * (g - 1) : 2 * p * 16 + 4 * g;
    if (ok) *reinterpret_cast<uint4*>(row + d0) = make_uint4(a[0], b[0], a[1], b[1]);
  }
}

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
