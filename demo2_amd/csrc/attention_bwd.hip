// Backward of the short-sequence ViT self-attention (L <= 160 tokens, head dim 64).
//
// Inputs per (image, head): Q, K, V (from the packed qkv), O, dO and the forward's LSE.  Outputs dQ, dK, dV packed
// like qkv.  Recompute instead of storing the L x L probabilities:
//     S = Q K^T * scale,  P = exp(S - LSE),  dV = P^T dO,  dP = dO V^T,  delta = rowsum(dO * O),
//     dS = P * (dP - delta) * scale,  dQ = dS K,  dK = dS^T Q.
//
// bf16 kernel: one 256-thread workgroup per (image, head); Q, dO, K, V zero-padded to NT*16 rows in LDS (128-B rows,
// 16-B chunk index ^= row&6: conflict free for ds_read_b128 row reads AND for the transposing reads below).
// Orientation follows the CDNA4 "key on the lane" recipe.
//   pass A (a wave owns KEY tiles, walks all queries):  S and dP tiles are mfma(A = Q / dO rows, B = K / V rows), so an
//     accumulator holds 4 QUERIES (registers) x 1 KEY (lane); converted to bf16, two such tiles are directly the B operand of
//         dV^T += dO^T P,   dK^T += Q^T dS        (sum over queries; A = ds_read_b64_tr_b16 of dO / Q)
//     with no cross-lane movement and no cross-wave sum.
//   pass B (a wave owns QUERY tiles, walks all keys):   the mirrored form, S^T / dP^T = mfma(A = K / V rows, B = Q / dO rows)
//     = 4 KEYS (registers) x 1 QUERY (lane), recomputed rather than exchanged: its bf16 image is the B operand of
//         dQ^T += K^T dS^T                         (sum over keys; A = transposing read of K).
//     Recomputing S and dP costs 8 MFMAs per 32 keys; the alternative (dS^T through LDS) costs 56 KB of LDS and a
//     barrier, and it is LDS capacity that decides whether TWO workgroups fit a CU: with NT = 9 (L <= 144, the 129-token
//     ViT sequence) the footprint is 73 KB, so one workgroup's global-load phase hides under the other's MFMA phase.
// The k index inside every 32-deep MFMA step is permuted identically on both operands ({4g+j, 16+4g+j}), which the sum
// does not see.  Tiles past NT (odd NT: second half of the last pair) are masked to zero contributions.
//
// generic kernel: fp32 math on the vector ALUs (parity mode and shapes outside the MFMA kernel's contract).
#include "common.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

constexpr int LP_MAX = 160;                   // 10 tiles of 16 rows

__device__ __forceinline__ int swz(int r, int c) { return r * 128 + ((c ^ (r & 6)) << 4); }

__device__ __forceinline__ bf16x8 tr_frag(const char* base, int off_lo, int off_hi) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + off_lo));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + off_hi));
  return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// second half only when that tile exists (wave-uniform), else zeros
__device__ __forceinline__ bf16x8 tr_frag_opt(const char* base, int off_lo, int off_hi, bool hi_ok) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + off_lo));
  s16x4 hi = {0, 0, 0, 0};
  if (hi_ok) hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + off_hi));
  return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <int NT>
__global__ __launch_bounds__(256, NT <= 9 ? 2 : 1) void attn_bwd_bf16_kernel(const bf16_t* __restrict__ qkv,
                                                                             const bf16_t* __restrict__ out,
                                                                             const bf16_t* __restrict__ dout,
                                                                             const float* __restrict__ lse,
                                                                             bf16_t* __restrict__ dqkv, int L, int H, float scale) {
  constexpr int LP = NT * 16, MAT_BYTES = LP * 128, NIT = (LP * 8 + 255) / 256;
  __shared__ __attribute__((aligned(1024))) char smem[4 * MAT_BYTES + 2 * LP * 4];
  char* Qs = smem;
  char* Gs = smem + MAT_BYTES;                // dO
  char* Ks = smem + 2 * MAT_BYTES;
  char* Vs = smem + 3 * MAT_BYTES;
  float* lse2 = reinterpret_cast<float*>(smem + 4 * MAT_BYTES);   // LSE * log2(e)
  float* dlt = lse2 + LP;                     // delta

  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int D = H * 64;
  const int64_t ld3 = 3 * (int64_t)D;
  const bf16_t* qbase = qkv + (int64_t)b * L * ld3 + h * 64;
  const bf16_t* obase = out + (int64_t)b * L * D + h * 64;
  const bf16_t* gbase = dout + (int64_t)b * L * D + h * 64;
  const int tid = threadIdx.x;

  // ---- stage Q, K, V, dO (all loads issued before the LDS writes), delta = rowsum(dO * O) ------------------
  {
    uint4 vq[NIT], vk[NIT], vv[NIT], vg[NIT], vo[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * 256, r = idx >> 3, c = idx & 7;
      const uint4 z = make_uint4(0u, 0u, 0u, 0u);
      vq[it] = vk[it] = vv[it] = vg[it] = vo[it] = z;
      if (r < L) {
        const bf16_t* p = qbase + (int64_t)r * ld3 + c * 8;
        vq[it] = *reinterpret_cast<const uint4*>(p);
        vk[it] = *reinterpret_cast<const uint4*>(p + D);
        vv[it] = *reinterpret_cast<const uint4*>(p + 2 * D);
        vg[it] = *reinterpret_cast<const uint4*>(gbase + (int64_t)r * D + c * 8);
        vo[it] = *reinterpret_cast<const uint4*>(obase + (int64_t)r * D + c * 8);
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * 256, r = idx >> 3, c = idx & 7;
      const uint32_t gw[4] = {vg[it].x, vg[it].y, vg[it].z, vg[it].w};
      const uint32_t ow[4] = {vo[it].x, vo[it].y, vo[it].z, vo[it].w};
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        d = fmaf(__uint_as_float(gw[e] << 16), __uint_as_float(ow[e] << 16), d);
        d = fmaf(__uint_as_float(gw[e] & 0xffff0000u), __uint_as_float(ow[e] & 0xffff0000u), d);
      }
      d += __shfl_xor(d, 1, 64);
      d += __shfl_xor(d, 2, 64);
      d += __shfl_xor(d, 4, 64);
      if (r < LP) {
        const int o = swz(r, c);
        *reinterpret_cast<uint4*>(Qs + o) = vq[it];
        *reinterpret_cast<uint4*>(Ks + o) = vk[it];
        *reinterpret_cast<uint4*>(Vs + o) = vv[it];
        *reinterpret_cast<uint4*>(Gs + o) = vg[it];
        if (c == 0) dlt[r] = d;
      }
    }
    if (tid < LP) lse2[tid] = tid < L ? lse[((int64_t)b * H + h) * L + tid] * 1.4426950408889634f : 0.f;
  }
  __syncthreads();

  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  // row-read fragment offsets for rows aligned to 16: row li, chunk (kk*4 + g)
  const int fo0 = swz(li, g), fo1 = swz(li, 4 + g);
  // transposing-read offsets: lane idx = 4q+pp -> row 4g+q (second read +16 rows), 8 B at column dt*16 + 4pp
  const int tq = li >> 2, tp = li & 3, trow = 4 * g + tq;
  int to[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) to[dt] = swz(trow, dt * 2 + (tp >> 1)) + ((tp & 1) << 3);
  const float c2 = scale * 1.4426950408889634f;
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  // ================= pass A: this wave's key tiles (kt = wave, wave+4, wave+8 < NT); all query pairs ==============
  {
    f32x4 dVa[3][4], dKa[3][4];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) { dVa[a][dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dKa[a][dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    // body of one query pair; T1 = the pair's second tile exists -- a compile-time fact (false only for the last pair of an
    // odd NT), so the full pairs carry no masking selects at all
    auto pass_a = [&](int qp, auto t1c) {
      constexpr bool t1 = decltype(t1c)::value;
      // row fragments of the two query tiles (A operands of S and dP)
      bf16x8 qf[2][2], gf[2][2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int rb = (qp * 2 + hh) * 2048;
        if (hh == 0 || t1) {
          qf[hh][0] = *reinterpret_cast<const bf16x8*>(Qs + rb + fo0);
          qf[hh][1] = *reinterpret_cast<const bf16x8*>(Qs + rb + fo1);
          gf[hh][0] = *reinterpret_cast<const bf16x8*>(Gs + rb + fo0);
          gf[hh][1] = *reinterpret_cast<const bf16x8*>(Gs + rb + fo1);
        } else {
          qf[hh][0] = qf[hh][1] = gf[hh][0] = gf[hh][1] = zero8;
        }
      }
      // transposed fragments (A operands of dV^T and dK^T): k = queries {4g+j, 16+4g+j} of this pair
      bf16x8 qt[4], gt[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        qt[dt] = tr_frag_opt(Qs + qp * 4096, to[dt], to[dt] + 2048, t1);
        gt[dt] = tr_frag_opt(Gs + qp * 4096, to[dt], to[dt] + 2048, t1);
      }
      // per-lane row constants: queries qbase + 4g + r for both tiles
      float l2[2][4], dl[2][4];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        // a tile that does not exist (odd NT, second half of the last pair) gets LSE = +inf: p = exp2(-inf) = 0 and
        // dS = 0 with no per-element selects
        float4 a4 = make_float4(INFINITY, INFINITY, INFINITY, INFINITY), d4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (hh == 0 || t1) {
          a4 = *reinterpret_cast<const float4*>(lse2 + (qp * 2 + hh) * 16 + 4 * g);
          d4 = *reinterpret_cast<const float4*>(dlt + (qp * 2 + hh) * 16 + 4 * g);
        }
        l2[hh][0] = a4.x; l2[hh][1] = a4.y; l2[hh][2] = a4.z; l2[hh][3] = a4.w;
        dl[hh][0] = d4.x; dl[hh][1] = d4.y; dl[hh][2] = d4.z; dl[hh][3] = d4.w;
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const int kt = wave + 4 * a;
        if (kt < NT) {                                       // wave-uniform
          const bf16x8 kf0 = *reinterpret_cast<const bf16x8*>(Ks + kt * 2048 + fo0);
          const bf16x8 kf1 = *reinterpret_cast<const bf16x8*>(Ks + kt * 2048 + fo1);
          const bf16x8 vf0 = *reinterpret_cast<const bf16x8*>(Vs + kt * 2048 + fo0);
          const bf16x8 vf1 = *reinterpret_cast<const bf16x8*>(Vs + kt * 2048 + fo1);
          u32x4 pw, dw;                                      // bf16 pairs, packed two per v_cvt_pk_bf16_f32
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[hh][0], kf0, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[hh][1], kf1, s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[hh][0], vf0, dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[hh][1], vf1, dp, 0, 0, 0);
            float pv[4], dv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              pv[r] = __builtin_amdgcn_exp2f(s[r] * c2 - l2[hh][r]);
              dv[r] = pv[r] * (dp[r] - dl[hh][r]);               // (* scale: folded into dK at the store -- 16 multiplies per lane and key tile instead of one per element)
            }
            pw[hh * 2] = pack_bf16x2(pv[0], pv[1]); pw[hh * 2 + 1] = pack_bf16x2(pv[2], pv[3]);
            dw[hh * 2] = pack_bf16x2(dv[0], dv[1]); dw[hh * 2 + 1] = pack_bf16x2(dv[2], dv[3]);
          }
          const bf16x8 pb = __builtin_bit_cast(bf16x8, pw), db = __builtin_bit_cast(bf16x8, dw);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            dVa[a][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt[dt], pb, dVa[a][dt], 0, 0, 0);
            dKa[a][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt[dt], db, dKa[a][dt], 0, 0, 0);
          }
        }
      }
    };
    for (int qp = 0; qp < NT / 2; ++qp) pass_a(qp, std::true_type{});
    if constexpr (NT & 1) pass_a(NT / 2, std::false_type{});
    // dK^T / dV^T tiles: D[row = d = 4g + reg][col = key = li] -> 4 consecutive d per lane, 8 after the row-pair exchange (store_rows16)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int kt = wave + 4 * a;
      if (kt < NT) {
        const int key = kt * 16 + li;
        bf16_t* row = dqkv + ((int64_t)b * L + (key < L ? key : 0)) * ld3 + h * 64;
        store_rows16(row + D, g, dKa[a], scale, key < L);
        store_rows16(row + 2 * D, g, dVa[a], 1.0f, key < L);
      }
    }
  }

  // ================= pass B: this wave's query tiles; all key pairs ================================================
  // tiles are dealt in the opposite wave order to pass A (3 - wave, 7 - wave, 11 - wave): with NT = 9 the ninth tile goes
  // to wave 0 in pass A and to wave 3 here, so no wave carries 3 + 3 tiles (no barrier separates the passes)
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int qtile = (3 - wave) + 4 * a;
    if (qtile < NT && qtile * 16 < L) {                      // wave-uniform
      const bf16x8 qf0 = *reinterpret_cast<const bf16x8*>(Qs + qtile * 2048 + fo0);
      const bf16x8 qf1 = *reinterpret_cast<const bf16x8*>(Qs + qtile * 2048 + fo1);
      const bf16x8 gf0 = *reinterpret_cast<const bf16x8*>(Gs + qtile * 2048 + fo0);
      const bf16x8 gf1 = *reinterpret_cast<const bf16x8*>(Gs + qtile * 2048 + fo1);
      const float l2q = lse2[qtile * 16 + li], dlq = dlt[qtile * 16 + li];      // this lane's query
      f32x4 dq[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      auto pass_b = [&](int kp, auto t1c) {
        constexpr bool t1 = decltype(t1c)::value;
        u32x4 dsw;                   // (a missing key tile needs no masking here: its K^T fragment half below is zero)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const bool tv = hh == 0 || t1;
          const int kb = (kp * 2 + hh) * 2048;
          bf16x8 kf0 = zero8, kf1 = zero8, vf0 = zero8, vf1 = zero8;
          if (tv) {
            kf0 = *reinterpret_cast<const bf16x8*>(Ks + kb + fo0);
            kf1 = *reinterpret_cast<const bf16x8*>(Ks + kb + fo1);
            vf0 = *reinterpret_cast<const bf16x8*>(Vs + kb + fo0);
            vf1 = *reinterpret_cast<const bf16x8*>(Vs + kb + fo1);
          }
          // D[row = key 4g + reg][col = query li]
          f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf0, qf0, s, 0, 0, 0);
          s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf1, qf1, s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf0, gf0, dp, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf1, gf1, dp, 0, 0, 0);
          float dv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) dv[r] = __builtin_amdgcn_exp2f(s[r] * c2 - l2q) * (dp[r] - dlq);      // (* scale: at the dQ store)
          dsw[hh * 2] = pack_bf16x2(dv[0], dv[1]);
          dsw[hh * 2 + 1] = pack_bf16x2(dv[2], dv[3]);
        }
        const bf16x8 dsb = __builtin_bit_cast(bf16x8, dsw);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const bf16x8 ka = tr_frag_opt(Ks + kp * 4096, to[dt], to[dt] + 2048, t1);
          dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, dsb, dq[dt], 0, 0, 0);
        }
      };
      for (int kp = 0; kp < NT / 2; ++kp) pass_b(kp, std::true_type{});
      if constexpr (NT & 1) pass_b(NT / 2, std::false_type{});
      const int q = qtile * 16 + li;
      store_rows16(dqkv + ((int64_t)b * L + (q < L ? q : 0)) * ld3 + h * 64, g, dq, scale, q < L);
    }
  }
}

// ---- generic fp32-math kernel: one workgroup per (image, head), one wave per query row --------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_generic_kernel(const T* __restrict__ qkv, const T* __restrict__ out,
                                                               const T* __restrict__ dout, const float* __restrict__ lse,
                                                               T* __restrict__ dqkv, int L, int H, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ks = sm;                  // [L][65]
  float* Vs = Ks + L * 65;         // [L][65]
  float* dKs = Vs + L * 65;        // [L][64]
  float* dVs = dKs + L * 64;       // [L][64]
  float* wv = dVs + L * 64;        // per wave: q[64], do[64], ds[L], p[L]
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int D = H * 64;
  const int64_t ld3 = 3 * (int64_t)D;
  const T* base = qkv + (int64_t)b * L * ld3 + h * 64;
  for (int i = threadIdx.x; i < L * 64; i += blockDim.x) {
    const int r = i >> 6, c = i & 63;
    Ks[r * 65 + c] = io<T>::ld(base + (int64_t)r * ld3 + D + c);
    Vs[r * 65 + c] = io<T>::ld(base + (int64_t)r * ld3 + 2 * D + c);
    dKs[i] = 0.f;
    dVs[i] = 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* qw = wv + wave * (128 + 2 * L);
  float* gw = qw + 64;
  float* dsw = gw + 64;
  float* pw = dsw + L;
  for (int q = wave; q < L; q += 4) {
    const int64_t orow = ((int64_t)b * L + q) * D + h * 64;
    qw[lane] = io<T>::ld(base + (int64_t)q * ld3 + lane);
    const float gl = io<T>::ld(dout + orow + lane);
    gw[lane] = gl;
    const float delta = wave_sum(gl * io<T>::ld(out + orow + lane));
    const float lq = lse[((int64_t)b * H + h) * L + q];
    __builtin_amdgcn_wave_barrier();
    for (int key = lane; key < L; key += 64) {
      float s = 0.f, dp = 0.f;
      for (int d = 0; d < 64; ++d) {
        s = fmaf(qw[d], Ks[key * 65 + d], s);
        dp = fmaf(gw[d], Vs[key * 65 + d], dp);
      }
      const float p = __expf(s * scale - lq);
      pw[key] = p;
      dsw[key] = p * (dp - delta) * scale;
    }
    __builtin_amdgcn_wave_barrier();
    float dq = 0.f;
    for (int key = 0; key < L; ++key) {
      const float ds = dsw[key], p = pw[key];
      dq = fmaf(ds, Ks[key * 65 + lane], dq);
      atomicAdd(&dKs[key * 64 + lane], ds * qw[lane]);
      atomicAdd(&dVs[key * 64 + lane], p * gl);
    }
    io<T>::st(dqkv + ((int64_t)b * L + q) * ld3 + h * 64 + lane, dq);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  for (int i = threadIdx.x; i < L * 64; i += blockDim.x) {
    const int r = i >> 6, c = i & 63;
    T* row = dqkv + ((int64_t)b * L + r) * ld3 + h * 64 + c;
    io<T>::st(row + D, dKs[i]);
    io<T>::st(row + 2 * D, dVs[i]);
  }
}

}  // namespace

extern "C" int demo_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int32_t Bn,
                                  int32_t L, int32_t H, int32_t dh, int32_t dt, void* stream) {
  DEMO_REQUIRE(dh == 64, "demo_attention_bwd: head dim %d unsupported (64 only)", dh);
  DEMO_REQUIRE(L >= 1 && L <= 256, "demo_attention_bwd: L=%d unsupported (1..256)", L);
  DEMO_REQUIRE(lse != nullptr, "demo_attention_bwd: the forward LSE is required");
  hipStream_t s = (hipStream_t)stream;
  const float scale = 0.125f;
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (dt == DEMO_BF16 && L <= LP_MAX && al(qkv) && al(out) && al(dout) && al(dqkv)) {
    if (L <= 144)       // 9 row tiles: 73 KB of LDS, two workgroups per CU
      hipLaunchKernelGGL(attn_bwd_bf16_kernel<9>, dim3(Bn * H), dim3(256), 0, s, (const bf16_t*)qkv, (const bf16_t*)out,
                         (const bf16_t*)dout, lse, (bf16_t*)dqkv, L, H, scale);
    else
      hipLaunchKernelGGL(attn_bwd_bf16_kernel<10>, dim3(Bn * H), dim3(256), 0, s, (const bf16_t*)qkv, (const bf16_t*)out,
                         (const bf16_t*)dout, lse, (bf16_t*)dqkv, L, H, scale);
  } else {
    const size_t shm = sizeof(float) * (2 * (size_t)L * 65 + 2 * (size_t)L * 64 + 4 * (128 + 2 * (size_t)L));
    DEMO_REQUIRE(shm <= 160 * 1024, "demo_attention_bwd: L=%d needs %zu B of LDS", L, shm);
    hipError_t e = dt == DEMO_BF16
        ? hipFuncSetAttribute((const void*)attn_bwd_generic_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)
        : hipFuncSetAttribute((const void*)attn_bwd_generic_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) DEMO_FAIL(-2, "demo_attention_bwd: cannot reserve %zu B of LDS: %s", shm, hipGetErrorString(e));
    if (dt == DEMO_BF16)
      hipLaunchKernelGGL(attn_bwd_generic_kernel<bf16_t>, dim3(Bn * H), dim3(256), shm, s, (const bf16_t*)qkv, (const bf16_t*)out,
                         (const bf16_t*)dout, lse, (bf16_t*)dqkv, L, H, scale);
    else
      hipLaunchKernelGGL(attn_bwd_generic_kernel<float>, dim3(Bn * H), dim3(256), shm, s, (const float*)qkv, (const float*)out,
                         (const float*)dout, lse, (float*)dqkv, L, H, scale);
  }
  DEMO_LAUNCH_CHECK("demo_attention_bwd");
  return 0;
}
