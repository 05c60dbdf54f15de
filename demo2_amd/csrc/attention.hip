// ViT self-attention for short sequences (L <= 160: 128 patches + cls = 129), head dim 64.
//
// bf16 kernel: one 256-thread workgroup (4 waves) per (image, head).  Q, K, V of that head live
// entirely in LDS (3 x 160 rows x 128 B = 60 KiB, two workgroups per CU), no tiling over keys and no online
// rescaling is needed.  The 10 query tiles of 16 rows are dealt to the 4 waves.
//   S^T = K Q^T   via v_mfma_f32_16x16x32_bf16 with K rows as the A operand, so that a lane
//                 holds ONE query (its column) and 4 keys per tile in registers: the softmax
//                 row reductions are register maxima/sums plus two cross-lane shuffles;
//   O^T = V^T P^T the S^T accumulators, converted to bf16, are directly the B operand (k index
//                 permuted inside each 32-key step); V^T fragments with the same permutation come
//                 from ds_read_b64_tr_b16 (hardware transpose read) of the row-major V tile.
// LDS rows are 128 B, so both images carry an XOR swizzle on the 16-B chunk index
// (Q/K: (row>>1)&7 for ds_read_b128 row reads; V: row&6 for the transposed reads).
//
// generic kernel: fp32 math on the vector ALUs for the fp32 parity mode (any L up to 256).
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

constexpr int QROWS = 160, KROWS = 160;
constexpr int Q_BYTES = QROWS * 128, K_BYTES = KROWS * 128;

// ---- occupancy form (the default): K and V only in LDS, NT key tiles (9 for the 129-token ViT sequence: 36 KB -> FOUR workgroups per
// CU instead of two); a wave's Q fragments come straight from global memory (a lane's MFMA operand is 16 contiguous bytes of its
// query row), requested before the K/V staging so their latency hides under it.  The kernel is HBM-bound and a workgroup's life is
// load -> compute -> store: with two workgroups per CU the memory system idles whenever both compute (3.3 TB/s, PMC bytes / time);
// four keep loads in flight.
template <int NT>
__global__ __launch_bounds__(256, 4) void attn_fwd_bf16_kv_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                  float* __restrict__ lse, int L, int H, float scale) {
  constexpr int ROWS = NT * 16, MAT = ROWS * 128, NIT = (ROWS * 8 + 255) / 256;
  __shared__ __attribute__((aligned(1024))) char smem[2 * MAT];
  char* Ks = smem;
  char* Vs = smem + MAT;
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int D = H * 64;
  const int64_t ld = 3 * (int64_t)D;
  const bf16_t* base = qkv + (int64_t)b * L * ld + h * 64;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, li = lane & 15;

  // this wave's query tiles (wave, wave + 4, wave + 8): fragment rows li, k-halves {g*8.., 32 + g*8..}
  bf16x8 qf[3][2];
#pragma unroll
  for (int qq = 0; qq < 3; ++qq) {
    const int r = (wave + 4 * qq) * 16 + li;
    uint4 a = make_uint4(0u, 0u, 0u, 0u), c = a;
    if (r < L) {
      const bf16_t* p = base + (int64_t)r * ld + g * 8;
      a = *reinterpret_cast<const uint4*>(p);
      c = *reinterpret_cast<const uint4*>(p + 32);
    }
    qf[qq][0] = __builtin_bit_cast(bf16x8, a);
    qf[qq][1] = __builtin_bit_cast(bf16x8, c);
  }
  {
    uint4 vk[NIT], vv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = threadIdx.x + it * 256, r = idx >> 3, c = idx & 7;
      vk[it] = vv[it] = make_uint4(0u, 0u, 0u, 0u);
      if (r < L) {
        const bf16_t* p = base + (int64_t)r * ld + c * 8;
        vk[it] = *reinterpret_cast<const uint4*>(p + D);
        vv[it] = *reinterpret_cast<const uint4*>(p + 2 * D);
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = threadIdx.x + it * 256, r = idx >> 3, c = idx & 7;
      if (r < ROWS) {
        *reinterpret_cast<uint4*>(Ks + r * 128 + ((c ^ ((r >> 1) & 7)) << 4)) = vk[it];
        *reinterpret_cast<uint4*>(Vs + r * 128 + ((c ^ (r & 6)) << 4)) = vv[it];
      }
    }
  }
  __syncthreads();

  const int sw = (lane >> 1) & 7;
  const int fo0 = li * 128 + ((g ^ sw) << 4);
  const int fo1 = li * 128 + (((4 + g) ^ sw) << 4);
  const int tq = li >> 2, tp = li & 3;
  const int trow = 4 * g + tq;
  int vo[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) vo[dt] = trow * 128 + (((dt * 2 + (tp >> 1)) ^ (trow & 6)) << 4) + ((tp & 1) << 3);
  const float c2 = scale * 1.4426950408889634f;   // softmax in base 2
  constexpr int NP = (NT + 1) / 2;                // 32-key steps of the PV product (the second half of an odd last step is masked)

#pragma unroll
  for (int qq = 0; qq < 3; ++qq) {
    const int qt = wave + 4 * qq;
    if (qt * 16 >= L || qt >= NT) break;           // wave-uniform
    f32x4 s[2 * NP];
#pragma unroll
    for (int kt = 0; kt < 2 * NP; ++kt) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      if (kt < NT) {
        const bf16x8 kf0 = *reinterpret_cast<const bf16x8*>(Ks + kt * 2048 + fo0);
        const bf16x8 kf1 = *reinterpret_cast<const bf16x8*>(Ks + kt * 2048 + fo1);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf0, qf[qq][0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf1, qf[qq][1], a, 0, 0, 0);
      }
      s[kt] = a;
    }
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2 * NP; ++kt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + 4 * g + r;
        if (key >= L) s[kt][r] = -INFINITY;
        m = fmaxf(m, s[kt][r]);
      }
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
    const float mc = m * c2;
#pragma unroll
    for (int kt = 0; kt < 2 * NP; ++kt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(s[kt][r] * c2 - mc);
        s[kt][r] = p;
        sum += p;
      }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);

    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NP; ++ks) {
      const uint32_t pw[4] = {pack_bf16x2(s[2 * ks][0], s[2 * ks][1]), pack_bf16x2(s[2 * ks][2], s[2 * ks][3]),
                              pack_bf16x2(s[2 * ks + 1][0], s[2 * ks + 1][1]), pack_bf16x2(s[2 * ks + 1][2], s[2 * ks + 1][3])};
      const bf16x8 pb = __builtin_bit_cast(bf16x8, (u32x4){pw[0], pw[1], pw[2], pw[3]});
      const bool hi_ok = 2 * ks + 1 < NT;          // compile-time after unrolling: the V rows past NT tiles are not in LDS
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(Vs + ks * 4096 + vo[dt]));
        s16x4 hi = {0, 0, 0, 0};
        if (hi_ok) hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(Vs + ks * 4096 + 2048 + vo[dt]));
        const bf16x8 va = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pb, o[dt], 0, 0, 0);
      }
    }
    const int q = qt * 16 + li;
    // 16-B row stores (store_rows16, common.h): the lanes exchanged hold the same query, so each scales by its own 1 / sum first
    store_rows16(out + ((int64_t)b * L + (q < L ? q : 0)) * D + h * 64, g, o, 1.f / sum, q < L);
    if (q < L && lse && g == 0) lse[((int64_t)b * H + h) * L + q] = m * scale + __logf(sum);
  }
}

// ---- round-1 form (Q, K and V in LDS, 160 rows each: 60 KB, two workgroups per CU); kept selectable for A/B runs
__global__ __launch_bounds__(256, 2) void attn_fwd_bf16_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                               float* __restrict__ lse, int L, int H, float scale) {
  __shared__ __attribute__((aligned(1024))) char smem[Q_BYTES + 2 * K_BYTES];
  char* Qs = smem;
  char* Ks = smem + Q_BYTES;
  char* Vs = smem + Q_BYTES + K_BYTES;

  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int D = H * 64;
  const int64_t ld = 3 * (int64_t)D;
  const bf16_t* base = qkv + (int64_t)b * L * ld + h * 64;

  // ---- stage Q, K, V (zero padded to 160 rows): all 15 global loads of a thread are issued before the first LDS
  //      write, so the HBM latency is paid once per workgroup instead of once per 16-B chunk ----------------------
  {
    uint4 vq[5], vk[5], vv[5];
#pragma unroll
    for (int it = 0; it < 5; ++it) {
      const int idx = threadIdx.x + it * 256, r = idx >> 3, c = idx & 7;
      vq[it] = vk[it] = vv[it] = make_uint4(0u, 0u, 0u, 0u);
      if (r < L) {
        const bf16_t* p = base + (int64_t)r * ld + c * 8;
        vq[it] = *reinterpret_cast<const uint4*>(p);
        vk[it] = *reinterpret_cast<const uint4*>(p + D);
        vv[it] = *reinterpret_cast<const uint4*>(p + 2 * D);
      }
    }
#pragma unroll
    for (int it = 0; it < 5; ++it) {
      const int idx = threadIdx.x + it * 256, r = idx >> 3, c = idx & 7;
      const int oqk = r * 128 + ((c ^ ((r >> 1) & 7)) << 4);
      *reinterpret_cast<uint4*>(Qs + oqk) = vq[it];
      *reinterpret_cast<uint4*>(Ks + oqk) = vk[it];
      *reinterpret_cast<uint4*>(Vs + r * 128 + ((c ^ (r & 6)) << 4)) = vv[it];
    }
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int sw = (lane >> 1) & 7;
  // row-read fragment offsets (rows aligned to 16): chunk (kk*4+g) ^ sw
  const int fo0 = li * 128 + ((g ^ sw) << 4);
  const int fo1 = li * 128 + (((4 + g) ^ sw) << 4);
  // transposed-read offsets into V for d-tile dt: row (4g + q), cols dt*16 + 4*pp
  const int tq = li >> 2, tp = li & 3;
  const int trow = 4 * g + tq;
  int vo[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) vo[dt] = trow * 128 + (((dt * 2 + (tp >> 1)) ^ (trow & 6)) << 4) + ((tp & 1) << 3);

  const float c2 = scale * 1.4426950408889634f;   // softmax in base 2

  for (int qq = 0; qq < 3; ++qq) {
    const int qt = wave + 4 * qq;                  // 10 query tiles over 4 waves: {0,4,8} {1,5,9} {2,6} {3,7}
    if (qt * 16 >= L || qt >= 10) break;           // wave-uniform
    const bf16x8 qf0 = *reinterpret_cast<const bf16x8*>(Qs + qt * 2048 + fo0);
    const bf16x8 qf1 = *reinterpret_cast<const bf16x8*>(Qs + qt * 2048 + fo1);

    f32x4 s[10];
#pragma unroll
    for (int kt = 0; kt < 10; ++kt) {
      const bf16x8 kf0 = *reinterpret_cast<const bf16x8*>(Ks + kt * 2048 + fo0);
      const bf16x8 kf1 = *reinterpret_cast<const bf16x8*>(Ks + kt * 2048 + fo1);
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf0, qf0, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf1, qf1, a, 0, 0, 0);
      s[kt] = a;
    }
    // mask padded keys, row max
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 10; ++kt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + 4 * g + r;
        if (key >= L) s[kt][r] = -INFINITY;
        m = fmaxf(m, s[kt][r]);
      }
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
    const float mc = m * c2;
#pragma unroll
    for (int kt = 0; kt < 10; ++kt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(s[kt][r] * c2 - mc);
        s[kt][r] = p;
        sum += p;
      }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);

    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      bf16x8 pb;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pb[r] = (short)f2bf(s[2 * ks][r]);
        pb[4 + r] = (short)f2bf(s[2 * ks + 1][r]);
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(Vs + ks * 4096 + vo[dt]));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(Vs + ks * 4096 + 2048 + vo[dt]));
        bf16x8 va;
        va[0] = lo[0]; va[1] = lo[1]; va[2] = lo[2]; va[3] = lo[3];
        va[4] = hi[0]; va[5] = hi[1]; va[6] = hi[2]; va[7] = hi[3];
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pb, o[dt], 0, 0, 0);
      }
    }
    const int q = qt * 16 + li;
    // 16-B row stores (store_rows16, common.h): the lanes exchanged hold the same query, so each scales by its own 1 / sum first
    store_rows16(out + ((int64_t)b * L + (q < L ? q : 0)) * D + h * 64, g, o, 1.f / sum, q < L);
    if (q < L && lse && g == 0) lse[((int64_t)b * H + h) * L + q] = m * scale + __logf(sum);
  }
}

// ---- generic fp32-math kernel (T in/out), one workgroup per (image, head), L <= 256 -----------
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_generic_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                               float* __restrict__ lse, int L, int H, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ks = sm;                    // [L][65]
  float* Vs = Ks + L * 65;           // [L][64]
  float* Ps = Vs + L * 64;           // [4 waves][L]
  float* Qw = Ps + 4 * L;            // [4 waves][64]
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int D = H * 64;
  const int64_t ld = 3 * (int64_t)D;
  const T* base = qkv + (int64_t)b * L * ld + h * 64;
  for (int i = threadIdx.x; i < L * 64; i += blockDim.x) {
    const int r = i >> 6, c = i & 63;
    Ks[r * 65 + c] = io<T>::ld(base + (int64_t)r * ld + D + c);
    Vs[r * 64 + c] = io<T>::ld(base + (int64_t)r * ld + 2 * D + c);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* pw = Ps + wave * L;
  float* qw = Qw + wave * 64;
  for (int q = wave; q < L; q += 4) {
    qw[lane] = io<T>::ld(base + (int64_t)q * ld + lane);
    __builtin_amdgcn_wave_barrier();
    float sc[4];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = lane + i * 64;
      float a = -INFINITY;
      if (key < L) {
        a = 0.f;
        for (int d = 0; d < 64; ++d) a = fmaf(qw[d], Ks[key * 65 + d], a);
        a *= scale;
      }
      sc[i] = a;
      m = fmaxf(m, a);
    }
    m = wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = lane + i * 64;
      if (key < L) {
        const float p = __expf(sc[i] - m);
        pw[key] = p;
        sum += p;
      }
    }
    sum = wave_sum(sum);
    __builtin_amdgcn_wave_barrier();
    float o = 0.f;
    for (int key = 0; key < L; ++key) o = fmaf(pw[key], Vs[key * 64 + lane], o);
    io<T>::st(out + ((int64_t)b * L + q) * D + h * 64 + lane, o / sum);
    if (lse && lane == 0) lse[((int64_t)b * H + h) * L + q] = m + __logf(sum);
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

static int g_attn_variant = 1;     // 1: K/V-in-LDS occupancy form, 0: round-1 form
extern "C" int demo_attention_set_variant(int v) { g_attn_variant = v; return 0; }

extern "C" int demo_attention_fwd(const void* qkv, void* out, float* lse, int32_t Bn, int32_t L, int32_t H, int32_t dh,
                                  int32_t dt, void* stream) {
  DEMO_REQUIRE(dh == 64, "demo_attention_fwd: head dim %d unsupported (64 only)", dh);
  DEMO_REQUIRE(L >= 1 && L <= 256, "demo_attention_fwd: L=%d unsupported (1..256)", L);
  hipStream_t s = (hipStream_t)stream;
  const float scale = 0.125f;
  if (dt == DEMO_BF16 && L <= QROWS && (((uintptr_t)qkv & 15) == 0) && (((uintptr_t)out & 7) == 0)) {
    if (g_attn_variant == 1 && L <= 144)
      hipLaunchKernelGGL(attn_fwd_bf16_kv_kernel<9>, dim3(Bn * H), dim3(256), 0, s, (const bf16_t*)qkv, (bf16_t*)out, lse, L, H, scale);
    else if (g_attn_variant == 1)
      hipLaunchKernelGGL(attn_fwd_bf16_kv_kernel<10>, dim3(Bn * H), dim3(256), 0, s, (const bf16_t*)qkv, (bf16_t*)out, lse, L, H, scale);
    else
      hipLaunchKernelGGL(attn_fwd_bf16_kernel, dim3(Bn * H), dim3(256), 0, s, (const bf16_t*)qkv, (bf16_t*)out, lse, L, H, scale);
  } else {
    const size_t shm = sizeof(float) * ((size_t)L * 65 + (size_t)L * 64 + 4 * (size_t)L + 4 * 64);
    if (shm > 48 * 1024) {   // opt in to the large-LDS carve-out (160 KiB per CU on gfx950)
      hipError_t e = dt == DEMO_BF16
          ? hipFuncSetAttribute((const void*)attn_fwd_generic_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)
          : hipFuncSetAttribute((const void*)attn_fwd_generic_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      if (e != hipSuccess) DEMO_FAIL(-2, "demo_attention_fwd: cannot reserve %zu B of LDS: %s", shm, hipGetErrorString(e));
    }
    if (dt == DEMO_BF16)
      hipLaunchKernelGGL(attn_fwd_generic_kernel<bf16_t>, dim3(Bn * H), dim3(256), shm, s, (const bf16_t*)qkv, (bf16_t*)out, lse, L, H, scale);
    else
      hipLaunchKernelGGL(attn_fwd_generic_kernel<float>, dim3(Bn * H), dim3(256), shm, s, (const float*)qkv, (float*)out, lse, L, H, scale);
  }
  DEMO_LAUNCH_CHECK("demo_attention_fwd");
  return 0;
}
