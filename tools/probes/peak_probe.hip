// Measured ceilings of the box the benchmarks run on ("confirm the vendor peaks with a microbenchmark before quoting
// fractions").  Stand-alone:  hipcc --offload-arch=gfx950 -O3 tools/probes/peak_probe.hip -o tools/probes/peak_probe && ./peak_probe
//   1. bf16 MFMA: every wave issues v_mfma_f32_16x16x32_bf16 back to back on register operands (8 independent accumulators, random
//      data, no memory traffic), 1 / 2 waves per SIMD on all CUs -> TFLOP/s the matrix pipes deliver at the clock the chip holds;
//   2. HBM: a grid-stride float4 copy over buffers far larger than the 256-MB Infinity Cache (read + write) and a read-only sum.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(512) void mfma_loop(const uint32_t* __restrict__ seed, float* __restrict__ out, int iters) {
  // operands: pseudo-random bf16 bit patterns of moderate magnitude (exponent 0x3f..), different per lane
  uint32_t s = seed[threadIdx.x & 63] + blockIdx.x * 2654435761u + threadIdx.x;
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    s = s * 1664525u + 1013904223u;
    a[i] = (short)(0x3f00 | ((s >> 8) & 0xff));
    s = s * 1664525u + 1013904223u;
    b[i] = (short)(0x3f00 | ((s >> 8) & 0xff) | ((s >> 1) & 0x8000));
  }
  f32x4 acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);
  }
  float r = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) r += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  if (r == 12345.678f) out[0] = r;             // keeps the loop alive without a store on the hot path
}

__global__ __launch_bounds__(256) void copy4(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n4) {
  for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void read4(const float4* __restrict__ src, float* __restrict__ out, int64_t n4) {
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = src[i];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = s.x;
}

static float time_ms(hipEvent_t e0, hipEvent_t e1) { float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms; }

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device: %s, %d CUs, clock %.2f GHz (reported max)\n", prop.gcnArchName, cus, prop.clockRate / 1e6);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  uint32_t hseed[64];
  for (int i = 0; i < 64; ++i) hseed[i] = 1234567u * (i + 1);
  uint32_t* dseed; float* dout;
  CK(hipMalloc(&dseed, sizeof(hseed))); CK(hipMalloc(&dout, 64));
  CK(hipMemcpy(dseed, hseed, sizeof(hseed), hipMemcpyHostToDevice));
  const int iters = 20000;
  for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
    const int threads = 256 * waves_per_simd;               // 4 SIMDs per CU
    hipLaunchKernelGGL(mfma_loop, dim3(cus), dim3(threads), 0, 0, dseed, dout, 2000);    // warm-up (clock ramp)
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(mfma_loop, dim3(cus), dim3(threads), 0, 0, dseed, dout, iters);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      const float ms = time_ms(e0, e1);
      best = ms < best ? ms : best;
    }
    const double flop = 2.0 * 16 * 16 * 32 * 8.0 * iters * (threads / 64) * cus;
    printf("bf16 MFMA 16x16x32, %d wave(s)/SIMD, random operands: %.1f ms -> %.0f TFLOP/s (%.1f %% of 2500)\n", waves_per_simd, best,
           flop / (best * 1e-3) / 1e12, flop / (best * 1e-3) / 1e12 / 25.0);
  }
  const int64_t bytes = 4LL << 30;                            // 4 GiB per buffer
  float4 *src, *dst;
  CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes));
  CK(hipMemset(src, 1, bytes)); CK(hipMemset(dst, 0, bytes));
  const int64_t n4 = bytes / 16;
  for (int pass = 0; pass < 2; ++pass) {
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0));
      if (pass == 0) hipLaunchKernelGGL(copy4, dim3(cus * 16), dim3(256), 0, 0, src, dst, n4);
      else hipLaunchKernelGGL(read4, dim3(cus * 16), dim3(256), 0, 0, src, dout, n4);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      const float ms = time_ms(e0, e1);
      best = ms < best ? ms : best;
    }
    const double moved = pass == 0 ? 2.0 * bytes : 1.0 * bytes;
    printf("HBM %s of 4 GiB: %.2f ms -> %.2f TB/s\n", pass == 0 ? "copy (read + write)" : "read-only sum", best, moved / (best * 1e-3) / 1e12);
  }
  return 0;
}
