// Probe: what bounds the 256x256 GEMM epilogue's output stores -- the CU's own store path or the chip (L2 / fabric / HBM)?
// One 512-thread workgroup per CU-slot writes one 256 x 256 bf16 tile (128 KB) with 16-B stores in three patterns, for grids of
// 32 .. 1024 workgroups (<= 256: one workgroup per CU, all concurrent).  If time per launch does not depend on the number of
// concurrent workgroups, the limit is per CU; if it grows with them, it is shared.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/store_probe.hip -o tools/probes/store_probe && tools/probes/store_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// PATTERN 0: the register epilogue's map -- a wave instruction covers 16 rows x 64 B (4 lanes per row), two instructions complete a
//            128-B line, rows N*2 bytes apart;  1: whole 512-B tile rows (32 lanes per row, 2 rows per instruction);
//            2: the tile as ONE contiguous 128-KB block (blocked output layout)
template <int PATTERN>
__global__ __launch_bounds__(512, 1) void writer(char* __restrict__ C, int N, int ntn, int keep_lds, int xcd_mask = 0xff) {
  if (!((xcd_mask >> (blockIdx.x & 7)) & 1)) return;          // (round 3) only the workgroups dealt to the XCDs of the mask store
  extern __shared__ char sm[];
  if (keep_lds == 12345) sm[threadIdx.x] = 1;                // keep the 160-KB allocation: one workgroup per CU
  const int tile = blockIdx.x, tm = tile / ntn, tn = tile % ntn;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u32x4 v = {(uint32_t)tile, (uint32_t)lane, (uint32_t)wave, 7u};
  if (PATTERN == 0) {
    const int wr = wave >> 2, wc = wave & 3, frow = lane & 15, g4 = lane >> 4;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int64_t row = (int64_t)tm * 256 + wr * 128 + i * 16 + frow, col = tn * 256 + wc * 64 + h * 32 + g4 * 8;
        *reinterpret_cast<u32x4*>(C + (row * N + col) * 2) = v;
      }
  } else if (PATTERN == 1) {
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int64_t row = (int64_t)tm * 256 + it * 16 + (threadIdx.x >> 5), col = tn * 256 + (threadIdx.x & 31) * 8;
      *reinterpret_cast<u32x4*>(C + (row * N + col) * 2) = v;
    }
  } else if (PATTERN == 2) {
#pragma unroll
    for (int it = 0; it < 16; ++it)
      *reinterpret_cast<u32x4*>(C + (int64_t)tile * 131072 + (it * 512 + threadIdx.x) * 16) = v;
  } else {                                                   // 3: the register epilogue after a lane-pair exchange: 8 rows x 128 B (whole lines)
    const int wr = wave >> 2, wc = wave & 3, frow = lane & 15, g4 = lane >> 4;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int64_t row = (int64_t)tm * 256 + wr * 128 + i * 16 + h * 8 + (frow & 7), col = tn * 256 + wc * 64 + (frow >> 3) * 32 + g4 * 8;
        *reinterpret_cast<u32x4*>(C + (row * N + col) * 2) = v;
      }
  }
}
template <int PATTERN> float run(char* C, int N, int ntn, int wgs, int reps) {
  hipFuncSetAttribute((const void*)writer<PATTERN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(writer<PATTERN>, dim3(wgs), dim3(512), 160 * 1024, 0, C, N, ntn, 0);
  hipEventRecord(e0);
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(writer<PATTERN>, dim3(wgs), dim3(512), 160 * 1024, 0, C, N, ntn, 0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / reps;
}
// (round 3) is the ~6 TB/s store limit chip-wide or per XCD?  256 workgroups launched, only those of 1 / 2 / 4 / 8 XCDs store.
static float run_mask(char* C, int N, int ntn, int mask, int reps) {
  hipFuncSetAttribute((const void*)writer<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(writer<1>, dim3(256), dim3(512), 160 * 1024, 0, C, N, ntn, 0, mask);
  hipEventRecord(e0);
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(writer<1>, dim3(256), dim3(512), 160 * 1024, 0, C, N, ntn, 0, mask);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / reps;
}

int main() {
  const int N = 3072, ntn = N / 256, M = 49536;
  char* C;
  hipMalloc(&C, (size_t)M * N * 2);
  const char* names[4] = {"16 rows x 64 B per instruction (register epilogue)", "2 rows x 512 B per instruction", "contiguous 128-KB block",
                          "8 rows x 128 B per instruction (lane-pair exchange)"};
  for (int mask : {0x01, 0x03, 0x0f, 0xff}) {
    int n = 0;
    for (int b = 0; b < 8; ++b) n += (mask >> b) & 1;
    const float t = run_mask(C, N, ntn, mask, 20);
    printf("256 workgroups launched, XCD mask 0x%02x (%d workgroups store, 32 per XCD): %7.2f us per launch  %6.0f GB/s\n", mask, n * 32, t, n * 32 * 131072.0 / t / 1e3);
  }
  for (int wgs : {1, 32, 64, 128, 256, 512, 1024, 2328}) {
    const float t[4] = {run<0>(C, N, ntn, wgs, 20), run<1>(C, N, ntn, wgs, 20), run<2>(C, N, ntn, wgs, 20), run<3>(C, N, ntn, wgs, 20)};
    for (int p = 0; p < 4; ++p) {
      const double rounds = wgs <= 256 ? 1.0 : wgs / 256.0;
      printf("%5d workgroups  %-52s %7.2f us per launch  %6.2f us per round  %6.0f GB/s  %5.1f B/clk/CU (2.4 GHz, busy CUs)\n", wgs, names[p], t[p],
             t[p] / rounds, wgs * 131072.0 / t[p] / 1e3, 131072.0 * rounds / (t[p] * 2400.0));
    }
  }
  return 0;
}
