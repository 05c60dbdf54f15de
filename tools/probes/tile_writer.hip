// Probe: how fast can 256x256 bf16 output tiles (512-B row pieces at stride N*2) be written, one workgroup per CU,
// all CUs at once -- the store pattern of the 256^2 GEMM epilogue without any of its arithmetic.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/tile_writer.hip -o /tmp/tile_writer && /tmp/tile_writer
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(512, 1) void writer(uint4* __restrict__ C, int M, int N, int ntm, int ntn, int lds_pad) {
  extern __shared__ char sm[];
  if (lds_pad == 12345) sm[threadIdx.x] = 1;              // keep the 160 KB allocation (1 workgroup per CU)
  const int tid = blockIdx.x;
  const int tm = tid / ntn, tn = tid % ntn;
  const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;  // 32 threads x 16 B = 512 B per row, 16 rows per pass
  const uint4 v = make_uint4(tid, c, r0, 7);
  for (int it = 0; it < 16; ++it) {
    const int m = tm * 256 + it * 16 + r0;
    if (m < M) C[((int64_t)m * N + tn * 256) / 8 + c] = v;
  }
}
int main() {
  const int M = 49536;
  for (int N : {768, 2304, 3072}) {
    const int ntm = (M + 255) / 256, ntn = N / 256;
    uint4* C;
    hipMalloc(&C, (size_t)M * N * 2);
    hipFuncSetAttribute((const void*)writer, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int lds : {0, 160 * 1024}) {
      for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(writer, dim3(ntm * ntn), dim3(512), lds, 0, C, M, N, ntm, ntn, 0);
      hipEventRecord(e0);
      for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(writer, dim3(ntm * ntn), dim3(512), lds, 0, C, M, N, ntm, ntn, 0);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("N=%d lds=%6d: %7.1f us per %5.1f MB  -> %6.0f GB/s\n", N, lds, ms * 100, (double)M * N * 2 / 1e6, (double)M * N * 2 / (ms / 10 * 1e-3) / 1e9);
    }
    hipFree(C);
  }
  return 0;
}
