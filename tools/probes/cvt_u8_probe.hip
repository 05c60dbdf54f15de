// How does v_cvt_pk_u8_f32 round?  (The DEMO_D8 encoder rounds explicitly with v_rndne_f32 in front of it: is that needed?)
// hipcc --offload-arch=gfx950 -O3 tools/probes/cvt_u8_probe.hip -o tools/probes/cvt_u8_probe && tools/probes/cvt_u8_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* x, unsigned* raw, unsigned* rne, int n) {
  const int i = threadIdx.x;
  if (i < n) {
    raw[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 0, 0u);
    rne[i] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(x[i]), 0, 0u);
  }
}
int main() {
  const float h[] = {0.0f, 0.25f, 0.5f, 0.75f, 1.0f, 1.5f, 2.5f, 2.500001f, 2.7f, 3.5f, 26.0f, 127.5f, 128.5f, 254.5f, 254.7f, 255.5f, 300.0f, -0.3f, -0.7f, -3.0f, 0.49999997f};
  const int n = sizeof(h) / sizeof(h[0]);
  float* d; unsigned *a, *b;
  hipMalloc(&d, sizeof(h)); hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, a, b, n);
  unsigned ha[64], hb[64];
  hipMemcpy(ha, a, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hb, b, n * 4, hipMemcpyDeviceToHost);
  int same = 1;
  for (int i = 0; i < n; ++i) {
    printf("%12.8f  v_cvt_pk_u8_f32 -> %3u   rint first -> %3u%s\n", h[i], ha[i] & 255u, hb[i] & 255u, ha[i] == hb[i] ? "" : "   DIFFERENT");
    same &= ha[i] == hb[i];
  }
  printf(same ? "the conversion rounds to nearest even by itself\n" : "the conversion does NOT round like rint: keep v_rndne_f32 in front of it\n");
  return 0;
}
