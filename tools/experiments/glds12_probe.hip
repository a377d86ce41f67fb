// Probe: global_load_lds_dwordx3 (12-byte LDS-DMA) on gfx950 -- source alignment, destination above 64 KB, masked lanes.
//   hipcc --offload-arch=gfx950 -O2 tools/experiments/glds12_probe.hip -o build_variants/glds12_probe && ./build_variants/glds12_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const float* g, float* out, int src_off, int lds_off_floats, int rows) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63;
  for (int k = threadIdx.x; k < 64 * 3 + 32; k += blockDim.x) sm[lds_off_floats + k] = -1.f;
  __syncthreads();
  const float* src = g + (lane >> 1) * 40 + 3 * (lane & 1) + src_off;
  if ((lane >> 1) < rows) __builtin_amdgcn_global_load_lds(src, sm + lds_off_floats, 12, 0, 0);
  __syncthreads();
  for (int k = threadIdx.x; k < 64 * 3 + 32; k += blockDim.x) out[k] = sm[lds_off_floats + k];
}
int main() {
  const int N = 64 * 40 + 64;
  std::vector<float> h(N);
  for (int i = 0; i < N; ++i) h[i] = (float)i;
  float *g, *o;
  hipMalloc(&g, N * 4);
  hipMalloc(&o, 256 * 4);
  hipMemcpy(g, h.data(), N * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  for (int src_off : {0, 1, 2, 3, 4}) for (int lds_off : {0, 12000, 28800, 35000}) for (int rows : {32, 16}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 150 * 1024, 0, g, o, src_off, lds_off, rows);
    std::vector<float> r(224);
    hipMemcpy(r.data(), o, 224 * 4, hipMemcpyDeviceToHost);
    int bad = 0, first = -1;
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 3; ++e) {
      const float want = (l >> 1) < rows ? (float)((l >> 1) * 40 + 3 * (l & 1) + src_off + e) : -1.f;
      if (r[l * 3 + e] != want) { if (first < 0) first = l * 3 + e; ++bad; }
    }
    for (int k = 192; k < 224; ++k) if (r[k] != -1.f) ++bad;
    printf("src_off %d floats, lds_off %6d floats (%6d B), rows %2d: %d wrong", src_off, lds_off, lds_off * 4, rows, bad);
    if (bad) printf("  first at %d: got %g %g %g %g", first, r[first], r[first + 1], r[first + 2], r[first + 3]);
    printf("\n");
  }
  return 0;
}
