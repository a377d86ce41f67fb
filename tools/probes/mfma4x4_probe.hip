// Probe: lane / register layout and issue cost of v_mfma_f32_4x4x1_16b_f32 on gfx950.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma4x4_probe.hip -o gpurun_out/mfma4x4_probe && gpurun_out/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout(float* out) {
  const int l = threadIdx.x;
  // A value encodes (block, i) = lane / 4, lane % 4 ; B value encodes (block, j)
  const float a = 1.0f + (l & 3);             // A[i] = 1 + i
  const float b = 10.0f * (1 + (l & 3)) + 100.0f * (l >> 2);   // B[j] = 10 (1 + j) + 100 block
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}

template <int KIND>
__global__ void timing(float* out, long long* cyc, int iters) {
  const int l = threadIdx.x;
  float a = 1.0f + l * 1e-3f, b = 2.0f - l * 1e-3f;
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
    } else if (KIND == 1) {
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);   // one dependent chain
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
    } else {
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  f32x4 s = c0 + c1 + c2 + c3;
  out[l] = s[0] + s[1] + s[2] + s[3];
  if (l == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  float* d; long long* dc;
  hipMalloc(&d, 64 * 4 * sizeof(float)); hipMalloc(&dc, 8 * sizeof(long long));
  layout<<<1, 64>>>(d);
  std::vector<float> h(256);
  hipMemcpy(h.data(), d, 256 * sizeof(float), hipMemcpyDeviceToHost);
  // expectation: out[lane][r] = A[i = r] * B[j = lane % 4] of block lane / 4 = (1 + r) * (10 (1 + lane%4) + 100 (lane/4))
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    float want = (1.0f + r) * (10.0f * (1 + (l & 3)) + 100.0f * (l >> 2));
    if (h[l * 4 + r] != want) ++bad;
  }
  printf("layout D[i=reg][j=lane%%4] per block lane/4: %s (%d mismatches); lane5 = %g %g %g %g\n", bad ? "NO" : "yes", bad,
         h[20], h[21], h[22], h[23]);
  const int iters = 4096;
  for (int kind = 0; kind < 3; ++kind) {
    if (kind == 0) timing<0><<<1, 64>>>(d, dc, iters);
    if (kind == 1) timing<1><<<1, 64>>>(d, dc, iters);
    if (kind == 2) timing<2><<<1, 64>>>(d, dc, iters);
    long long c; hipMemcpy(&c, dc, sizeof(c), hipMemcpyDeviceToHost);
    printf("kind %d (%s): %.2f cycles per MFMA (one wave)\n", kind,
           kind == 0 ? "4x4x1 independent" : kind == 1 ? "4x4x1 dependent" : "16x16x4 independent", (double)c / (iters * 4.0));
  }
  return 0;
}
