// How fast ONE wave per SIMD issues v_mfma_f32_16x16x32_f16: a dependent chain on one accumulator, three independent accumulators,
// and the same with two ds_read_b128 per three MFMAs feeding the B operand (the inner loop of the f16 kernels of csrc/c3d2.hip).
//   hipcc --offload-arch=gfx950 -O3 mfma_issue_probe.hip -o mfma_issue_probe && ./mfma_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(float* out, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned lds[16384];
  for (int k = threadIdx.x; k < 16384; k += blockDim.x) lds[k] = 0x3c003c00u;   // halves 1.0
  __syncthreads();
  const int lane = threadIdx.x & 63;
  u32x4 a = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, b = a, b2 = a;
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0;
  const u32x4* src = reinterpret_cast<const u32x4*>(lds) + lane;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < (MODE >= 4 ? 0 : iters); ++it) {
#pragma unroll
    for (int s = 0; s < 14; ++s) {
      if (MODE >= 2) {
        b = src[64 * ((2 * s) & 31)];
        b2 = src[64 * ((2 * s + 1) & 31)];
      }
      __builtin_amdgcn_sched_barrier(0);
      if (MODE == 0 || MODE == 2) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b2), c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, b2), __builtin_bit_cast(f16x8, b), c0, 0, 0, 0);
      } else {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b2), c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, b2), __builtin_bit_cast(f16x8, b), c2, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (MODE == 4 || MODE == 5) {   // fragments five steps ahead of the MFMAs that use them (six rotating sets); 5: one read per step only
    u32x4 fh[6], fl[6];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      fh[q] = src[64 * q];
      fl[q] = src[64 * (q + 8)];
    }
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 12; ++s) {
        fh[(s + 5) % 6] = src[64 * ((s + 5) & 31)];
        if (MODE == 4) fl[(s + 5) % 6] = src[64 * ((s + 13) & 31)];
        __builtin_amdgcn_sched_barrier(0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, fh[s % 6]), c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, fl[s % 6]), c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, b2), __builtin_bit_cast(f16x8, fh[s % 6]), c2, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  f32x4 c = c0 + c1 + c2;
  out[blockIdx.x * blockDim.x + threadIdx.x] = c[0] + c[1] + c[2] + c[3];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int waves_per_simd) {
  float* out;
  long long* cyc;
  const int grid = 256, block = 256 * waves_per_simd, iters = 2000;
  hipMalloc(&out, sizeof(float) * grid * block);
  hipMalloc(&cyc, sizeof(long long) * grid);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<MODE><<<grid, block>>>(out, cyc, 10);
  hipEventRecord(e0);
  probe<MODE><<<grid, block>>>(out, cyc, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  long long h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double mf = (MODE >= 4 ? 36.0 : 42.0) * iters * waves_per_simd;
  printf("%-44s %d wave(s)/SIMD: %.3f ms, s_memtime %.1f per MFMA and SIMD, wall %.1f ns per MFMA and SIMD\n", name, waves_per_simd, ms, h[0] / mf, ms * 1e6 / mf);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run<0>("one accumulator, operands in registers", w);
    run<1>("three accumulators, operands in registers", w);
    run<2>("one accumulator, 2 ds_read_b128 per 3 MFMAs", w);
    run<3>("three accumulators, 2 ds_read_b128 per 3", w);
    run<4>("three acc., 2 reads per 3, FIVE steps ahead", w);
    run<5>("three acc., 1 read per 3, five steps ahead", w);
  }
  return 0;
}
