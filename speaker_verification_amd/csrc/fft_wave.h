// 512-point complex FFT spread over one 64-lane wave (gfx950), shared by the fused front
// end (frontend.hip) and the stand-alone spectrum entry point (stages.hip).
//
// 8 points per lane, three radix-8 passes in registers, two transposes through a 576-entry
// float2 LDS scratch whose paddings (row stride 72, then 9) make every ds_read_b64 /
// ds_write_b64 of the transposes bank-conflict free.  Only this wave touches its scratch, and
// the LDS unit executes one wave's DS instructions in order, so the transposes need no
// s_barrier: wave_sync() merely stops the compiler from moving LDS accesses across it.
#pragma once
#include <hip/hip_runtime.h>

namespace svk_fft {

constexpr int SCR = 576;  // float2 entries of scratch the caller provides

// Order this wave's LDS accesses (compiler fence + wave-level barrier; no instruction cost).
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float2 operator+(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 operator-(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }  // * (-i)

// Forward 8-point DFT in place, natural order out.
__device__ __forceinline__ void dft8(float2 (&v)[8]) {
  const float S = 0.70710678118654752440f;
  float2 a0 = v[0] + v[4], a1 = v[0] - v[4], a2 = v[2] + v[6], a3 = mul_mi(v[2] - v[6]);
  float2 a4 = v[1] + v[5], a5 = v[1] - v[5], a6 = v[3] + v[7], a7 = mul_mi(v[3] - v[7]);
  float2 b0 = a0 + a2, b2 = a0 - a2, b1 = a1 + a3, b3 = a1 - a3;
  float2 c4 = a4 + a6, c6 = mul_mi(a4 - a6), t5 = a5 + a7, t7 = a5 - a7;
  float2 c5 = make_float2((t5.x + t5.y) * S, (t5.y - t5.x) * S);   // * W8
  float2 c7 = make_float2((t7.y - t7.x) * S, -(t7.x + t7.y) * S);  // * W8^3
  v[0] = b0 + c4; v[4] = b0 - c4;
  v[1] = b1 + c5; v[5] = b1 - c5;
  v[2] = b2 + c6; v[6] = b2 - c6;
  v[3] = b3 + c7; v[7] = b3 - c7;
}

// 512-point forward complex FFT across one wave.
// in : v[a] = z[64 a + lane]        out: v[q] = Z[lane + 64 q]
__device__ __forceinline__ void fft512_wave(float2 (&v)[8], float2* scr, int lane, const float2 (&t1)[8],
                                            const float2 (&t2)[8]) {
  dft8(v);
#pragma unroll
  for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], t1[r]);
  wave_sync();  // scratch free (previous readers done)
#pragma unroll
  for (int r = 0; r < 8; ++r) scr[r * 72 + lane] = v[r];
  wave_sync();
  const int r2 = lane >> 3, p = lane & 7;
#pragma unroll
  for (int b = 0; b < 8; ++b) v[b] = scr[r2 * 72 + 8 * b + p];
  dft8(v);
#pragma unroll
  for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], t2[r]);
  wave_sync();
#pragma unroll
  for (int r1 = 0; r1 < 8; ++r1) scr[(8 * r1 + r2) * 9 + p] = v[r1];
  wave_sync();
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = scr[lane * 9 + q];
  dft8(v);
}

// Partner of bin k = lane + 64 j is bin 512 - k: register 7 - j of lane 64 - lane; lane 0
// pairs with itself one register later (bin 64 j <-> bin 64 (8 - j)).
__device__ __forceinline__ float2 shfl2(float2 v, int src) {
  return make_float2(__shfl(v.x, src, 64), __shfl(v.y, src, 64));
}

}  // namespace svk_fft
