// 512-point complex FFT spread over one 64-lane wave (gfx950), shared by the fused front
// end (frontend.hip) and the stand-alone spectrum entry point (stages.hip).
//
// 8 points per lane, three radix-8 passes in registers, two transposes through a 576-entry
// float2 LDS scratch whose paddings (row stride 72, then 9) make every ds_read_b64 /
// ds_write_b64 of the transposes bank-conflict free.  Only this wave touches its scratch, and
// the LDS unit executes one wave's DS instructions in order, so the transposes need no
// s_barrier: wave_sync() merely stops the compiler from moving LDS accesses across it.
//
// A complex number is a 2-float vector so that the arithmetic maps onto the PACKED f32 VALU
// ops (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two floats per lane per issue slot).  A
// gfx950 SIMD issues one VALU instruction per 4 cycles whatever it does, and this kernel is
// bound by that issue rate, so halving the instruction count is what matters: a complex add
// is one instruction, a complex multiply two (the swap / negate of an operand rides on the
// op_sel / neg modifiers).
#pragma once
#include <hip/hip_runtime.h>

namespace svk_fft {

typedef float cplx __attribute__((ext_vector_type(2)));  // .x = re, .y = im

constexpr int SCR = 576;  // cplx entries of scratch the caller provides

// Order this wave's LDS accesses (compiler fence + wave-level barrier; no instruction cost).
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__host__ __device__ __forceinline__ cplx mk(float re, float im) { return (cplx){re, im}; }
// (a.x + i a.y) * (b.x + i b.y) = a.x * (b.x, b.y) + a.y * (-b.y, b.x)
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
  const cplx ax = __builtin_shufflevector(a, a, 0, 0), ay = __builtin_shufflevector(a, a, 1, 1);
  const cplx bs = mk(-b.y, b.x);
  return __builtin_elementwise_fma(ay, bs, ax * b);
}
// a + (-i) b  and  a - (-i) b:   (-i) b = (b.y, -b.x)
__device__ __forceinline__ cplx add_mi(cplx a, cplx b) { return a + mk(b.y, -b.x); }
__device__ __forceinline__ cplx sub_mi(cplx a, cplx b) { return a - mk(b.y, -b.x); }

// Forward 8-point DFT in place, natural order out (two radix-4 halves + one radix-2 layer).
__device__ __forceinline__ void dft8(cplx (&v)[8]) {
  const float S = 0.70710678118654752440f;
  const cplx a0 = v[0] + v[4], a1 = v[0] - v[4], a2 = v[2] + v[6], d26 = v[2] - v[6];
  const cplx a4 = v[1] + v[5], a5 = v[1] - v[5], a6 = v[3] + v[7], d37 = v[3] - v[7];
  const cplx b0 = a0 + a2, b2 = a0 - a2, b1 = add_mi(a1, d26), b3 = sub_mi(a1, d26);
  const cplx c4 = a4 + a6, d46 = a4 - a6, t5 = add_mi(a5, d37), t7 = sub_mi(a5, d37);
  // t5 * W8 = ((x + y), (y - x)) S ;  t7 * W8^3 = ((y - x), -(x + y)) S
  const cplx c5 = mk(t5.x + t5.y, t5.y - t5.x) * S;
  const cplx c7 = mk(t7.y - t7.x, -(t7.x + t7.y)) * S;
  v[0] = b0 + c4;
  v[4] = b0 - c4;
  v[1] = b1 + c5;
  v[5] = b1 - c5;
  v[2] = add_mi(b2, d46);
  v[6] = sub_mi(b2, d46);
  v[3] = b3 + c7;
  v[7] = b3 - c7;
}

// 512-point forward complex FFT across one wave.
// in : v[a] = z[64 a + lane]        out: v[q] = Z[lane + 64 q]
__device__ __forceinline__ void fft512_wave(cplx (&v)[8], cplx* scr, int lane, const cplx (&t1)[8],
                                            const cplx (&t2)[8]) {
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
__device__ __forceinline__ cplx shfl2(cplx v, int src) { return mk(__shfl(v.x, src, 64), __shfl(v.y, src, 64)); }

}  // namespace svk_fft
