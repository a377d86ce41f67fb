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

// Packed-f32 helpers written as single instructions: hipcc does not fold the swap / negate of a
// complex operand into the op_sel / neg modifiers of v_pk_*_f32 (it spent 55 of the FFT's 180
// VALU instructions on v_mov / v_xor for them), so the modifier forms are spelled out.
// op_sel picks the dword of each 64-bit source that feeds the LOW result, op_sel_hi the HIGH.
#define SVK_PK2(name, text)                                            \
  __device__ __forceinline__ cplx name(cplx a, cplx b) {               \
    cplx r;                                                            \
    asm(text : "=v"(r) : "v"(a), "v"(b));                              \
    return r;                                                          \
  }
// a + (-i) b = (a.x + b.y, a.y - b.x)
SVK_PK2(add_mi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")
// a - (-i) b = (a.x - b.y, a.y + b.x)
SVK_PK2(sub_mi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")
// a + conj(b) = (a.x + b.x, a.y - b.y)
SVK_PK2(add_conj, "v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]")
// (a.y + b.y, a.x - b.x)
SVK_PK2(swap_add_conj, "v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[0,1]")
#undef SVK_PK2
// t (1 - i) = (t.x + t.y, t.y - t.x): W8 up to the factor 1/sqrt 2
__device__ __forceinline__ cplx rot_w8(cplx t) { return add_mi(t, t); }
// t (-1 - i) = (t.y - t.x, -t.x - t.y): W8^3 up to the factor 1/sqrt 2
__device__ __forceinline__ cplx rot_w8_3(cplx t) {
  cplx r;
  asm("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[1,1]" : "=v"(r) : "v"(t));
  return r;
}
// (a.x + i a.y)(b.x + i b.y): two instructions, no temporaries for the swapped / negated b
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
  cplx t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));  // (a.x b.x, a.x b.y)
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
      : "=v"(r) : "v"(a), "v"(b), "v"(t));                                                  // (-a.y b.y, a.y b.x) + t
  return r;
}

// a * conj(b) = (a.x b.x + a.y b.y, a.y b.x - a.x b.y): the same two instructions with other modifiers
__device__ __forceinline__ cplx cmul_conj(cplx a, cplx b) {
  cplx t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));  // (a.x b.x, -a.x b.y)
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(t));  // + (a.y b.y, a.y b.x)
  return r;
}

// Forward 8-point DFT in place, natural order out: 26 packed instructions.  NZ < 8 says inputs
// v[NZ..7] are exact zeros (the zero padding of a frame up to the FFT length, which the first
// radix-8 pass of the wave FFT sees as whole registers): their butterflies of the first stage
// drop out (x + 0 is not something the compiler may fold: -0 + 0 = +0).
template <int NZ = 8>
__device__ __forceinline__ void dft8(cplx (&v)[8]) {
  const cplx S = mk(0.70710678118654752440f, 0.70710678118654752440f);
  const cplx a0 = NZ > 4 ? v[0] + v[4] : v[0], a1 = NZ > 4 ? v[0] - v[4] : v[0];
  const cplx a2 = NZ > 6 ? v[2] + v[6] : v[2], d26 = NZ > 6 ? v[2] - v[6] : v[2];
  const cplx a4 = NZ > 5 ? v[1] + v[5] : v[1], a5 = NZ > 5 ? v[1] - v[5] : v[1];
  const cplx a6 = NZ > 7 ? v[3] + v[7] : v[3], d37 = NZ > 7 ? v[3] - v[7] : v[3];
  const cplx b0 = a0 + a2, b2 = a0 - a2, b1 = add_mi(a1, d26), b3 = sub_mi(a1, d26);
  const cplx c4 = a4 + a6, d46 = a4 - a6, t5 = add_mi(a5, d37), t7 = sub_mi(a5, d37);
  const cplx c5 = rot_w8(t5), c7 = rot_w8_3(t7);  // still to be scaled by 1/sqrt 2: folded into the fma below
  v[0] = b0 + c4;
  v[4] = b0 - c4;
  v[1] = __builtin_elementwise_fma(c5, S, b1);
  v[5] = __builtin_elementwise_fma(c5, -S, b1);
  v[2] = add_mi(b2, d46);
  v[6] = sub_mi(b2, d46);
  v[3] = __builtin_elementwise_fma(c7, S, b3);
  v[7] = __builtin_elementwise_fma(c7, -S, b3);
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

// Two independent transforms through ONE scratch, software-pipelined: while the LDS round trip
// of one transform's transpose is in flight the other transform's radix-8 pass issues.  With 3
// waves per SIMD the other waves alone do not cover those round trips.  Sharing the scratch is
// safe because the LDS unit executes one wave's DS instructions in program order: b's transpose
// writes are issued after a's transpose reads and therefore land after them.
// NZ: inputs a[NZ..7], b[NZ..7] are exact zeros (see dft8).
template <int NZ = 8>
__device__ __forceinline__ void fft512_wave_x2(cplx (&a)[8], cplx (&b)[8], cplx* scr, int lane, const cplx (&t1)[8],
                                               const cplx (&t2)[8]) {
  const int r2 = lane >> 3, p = lane & 7;
  dft8<NZ>(a);
#pragma unroll
  for (int r = 1; r < 8; ++r) a[r] = cmul(a[r], t1[r]);
  wave_sync();
#pragma unroll
  for (int r = 0; r < 8; ++r) scr[r * 72 + lane] = a[r];
  wave_sync();
#pragma unroll
  for (int c = 0; c < 8; ++c) a[c] = scr[r2 * 72 + 8 * c + p];
  dft8<NZ>(b);
#pragma unroll
  for (int r = 1; r < 8; ++r) b[r] = cmul(b[r], t1[r]);
  wave_sync();
#pragma unroll
  for (int r = 0; r < 8; ++r) scr[r * 72 + lane] = b[r];
  wave_sync();
#pragma unroll
  for (int c = 0; c < 8; ++c) b[c] = scr[r2 * 72 + 8 * c + p];
  dft8(a);
#pragma unroll
  for (int r = 1; r < 8; ++r) a[r] = cmul(a[r], t2[r]);
  wave_sync();
#pragma unroll
  for (int r1 = 0; r1 < 8; ++r1) scr[(8 * r1 + r2) * 9 + p] = a[r1];
  wave_sync();
#pragma unroll
  for (int q = 0; q < 8; ++q) a[q] = scr[lane * 9 + q];
  dft8(b);
#pragma unroll
  for (int r = 1; r < 8; ++r) b[r] = cmul(b[r], t2[r]);
  wave_sync();
#pragma unroll
  for (int r1 = 0; r1 < 8; ++r1) scr[(8 * r1 + r2) * 9 + p] = b[r1];
  wave_sync();
#pragma unroll
  for (int q = 0; q < 8; ++q) b[q] = scr[lane * 9 + q];
  dft8(a);
  dft8(b);
}

// Partner of bin k = lane + 64 j is bin 512 - k: register 7 - j of lane 64 - lane; lane 0
// pairs with itself one register later (bin 64 j <-> bin 64 (8 - j)).
__device__ __forceinline__ cplx shfl2(cplx v, int src) { return mk(__shfl(v.x, src, 64), __shfl(v.y, src, 64)); }

}  // namespace svk_fft
