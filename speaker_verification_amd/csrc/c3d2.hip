// The C3D2 embedding network's first three blocks (model.py:110-131, :141-164).
//   c3d2_stage1h_kernel   cube + conv1_1 + conv1_2 + pool1 on v_mfma_f32_16x16x32_f16 through two-piece f16 products (round 4)
//   c3d2_conv21w_kernel   conv2_1, v_mfma_f32_16x16x4_f32, depth-transformed (Winograd F(2, 3) along depth)
//   c3d2_conv22w_kernel   conv2_2 + pool2, depth-transformed
//   c3d2_conv31w_kernel   conv3_1, depth-transformed; writes the chunked, column-major layout c3d2_tail_kernel<Conv32T> stages
// (conv3_2, conv4_1, conv4_2 and FC5 live in c3d2_tail.hip.)  BatchNorm (eval mode) is folded into weights and biases by
// the host (model.FusedEmbedder).  Work items come from device-wide counters.  What earlier rounds built and superseded is
// under tools/experiments/ with its measured numbers: the direct-form f32 kernels, the t-plane first block, the K-split conv3_2
// (c3d2_superseded_r3.patch) and the f32 first block through the depth transform, round 4's 7.12 ms kernel
// (stage1_f32_winograd.patch).
//
// The first block as ONE gfx950 kernel:
//   feature rows + crop starts -> cube (utils.py:351-379) -> conv1_1 (1 -> 16, k(3,1,5)) + BN + PReLU
//   -> conv1_2 (16 -> 16, k(3,9,1), stride (1,2,1)) + BN + PReLU -> MaxPool3d((1,1,2))
// (/root/reference/model.py:110-117 and :141-150).  These two layers are 46 % of the network's multiply-adds, and
// conv1_1's output is the network's largest tensor (3.3 MB per cube); here it only ever exists as a 100 KB tile in LDS.
//
// Work item = (cube u, pooled output column j, half q of the output depths): conv1_2 outputs
//   d in [8q, 8q + 8), h in [0, 36), w in {2j, 2j + 1}  ->  pooled column j, 16 channels.
// A persistent workgroup of 8 waves (two per SIMD; it owns the CU's LDS) loops over items:
//   1. the 12 x 80 x 6 cube patch the item needs is moved into LDS by LDS-DMA inside the previous item's matrix work and
//      converted in place to (h, l) half pairs by the waves that fetched it;
//   2. conv1_1 as a GEMM [16 channels] x [K = 32: 15 taps + pad, h | l] x [16 pixels], + PReLU, split into (h, l), written to
//      the act1 tile in LDS: 10 depths x 80 rows x 2 columns x 16 channels;
//   3. conv1_2 as an implicit GEMM in the direct form, two taps per K = 32 block, the weights of all 27 taps in 112 VGPRs;
//   4. bias (in the accumulator), PReLU, max over the column pair (adjacent lanes: one DPP instruction), 16-byte stores.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "svk_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int NCROP = 20, NFRAME = 80, NCOEF = 40;  // cube geometry (utils.py:20-21)
constexpr int TD = 8;                                // conv1_2 output depths per item
constexpr int DIN = TD + 2;                          // act1 depths per item
constexpr int PD = TD + 4;                           // cube patch depths per item
constexpr int OD = 16, OH = 36, OWP = 18;            // output: depths, rows, pooled columns
// output strides (floats) of [n][16 d][36 h][18 w][16 c] (channels-last memory of a (n, 16, 16, 36, 18) tensor): pooled column,
// row parity, row pair, depth, cube.  Compile-time: as kernel parameters they were 64-bit scalar multiplies per item
constexpr int S_W = 16, S_PAR = OWP * 16, S_HP = 2 * OWP * 16, S_D = OH * OWP * 16, S_N = OD * OH * OWP * 16;
// A work item (cube u, rem = 18 q + j: half q of the output depths, pooled column j), decoded ONCE when its index is known (two
// items ahead) and handed down: item / 36, % 36, / 18 for the item, the next one (patch fetch) and the one after (crop starts)
// were three division chains of scalar instructions per item, in front of the barrier where nothing hides them.
struct ItemPos {
  int u, rem;
  __device__ __forceinline__ int q() const { return rem >= 18 ? 1 : 0; }
  __device__ __forceinline__ int j() const { return rem >= 18 ? rem - 18 : rem; }
  __device__ static __forceinline__ ItemPos of(int item) {
    const int u = item / 36;
    return ItemPos{u, item - 36 * u};
  }
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
struct Stage1Params {
  const float* feat;
  const int32_t* crop;
  int32_t n_utt, max_frames;
  const u32x4* w1blk;    // [2][64]: conv1_1's A blocks (8 halves per lane): [H taps 0-15 | H taps 0-15], [L taps 0-15 | 0]
  const float* bias1;
  const float* slope1;
  const u32x4* w2blk;    // [14 pairs][2][64]: [H_a | H_b], [L_a | L_b]; lane (co = l & 15, kk): ci = 8 (kk & 1) + e, tap a (kk < 2) / b
  const float* bias2;
  const float* slope2;
  float* out;
  unsigned* queue;
};

// In-kernel phase stamps (s_memtime) for `make TUNING=1` builds; compiled out of the shipped library.
#ifdef SVK_TUNING
#define SVK_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define SVK_STAMP_ADD(slot, a, b) do { stamp_acc[slot] += (b) - (a); } while (0)   /* wave-uniform: scalar registers */
#else
#define SVK_STAMP(var) do { } while (0)
#define SVK_STAMP_ADD(slot, a, b) do { } while (0)
#endif

__device__ __forceinline__ float prelu(float v, float slope) { return v > 0.f ? v : slope * v; }
// 0 <= slope <= 1 (nn.PReLU starts at 0.25 and trained slopes stay there): prelu(v) = max(v, slope v), two
// instructions instead of compare / multiply / select (+ a wait state); bit-identical for finite v.
template <bool SLOPE01>
__device__ __forceinline__ float prelu_t(float v, float slope) {
  return SLOPE01 ? fmaxf(v, slope * v) : prelu(v, slope);
}

// a per-thread constant plus an immediate.
// (a VECTOR load by lanes 0 .. 11, not twelve scalar loads: scalar loads return out of order, so while any is in
// flight every LDS wait of the wave becomes lgkmcnt(0) -- the first gather read of the conv1_1 phase then stalled for
// the crop table's whole L2 round trip, 2 500 cycles per item by the in-kernel stamps)
__device__ __forceinline__ int fetch_starts(const Stage1Params& p, ItemPos it, int lane) {
  const int32_t* cr = p.crop + (int64_t)it.u * NCROP + TD * it.q();
  return cr[lane < PD ? lane : 0];
}

constexpr int WPW = 8;                                     // floats per patch row in LDS: [ww 0 1 2 | - | ww 3 4 5 | -]
constexpr int WP_FLOATS = PD * NFRAME * WPW;               // 7 680

// The item's cube patch by LDS-DMA (global_load_lds_dwordx3; round 3): patch[dd][h][.] = feat[u][crop[u][8 q + dd] + h][2 j ..
// 2 j + 5].  A DMA lane's 12 bytes land at a wave-uniform LDS base + 16 lane (measured: tools/experiments/glds12_probe.hip
// -- the fourth word of every 16 bytes is left alone), so two lanes carry a row's two halves, the row is 8 floats in LDS
// and one instruction moves 32 rows: no staging registers, no parking writes, no per-lane address arithmetic (the register
// path before it: twelve 8-byte loads + six ds_write_b64 per thread).  The source needs 4-byte alignment only.  Depth
// dd's 80 rows are three pieces (32 + 32 + 16 rows, the last with half the lanes).
// WHO fetches matters more than how: the workgroup's four OLDER waves (part 0) finish their tiles ~5 k cycles before the
// younger four and wait at the item's last barrier, so they carry the whole fetch -- nine pieces each: depths pair, pair + 4,
// pair + 8, the piece's part a compile-time constant -- and the younger waves, whose tiles end the item, none: the
// scalar address work and the issue of a fetch spread over all eight waves cost 2.5 % of the kernel (7.70 -> 7.51 ms; a
// build with no fetch at all: 7.33).  A piece that is not wholly inside the clip (a wild crop start: the C-ABI takes any
// int32) goes the slow way, lane by lane, with zeros outside -- a wave-uniform branch the pipeline's own crops never take.
struct PatchPiece { const float* src; float* dst; int rows; bool inside; int start, h0; };
__device__ __forceinline__ void patch_piece_issue(const Stage1Params& p, const PatchPiece& pc, int lane) {
  const int rl = lane >> 1, half = lane & 1;
  if (pc.inside) {
    if (rl < pc.rows) __builtin_amdgcn_global_load_lds(pc.src + rl * NCOEF + 3 * half, pc.dst, 12, 0, 0);
  } else if (rl < pc.rows) {
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    if ((unsigned)pc.start < (unsigned)p.max_frames && pc.h0 + rl < p.max_frames - pc.start) {
      const float* src = pc.src + rl * NCOEF + 3 * half;
      v0 = src[0];
      v1 = src[1];
      v2 = src[2];
    }
    float* d = pc.dst + rl * WPW + 4 * half;
    d[0] = v0;
    d[1] = v1;
    d[2] = v2;
  }
}
// the nine pieces of wave `pair` (a part-0 wave).  Every crop start is read BEFORE the first DMA: with one in flight the
// compiler drains vmcnt in front of any use of an ordinary load's result -- `starts_v` is one -- which would serialise them.
__device__ __forceinline__ void dma_patch_w(const Stage1Params& p, ItemPos it, int starts_v, int pair, int lane, float* patch) {
  const float* base = p.feat + (int64_t)it.u * ((int64_t)p.max_frames * NCOEF) + 2 * it.j();
  int st[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) st[g] = __builtin_amdgcn_readlane(starts_v, pair + 4 * g);
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int part = 0; part < 3; ++part) {
      PatchPiece pc;
      pc.rows = part < 2 ? 32 : 16;
      pc.h0 = 32 * part;
      pc.start = st[g];
      pc.dst = patch + ((pair + 4 * g) * NFRAME + pc.h0) * WPW;
      pc.inside = (unsigned)st[g] < (unsigned)p.max_frames && pc.h0 + pc.rows <= p.max_frames - st[g];   // (cannot overflow for any int32 start)
      pc.src = base + (int64_t)(st[g] + pc.h0) * NCOEF;
      patch_piece_issue(p, pc, lane);
    }
}

// Input transform of the depth-Winograd form for one element pair (hf = 0: elements 0, 1; 1: elements 2, 3) of the four
// depth fragments x: t0 = x0 - x2, t1 = x1 + x2, t2 = x2 - x1, t3 = x1 - x3, as four v_pk_add_f32 (written out: the
// compiler emits the packed form now and then for sums and never for differences, which use the negate modifiers).
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// prelu for 0 <= slope <= 1 straight off MFMA accumulators: fmaxf() on a value the compiler cannot prove canonical costs a
// third instruction (v_max x, x in front of the real one) and the product is one v_mul per value; written as vectors it is one
// v_pk_mul_f32 per PAIR + one v_max_f32 per value (12 -> 6 instructions per conv1_1 tile; the same product, the same
// maximum: bit-identical for every finite and infinite input, NaN stays NaN).
// (the product is left to the compiler -- it selects v_pk_mul_f32 for a two-float vector product and, unlike for an asm
// statement, counts the wait states between an MFMA and the first instruction that reads its result; the v_max behind it
// depends on that product, so it is issued later still)
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) { return a * b; }
__device__ __forceinline__ float max_raw(float a, float b) {
  float d;
  asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
template <bool SLOPE01>
__device__ __forceinline__ f32x4 prelu4(f32x4 v, f32x4 slope) {
  f32x4 o;
  if (SLOPE01) {
    const f32x2 m0 = pk_mul(__builtin_shufflevector(v, v, 0, 1), __builtin_shufflevector(slope, slope, 0, 1));
    const f32x2 m1 = pk_mul(__builtin_shufflevector(v, v, 2, 3), __builtin_shufflevector(slope, slope, 2, 3));
    o[0] = max_raw(v[0], m0[0]);
    o[1] = max_raw(v[1], m0[1]);
    o[2] = max_raw(v[2], m1[0]);
    o[3] = max_raw(v[3], m1[1]);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = prelu(v[r], slope[r]);
  }
  return o;
}
// The packed product on values a VALU instruction produced (never straight off an MFMA: see pk_mul): as an asm statement it is
// ONE v_pk_mul_f32 whatever pair of registers the allocator picked (the compiler's own choice falls back to two v_mul_f32 now and
// then).
__device__ __forceinline__ f32x2 pk_mul_valu(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f32x2 lo2(f32x4 v) { return __builtin_shufflevector(v, v, 0, 1); }
__device__ __forceinline__ f32x2 hi2(f32x4 v) { return __builtin_shufflevector(v, v, 2, 3); }
// The output transform y(2 P) = (a0 + a1) + a2, y(2 P + 1) = (a1 - a2) - a3 of four accumulators as eight packed adds in TWO
// rounds of four independent ones: the compiler puts a wait state in front of an asm statement that reads the register the
// instruction before it wrote, so a dependent chain written link by link costs an s_nop per link.
__device__ __forceinline__ void wino_output(const f32x4 (&a)[4], f32x2 (&y0)[2], f32x2 (&y1)[2]) {
  const f32x2 u0 = pk_add(lo2(a[0]), lo2(a[1])), u1 = pk_add(hi2(a[0]), hi2(a[1]));
  const f32x2 w0 = pk_sub(lo2(a[1]), lo2(a[2])), w1 = pk_sub(hi2(a[1]), hi2(a[2]));
  y0[0] = pk_add(u0, lo2(a[2]));
  y0[1] = pk_add(u1, hi2(a[2]));
  y1[0] = pk_sub(w0, lo2(a[3]));
  y1[1] = pk_sub(w1, hi2(a[3]));
}
// PReLU of the two outputs of wino_output() with a lane's four slopes: four packed products, then eight v_max_f32 (12
// instructions for 8 values; fmaxf(v, slope * v) compiled to 24: a product and a canonicalising v_max x, x per value on top)
template <bool SLOPE01>
__device__ __forceinline__ void prelu_pairs(f32x2 (&y0)[2], f32x2 (&y1)[2], f32x4 slope, f32x4& o0, f32x4& o1) {
  if (SLOPE01) {
    const f32x2 m00 = pk_mul_valu(y0[0], lo2(slope)), m01 = pk_mul_valu(y0[1], hi2(slope));
    const f32x2 m10 = pk_mul_valu(y1[0], lo2(slope)), m11 = pk_mul_valu(y1[1], hi2(slope));
    o0[0] = max_raw(y0[0][0], m00[0]);
    o0[1] = max_raw(y0[0][1], m00[1]);
    o0[2] = max_raw(y0[1][0], m01[0]);
    o0[3] = max_raw(y0[1][1], m01[1]);
    o1[0] = max_raw(y1[0][0], m10[0]);
    o1[1] = max_raw(y1[0][1], m10[1]);
    o1[2] = max_raw(y1[1][0], m11[0]);
    o1[3] = max_raw(y1[1][1], m11[1]);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o0[r] = prelu(y0[r >> 1][r & 1], slope[r]);
      o1[r] = prelu(y1[r >> 1][r & 1], slope[r]);
    }
  }
}
__device__ __forceinline__ void wino_input_pair(const f32x4 (&x)[4], int hf, f32x2 (&t)[4][2]) {
  f32x2 xh[4];
#pragma unroll
  for (int dd = 0; dd < 4; ++dd) xh[dd] = hf ? __builtin_shufflevector(x[dd], x[dd], 2, 3) : __builtin_shufflevector(x[dd], x[dd], 0, 1);
  t[0][hf] = pk_sub(xh[0], xh[2]);
  t[1][hf] = pk_add(xh[1], xh[2]);
  t[2][hf] = pk_sub(xh[2], xh[1]);
  t[3][hf] = pk_sub(xh[1], xh[3]);
}


// =====================================================================================================
// The first block on the F16 matrix pipe through TWO-PIECE products (round 4, second half).  An f32 value x is carried as
// the pair (h, l) of halves with h = f16(x), l = f16(x - h): 22 significant bits in the same four bytes, and
//     x w  =  h_x h_w + l_x h_w + h_x l_w      (+ l_x l_w, 2^-22 of the product: dropped)
// is three f16 products, exact in the f32 the MFMA accumulates in.  Measured on this network's layers (CPU emulation with
// the trained checkpoint): 0.9 - 4.6e-7 of the activation scale from the f64 convolution -- the f32 direct form itself is
// 1.8 - 7.9e-7 (its error is the accumulation's).  `v_mfma_f32_16x16x32_f16` issues every 16 cycles with K = 32: 16 x the
// multiply-adds per cycle of `v_mfma_f32_16x16x4_f32`, so three piece products cost 3 / 16 of one f32 product, and
//   * the pieces are made where a value is PRODUCED (conv1_1's epilogue: 2.5 vector instructions per value with
//     v_cvt_pk_f16_f32; the patch is converted in place once per item), never at a fragment read;
//   * the depth transform is gone (its adds do not distribute over pieces): the direct form's 27 taps, two taps per K = 32
//     block -- [h_a | h_b] x [H_a | H_b],  [l_a | l_b] x [H_a | H_b],  [h_a | h_b] x [L_a | L_b] -- 42 MFMAs of 16 cycles per
//     tile of 16 output positions where the f32 kernel issues 144 of 32;
//   * act1 = 64 bytes per pixel as before, as FOUR planes of 16-byte slots (h c0-7, h c8-15, l c0-7, l c8-15), each split by
//     the parity of the row: slot ((quarter * 2 + (r & 1)) * 10 + dd) * 80 + (r >> 1) * 2 + col.  A B fragment is one
//     ds_read_b128, and the 16 positions of a tile (8 output rows x 2 columns, input rows 2 apart: one parity) are 16
//     CONSECUTIVE slots for every lane group of the LDS (the channel half and the tap of a K = 32 block pick planes, not
//     slots): conflict-free.  (Pixels as 64-byte records put eight rows of a tile on the same banks: 8-way conflicts.);
//   * a tile is ANY 16 positions (the operand address is per lane): the 576 positions of an item are 36 full tiles, no
//     remainder tiles, no exchange buffer.
// Same boundary as svk_c3d2_stage1 (f32 feature rows + crop starts in, f32 pooled activation out).
// =====================================================================================================
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int HACT_WORDS = 16 * DIN * NFRAME * 2;            // 25 600 32-bit words = 6 400 slots of 16 bytes
constexpr int HPLANE = DIN * NFRAME;                         // slots per (quarter, row parity) plane: [10 dd][40 r / 2][2 col]
constexpr int HPAIRS = 14;                                   // tap pairs of conv1_2 (27 taps + one empty)

// (h, l) of two f32 values as two packed-half words: {h0, h1}, {l0, l1}
__device__ __forceinline__ void split2(f32x2 v, unsigned& h, unsigned& l) {
  const f16x2 hh = __builtin_convertvector(v, f16x2);
  const f32x2 back = __builtin_convertvector(hh, f32x2);
  const f16x2 ll = __builtin_convertvector(v - back, f16x2);
  h = __builtin_bit_cast(unsigned, hh);
  l = __builtin_bit_cast(unsigned, ll);
}

// max(x, x of lane ^ 1) as ONE instruction (DPP quad_perm [1, 0, 3, 2] on the first source).  Written out: four calls of
// __builtin_amdgcn_mov_dpp on the four registers of an accumulator came back as one v_mov_b32_dpp of the first (ROCm 7.2).
// (the s_nop: a DPP read of a register the previous vector instruction wrote needs two wait states, and the compiler does not
// count them for asm statements)
__device__ __forceinline__ float max_with_lane_xor1(float x) {
  float d;
  asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(x));
  return d;
}

template <bool SLOPE01>
__global__ __launch_bounds__(512) void c3d2_stage1h_kernel(const Stage1Params p) {
  extern __shared__ __attribute__((aligned(16))) float smem_c3d2[];
  unsigned* const act = reinterpret_cast<unsigned*>(smem_c3d2);   // [HACT_WORDS]
  float* const patch = smem_c3d2 + HACT_WORDS;                    // [WP_FLOATS]: [12 dd][80 h][8], f32 from the DMA, then (l << 16 | h) words
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, kk = lane >> 4;
  const int pair = wave & 3, part = wave >> 2;
  const int n_items = p.n_utt * 36;

  u32x4 W2[HPAIRS][2];
#pragma unroll
  for (int pr = 0; pr < HPAIRS; ++pr) {
    W2[pr][0] = p.w2blk[(2 * pr) * 64 + lane];
    W2[pr][1] = p.w2blk[(2 * pr + 1) * 64 + lane];
  }
  const u32x4 W1a = p.w1blk[lane], W1b = p.w1blk[64 + lane];
  f32x4 b1v, sl1v, b2v, sl2v;   // a lane holds channels 4 kk .. 4 kk + 3 of ONE position (A = the weights)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    b1v[r] = p.bias1[4 * kk + r];
    sl1v[r] = p.slope1[4 * kk + r];
    b2v[r] = p.bias2[4 * kk + r];
    sl2v[r] = p.slope2[4 * kk + r];
  }

  int starts = 0;
  __shared__ int q_item3;
  int item = blockIdx.x, item1 = item + (int)gridDim.x, item2 = item1 + (int)gridDim.x;
  ItemPos cur = ItemPos::of(item), nx = ItemPos::of(item1), nx2 = ItemPos::of(item2);
  // The patch in place, f32 -> (l << 16 | h) words: by the wave that FETCHED the words (its own vmcnt(0) is all it needs: no
  // barrier of its own), depths pair, pair + 4, pair + 8 = 3 x 640 words = nine 16-byte trips per lane, all nine reads in flight
  // before the first conversion.  (As a pass of all eight waves in front of conv1_1, behind a barrier: 0.45 of 4.11 ms.)
  auto convert_own = [&]() {
    f32x4 v[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int w = (pair + 4 * (k / 3)) * (NFRAME * WPW) + 256 * (k % 3) + 4 * lane;
      if (k % 3 < 2 || lane < 32) v[k] = *reinterpret_cast<const f32x4*>(patch + w);
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int w = (pair + 4 * (k / 3)) * (NFRAME * WPW) + 256 * (k % 3) + 4 * lane;
      unsigned h0, l0, h1, l1;
      split2(__builtin_shufflevector(v[k], v[k], 0, 1), h0, l0);
      split2(__builtin_shufflevector(v[k], v[k], 2, 3), h1, l1);
      u32x4 o;   // word = the value's own pair: low half h, high half l
      o[0] = __builtin_amdgcn_perm(l0, h0, 0x05040100u);
      o[1] = __builtin_amdgcn_perm(l0, h0, 0x07060302u);
      o[2] = __builtin_amdgcn_perm(l1, h1, 0x05040100u);
      o[3] = __builtin_amdgcn_perm(l1, h1, 0x07060302u);
      if (k % 3 < 2 || lane < 32) *reinterpret_cast<u32x4*>(patch + w) = o;
    }
  };
  if (item < n_items) {
    starts = fetch_starts(p, cur, lane);
    if (part == 0) dma_patch_w(p, cur, starts, pair, lane, patch);
    if (item1 < n_items) starts = fetch_starts(p, nx, lane);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (part == 0 && item < n_items) convert_own();
  __syncthreads();
  while (item < n_items) {
    const int next = item1;
    unsigned q_ticket = 0;
    if (threadIdx.x == 0 && p.queue) q_ticket = atomicAdd(p.queue, 1u);

    // ---- (1) conv1_1 + PReLU -> act1 as (h, l): 100 tiles of 16 pixels, tile tt = wave + 8 m ----
    {
      // B = [h taps 0-7 | h taps 8-15 | l taps 0-7 | l taps 8-15] by kk; tap t = (kd, kw) = (t / 5, t % 5), t = 15: the zero column
      const unsigned* pw[8];
      const unsigned* const pbase = reinterpret_cast<const unsigned*>(patch) + 8 * WPW * wave + (i >> 1) * WPW;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int t0 = e, t1 = 8 + e;                                  // kk & 1 = 0 / 1
        const int o0 = (t0 / 5) * (NFRAME * WPW), c0 = t0 % 5;
        const int o1 = t1 < 15 ? (t1 / 5) * (NFRAME * WPW) : 0, c1 = t1 < 15 ? t1 % 5 : 0;
        const int colA = (i & 1) + c0, colB = (i & 1) + c1;
        const int offA = o0 + colA + (colA >= 3 ? 1 : 0), offB = o1 + colB + (colB >= 3 ? 1 : 0);
        pw[e] = pbase + ((kk & 1) ? offB : offA);
      }
      const unsigned sel = kk < 2 ? 0x05040100u : 0x07060302u;         // the h halves / the l halves of two words
      // pixel 16 tt + i = (dd = tt / 10, r = 8 (tt % 10) + (i >> 1), col = i & 1): slot 8 tt + 2 (i >> 2) + (i & 1) of the plane
      // (quarter kk >> 1 [+ 2 for l], parity (i >> 1) & 1); the lane's four channels are bytes 8 (kk & 1) .. + 7 of the slot
      unsigned* const aw = act + 4 * ((((kk >> 1) * 2 + ((i >> 1) & 1)) * HPLANE) + 8 * wave + 2 * (i >> 2) + (i & 1)) + 2 * (kk & 1);
      auto tile_group = [&](auto nt_tag, int m0) {
        constexpr int NT = decltype(nt_tag)::value;
        unsigned w[NT][8];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int e = 0; e < 8; ++e) w[t][e] = pw[e][64 * WPW * (m0 + t)];
        f32x4 acc[NT];
        u32x4 B[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
          for (int jx = 0; jx < 4; ++jx) B[t][jx] = __builtin_amdgcn_perm(w[t][2 * jx + 1], w[t][2 * jx], sel);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, W1a), __builtin_bit_cast(f16x8, B[t]), b1v, 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, W1b), __builtin_bit_cast(f16x8, B[t]), acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const f32x4 y = prelu4<SLOPE01>(acc[t], sl1v);
          unsigned h0, l0, h1, l1;
          split2(__builtin_shufflevector(y, y, 0, 1), h0, l0);
          split2(__builtin_shufflevector(y, y, 2, 3), h1, l1);
          *reinterpret_cast<u32x2*>(aw + 4 * 64 * (m0 + t)) = (u32x2){h0, h1};
          *reinterpret_cast<u32x2*>(aw + 4 * 64 * (m0 + t) + 4 * 4 * HPLANE) = (u32x2){l0, l1};
        }
      };
#pragma unroll
      for (int m0 = 0; m0 < 12; m0 += 4) tile_group(std::integral_constant<int, 4>{}, m0);
      if (wave < 4) tile_group(std::integral_constant<int, 1>{}, 12);
    }
    if (threadIdx.x == 0) q_item3 = p.queue ? (int)q_ticket + 3 * (int)gridDim.x : item2 + (int)gridDim.x;
    __syncthreads();   // act1 is complete; the patch buffer is free
    const int item3 = q_item3;

    // ---- (2) conv1_2 + PReLU + pool: 36 tiles of 16 positions, position P = 16 t + i -> (depth P / 72, row, column) ----
    {
      if (part == 0 && next < n_items) {
        dma_patch_w(p, nx, starts, pair, lane, patch);
        if (item2 < n_items) starts = fetch_starts(p, nx2, lane);
      }
      const int u = cur.u, q = cur.q(), j = cur.j();
      float* const obase = p.out + (int64_t)u * S_N + (TD * q) * S_D + j * S_W + 4 * kk;
      // tiles t = wave + 8 m (m < 4); the last four go to the YOUNGER waves (the older ones fetch and convert the next patch)
#pragma unroll 1
      for (int m = 0; m < 4 + part; ++m) {
        const int t = m < 4 ? wave + 8 * m : 28 + wave;
        const int P = 16 * t + i;
        const int dq = (P * 911) >> 16, rem = P - 72 * dq, row = rem >> 1;          // P / 72 for P < 576
        // pixel (dd = dq + kd, r = 2 row + kh, col), channels 8 (kk & 1) .. + 7: slot (((kk & 1) * 2 + (kh & 1)) * 10 + dd) * 80 +
        // (row + kh / 2) * 2 + col of the h planes; the l planes 4 HPLANE slots on.  The taps of a pair (kd, 2 m), (kd, 2 m + 1)
        // differ by the parity plane; the pair (0, 8) | (1, 8) by one depth
        const int base = 16 * (((kk & 1) * 2) * HPLANE + dq * 80 + 2 * row + (i & 1));
        const char* const a2 = reinterpret_cast<const char*>(act) + base + (kk >= 2 ? 16 * HPLANE : 0);
        const char* const a3 = reinterpret_cast<const char*>(act) + base + (kk >= 2 ? 16 * 80 : 0);
        f32x4 acc = b2v;
        // pair pr: 0 .. 11 = (kd = pr / 4, kh = 2 (pr % 4) | + 1) off a2; 12 = taps (0, 8) | (1, 8) off a3; 13 = tap (2, 8) | none
        auto rd = [&](int pr, int piece) -> u32x4 {
          const char* ad = pr < 12 ? a2 + 1280 * (pr / 4) + 32 * (pr % 4) : pr == 12 ? a3 + 32 * 4 : a3 + 2 * 1280 + 32 * 4 - (kk >= 2 ? 16 * 80 : 0);
          return *reinterpret_cast<const u32x4*>(ad + 16 * 4 * HPLANE * piece);
        };
        // fragments TWO pairs ahead (three rotating sets): a pair is 48 cycles of MFMA, less than an LDS round trip
        u32x4 bh[3], bl[3];
        bh[0] = rd(0, 0);
        bl[0] = rd(0, 1);
        bh[1] = rd(1, 0);
        bl[1] = rd(1, 1);
#pragma unroll
        for (int pr = 0; pr < HPAIRS; ++pr) {
          if (pr + 2 < HPAIRS) {
            bh[(pr + 2) % 3] = rd(pr + 2, 0);
            bl[(pr + 2) % 3] = rd(pr + 2, 1);
          }
          __builtin_amdgcn_sched_barrier(0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, W2[pr][0]), __builtin_bit_cast(f16x8, bh[pr % 3]), acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, W2[pr][0]), __builtin_bit_cast(f16x8, bl[pr % 3]), acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, W2[pr][1]), __builtin_bit_cast(f16x8, bh[pr % 3]), acc, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        // PReLU, max over the column pair (lanes i, i ^ 1: the same depth and row), the even lane stores its four channels
        const f32x4 y = prelu4<SLOPE01>(acc, sl2v);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = max_with_lane_xor1(y[r]);
        if ((i & 1) == 0) *reinterpret_cast<f32x4*>(obase + dq * S_D + row * S_PAR) = o;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces have landed
    if (part == 0 && next < n_items) convert_own();
    __syncthreads();  // the next patch is in place and converted; act1 may be overwritten
    item = item1;
    item1 = item2;
    item2 = item3;
    cur = nx;
    nx = nx2;
    nx2 = ItemPos::of(item3);
  }
}

}  // namespace

extern "C" {

size_t svk_c3d2_stage1_lds_bytes(void) { return sizeof(float) * (size_t)(HACT_WORDS + WP_FLOATS); }

int svk_c3d2_stage1(svk_ctx* ctx, const float* d_feat, int32_t n_utt, int32_t max_frames, int32_t n_cols,
                     const int32_t* d_crop_idx, int32_t n_crops, int32_t crop_frames, const void* d_w1blk,
                     const float* d_bias1, const float* d_slope1, const void* d_w2blk, const float* d_bias2,
                     const float* d_slope2, int32_t flags, float* d_out) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  const bool slope01 = (flags & 2) != 0;
  SVK_REQUIRE(ctx, (flags & ~2) == 0, "flags: only bit 1 (slopes in [0, 1]) is defined");
  SVK_REQUIRE(ctx, n_utt >= 0 && max_frames >= 1, "shape");
  if (n_cols != NCOEF || n_crops != NCROP || crop_frames != NFRAME)
    return svk_fail(ctx, SVK_ERR_UNSUPPORTED,
                    "svk_c3d2_stage1 is built for the 20 x 80 x 40 cube of utils.py:20-21 (got %d x %d x %d)", n_crops,
                    crop_frames, n_cols);
  if (n_utt == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_feat && d_crop_idx && d_w1blk && d_bias1 && d_slope1 && d_w2blk && d_bias2 && d_slope2 && d_out,
              "NULL buffer");
  SVK_REQUIRE(ctx, (reinterpret_cast<uintptr_t>(d_feat) & 7) == 0 && ((reinterpret_cast<uintptr_t>(d_w1blk) | reinterpret_cast<uintptr_t>(d_w2blk) |
                                                                    reinterpret_cast<uintptr_t>(d_out)) & 15) == 0,
              "d_feat must be 8-byte, the weight blocks and d_out 16-byte aligned");
  SVK_REQUIRE(ctx, (int64_t)n_utt * 36 + 4 * (int64_t)ctx->num_cu < ((int64_t)1 << 31), "too many cubes for one launch");
  Stage1Params p;
  p.feat = d_feat;
  p.crop = d_crop_idx;
  p.n_utt = n_utt;
  p.max_frames = max_frames;
  p.w1blk = static_cast<const u32x4*>(d_w1blk);
  p.bias1 = d_bias1;
  p.slope1 = d_slope1;
  p.w2blk = static_cast<const u32x4*>(d_w2blk);
  p.bias2 = d_bias2;
  p.slope2 = d_slope2;
  p.out = d_out;
  const size_t lds = svk_c3d2_stage1_lds_bytes();
  if (lds + 64 > (size_t)ctx->lds_per_cu)
    return svk_fail(ctx, SVK_ERR_UNSUPPORTED, "svk_c3d2_stage1 needs %zu bytes of LDS per workgroup (device: %d)", lds, ctx->lds_per_cu);
  void (*kern)(const Stage1Params) = slope01 ? c3d2_stage1h_kernel<true> : c3d2_stage1h_kernel<false>;
  SVK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t items = (int64_t)n_utt * 36;
  const unsigned grid = (unsigned)std::min<int64_t>(items, ctx->num_cu);
  p.queue = getenv("SVK_C3D2_STATIC_ITEMS") ? nullptr : reinterpret_cast<unsigned*>(static_cast<char*>(ctx->scratch) + 112);
  if (p.queue) SVK_HIP(ctx, hipMemsetAsync(p.queue, 0, 4, ctx->stream));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, ctx->stream, p);
  SVK_LAUNCH_CHECK(ctx);
  return SVK_OK;
}

}  // extern "C"

// =====================================================================================================
// The second block: conv2_1 (16 -> 32, kernel (3,1,4)) + BN + PReLU, conv2_2 (32 -> 32, kernel (3,8,1),
// stride (1,2,1)) + BN + PReLU + MaxPool3d((1,1,2))  (/root/reference/model.py:119-124, :151-158), two
// kernels of the same shape as the matrix phase above: the input region of a work item is staged in LDS
// as padded 'pixels' (one pixel = the channel vector of one (d, h, w) position), the weights of the
// wave's output-channel tile(s) sit in registers, and the A operand of every tap is one ds_read_b128
// at a fixed offset from the M tile's base address.
// =====================================================================================================
namespace {

constexpr int S2_D = 16, S2_H = 36, S2_W = 18;       // input of conv2_1 (after pool1), 16 channels
constexpr int A2_D = 14, A2_W = 15;                  // conv2_1 output (32 channels), rows = S2_H
constexpr int O2_D = 12, O2_H = 15, O2_W = 7;        // after conv2_2 + pool2 (32 channels)

struct Conv21Params {
  const float* in;      // [n][16][36][18][16]
  const f32x4* wfrag;   // [2 nt][12 taps][64]: lane (co = 16 nt + (l & 15), kk = l >> 4), e: W[co][4 kk + e][kd][kw], tap = 4 kd + kw
  const float* bias;    // [32]
  const float* slope;   // [32]
  float* out;           // [n][14][36][15][32]
  int32_t n_utt;
  unsigned* queue;      // work-item counter (zeroed by the host before the launch), or NULL = static round-robin
};

// ---- conv2_1 through Winograd's F(2, 3) along depth (see c3d2_stage1w_kernel): per output depth pair P and row, the
// planes x0 .. x3 = depths 2 P .. 2 P + 3 give t0 .. t3, four accumulators per N tile, 128 MFMAs where the direct form
// issues 192; a transformed fragment feeds 8 MFMAs (two N tiles), so the packed adds are 1 per 4 MFMAs.  Item = (cube,
// block of 4 rows): 7 pairs x 4 rows = 28 M tiles, 7 per wave; 78 KB of LDS, two workgroups per CU; the transformed
// weights (128 VGPRs) are derived from the direct fragments in the prologue. ----
constexpr int C21W_TH = 4;
constexpr int C21W_PIX = S2_D * C21W_TH * S2_W;      // 1152 pixels of 16 channels, stored at 16 p + 4 (p >> 2)
constexpr int C21W_LDS_FLOATS = 17 * (C21W_PIX + 4);
constexpr int C21W_DSTEP = 17 * C21W_TH * S2_W;      // floats per depth plane (72 pixels = 18 groups of 4)

template <bool SLOPE01>
__global__ __launch_bounds__(256, 2) void c3d2_conv21w_kernel(const Conv21Params p) {
  extern __shared__ __attribute__((aligned(16))) float smem_c21w[];
  float* reg = smem_c21w;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, kk = lane >> 4;
  f32x4 G[2][4][4];   // [nt][k][kw]
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int kw = 0; kw < 4; ++kw) {
      const f32x4 g0 = p.wfrag[(nt * 12 + kw) * 64 + lane], g1 = p.wfrag[(nt * 12 + 4 + kw) * 64 + lane],
                  g2 = p.wfrag[(nt * 12 + 8 + kw) * 64 + lane];
      G[nt][0][kw] = g0;
      G[nt][1][kw] = 0.5f * ((g0 + g2) + g1);
      G[nt][2][kw] = 0.5f * ((g0 + g2) - g1);
      G[nt][3][kw] = g2;
    }
  // M = channel (A = the weights G), N = output column w' (B = the transformed fragments): a lane ends up with channels
  // 16 nt + 4 kk .. + 3 of column i -- 16 contiguous bytes of the channels-last output: one 16-byte store per (N tile, depth)
  // where M = position issued four 4-byte stores (round 4; the same products in the same order: bit-identical)
  f32x4 b4[2], sl4[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      b4[nt][r] = p.bias[16 * nt + 4 * kk + r];
      sl4[nt][r] = p.slope[16 * nt + 4 * kk + r];
    }
  constexpr int BLOCKS = S2_H / C21W_TH;   // 9 row blocks per cube
  const int n_items = p.n_utt * BLOCKS;
  // Work items come from a device-wide counter, not from a fixed stride: of the two workgroups that share a CU the older
  // one gets about two MFMA issue slots in three, so with equal shares it finished 20 % early and its partner ran the
  // tail alone, latency-exposed (in-kernel stamps: 1.43 M vs 1.76 M loop cycles).  The next index is requested at the
  // top of an item (it returns during the staging) and published through LDS with the staging barrier.
  __shared__ int q_next;
  int item = blockIdx.x;
  while (item < n_items) {
    unsigned q_ticket = 0;
    if (threadIdx.x == 0 && p.queue) q_ticket = atomicAdd(p.queue, 1u);
    const int u = item / BLOCKS, hb = (item - u * BLOCKS) * C21W_TH;
    // stage [16 d][4 rows][18 w][16 c]: 18 16-byte pieces per thread, six in flight at a time
    const float* src = p.in + (int64_t)u * (S2_D * S2_H * S2_W * 16);
    constexpr int NV = 6;
    static_assert(C21W_PIX * 4 % (256 * NV) == 0, "whole rounds");
    // piece e = t + 256 k of thread t: 16-byte piece e & 3 of pixel (t >> 2) + 64 k.  The LDS address is linear in k (64 k is a
    // multiple of 4: 16 pix + 4 (pix >> 2) moves by 1 088 k floats); the source is (576 d + 18 hb + pix) * 16 with d = pix / 72,
    // and for a compile-time k that is a constant plus ONE comparison of t >> 2 with the window's depth boundary: two vector
    // instructions per piece where the division took a dozen (none of them overlaps an MFMA on this chip).
    {
      int tl = threadIdx.x;
      asm volatile("" : "+v"(tl));   // (keeps this arithmetic inside the item loop: hoisted, it costs registers)
      const int piece = tl & 3, pix0 = tl >> 2;
      float* const a0s = reg + 16 * pix0 + 4 * (pix0 >> 2) + 4 * piece;
      const float* const g0 = src + (int64_t)(hb * S2_W + pix0) * 16 + 4 * piece;
      constexpr int PER_D = C21W_TH * S2_W;            // 72 staged pixels per depth, 648 in the tensor
      constexpr int DSTEP = (S2_H * S2_W - PER_D) * 16;  // floats the source gains per depth on top of 16 pix
#pragma unroll
      for (int r0 = 0; r0 < 18; r0 += NV) {
        f32x4 sv[NV];
#pragma unroll
        for (int k = r0; k < r0 + NV; ++k) {
          const int d_lo = (64 * k) / PER_D, cross = PER_D * (d_lo + 1) - 64 * k;   // pix0 >= cross: the next depth
          const float* g = g0 + 1024 * k + DSTEP * d_lo;
          if (cross < 64) g = pix0 >= cross ? g + DSTEP : g;
          sv[k - r0] = *reinterpret_cast<const f32x4*>(g);
        }
#pragma unroll
        for (int k = r0; k < r0 + NV; ++k) *reinterpret_cast<f32x4*>(a0s + 1088 * k) = sv[k - r0];
      }
    }
    if (threadIdx.x == 0) q_next = p.queue ? (int)q_ticket + (int)gridDim.x : item + (int)gridDim.x;
    __syncthreads();
    const int item_next = q_next;
    // M tiles = (pair P, row hl): 16 pixels w' = 0 .. 15 (15 is a dummy), tile = 4 P + hl = 4 P + wave: this wave's seven tiles
    // differ in P alone, and pixel 144 P + r sits at 2 448 P + 16 r + 4 (r >> 2) -- the four tap addresses of a lane are those
    // of its first tile plus 2 448 floats per tile (four adds per tile; recomputed from the tile index they were 22 vector
    // instructions per 128 MFMAs, none of which an MFMA hides)
    int ao[4];   // (offsets into `reg`, not pointers: a pointer carried round the loop loses its LDS address space -- flat loads)
    {
      int il = i;
      asm volatile("" : "+v"(il));   // (inside the item loop: hoisted out of it, the four addresses cost registers all kernel long)
#pragma unroll
      for (int kw = 0; kw < 4; ++kw) {
        const int r = S2_W * wave + il + kw;
        ao[kw] = 16 * r + 4 * (r >> 2) + 4 * kk;
      }
    }
    // the first tile's first tap; every later tile's is read under the previous tile's last 32 MFMAs
    f32x4 x[4];
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) x[dd] = *reinterpret_cast<const f32x4*>(reg + ao[0] + C21W_DSTEP * dd);
#pragma unroll 1
    for (int P = 0; P < 7; ++P) {
      // (a1 enters both outputs with a plus sign: the bias rides in its accumulator)
      f32x4 acc[2][4];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[nt][k] = k == 1 ? b4[nt] : (f32x4){0.f, 0.f, 0.f, 0.f};
      f32x2 t[4][2];
#pragma unroll
      for (int kw = 0; kw < 4; ++kw) {
        // this tap's transformed fragments (8 packed adds, one burst), the next tap's reads, 32 MFMAs
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) wino_input_pair(x, hf, t);
        __builtin_amdgcn_sched_barrier(0);
        if (kw + 1 < 4) {
#pragma unroll
          for (int dd = 0; dd < 4; ++dd) x[dd] = *reinterpret_cast<const f32x4*>(reg + ao[kw + 1] + C21W_DSTEP * dd);
        } else if (P + 1 < 7) {
#pragma unroll
          for (int dd = 0; dd < 4; ++dd) x[dd] = *reinterpret_cast<const f32x4*>(reg + ao[0] + 2 * C21W_DSTEP + C21W_DSTEP * dd);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int k = 0; k < 4; ++k)
              acc[nt][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(G[nt][k][kw][e], t[k][e >> 1][e & 1], acc[nt][k], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int kw = 0; kw < 4; ++kw) ao[kw] += 2 * C21W_DSTEP;
      // rows 4 kk + r = channel 16 nt + 4 kk + r; column i = output column w' (15 is the dummy); depths 2 P and 2 P + 1
      // (wave-uniform 64-bit bases + one 32-bit lane offset + immediates: the stores need no per-store address VALU)
      float* const o0 = p.out + (((int64_t)u * A2_D + 2 * P) * S2_H + hb + wave) * (A2_W * 32);
      float* const o1 = o0 + (int64_t)S2_H * (A2_W * 32);
      const int lo = i * 32 + 4 * kk;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        // y0 = (a0 + a1) + a2, y1 = (a1 - a2) - a3 as packed adds (every VALU instruction here is paid in MFMA slots)
        f32x2 s0[2], s1[2];
        wino_output(acc[nt], s0, s1);
        f32x4 y0, y1;
        prelu_pairs<SLOPE01>(s0, s1, sl4[nt], y0, y1);
        if (i < A2_W) {
          *reinterpret_cast<f32x4*>(o0 + lo + 16 * nt) = y0;
          *reinterpret_cast<f32x4*>(o1 + lo + 16 * nt) = y1;
        }
      }
    }
    __syncthreads();
    item = item_next;
  }
}

struct Conv22Params {
  const float* in;      // [n][14][36][15][32]
  const f32x4* wfrag;   // [2 nt][24 taps][2 chunks][64]: e: W[16 nt + (l & 15)][16 chunk + 4 (l >> 4) + e][kd][kh], tap = 8 kd + kh
  const float* bias;    // [32]
  const float* slope;   // [32]
  float* out;           // [n][12][15][7][32]
  int32_t n_utt;
  unsigned long long* stamps;   // tuning builds only: [grid][4 waves][4] summed phase cycles
  unsigned* queue;              // work-item counter (zeroed before the launch), or NULL
};

// ---- conv2_2 + pool2 through the depth transform.  Its transformed weights are 4 x 8 x 32 x 32 floats = 512 VGPRs
// x 64 lanes: exactly the registers of four waves at two workgroups per CU, so every weight lives in ONE wave and each
// wave = (N tile nt, 16-channel K chunk ch) holds 32 fragments and produces PARTIAL sums over its half of K.  Item =
// (cube, pooled column j, third q of the output depths): two depth pairs x two columns = four M tiles (16 output rows,
// 15 real) of 8 row taps x 16 MFMAs per wave.  The two waves of an N tile swap partial sums through LDS: each writes the
// output-transformed sums of the pair it does not finish, keeps those of the pair it does (ch finishes pair ch), and
// after the barrier adds its partner's, then bias (carried by chunk 0's accumulator), PReLU, max over the column pair. ----
constexpr int C22W_TD = 4;                                   // output depths per item
constexpr int C22W_PIX = (C22W_TD + 2) * 2 * S2_H;           // [6 d][2 w][36 h] pixels of 32 channels at 32 p + 4 (p >> 1)
constexpr int C22W_IN_FLOATS = 34 * (C22W_PIX + 4);
constexpr int C22W_XCH_FLOATS = 2 * 2 * 2 * 2 * 64 * 4;      // [nt][pair][column][y][lane] f32x4
constexpr int C22W_LDS_FLOATS = C22W_IN_FLOATS + C22W_XCH_FLOATS;
constexpr int C22W_PLANE = 34 * 2 * S2_H, C22W_COL = 34 * S2_H;   // floats per depth plane / per column inside it

template <bool SLOPE01>
__global__ __launch_bounds__(256, 2) void c3d2_conv22w_kernel(const Conv22Params p) {
  extern __shared__ __attribute__((aligned(16))) float smem_c22w[];
  float* reg = smem_c22w;
  float* exch = reg + C22W_IN_FLOATS;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, kk = lane >> 4;
  const int nt = wave & 1, ch = wave >> 1;
#ifdef SVK_TUNING
  const unsigned long long clk_entry = __builtin_amdgcn_s_memrealtime();   // 100 MHz, one counter for the whole chip
#endif
  f32x4 G[4][8];   // [k][kh], this wave's N tile and K chunk
#pragma unroll
  for (int kh = 0; kh < 8; ++kh) {
    const f32x4 g0 = p.wfrag[((nt * 24 + kh) * 2 + ch) * 64 + lane], g1 = p.wfrag[((nt * 24 + 8 + kh) * 2 + ch) * 64 + lane],
                g2 = p.wfrag[((nt * 24 + 16 + kh) * 2 + ch) * 64 + lane];
    G[0][kh] = g0;
    G[1][kh] = 0.5f * ((g0 + g2) + g1);
    G[2][kh] = 0.5f * ((g0 + g2) - g1);
    G[3][kh] = g2;
  }
  const float b = ch == 0 ? p.bias[16 * nt + i] : 0.f, sl = p.slope[16 * nt + i];
  const float* const a0 = reg + 68 * i + 4 * kk + 16 * ch;   // row 2 i of column 0 of plane 0, this lane's K piece
  constexpr int PER_CUBE = O2_W * (O2_D / C22W_TD);           // 7 x 3 items
  const int n_items = p.n_utt * PER_CUBE;
#ifdef SVK_TUNING
  unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long clk_c0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  __shared__ int q_next;   // dynamic work items: see c3d2_conv21w_kernel
  int item = blockIdx.x;
  while (item < n_items) {
    SVK_STAMP(ts0);
    unsigned q_ticket = 0;
    if (threadIdx.x == 0 && p.queue) q_ticket = atomicAdd(p.queue, 1u);
    const int u = item / PER_CUBE, rem = item - u * PER_CUBE, q = rem / O2_W, j = rem - q * O2_W;
    // stage [6 d][36 h][2 w][32 c] of the input as pixels p = (dl * 2 + w) * 36 + h (h fastest): 13.5 16-byte pieces per
    // thread, seven in flight at a time
    const float* src = p.in + ((int64_t)u * A2_D + C22W_TD * q) * (S2_H * A2_W * 32) + 2 * j * 32;
    constexpr int NV = 7;
    // piece e = t + 256 k of thread t: (16-byte piece e & 7, column wq = (e >> 3) & 1) are the thread's own, dh = (t >> 4) +
    // 16 k.  Its LDS pixel is dh + 36 wq + 36 dl with dl = dh / 36, and 16 k + 36 dl is even, so the address is
    // A0 + 544 k + 1 224 dl: for a compile-time k, dl is a constant plus ONE comparison of t >> 4 with the window's depth
    // boundary -- two vector instructions per piece where the general form (a division by 36, the pixel, the pad) took a dozen
    // (a fifth of this kernel's vector instructions, none of which overlaps an MFMA).
    // (Measured, round 4: the loads of the NEXT item issued right behind the second barrier, in front of the epilogue's stores --
    // 8 of the 14 pieces, more spills -- move 2.3 k cycles from this stamp into the epilogue and leave the kernel where it was.)
    {
      int tl = threadIdx.x;
      asm volatile("" : "+v"(tl));   // (keeps this arithmetic inside the item loop: hoisted, it costs registers the kernel spills)
      const int piece = tl & 7, wq = (tl >> 3) & 1, dh0 = tl >> 4;
      const int pixb = dh0 + 36 * wq;
      float* const a0s = reg + 32 * pixb + 4 * (pixb >> 1) + 4 * piece;
      const float* const g0 = src + dh0 * (A2_W * 32) + wq * 32 + 4 * piece;
#pragma unroll
      for (int r0 = 0; r0 < 14; r0 += NV) {
        f32x4 sv[NV];
#pragma unroll
        for (int k = r0; k < r0 + NV; ++k)
          if (k < 13 || tl < 128) sv[k - r0] = *reinterpret_cast<const f32x4*>(g0 + (int64_t)(16 * k) * (A2_W * 32));
#pragma unroll
        for (int k = r0; k < r0 + NV; ++k) {
          const int dl_lo = (16 * k) / S2_H, cross = S2_H * (dl_lo + 1) - 16 * k;   // dh0 >= cross: the next depth
          float* d = a0s + 544 * k + 1224 * dl_lo;
          if (cross < 16) d = dh0 >= cross ? d + 1224 : d;
          if (k < 13 || tl < 128) *reinterpret_cast<f32x4*>(d) = sv[k - r0];
        }
      }
    }
    SVK_STAMP(ts1);
    if (threadIdx.x == 0) q_next = p.queue ? (int)q_ticket + (int)gridDim.x : item + (int)gridDim.x;
    __syncthreads();
    SVK_STAMP(ts2);
    const int item_next = q_next;
    f32x4 own[2][2];   // [column][y]: the partial sums of the pair this wave finishes
    // the first tile's first tap; every later tile's is read under the previous tile's last 16 MFMAs
    f32x4 x[4];
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) x[dd] = *reinterpret_cast<const f32x4*>(a0 + 2 * C22W_PLANE * (1 - ch) + C22W_PLANE * dd);
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      const int P = pass ? ch : 1 - ch;   // the partner's pair first
#pragma unroll
      for (int wq = 0; wq < 2; ++wq) {
        const float* ap = a0 + 2 * C22W_PLANE * P + C22W_COL * wq;
        f32x4 acc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = k == 1 ? (f32x4){b, b, b, b} : (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x2 t[4][2];
#pragma unroll
        for (int kh = 0; kh < 8; ++kh) {
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) wino_input_pair(x, hf, t);
          __builtin_amdgcn_sched_barrier(0);
          if (kh + 1 < 8) {
            const int off = 32 * (kh + 1) + 4 * ((kh + 1) >> 1);
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) x[dd] = *reinterpret_cast<const f32x4*>(ap + C22W_PLANE * dd + off);
          } else if (wq == 0) {          // the pair's second column
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) x[dd] = *reinterpret_cast<const f32x4*>(ap + C22W_COL + C22W_PLANE * dd);
          } else if (pass == 0) {        // the second pass's pair (ch), column 0
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) x[dd] = *reinterpret_cast<const f32x4*>(a0 + 2 * C22W_PLANE * ch + C22W_PLANE * dd);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int k = 0; k < 4; ++k)
              acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(t[k][e >> 1][e & 1], G[k][kh][e], acc[k], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        f32x2 s0[2], s1[2];
        wino_output(acc, s0, s1);
        const f32x4 y0 = __builtin_shufflevector(s0[0], s0[1], 0, 1, 2, 3), y1 = __builtin_shufflevector(s1[0], s1[1], 0, 1, 2, 3);
        if (pass == 0) {
          float* xo = exch + ((((nt * 2 + P) * 2 + wq) * 2) * 64 + lane) * 4;
          *reinterpret_cast<f32x4*>(xo) = y0;
          *reinterpret_cast<f32x4*>(xo + 256) = y1;
        } else {
          own[wq][0] = y0;
          own[wq][1] = y1;
        }
      }
    }
    SVK_STAMP(ts3);
    __syncthreads();   // the partner's partial sums are in LDS; nobody reads the input region any more
    SVK_STAMP(ts4);
    SVK_STAMP_ADD(0, ts0, ts1);  // staging
    SVK_STAMP_ADD(1, ts1, ts2);  // barrier 1
    SVK_STAMP_ADD(2, ts2, ts3);  // 2 passes x 2 tiles of 8 taps x 16 MFMAs + transforms + exchange writes
    SVK_STAMP_ADD(3, ts3, ts4);  // barrier 2 (the epilogue that follows is counted with the next item's staging)
    {
      const float* xi = exch + ((((nt * 2 + ch) * 2) * 2) * 64 + lane) * 4;   // pair ch, written by wave (nt, 1 - ch)
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        const f32x4 v0 = own[0][y] + *reinterpret_cast<const f32x4*>(xi + 256 * y);
        const f32x4 v1 = own[1][y] + *reinterpret_cast<const f32x4*>(xi + 512 + 256 * y);
        // rows 4 kk + r = output row h'; pool over the column pair, PReLU first (model.py:156-158)
        float* const o = p.out + ((((int64_t)u * O2_D + C22W_TD * q + 2 * ch + y) * O2_H) * O2_W + j) * 32 + 16 * nt;   // uniform
        const int olane = 4 * kk * (O2_W * 32) + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int hq = 4 * kk + r;
          if (hq < O2_H) o[olane + r * (O2_W * 32)] = fmaxf(prelu_t<SLOPE01>(v0[r], sl), prelu_t<SLOPE01>(v1[r], sl));
        }
      }
    }
    item = item_next;
  }
#ifdef SVK_TUNING
  if (p.stamps && lane == 0) {
    for (int k = 0; k < 2; ++k) p.stamps[((size_t)blockIdx.x * 4 + wave) * 4 + k] = stamp_acc[k];
    p.stamps[((size_t)blockIdx.x * 4 + wave) * 4 + 2] = stamp_acc[2] + stamp_acc[3];   // (MFMA passes + barrier 2 in one slot;
    // slot 3 of wave 0 / 1 carries the clock pair: shader cycles and 100 MHz ticks over the loop)
    // (waves 2 / 3: the absolute 100 MHz time of the workgroup's entry into the kernel / of its leaving the loop)
    p.stamps[((size_t)blockIdx.x * 4 + wave) * 4 + 3] = wave == 0 ? __builtin_amdgcn_s_memtime() - clk_c0
                                                        : wave == 1 ? __builtin_amdgcn_s_memrealtime() - clk_r0
                                                        : wave == 2 ? clk_entry : __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// ---- conv3_1 (32 -> 64, kernel (3,1,3)) + BN + PReLU (model.py:126-128, :159-161), depth-transformed like the kernels
// above.  No taps along h, so item = (cube, block of 3 rows): region 12 depths x 3 rows x 7 columns of 32 channels
// (36 KB as pixels of 36 floats: pixel stride 9 sixteen-byte slots, odd, so pixels that differ mod 16 never share a bank
// slot; the 15 pixels of a tile span 19, three lanes of a quarter wave take a second pass).  M tile = 3 rows x 5 output
// columns of one depth pair (15 positions + 1 dummy), five tiles per item; wave = N tile (16 of the 64 output channels),
// 4 k x 3 kw x 2 chunks = 24 weight fragments = 96 VGPRs; 96 MFMAs per tile where the direct form issues 144. ----
constexpr int C31_PIXF = 36;
constexpr int C31_PIX = 12 * 3 * 7;                           // pixel p = (d * 3 + hl) * 7 + w
constexpr int C31_LDS_FLOATS = C31_PIXF * C31_PIX;
constexpr int C31_PLANE = C31_PIXF * 21;                      // floats per depth plane

struct Conv31Params {
  const float* in;      // [n][12][15][7][32]
  const f32x4* wfrag;   // [4 nt][9 taps][2 chunks][64]: e: W[16 nt + (l & 15)][16 chunk + 4 (l >> 4) + e][kd][kw], tap = 3 kd + kw
  const float* bias;    // [64]
  const float* slope;   // [64]
  float* out;           // chunked and column-major: [n][10][8 chunks][5 w][15 h][8] (what svk_c3d2_conv32t stages)
  int32_t n_utt;
  unsigned* queue;      // work-item counter (zeroed before the launch), or NULL
};

template <bool SLOPE01>
__global__ __launch_bounds__(256, 3) void c3d2_conv31w_kernel(const Conv31Params p) {
  extern __shared__ __attribute__((aligned(16))) float smem_c31[];
  float* reg = smem_c31;
  const int lane = threadIdx.x & 63, nt = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, kk = lane >> 4;
  f32x4 G[4][3][2];   // [k][kw][chunk]
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      const f32x4 g0 = p.wfrag[((nt * 9 + kw) * 2 + ch) * 64 + lane], g1 = p.wfrag[((nt * 9 + 3 + kw) * 2 + ch) * 64 + lane],
                  g2 = p.wfrag[((nt * 9 + 6 + kw) * 2 + ch) * 64 + lane];
      G[0][kw][ch] = g0;
      G[1][kw][ch] = 0.5f * ((g0 + g2) + g1);
      G[2][kw][ch] = 0.5f * ((g0 + g2) - g1);
      G[3][kw][ch] = g2;
    }
  // M = channel (A = the weights G), N = position (B = the transformed fragments): a lane ends up with channels 16 nt + 4 kk .. + 3
  // of ONE position -- 16 contiguous bytes of the chunked output (round 4, as in c3d2_conv21w_kernel; bit-identical)
  f32x4 b4, sl4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    b4[r] = p.bias[16 * nt + 4 * kk + r];
    sl4[r] = p.slope[16 * nt + 4 * kk + r];
  }
  // B column of this lane: position m = i -> (row hl = m / 5, column w' = m % 5); m = 15 is a dummy (reads position 14)
  const int mi = i < 15 ? i : 14;
  const float* const a0 = reg + C31_PIXF * ((mi / 5) * 7 + mi % 5) + 4 * kk;
  const int n_items = p.n_utt * 5;
  __shared__ int q_next;   // dynamic work items: see c3d2_conv21w_kernel (three workgroups share a CU here)
  int item = blockIdx.x;
  while (item < n_items) {
    unsigned q_ticket = 0;
    if (threadIdx.x == 0 && p.queue) q_ticket = atomicAdd(p.queue, 1u);
    const int u = item / 5, rb = item - u * 5;
    // stage [12 d][3 rows][7 w][32 c]: per depth 672 contiguous floats = 168 sixteen-byte pieces; 2 016 in all, eight
    // per thread, four in flight at a time (168 VGPRs at three workgroups per CU)
    const float* src = p.in + ((int64_t)u * 12 * 15 + 3 * rb) * (7 * 32);
    // piece e = t + 256 k of thread t.  168 d is a multiple of 8, so the LDS address 36 (21 d + (r >> 3)) + 4 (r & 7) with
    // r = e - 168 d is 36 (e >> 3) + 4 (e & 7): no d in it, linear in k.  The source is 4 e + 2 688 d floats, and for a
    // compile-time k the depth d = e / 168 is a constant plus at most two comparisons of t with the window's boundaries.
    {
      int tl = threadIdx.x;
      asm volatile("" : "+v"(tl));   // (keeps this arithmetic inside the item loop)
      float* const a0s = reg + C31_PIXF * (tl >> 3) + 4 * (tl & 7);
      const float* const g0 = src + 4 * tl;
      constexpr int DSTEP = 15 * 7 * 32 - 4 * 168;   // floats the source gains per depth on top of 4 e
#pragma unroll
      for (int r0 = 0; r0 < 8; r0 += 4) {
        f32x4 sv[4];
#pragma unroll
        for (int k = r0; k < r0 + 4; ++k) {
          const int d_lo = (256 * k) / 168, c1 = 168 * (d_lo + 1) - 256 * k, c2 = c1 + 168;   // t >= c1 (c2): one (two) depths on
          const float* g = g0 + 1024 * k + DSTEP * d_lo;
          if (c1 < 256) g = tl >= c1 ? g + DSTEP : g;
          if (c2 < 256) g = tl >= c2 ? g + DSTEP : g;
          if (256 * k + 255 < 2016 || tl < 2016 - 256 * k) sv[k - r0] = *reinterpret_cast<const f32x4*>(g);
        }
#pragma unroll
        for (int k = r0; k < r0 + 4; ++k)
          if (256 * k + 255 < 2016 || tl < 2016 - 256 * k) *reinterpret_cast<f32x4*>(a0s + 32 * C31_PIXF * k) = sv[k - r0];
      }
    }
    if (threadIdx.x == 0) q_next = p.queue ? (int)q_ticket + (int)gridDim.x : item + (int)gridDim.x;
    __syncthreads();
    const int item_next = q_next;
    // the first pair's first fragments; every later pair's are read under the previous pair's last 16 MFMAs
    f32x4 x[4];
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) x[dd] = *reinterpret_cast<const f32x4*>(a0 + C31_PLANE * dd);
#pragma unroll 1
    for (int P = 0; P < 5; ++P) {
      const float* ap = a0 + 2 * C31_PLANE * P;
      f32x4 acc[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = k == 1 ? b4 : (f32x4){0.f, 0.f, 0.f, 0.f};
      f32x2 t[4][2];
#pragma unroll
      for (int st = 0; st < 6; ++st) {   // step = (column tap kw, 16-channel chunk)
        const int kw = st >> 1, ch = st & 1;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) wino_input_pair(x, hf, t);
        __builtin_amdgcn_sched_barrier(0);
        if (st + 1 < 6) {
          const int off = C31_PIXF * ((st + 1) >> 1) + 16 * ((st + 1) & 1);
#pragma unroll
          for (int dd = 0; dd < 4; ++dd) x[dd] = *reinterpret_cast<const f32x4*>(ap + C31_PLANE * dd + off);
        } else if (P + 1 < 5) {
#pragma unroll
          for (int dd = 0; dd < 4; ++dd) x[dd] = *reinterpret_cast<const f32x4*>(ap + 2 * C31_PLANE + C31_PLANE * dd);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int k = 0; k < 4; ++k)
            acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(G[k][kw][ch][e], t[k][e >> 1][e & 1], acc[k], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // rows 4 kk + r = channel 16 nt + 4 kk + r; column i = position m -> (row 3 rb + m / 5, column m % 5); depths 2 P, 2 P + 1.
      // Chunked, column-major output [d][chunk = 2 nt + (kk >> 1)][w][h][8]: the lane's position offset (m % 5) * 15 + m / 5 is a
      // per-lane constant (the M = position order needed a byte table per store)
      float* const o = p.out + (((int64_t)u * 10 + 2 * P) * 8 + 2 * nt) * (5 * 15 * 8);   // wave-uniform
      const int olane = (kk >> 1) * (5 * 15 * 8) + 4 * (kk & 1) + ((mi % 5) * 15 + mi / 5 + 3 * rb) * 8;
      constexpr int ostep = 8 * 5 * 15 * 8;   // one output depth further
      f32x2 s0[2], s1[2];
      wino_output(acc, s0, s1);
      f32x4 y0, y1;
      prelu_pairs<SLOPE01>(s0, s1, sl4, y0, y1);
      if (i < 15) {
        *reinterpret_cast<f32x4*>(o + olane) = y0;
        *reinterpret_cast<f32x4*>(o + ostep + olane) = y1;
      }
    }
    __syncthreads();
    item = item_next;
  }
}

}  // namespace

extern "C" int svk_c3d2_stage2(svk_ctx* ctx, const float* d_in, int32_t n_utt, const float* d_w21frag,
                               const float* d_bias21, const float* d_slope21, const float* d_w22frag,
                               const float* d_bias22, const float* d_slope22, int32_t flags, float* d_act2, float* d_out) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, n_utt >= 0, "n_utt negative");
  SVK_REQUIRE(ctx, (flags & ~2) == 0, "flags: only bit 1 (slopes in [0, 1]) is defined");
  if (n_utt == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_in && d_w21frag && d_bias21 && d_slope21 && d_w22frag && d_bias22 && d_slope22 && d_act2 && d_out,
              "NULL buffer");
  SVK_REQUIRE(ctx, ((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_act2) |
                     reinterpret_cast<uintptr_t>(d_w21frag) | reinterpret_cast<uintptr_t>(d_w22frag)) & 15) == 0,
              "buffers must be 16-byte aligned");
  SVK_REQUIRE(ctx, (int64_t)n_utt * 21 < ((int64_t)1 << 31), "too many cubes for one launch");
  // work-item counters of the kernels that share a CU between workgroups (slots of the handle's 256-byte scratch; svk_log_power
  // owns the first word): zeroed in stream order before the launches.  SVK_C3D2_STATIC_ITEMS: items at a fixed stride instead
  // (the determinism test: the same results bit for bit whichever workgroup takes an item)
  const bool static_items = getenv("SVK_C3D2_STATIC_ITEMS") != nullptr;
  unsigned* const queues = static_items ? nullptr : reinterpret_cast<unsigned*>(static_cast<char*>(ctx->scratch) + 64);
  if (queues) SVK_HIP(ctx, hipMemsetAsync(queues, 0, 16, ctx->stream));
  const bool slope01 = (flags & 2) != 0;
  {
    Conv21Params p{d_in, reinterpret_cast<const f32x4*>(d_w21frag), d_bias21, d_slope21, d_act2, n_utt, queues};
    void (*kern)(const Conv21Params) = slope01 ? c3d2_conv21w_kernel<true> : c3d2_conv21w_kernel<false>;
    const size_t lds = sizeof(float) * (size_t)C21W_LDS_FLOATS;
    if (lds > (size_t)ctx->lds_per_cu)
      return svk_fail(ctx, SVK_ERR_UNSUPPORTED, "svk_c3d2_stage2 (conv2_1) needs %zu bytes of LDS per workgroup (device: %d)",
                      lds, ctx->lds_per_cu);
    SVK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t items = (int64_t)n_utt * (S2_H / C21W_TH);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 256, lds) != hipSuccess ||
        per_cu < 1)
      per_cu = 2;
    hipLaunchKernelGGL(kern, dim3((unsigned)std::min<int64_t>(items, (int64_t)per_cu * ctx->num_cu)), dim3(256), lds,
                       ctx->stream, p);
    SVK_LAUNCH_CHECK(ctx);
  }
  {
    Conv22Params p{d_act2, reinterpret_cast<const f32x4*>(d_w22frag), d_bias22, d_slope22, d_out, n_utt, nullptr,
                   queues ? queues + 1 : nullptr};
    void (*kern)(const Conv22Params) = slope01 ? c3d2_conv22w_kernel<true> : c3d2_conv22w_kernel<false>;
    const size_t lds = sizeof(float) * (size_t)C22W_LDS_FLOATS;
    if (lds > (size_t)ctx->lds_per_cu)
      return svk_fail(ctx, SVK_ERR_UNSUPPORTED, "svk_c3d2_stage2 (conv2_2) needs %zu bytes of LDS per workgroup (device: %d)",
                      lds, ctx->lds_per_cu);
    SVK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t items = (int64_t)n_utt * (O2_W * (O2_D / C22W_TD));
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 256, lds) != hipSuccess ||
        per_cu < 1)
      per_cu = 2;
    const unsigned gridw = (unsigned)std::min<int64_t>(items, (int64_t)per_cu * ctx->num_cu);
#ifdef SVK_TUNING
    const bool want_stamps_w = getenv("SVK_C3D2_STAMPS") != nullptr;
    const size_t stamp_bytes_w = (size_t)gridw * 4 * 4 * sizeof(unsigned long long);
    if (want_stamps_w) {
      const int rc = svk_ensure_work(ctx, stamp_bytes_w);
      if (rc != SVK_OK) return rc;
      p.stamps = reinterpret_cast<unsigned long long*>(ctx->work);
    }
#endif
    hipLaunchKernelGGL(kern, dim3(gridw), dim3(256), lds, ctx->stream, p);
    SVK_LAUNCH_CHECK(ctx);
#ifdef SVK_TUNING
    if (want_stamps_w) {
      std::vector<unsigned long long> h((size_t)gridw * 16);
      SVK_HIP(ctx, hipStreamSynchronize(ctx->stream));
      SVK_HIP(ctx, hipMemcpy(h.data(), p.stamps, stamp_bytes_w, hipMemcpyDeviceToHost));
      const char* names[4] = {"staging (+ previous epilogue)", "barrier 1", "MFMA passes + exchange + barrier 2", ""};
      const double per = (double)items / gridw;
      {
        std::vector<double> mhz;
        for (unsigned b = 0; b < gridw; ++b) {
          const unsigned long long c = h[((size_t)b * 4 + 0) * 4 + 3], r = h[((size_t)b * 4 + 1) * 4 + 3];
          if (r) mhz.push_back(100.0 * (double)c / (double)r);
        }
        if (!mhz.empty()) {
          std::sort(mhz.begin(), mhz.end());
          fprintf(stderr, "conv22w in-kernel clock: median %.0f MHz (min %.0f, max %.0f)\n", mhz[mhz.size() / 2], mhz.front(), mhz.back());
        }
      }
      {   // the launch on the chip-wide 100 MHz counter (microseconds from the first workgroup's entry)
        std::vector<double> ent, beg, end;
        for (unsigned b = 0; b < gridw; ++b) {
          const double e = (double)h[((size_t)b * 4 + 2) * 4 + 3], x = (double)h[((size_t)b * 4 + 3) * 4 + 3];
          ent.push_back(e);
          end.push_back(x);
          beg.push_back(x - (double)h[((size_t)b * 4 + 1) * 4 + 3]);
        }
        const double t0 = *std::min_element(ent.begin(), ent.end());
        auto us = [&](std::vector<double>& v, const char* what) {
          std::sort(v.begin(), v.end());
          fprintf(stderr, "conv22w %s: first %.1f  median %.1f  last %.1f us after the first workgroup's entry\n", what, (v.front() - t0) / 100.0,
                  (v[v.size() / 2] - t0) / 100.0, (v.back() - t0) / 100.0);
        };
        us(ent, "kernel entry");
        us(beg, "loop start");
        us(end, "loop end");
      }
      {   // spread over workgroups of the loop's total cycles (wave 0): static item assignment makes the slowest one the kernel's time
        std::vector<double> tot;
        for (unsigned b = 0; b < gridw; ++b) tot.push_back((double)(h[(size_t)b * 16 + 0] + h[(size_t)b * 16 + 1] + h[(size_t)b * 16 + 2]));
        std::sort(tot.begin(), tot.end());
        fprintf(stderr, "conv22w loop cycles per workgroup: min %.0f  median %.0f  p90 %.0f  max %.0f\n", tot.front(), tot[tot.size() / 2],
                tot[tot.size() * 9 / 10], tot.back());
      }
      for (int w = 0; w < 4; ++w) {
        fprintf(stderr, "conv22w stamps wave %d (cycles per item, %d workgroups per CU):", w, per_cu);
        for (int k = 0; k < 3; ++k) {
          double sum = 0;
          for (unsigned b = 0; b < gridw; ++b) sum += (double)h[((size_t)b * 4 + w) * 4 + k];
          fprintf(stderr, "  %s %.0f", names[k], sum / gridw / per);
        }
        fprintf(stderr, "\n");
      }
    }
#endif
  }
  return SVK_OK;
}

extern "C" int svk_c3d2_conv31(svk_ctx* ctx, const float* d_in, int32_t n_utt, const float* d_wfrag, const float* d_bias,
                               const float* d_slope, int32_t flags, float* d_out) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, n_utt >= 0, "n_utt negative");
  SVK_REQUIRE(ctx, (flags & ~2) == 0, "flags: only bit 1 (slopes in [0, 1]) is defined");
  if (n_utt == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_in && d_wfrag && d_bias && d_slope && d_out, "NULL buffer");
  SVK_REQUIRE(ctx, ((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_wfrag)) & 15) == 0,
              "buffers must be 16-byte aligned");
  SVK_REQUIRE(ctx, (int64_t)n_utt * 5 < ((int64_t)1 << 31), "too many cubes for one launch");
  const bool static_items31 = getenv("SVK_C3D2_STATIC_ITEMS") != nullptr;
  unsigned* const queue31 = static_items31 ? nullptr : reinterpret_cast<unsigned*>(static_cast<char*>(ctx->scratch) + 80);
  if (queue31) SVK_HIP(ctx, hipMemsetAsync(queue31, 0, 4, ctx->stream));
  Conv31Params p{d_in, reinterpret_cast<const f32x4*>(d_wfrag), d_bias, d_slope, d_out, n_utt, queue31};
  void (*kern)(const Conv31Params) = (flags & 2) ? c3d2_conv31w_kernel<true> : c3d2_conv31w_kernel<false>;
  const size_t lds = sizeof(float) * (size_t)C31_LDS_FLOATS;
  if (lds > (size_t)ctx->lds_per_cu)
    return svk_fail(ctx, SVK_ERR_UNSUPPORTED, "svk_c3d2_conv31 needs %zu bytes of LDS per workgroup (device: %d)", lds,
                    ctx->lds_per_cu);
  SVK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t items = (int64_t)n_utt * 5;
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 256, lds) != hipSuccess ||
      per_cu < 1)
    per_cu = 2;
  hipLaunchKernelGGL(kern, dim3((unsigned)std::min<int64_t>(items, (int64_t)per_cu * ctx->num_cu)), dim3(256), lds, ctx->stream,
                     p);
  SVK_LAUNCH_CHECK(ctx);
  return SVK_OK;
}
