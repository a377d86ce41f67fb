// The C3D2 embedding network's first three blocks (model.py:110-131, :141-164).
//   c3d2_stage1h_kernel   cube + conv1_1 + conv1_2 + pool1 on v_mfma_f32_16x16x32_f16 through two-piece f16 products (round 4)
//   c3d2_conv21h_kernel   conv2_1, two-piece f16 products
//   c3d2_conv22h_kernel   conv2_2 + pool2, two-piece f16 products
//   c3d2_conv31h_kernel   conv3_1, two-piece f16 products; writes the chunked, column-major layout c3d2_tail_kernel<Conv32T> stages
// (conv3_2, conv4_1, conv4_2 and FC5 live in c3d2_tail.hip.)  BatchNorm (eval mode) is folded into weights and biases by
// the host (model.FusedEmbedder).  Work items come from device-wide counters.  What earlier rounds built and superseded is
// under tools/experiments/ with its measured numbers: the direct-form f32 kernels, the t-plane first block, the K-split conv3_2
// (c3d2_superseded_r3.patch) and the f32 first and second blocks through the depth transform, round 4's 7.12 + 5.09 ms kernels
// (stage1_f32_winograd.patch, stage2_f32_winograd.patch).
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
  const u32x4* w2blk;    // [14 pairs][2][64]: [H_a | H_b], [L_a | L_b]; lane (co = l & 15, kk): ci = 8 (kk & 1) + e, tap a (kk < 2) / b;
                         // pair 13 (one tap): [H_a | H_a], [L_a | 0]
  const float* bias2;
  const float* slope2;
  float* out;
  unsigned* queue;
};

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

// prelu for 0 <= slope <= 1 straight off MFMA accumulators: fmaxf() on a value the compiler cannot prove canonical costs a
// third instruction (v_max x, x in front of the real one) and the product is one v_mul per value; written as vectors it is one
// v_pk_mul_f32 per PAIR + one v_max_f32 per value (12 -> 6 instructions per four values; the same product, the same
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
  h = __builtin_bit_cast(unsigned, hh);
  // l = f16(x - f32(h)) as ONE instruction per value: v_fma_mix reads h as a half and x as a float, multiplies by -1 and rounds the
  // f32 result (exact: x - h has at most 13 significant bits) into one half of the destination -- where the compiler's own code is
  // two v_cvt_f32_f16, a packed subtract and v_cvt_pk_f16_f32.  Bit-identical on 2^22 random pairs incl. denormals, NaN, infinities.
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(v[0]));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(v[1]));
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
        // pair pr: 0 .. 11 = (kd = pr / 4, kh = 2 (pr % 4) | + 1) off a2; 12 = taps (0, 8) | (1, 8) off a3; 13 = the LAST tap (2, 8)
        // alone, as [h | l] in ONE fragment (lanes kk >= 2 read the l planes): [H | H] x [h | l] + [L | 0] x [h | l] are its three
        // piece products in two MFMAs and one read, where [h | -] and [l | -] against [H | 0], [L | 0] were three and two
        auto rd = [&](int pr, int piece) -> u32x4 {
          const char* ad = pr < 12 ? a2 + 1280 * (pr / 4) + 32 * (pr % 4) : pr == 12 ? a3 + 32 * 4
                                   : a3 + 2 * 1280 + 32 * 4 + (kk >= 2 ? 16 * 4 * HPLANE - 16 * 80 : 0);
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
            if (pr + 2 < HPAIRS - 1) bl[(pr + 2) % 3] = rd(pr + 2, 1);
          }
          __builtin_amdgcn_sched_barrier(0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, W2[pr][0]), __builtin_bit_cast(f16x8, bh[pr % 3]), acc, 0, 0, 0);
          if (pr < HPAIRS - 1)
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
constexpr int A2_D = 14, A2_W = 14;                  // conv2_1 output (32 channels), rows = S2_H.  The layer has 15 columns; pool2 drops
                                                     // conv2_2's 15th, which is all that reads conv2_1's 15th (kernel width 1): never computed
constexpr int O2_D = 12, O2_H = 15, O2_W = 7;        // after conv2_2 + pool2 (32 channels)

// ---- conv2_1 through two-piece f16 products (see c3d2_stage1h_kernel): direct form, 12 taps = 6 pairs (kd, kw | kw + 1) of
// K = 32 blocks, three MFMAs per pair and N tile (both N tiles of a wave share the B fragments).  Item = (cube, block of 4
// rows) as before; its input [16 d][4 rows][18 w][16 c] is split into (h, l) while it is staged and lies in LDS as four planes
// of 16-byte slots (h c0-7, h c8-15, l c0-7, l c8-15), slot = pixel (d * 4 + row) * 18 + col: the 16 positions of a tile --
// ANY 16 consecutive outputs of the item's 14 d x 4 rows x 14 columns = 784 = 49 full tiles -- read consecutive slots (+ 4 across a
// row end).  36 MFMAs of 16 cycles per tile where the depth-transformed f32 kernel issued 128 of 32 per 16 positions of a pair. ----
constexpr int C21H_PIX = S2_D * 4 * S2_W;            // 1 152 pixels = slots per plane
constexpr int C21H_LDS_WORDS = 4 * 4 * C21H_PIX;     // four planes of 16-byte slots: 73 728 bytes
constexpr int C21H_POS = A2_D * 4 * A2_W;            // 784 output positions per item

struct Conv21hParams {
  const float* in;      // [n][16][36][18][16]
  const u32x4* wblk;    // [2 nt][6 pairs][2][64]: lane (co = 16 nt + (l & 15), kk): e: W[co][8 (kk & 1) + e][kd][kw + (kk >= 2)], pair = 2 kd + kw / 2; H | L
  const float* bias;    // [32]
  const float* slope;   // [32]
  float* out;           // [n][14][36][14][32]
  int32_t n_utt;
  unsigned* queue;
};

template <bool SLOPE01>
__global__ __launch_bounds__(256, 2) void c3d2_conv21h_kernel(const Conv21hParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem_c21w[];
  unsigned* const reg = reinterpret_cast<unsigned*>(smem_c21w);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, kk = lane >> 4;
  u32x4 W[2][6][2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int pr = 0; pr < 6; ++pr) {
      W[nt][pr][0] = p.wblk[((nt * 6 + pr) * 2) * 64 + lane];
      W[nt][pr][1] = p.wblk[((nt * 6 + pr) * 2 + 1) * 64 + lane];
    }
  f32x4 b4[2], sl4[2];   // a lane holds channels 16 nt + 4 kk .. + 3 of ONE position
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      b4[nt][r] = p.bias[16 * nt + 4 * kk + r];
      sl4[nt][r] = p.slope[16 * nt + 4 * kk + r];
    }
  constexpr int BLOCKS = S2_H / 4;   // 9 row blocks per cube
  const int n_items = p.n_utt * BLOCKS;
  __shared__ int q_next;   // dynamic work items (two workgroups share a CU)
  int item = blockIdx.x;
  while (item < n_items) {
    unsigned q_ticket = 0;
    if (threadIdx.x == 0 && p.queue) q_ticket = atomicAdd(p.queue, 1u);
    const int u = item / BLOCKS, hb = (item - u * BLOCKS) * 4;
    // stage + split: piece e = t + 256 k of thread t = channels 4 (e & 3) .. + 3 of pixel (t >> 2) + 64 k (18 pieces per thread);
    // its four h halves are bytes 8 (piece & 1) .. + 7 of slot `pixel` in plane (piece >> 1), its l halves the same two planes on
    const float* src = p.in + (int64_t)u * (S2_D * S2_H * S2_W * 16);
    {
      int tl = threadIdx.x;
      asm volatile("" : "+v"(tl));   // (keeps this arithmetic inside the item loop)
      const int piece = tl & 3, pix0 = tl >> 2;
      unsigned* const a0s = reg + 4 * ((piece >> 1) * C21H_PIX + pix0) + 2 * (piece & 1);
      const float* const g0 = src + (int64_t)(hb * S2_W + pix0) * 16 + 4 * piece;
      constexpr int PER_D = 4 * S2_W;                    // 72 staged pixels per depth, 648 in the tensor
      constexpr int DSTEP = (S2_H * S2_W - PER_D) * 16;  // floats the source gains per depth on top of 16 pix
      constexpr int NV = 9;
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
        for (int k = r0; k < r0 + NV; ++k) {
          unsigned h0, l0, h1, l1;
          split2(__builtin_shufflevector(sv[k - r0], sv[k - r0], 0, 1), h0, l0);
          split2(__builtin_shufflevector(sv[k - r0], sv[k - r0], 2, 3), h1, l1);
          *reinterpret_cast<u32x2*>(a0s + 4 * 64 * k) = (u32x2){h0, h1};
          *reinterpret_cast<u32x2*>(a0s + 4 * 64 * k + 4 * 2 * C21H_PIX) = (u32x2){l0, l1};
        }
      }
    }
    if (threadIdx.x == 0) q_next = p.queue ? (int)q_ticket + (int)gridDim.x : item + (int)gridDim.x;
    __syncthreads();
    const int item_next = q_next;
    // tiles t = wave + 4 m of 16 positions P = 16 t + i -> (depth P / 56, row (P % 56) / 14, column P % 14); 49 tiles
    static_assert(C21H_POS % 16 == 0 && A2_W == 14, "the position decode below is for 14 columns");
#pragma unroll 1
    for (int t = wave; t < C21H_POS / 16; t += 4) {
      const int P = 16 * t + i;
      const int dq = (P * 1171) >> 16, rem = P - 56 * dq;                       // P / 56 for P < 784
      const int row = (rem * 4682) >> 16, col = rem - 14 * row;                 // rem / 14 for rem < 56
      // input pixel (dq + kd, row, col + kw): slot (dq * 4 + row) * 18 + col + 72 kd + kw of plane (kk & 1) [l: + 2]; tap b = + 1 slot
      const char* const a2 = reinterpret_cast<const char*>(reg) + 16 * ((kk & 1) * C21H_PIX + (dq * 4 + row) * S2_W + col + (kk >= 2 ? 1 : 0));
      auto rd = [&](int pr, int piece) -> u32x4 {
        return *reinterpret_cast<const u32x4*>(a2 + 16 * (72 * (pr >> 1) + 2 * (pr & 1)) + 16 * 2 * C21H_PIX * piece);
      };
      f32x4 acc[2] = {b4[0], b4[1]};
      u32x4 bh[3], bl[3];
      bh[0] = rd(0, 0);
      bl[0] = rd(0, 1);
      bh[1] = rd(1, 0);
      bl[1] = rd(1, 1);
#pragma unroll
      for (int pr = 0; pr < 6; ++pr) {
        if (pr + 2 < 6) {
          bh[(pr + 2) % 3] = rd(pr + 2, 0);
          bl[(pr + 2) % 3] = rd(pr + 2, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, W[nt][pr][0]), __builtin_bit_cast(f16x8, bh[pr % 3]), acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, W[nt][pr][0]), __builtin_bit_cast(f16x8, bl[pr % 3]), acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, W[nt][pr][1]), __builtin_bit_cast(f16x8, bh[pr % 3]), acc[nt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      float* const o = p.out + ((((int64_t)u * A2_D + dq) * S2_H + hb + row) * A2_W + col) * 32 + 4 * kk;
      *reinterpret_cast<f32x4*>(o) = prelu4<SLOPE01>(acc[0], sl4[0]);
      *reinterpret_cast<f32x4*>(o + 16) = prelu4<SLOPE01>(acc[1], sl4[1]);
    }
    __syncthreads();
    item = item_next;
  }
}

// ---- conv2_2 + pool2 through two-piece f16 products: direct form, K = 32 = the 32 input channels of ONE tap, three MFMAs per
// tap (H x h, H x l, L x h), 24 taps.  Item = (cube, pooled column j, third q of the output depths) as before; its input
// [6 d][36 h][2 w][32 c] is split while it is staged: eight planes (four channel quarters x {h, l}) of 16-byte slots, a plane
// split by the parity of the row (rows are 2 apart along a tile): slot ((r & 1) * 6 + d) * 46 + (r >> 1) * 2 + col.  The item's
// 4 d x 15 rows x 2 columns = 120 positions are 7.5 tiles of 16; wave = (N tile nt, every other tile): the 48 weight blocks of an
// N tile are 192 VGPRs.  Pool = max over adjacent lanes (the column pair), the even lane stores four channels. ----
// Depth pitch 46, not 36: a tile's positions run on from one depth's 30 to the next, and with 46 = 30 (mod 16) so do their slots mod 16
// -- the sixteen lanes of an LDS lane group stay on sixteen different 16-byte bank groups in the three of 7.5 tiles that straddle depths
constexpr int C22H_DP = 46;
constexpr int C22H_PLANE = 560;                      // 2 * 6 * 46 = 552 slots per plane, padded to a multiple of 16
constexpr int C22H_LDS_WORDS = 4 * 8 * C22H_PLANE;   // 71 680 bytes
constexpr int C22H_POS = 4 * O2_H * 2;               // 120 positions per item

struct Conv22hParams {
  const float* in;      // [n][14][36][14][32]
  const u32x4* wblk;    // [2 nt][24 taps][2][64]: lane (co = 16 nt + (l & 15), kk): e: W[co][8 kk + e][kd][kh], tap = 8 kd + kh; H | L
  const float* bias;    // [32]
  const float* slope;   // [32]
  float* out;           // [n][12][15][7][32]
  int32_t n_utt;
  unsigned* queue;
};

template <bool SLOPE01>
__global__ __launch_bounds__(256, 2) void c3d2_conv22h_kernel(const Conv22hParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem_c22w[];
  unsigned* const reg = reinterpret_cast<unsigned*>(smem_c22w);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, kk = lane >> 4;
  const int nt = wave & 1, half = wave >> 1;
  u32x4 W[24][2];
#pragma unroll
  for (int t = 0; t < 24; ++t) {
    W[t][0] = p.wblk[((nt * 24 + t) * 2) * 64 + lane];
    W[t][1] = p.wblk[((nt * 24 + t) * 2 + 1) * 64 + lane];
  }
  f32x4 b4, sl4;   // channels 16 nt + 4 kk .. + 3 of ONE position
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    b4[r] = p.bias[16 * nt + 4 * kk + r];
    sl4[r] = p.slope[16 * nt + 4 * kk + r];
  }
  constexpr int PER_CUBE = O2_W * (O2_D / 4);   // 7 x 3 items
  const int n_items = p.n_utt * PER_CUBE;
  __shared__ int q_next;
  int item = blockIdx.x;
  while (item < n_items) {
    unsigned q_ticket = 0;
    if (threadIdx.x == 0 && p.queue) q_ticket = atomicAdd(p.queue, 1u);
    const int u = item / PER_CUBE, rem = item - u * PER_CUBE, q = rem / O2_W, j = rem - q * O2_W;
    // stage + split: piece e = t + 256 k: channels 4 (e & 7) .. + 3 of pixel (dh = (t >> 4) + 16 k, column (e >> 3) & 1), dh = d * 36 + h
    const float* src = p.in + ((int64_t)u * A2_D + 4 * q) * (S2_H * A2_W * 32) + 2 * j * 32;
    {
      int tl = threadIdx.x;
      asm volatile("" : "+v"(tl));
      const int piece = tl & 7, wq = (tl >> 3) & 1, dh0 = tl >> 4;
      const float* const g0 = src + dh0 * (A2_W * 32) + wq * 32 + 4 * piece;
      constexpr int NV = 7;   // (all 14 in flight spill: the 192 weight registers stay live)
#pragma unroll
      for (int r0 = 0; r0 < 14; r0 += NV) {
        f32x4 sv[NV];
#pragma unroll
        for (int k = r0; k < r0 + NV; ++k)
          if (k < 13 || tl < 128) sv[k - r0] = *reinterpret_cast<const f32x4*>(g0 + (int64_t)(16 * k) * (A2_W * 32));
#pragma unroll
        for (int k = r0; k < r0 + NV; ++k) {
          // dh = dh0 + 16 k -> (d, h): 16 k = 36 d_lo + h_lo at compile time, one comparison for the carry
          const int d_lo = (16 * k) / S2_H, h_lo = 16 * k - S2_H * d_lo;
          int hh = dh0 + h_lo, d = d_lo;
          if (h_lo + 15 >= S2_H) {
            const bool carry = hh >= S2_H;
            hh = carry ? hh - S2_H : hh;
            d = carry ? d + 1 : d;
          }
          const int slot = ((hh & 1) * 6 + d) * C22H_DP + (hh >> 1) * 2 + wq;
          unsigned* const dst = reg + 4 * ((piece >> 1) * C22H_PLANE + slot) + 2 * (piece & 1);
          unsigned h0, l0, h1, l1;
          split2(__builtin_shufflevector(sv[k - r0], sv[k - r0], 0, 1), h0, l0);
          split2(__builtin_shufflevector(sv[k - r0], sv[k - r0], 2, 3), h1, l1);
          if (k < 13 || tl < 128) {
            *reinterpret_cast<u32x2*>(dst) = (u32x2){h0, h1};
            *reinterpret_cast<u32x2*>(dst + 4 * 4 * C22H_PLANE) = (u32x2){l0, l1};
          }
        }
      }
    }
    if (threadIdx.x == 0) q_next = p.queue ? (int)q_ticket + (int)gridDim.x : item + (int)gridDim.x;
    __syncthreads();
    const int item_next = q_next;
    // tiles t = half, half + 2, ...: positions P = 16 t + i -> (depth P / 30, row (P % 30) / 2, column P & 1)
#pragma unroll 1
    for (int t = half; t < (C22H_POS + 15) / 16; t += 2) {
      const int P = min(16 * t + i, C22H_POS - 1);
      const int dq = (P * 2185) >> 16, r30 = P - 30 * dq, row = r30 >> 1;      // P / 30 for P < 120
      // input pixel (dq + kd, 2 row + kh, col), channels 8 kk .. + 7: slot ((kh & 1) * 6 + dq + kd) * 46 + (row + kh / 2) * 2 + col of plane kk [l: + 4]
      const char* const a2 = reinterpret_cast<const char*>(reg) + 16 * (kk * C22H_PLANE + dq * C22H_DP + 2 * row + (i & 1));
      auto rd = [&](int tap, int piece) -> u32x4 {
        const int kd = tap >> 3, kh = tap & 7;
        return *reinterpret_cast<const u32x4*>(a2 + 16 * (((kh & 1) * 6 + kd) * C22H_DP + (kh >> 1) * 2) + 16 * 4 * C22H_PLANE * piece);
      };
      f32x4 acc = b4;
      u32x4 bh[3], bl[3];
      bh[0] = rd(0, 0);
      bl[0] = rd(0, 1);
      bh[1] = rd(1, 0);
      bl[1] = rd(1, 1);
#pragma unroll
      for (int tap = 0; tap < 24; ++tap) {
        if (tap + 2 < 24) {
          bh[(tap + 2) % 3] = rd(tap + 2, 0);
          bl[(tap + 2) % 3] = rd(tap + 2, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, W[tap][0]), __builtin_bit_cast(f16x8, bh[tap % 3]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, W[tap][0]), __builtin_bit_cast(f16x8, bl[tap % 3]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, W[tap][1]), __builtin_bit_cast(f16x8, bh[tap % 3]), acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      const f32x4 y = prelu4<SLOPE01>(acc, sl4);
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = max_with_lane_xor1(y[r]);
      if ((i & 1) == 0 && 16 * t + i < C22H_POS)
        *reinterpret_cast<f32x4*>(p.out + ((((int64_t)u * O2_D + 4 * q + dq) * O2_H + row) * O2_W + j) * 32 + 16 * nt + 4 * kk) = o;
    }
    __syncthreads();
    item = item_next;
  }
}

// ---- conv3_1 (32 -> 64, kernel (3,1,3)) + BN + PReLU (model.py:126-128, :159-161) through two-piece f16 products: direct
// form, K = 32 = the 32 input channels of ONE tap, three MFMAs per tap, 9 taps.  No taps along h, so item = (cube, block of 3
// rows): region 12 depths x 3 rows x 7 columns of 32 channels, split into (h, l) while staged: eight planes (four channel quarters
// x {h, l}) of 16-byte slots, slot = pixel (d * 3 + row) * 7 + col (32 KB; three workgroups per CU).  The item's 10 d x 3 rows x
// 5 columns = 150 positions are 9.4 tiles of 16 consecutive positions; wave = N tile (16 of the 64 output channels: 18 weight
// blocks = 72 VGPRs), every wave walks all ten tiles.  Output chunked and column-major, 16-byte stores. ----
constexpr int C31H_PLANE = 256;                               // 12 * 3 * 7 = 252 slots per plane, padded to a multiple of 16: the lanes of an LDS
                                                              // lane group sit in different planes (kk) and must not land on each other's slots
constexpr int C31H_LDS_WORDS = 4 * 8 * C31H_PLANE;            // 32 768 bytes
constexpr int C31H_POS = 10 * 3 * 5;                          // 150 positions per item

struct Conv31Params {
  const float* in;      // [n][12][15][7][32]
  const u32x4* wblk;    // [4 nt][9 taps][2][64]: lane (co = 16 nt + (l & 15), kk): e: W[co][8 kk + e][kd][kw], tap = 3 kd + kw; H | L
  const float* bias;    // [64]
  const float* slope;   // [64]
  float* out;           // chunked and column-major: [n][10][8 chunks][5 w][15 h][8] (what svk_c3d2_conv32t stages)
  int32_t n_utt;
  unsigned* queue;      // work-item counter (zeroed before the launch), or NULL
};

template <bool SLOPE01>
__global__ __launch_bounds__(256, 3) void c3d2_conv31h_kernel(const Conv31Params p) {
  extern __shared__ __attribute__((aligned(16))) float smem_c31[];
  unsigned* const reg = reinterpret_cast<unsigned*>(smem_c31);
  const int lane = threadIdx.x & 63, nt = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, kk = lane >> 4;
  u32x4 W[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    W[t][0] = p.wblk[((nt * 9 + t) * 2) * 64 + lane];
    W[t][1] = p.wblk[((nt * 9 + t) * 2 + 1) * 64 + lane];
  }
  f32x4 b4, sl4;   // channels 16 nt + 4 kk .. + 3 of ONE position
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    b4[r] = p.bias[16 * nt + 4 * kk + r];
    sl4[r] = p.slope[16 * nt + 4 * kk + r];
  }
  const int n_items = p.n_utt * 5;
  __shared__ int q_next;   // dynamic work items: see c3d2_conv21h_kernel (three workgroups share a CU here)
  int item = blockIdx.x;
  while (item < n_items) {
    unsigned q_ticket = 0;
    if (threadIdx.x == 0 && p.queue) q_ticket = atomicAdd(p.queue, 1u);
    const int u = item / 5, rb = item - u * 5;
    // stage + split [12 d][3 rows][7 w][32 c]: per depth 672 contiguous floats = 168 sixteen-byte pieces; 2 016 in all, eight per
    // thread.  piece e = t + 256 k = channels 4 (e & 7) .. + 3 of pixel e >> 3 (168 d is a multiple of 8: the pixel index has no d
    // in it and is linear in k); the source is 4 e + 2 688 d floats, and for a compile-time k the depth d = e / 168 is a constant
    // plus at most two comparisons of t with the window's boundaries
    const float* src = p.in + ((int64_t)u * 12 * 15 + 3 * rb) * (7 * 32);
    {
      int tl = threadIdx.x;
      asm volatile("" : "+v"(tl));   // (keeps this arithmetic inside the item loop)
      const int c4 = tl & 7;
      unsigned* const a0s = reg + 4 * ((c4 >> 1) * C31H_PLANE + (tl >> 3)) + 2 * (c4 & 1);
      const float* const g0 = src + 4 * tl;
      constexpr int DSTEP = 15 * 7 * 32 - 4 * 168;   // floats the source gains per depth on top of 4 e
      f32x4 sv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int d_lo = (256 * k) / 168, c1 = 168 * (d_lo + 1) - 256 * k, c2 = c1 + 168;   // t >= c1 (c2): one (two) depths on
        const float* g = g0 + 1024 * k + DSTEP * d_lo;
        if (c1 < 256) g = tl >= c1 ? g + DSTEP : g;
        if (c2 < 256) g = tl >= c2 ? g + DSTEP : g;
        if (256 * k + 255 < 2016 || tl < 2016 - 256 * k) sv[k] = *reinterpret_cast<const f32x4*>(g);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        unsigned h0, l0, h1, l1;
        split2(__builtin_shufflevector(sv[k], sv[k], 0, 1), h0, l0);
        split2(__builtin_shufflevector(sv[k], sv[k], 2, 3), h1, l1);
        if (256 * k + 255 < 2016 || tl < 2016 - 256 * k) {
          *reinterpret_cast<u32x2*>(a0s + 4 * 32 * k) = (u32x2){h0, h1};
          *reinterpret_cast<u32x2*>(a0s + 4 * 32 * k + 4 * 4 * C31H_PLANE) = (u32x2){l0, l1};
        }
      }
    }
    if (threadIdx.x == 0) q_next = p.queue ? (int)q_ticket + (int)gridDim.x : item + (int)gridDim.x;
    __syncthreads();
    const int item_next = q_next;
#pragma unroll 1
    for (int t = 0; t < (C31H_POS + 15) / 16; ++t) {
      const int P = min(16 * t + i, C31H_POS - 1);
      const int dq = (P * 4370) >> 16, r15 = P - 15 * dq;                       // P / 15 for P < 150
      const int row = (r15 * 13108) >> 16, col = r15 - 5 * row;                 // r15 / 5 for r15 < 15
      // input pixel (dq + kd, row, col + kw), channels 8 kk .. + 7: slot (dq * 3 + row) * 7 + col + 21 kd + kw of plane kk [l: + 4]
      const char* const a2 = reinterpret_cast<const char*>(reg) + 16 * (kk * C31H_PLANE + (dq * 3 + row) * 7 + col);
      auto rd = [&](int tap, int piece) -> u32x4 {
        return *reinterpret_cast<const u32x4*>(a2 + 16 * (21 * (tap / 3) + tap % 3) + 16 * 4 * C31H_PLANE * piece);
      };
      f32x4 acc = b4;
      u32x4 bh[3], bl[3];
      bh[0] = rd(0, 0);
      bl[0] = rd(0, 1);
      bh[1] = rd(1, 0);
      bl[1] = rd(1, 1);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap + 2 < 9) {
          bh[(tap + 2) % 3] = rd(tap + 2, 0);
          bl[(tap + 2) % 3] = rd(tap + 2, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, W[tap][0]), __builtin_bit_cast(f16x8, bh[tap % 3]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, W[tap][0]), __builtin_bit_cast(f16x8, bl[tap % 3]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, W[tap][1]), __builtin_bit_cast(f16x8, bh[tap % 3]), acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // chunked, column-major output [d][chunk = 2 nt + (kk >> 1)][w][h][8]: the lane's four channels are 16 contiguous bytes
      if (16 * t + i < C31H_POS) {
        float* const o = p.out + ((((int64_t)u * 10 + dq) * 8 + 2 * nt + (kk >> 1)) * 5 + col) * (15 * 8) + (3 * rb + row) * 8 + 4 * (kk & 1);
        *reinterpret_cast<f32x4*>(o) = prelu4<SLOPE01>(acc, sl4);
      }
    }
    __syncthreads();
    item = item_next;
  }
}

// ---- conv3_2 (64 -> 64, kernel (3,7,1)) + BN + PReLU (model.py:129-131, :162-164) through two-piece f16 products, direct form:
// 21 taps x two K = 32 blocks (channels 0-31 | 32-63) x three MFMAs.  No taps along w, and conv3_1 writes its output chunked and
// COLUMN-major, so item = (cube, column): 10 d x 15 h x 64 c = eighty 480-byte runs, split into (h, l) while staged: sixteen planes
// (eight channel chunks x {h, l}) of 16-byte slots, slot = d * 15 + h (38 KB).  Outputs 8 d x 9 h = 72 positions = 4.5 tiles.  The
// weight blocks of an N tile and ONE K block are 168 VGPRs: wave = (N tile nt, K block kb), eight waves; the two waves of an N tile
// swap partial sums through LDS (kb 0 finishes tiles 0 - 2, kb 1 tiles 3 - 4).  The next item's runs are loaded into registers in
// front of the tiles and parked behind them: one workgroup per CU, its memory latency under its own matrix work.
// (As an instance of the f32 batch-GEMM template of c3d2_tail.hip, Winograd F(2,3) along depth: 1.34 - 1.62 ms per 4 018 cubes.) ----
// depth pitch 25, not 15: a tile's positions run on from one depth's 9 rows to the next, and with 25 = 9 (mod 16) so do their slots
// mod 16 (see C22H_DP)
constexpr int C32H_DP = 25;
constexpr int C32H_PLANE = 256;                               // 10 * 25 = 250 slots per plane, padded to a multiple of 16 (see C31H_PLANE)
constexpr int C32H_LDS_WORDS = 4 * 16 * C32H_PLANE;           // 65 536 bytes
constexpr int C32H_XCH_FLOATS = 8 * 5 * 64 * 4;               // [wave = nt + 4 kb][tile][lane] f32x4: every wave's partial sums
constexpr int C32H_POS = 8 * 9;                               // 72 positions per item

struct Conv32hParams {
  const float* in;      // [n][10][8 chunks][5 w][15 h][8]
  const u32x4* wblk;    // [4 nt][2 kb][21 taps][2][64]: lane (co = 16 nt + (l & 15), kk): e: W[co][32 kb + 8 kk + e][kd][kh], tap = 7 kd + kh; H | L
  const float* bias;    // [64]
  const float* slope;   // [64]
  float* out;           // [n][8 d][8 chunks][45 = 9 h x 5 w][8]
  int32_t n_utt;
  unsigned* queue;
};

template <bool SLOPE01>
__global__ __launch_bounds__(512) void c3d2_conv32h_kernel(const Conv32hParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem_c32[];
  unsigned* const reg = reinterpret_cast<unsigned*>(smem_c32);
  float* const xch = smem_c32 + C32H_LDS_WORDS;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, kk = lane >> 4;
  const int nt = wave & 3, kb = wave >> 2;
  u32x4 W[21][2];
#pragma unroll
  for (int t = 0; t < 21; ++t) {
    W[t][0] = p.wblk[(((nt * 2 + kb) * 21 + t) * 2) * 64 + lane];
    W[t][1] = p.wblk[(((nt * 2 + kb) * 21 + t) * 2 + 1) * 64 + lane];
  }
  f32x4 b4, sl4;   // channels 16 nt + 4 kk .. + 3 of ONE position (the bias rides in K block 0's accumulators)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    b4[r] = kb == 0 ? p.bias[16 * nt + 4 * kk + r] : 0.f;
    sl4[r] = p.slope[16 * nt + 4 * kk + r];
  }
  const int n_items = p.n_utt * 5;
  __shared__ int q_next;
  // piece e = t + 512 k (five per thread, 2 400 in all): run e / 30 = d * 8 + chunk, h = (e % 30) / 2, channels 4 (e & 1) .. + 3 of the chunk
  f32x4 sv[5];
  auto load_item = [&](int it) {
    const int u = it / 5, w = it - 5 * u;
    const float* const src = p.in + (int64_t)u * (10 * 8 * 5 * 120) + w * 120;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int e = (int)threadIdx.x + 512 * k;
      const int run = (e * 2185) >> 16, r = e - 30 * run;       // e / 30 for e < 2 400
      if (k < 4 || threadIdx.x < 2400 - 4 * 512) sv[k] = *reinterpret_cast<const f32x4*>(src + run * 600 + 4 * r);
    }
  };
  auto park_item = [&]() {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int e = (int)threadIdx.x + 512 * k;
      const int run = (e * 2185) >> 16, r = e - 30 * run;
      const int d = run >> 3, chunk = run & 7;
      unsigned* const dst = reg + 4 * (chunk * C32H_PLANE + d * C32H_DP + (r >> 1)) + 2 * (r & 1);
      unsigned h0, l0, h1, l1;
      split2(__builtin_shufflevector(sv[k], sv[k], 0, 1), h0, l0);
      split2(__builtin_shufflevector(sv[k], sv[k], 2, 3), h1, l1);
      if (k < 4 || threadIdx.x < 2400 - 4 * 512) {
        *reinterpret_cast<u32x2*>(dst) = (u32x2){h0, h1};
        *reinterpret_cast<u32x2*>(dst + 4 * 8 * C32H_PLANE) = (u32x2){l0, l1};
      }
    }
  };
  // items: the first two of a workgroup at a fixed stride, every later one drawn from the device-wide counter ONE item ahead (the
  // next item's runs are fetched during this item; its ticket's round trip runs under this item's tiles)
  int item = blockIdx.x, item_next = item + (int)gridDim.x;
  if (item < n_items) {
    load_item(item);
    park_item();
  }
  __syncthreads();
  while (item < n_items) {
    unsigned q_ticket = 0;
    if (threadIdx.x == 0 && p.queue) q_ticket = atomicAdd(p.queue, 1u);
    const int u = item / 5, w = item - 5 * u;
    if (item_next < n_items) load_item(item_next);
#pragma unroll 1
    for (int t = 0; t < 5; ++t) {
      const int P = min(16 * t + i, C32H_POS - 1);
      const int dq = (P * 7282) >> 16, row = P - 9 * dq;                        // P / 9 for P < 72
      // input pixel (dq + kd, row + kh), channels 32 kb + 8 kk .. + 7: slot (dq + kd) * 25 + row + kh of plane 4 kb + kk [l: + 8]
      const char* const a2 = reinterpret_cast<const char*>(reg) + 16 * ((4 * kb + kk) * C32H_PLANE + dq * C32H_DP + row);
      auto rd = [&](int tap, int piece) -> u32x4 {
        return *reinterpret_cast<const u32x4*>(a2 + 16 * (C32H_DP * (tap / 7) + tap % 7) + 16 * 8 * C32H_PLANE * piece);
      };
      f32x4 a = b4;
      u32x4 bh[2], bl[2];   // (one tap ahead: a second set ahead is eight registers more)
      bh[0] = rd(0, 0);
      bl[0] = rd(0, 1);
#pragma unroll
      for (int tap = 0; tap < 21; ++tap) {
        if (tap + 1 < 21) {
          bh[(tap + 1) & 1] = rd(tap + 1, 0);
          bl[(tap + 1) & 1] = rd(tap + 1, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, W[tap][0]), __builtin_bit_cast(f16x8, bh[tap & 1]), a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, W[tap][0]), __builtin_bit_cast(f16x8, bl[tap & 1]), a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, W[tap][1]), __builtin_bit_cast(f16x8, bh[tap & 1]), a, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // partial sums over this wave's K block -> LDS (kept in registers for the tiles a wave finishes they cost 12 VGPRs this kernel
      // does not have: the 168 weight registers and the 20 of the next item's runs leave room for one accumulator)
      *reinterpret_cast<f32x4*>(xch + ((wave * 5 + t) * 64 + lane) * 4) = a;
    }
    __syncthreads();   // both K blocks' partial sums are in LDS; nobody reads the planes any more
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const bool mine = kb == 0 ? t < 3 : t >= 3;   // wave-uniform: K block 0's wave finishes tiles 0 - 2, K block 1's tiles 3 - 4
      if (mine) {
        const int P = 16 * t + i;
        const int Pc = min(P, C32H_POS - 1);
        const int dq = (Pc * 7282) >> 16, row = Pc - 9 * dq;
        const f32x4 v = *reinterpret_cast<const f32x4*>(xch + ((nt * 5 + t) * 64 + lane) * 4) +
                        *reinterpret_cast<const f32x4*>(xch + (((nt + 4) * 5 + t) * 64 + lane) * 4);
        if (P < C32H_POS) {
          float* const o = p.out + ((((int64_t)u * 8 + dq) * 8 + 2 * nt + (kk >> 1)) * 45 + row * 5 + w) * 8 + 4 * (kk & 1);
          *reinterpret_cast<f32x4*>(o) = prelu4<SLOPE01>(v, sl4);
        }
      }
    }
    if (item_next < n_items) park_item();
    if (threadIdx.x == 0) q_next = p.queue ? (int)q_ticket + 2 * (int)gridDim.x : item_next + (int)gridDim.x;
    __syncthreads();   // the next item's planes are written; the exchange buffer is free
    item = item_next;
    item_next = q_next;
  }
}

// ---- conv4_1 (64 -> 128, kernel (3,1,3)) + BN + PReLU (model.py:132-135, :165-166) through two-piece f16 products, direct form:
// 9 taps x two K = 32 blocks x three MFMAs.  Item = ONE CUBE: its whole input [8 d][8 chunks][45 = 9 h x 5 w][8] (92 KB) is split
// while staged into sixteen planes (eight channel chunks x {h, l}) of 16-byte slots, slot = d * 59 + 9 w + h; outputs 6 d x 9 h x 3 w =
// 162 positions = 10.1 tiles; wave = N tile (eight waves: 128 output channels; 36 weight blocks = 144 VGPRs), every wave walks all
// eleven tiles.  The next cube is loaded into registers in front of the tiles and parked behind them.
// (As an instance of the f32 batch-GEMM template of c3d2_tail.hip, Winograd F(2,3) along depth: 0.54 - 0.69 ms per 4 018 cubes.) ----
// A plane holds [8 d][5 w][9 h] at depth pitch 59: positions are walked (depth, column, row) with the row fastest, 27 per depth, and
// 59 = 27 (mod 16), so sixteen consecutive positions are sixteen consecutive slots mod 16 at every tap (see C22H_DP)
constexpr int C41H_DP = 59;
constexpr int C41H_PLANE = 480;                               // 8 * 59 = 472 slots per plane, padded to a multiple of 16
constexpr int C41H_LDS_WORDS = 4 * 16 * C41H_PLANE;           // 122 880 bytes
constexpr int C41H_POS = 6 * 9 * 3;                           // 162 positions per cube
constexpr int C41H_PIECES = 8 * 8 * 45 * 2;                   // 5 760 sixteen-byte pieces per cube

struct Conv41hParams {
  const float* in;      // [n][8][8 chunks][45][8]
  const u32x4* wblk;    // [8 nt][9 taps][2 kb][2][64]: lane (co = 16 nt + (l & 15), kk): e: W[co][32 kb + 8 kk + e][kd][kw], tap = 3 kd + kw; H | L
  const float* bias;    // [128]
  const float* slope;   // [128]
  float* out;           // [n][6 d][16 chunks][27 = 9 h x 3 w][8]
  int32_t n_utt;
  unsigned* queue;
};

template <bool SLOPE01>
__global__ __launch_bounds__(512) void c3d2_conv41h_kernel(const Conv41hParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem_c41[];
  unsigned* const reg = reinterpret_cast<unsigned*>(smem_c41);
  const int lane = threadIdx.x & 63, nt = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, kk = lane >> 4;
  u32x4 W[9][2][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      W[t][kb][0] = p.wblk[(((nt * 9 + t) * 2 + kb) * 2) * 64 + lane];
      W[t][kb][1] = p.wblk[(((nt * 9 + t) * 2 + kb) * 2 + 1) * 64 + lane];
    }
  f32x4 b4, sl4;   // channels 16 nt + 4 kk .. + 3 of ONE position
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    b4[r] = p.bias[16 * nt + 4 * kk + r];
    sl4[r] = p.slope[16 * nt + 4 * kk + r];
  }
  const int n_items = p.n_utt;
  __shared__ int q_next;
  // piece e = t + 512 k (twelve per thread, 5 760 in all): run e / 90 = d * 8 + chunk, pixel (e % 90) / 2, channels 4 (e & 1) .. + 3
  constexpr int NPC = (C41H_PIECES + 511) / 512;
  f32x4 sv[NPC];
  auto load_item = [&](int it) {
    const float* const src = p.in + (int64_t)it * (8 * 8 * 45 * 8) + 4 * (int)threadIdx.x;
#pragma unroll
    for (int k = 0; k < NPC; ++k)
      if (k < NPC - 1 || threadIdx.x < C41H_PIECES - (NPC - 1) * 512) sv[k] = *reinterpret_cast<const f32x4*>(src + 2048 * k);
  };
  auto park_item = [&]() {
#pragma unroll
    for (int k = 0; k < NPC; ++k) {
      const int e = (int)threadIdx.x + 512 * k;
      const int run = (e * 46604) >> 22, r = e - 90 * run;        // e / 90 for e < 5 760
      const int d = run >> 3, chunk = run & 7;
      const int pix = r >> 1, ph = (pix * 13) >> 6, pw = pix - 5 * ph;       // pixel = 5 h + w; pix / 5 for pix < 45
      unsigned* const dst = reg + 4 * (chunk * C41H_PLANE + d * C41H_DP + 9 * pw + ph) + 2 * (r & 1);
      unsigned h0, l0, h1, l1;
      split2(__builtin_shufflevector(sv[k], sv[k], 0, 1), h0, l0);
      split2(__builtin_shufflevector(sv[k], sv[k], 2, 3), h1, l1);
      if (k < NPC - 1 || threadIdx.x < C41H_PIECES - (NPC - 1) * 512) {
        *reinterpret_cast<u32x2*>(dst) = (u32x2){h0, h1};
        *reinterpret_cast<u32x2*>(dst + 4 * 8 * C41H_PLANE) = (u32x2){l0, l1};
      }
    }
  };
  int item = blockIdx.x, item_next = item + (int)gridDim.x;   // the counter is drawn one item ahead, as in c3d2_conv32h_kernel
  if (item < n_items) {
    load_item(item);
    park_item();
  }
  __syncthreads();
  while (item < n_items) {
    unsigned q_ticket = 0;
    if (threadIdx.x == 0 && p.queue) q_ticket = atomicAdd(p.queue, 1u);
    if (item_next < n_items) load_item(item_next);
#pragma unroll 1
    for (int t = 0; t < (C41H_POS + 15) / 16; ++t) {
      const int P = min(16 * t + i, C41H_POS - 1);
      const int dq = (P * 2428) >> 16, r27 = P - 27 * dq;                       // P / 27 for P < 162;  r27 = 9 wq + h
      const int wq = (r27 * 7282) >> 16, h = r27 - 9 * wq;                      // r27 / 9 for r27 < 27
      // input pixel (dq + kd, h, wq + kw), channels 32 kb + 8 kk .. + 7: slot (dq + kd) * 59 + 9 (wq + kw) + h of plane 4 kb + kk [l: + 8]
      const char* const a2 = reinterpret_cast<const char*>(reg) + 16 * (kk * C41H_PLANE + dq * C41H_DP + r27);
      auto rd = [&](int st, int piece) -> u32x4 {   // step st = 2 tap + kb
        const int tap = st >> 1, kb = st & 1;
        return *reinterpret_cast<const u32x4*>(a2 + 16 * (C41H_DP * (tap / 3) + 9 * (tap % 3)) + 16 * 4 * C41H_PLANE * kb + 16 * 8 * C41H_PLANE * piece);
      };
      f32x4 a = b4;
      u32x4 bh[2], bl[2];
      bh[0] = rd(0, 0);
      bl[0] = rd(0, 1);
#pragma unroll
      for (int st = 0; st < 18; ++st) {
        if (st + 1 < 18) {
          bh[(st + 1) & 1] = rd(st + 1, 0);
          bl[(st + 1) & 1] = rd(st + 1, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, W[st >> 1][st & 1][0]), __builtin_bit_cast(f16x8, bh[st & 1]), a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, W[st >> 1][st & 1][0]), __builtin_bit_cast(f16x8, bl[st & 1]), a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, W[st >> 1][st & 1][1]), __builtin_bit_cast(f16x8, bh[st & 1]), a, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (16 * t + i < C41H_POS) {
        float* const o = p.out + ((((int64_t)item * 6 + dq) * 16 + 2 * nt + (kk >> 1)) * 27 + 3 * h + wq) * 8 + 4 * (kk & 1);
        *reinterpret_cast<f32x4*>(o) = prelu4<SLOPE01>(a, sl4);
      }
    }
    __syncthreads();   // nobody reads the planes any more
    if (item_next < n_items) park_item();
    if (threadIdx.x == 0) q_next = p.queue ? (int)q_ticket + 2 * (int)gridDim.x : item_next + (int)gridDim.x;
    __syncthreads();   // the next cube's planes are written
    item = item_next;
    item_next = q_next;
  }
}

}  // namespace

extern "C" int svk_c3d2_stage2(svk_ctx* ctx, const float* d_in, int32_t n_utt, const void* d_w21blk,
                               const float* d_bias21, const float* d_slope21, const void* d_w22blk,
                               const float* d_bias22, const float* d_slope22, int32_t flags, float* d_act2, float* d_out) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, n_utt >= 0, "n_utt negative");
  SVK_REQUIRE(ctx, (flags & ~2) == 0, "flags: only bit 1 (slopes in [0, 1]) is defined");
  if (n_utt == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_in && d_w21blk && d_bias21 && d_slope21 && d_w22blk && d_bias22 && d_slope22 && d_act2 && d_out,
              "NULL buffer");
  SVK_REQUIRE(ctx, ((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_act2) |
                     reinterpret_cast<uintptr_t>(d_w21blk) | reinterpret_cast<uintptr_t>(d_w22blk) | reinterpret_cast<uintptr_t>(d_out)) & 15) == 0,
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
    Conv21hParams p{d_in, reinterpret_cast<const u32x4*>(d_w21blk), d_bias21, d_slope21, d_act2, n_utt, queues};
    void (*kern)(const Conv21hParams) = slope01 ? c3d2_conv21h_kernel<true> : c3d2_conv21h_kernel<false>;
    const size_t lds = sizeof(unsigned) * (size_t)C21H_LDS_WORDS;
    if (lds > (size_t)ctx->lds_per_cu)
      return svk_fail(ctx, SVK_ERR_UNSUPPORTED, "svk_c3d2_stage2 (conv2_1) needs %zu bytes of LDS per workgroup (device: %d)",
                      lds, ctx->lds_per_cu);
    SVK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t items = (int64_t)n_utt * (S2_H / 4);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 256, lds) != hipSuccess ||
        per_cu < 1)
      per_cu = 2;
    hipLaunchKernelGGL(kern, dim3((unsigned)std::min<int64_t>(items, (int64_t)per_cu * ctx->num_cu)), dim3(256), lds,
                       ctx->stream, p);
    SVK_LAUNCH_CHECK(ctx);
  }
  {
    Conv22hParams p{d_act2, reinterpret_cast<const u32x4*>(d_w22blk), d_bias22, d_slope22, d_out, n_utt, queues ? queues + 1 : nullptr};
    void (*kern)(const Conv22hParams) = slope01 ? c3d2_conv22h_kernel<true> : c3d2_conv22h_kernel<false>;
    const size_t lds = sizeof(unsigned) * (size_t)C22H_LDS_WORDS;
    SVK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t items = (int64_t)n_utt * 21;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 256, lds) != hipSuccess ||
        per_cu < 1)
      per_cu = 2;
    hipLaunchKernelGGL(kern, dim3((unsigned)std::min<int64_t>(items, (int64_t)per_cu * ctx->num_cu)), dim3(256), lds,
                       ctx->stream, p);
    SVK_LAUNCH_CHECK(ctx);
  }
  return SVK_OK;
}

extern "C" int svk_c3d2_conv31(svk_ctx* ctx, const float* d_in, int32_t n_utt, const void* d_wblk, const float* d_bias,
                               const float* d_slope, int32_t flags, float* d_out) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, n_utt >= 0, "n_utt negative");
  SVK_REQUIRE(ctx, (flags & ~2) == 0, "flags: only bit 1 (slopes in [0, 1]) is defined");
  if (n_utt == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_in && d_wblk && d_bias && d_slope && d_out, "NULL buffer");
  SVK_REQUIRE(ctx, ((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_wblk) | reinterpret_cast<uintptr_t>(d_out)) & 15) == 0,
              "buffers must be 16-byte aligned");
  SVK_REQUIRE(ctx, (int64_t)n_utt * 5 < ((int64_t)1 << 31), "too many cubes for one launch");
  const bool static_items31 = getenv("SVK_C3D2_STATIC_ITEMS") != nullptr;
  unsigned* const queue31 = static_items31 ? nullptr : reinterpret_cast<unsigned*>(static_cast<char*>(ctx->scratch) + 80);
  if (queue31) SVK_HIP(ctx, hipMemsetAsync(queue31, 0, 4, ctx->stream));
  Conv31Params p{d_in, reinterpret_cast<const u32x4*>(d_wblk), d_bias, d_slope, d_out, n_utt, queue31};
  void (*kern)(const Conv31Params) = (flags & 2) ? c3d2_conv31h_kernel<true> : c3d2_conv31h_kernel<false>;
  const size_t lds = sizeof(unsigned) * (size_t)C31H_LDS_WORDS;
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

extern "C" int svk_c3d2_conv32t(svk_ctx* ctx, const float* d_in, int32_t n_utt, const void* d_wblk, const float* d_bias,
                                const float* d_slope, int32_t flags, float* d_out) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, n_utt >= 0, "n_utt negative");
  SVK_REQUIRE(ctx, (flags & ~2) == 0, "flags: only bit 1 (slopes in [0, 1]) is defined");
  if (n_utt == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_in && d_wblk && d_bias && d_slope && d_out, "NULL buffer");
  SVK_REQUIRE(ctx, ((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_wblk) | reinterpret_cast<uintptr_t>(d_out)) & 15) == 0,
              "buffers must be 16-byte aligned");
  SVK_REQUIRE(ctx, (int64_t)n_utt * 5 + 2 * (int64_t)ctx->num_cu < ((int64_t)1 << 31), "too many cubes for one launch");
  unsigned* const queue = getenv("SVK_C3D2_STATIC_ITEMS") ? nullptr : reinterpret_cast<unsigned*>(static_cast<char*>(ctx->scratch) + 96);
  if (queue) SVK_HIP(ctx, hipMemsetAsync(queue, 0, 4, ctx->stream));
  Conv32hParams p{d_in, reinterpret_cast<const u32x4*>(d_wblk), d_bias, d_slope, d_out, n_utt, queue};
  void (*kern)(const Conv32hParams) = (flags & 2) ? c3d2_conv32h_kernel<true> : c3d2_conv32h_kernel<false>;
  const size_t lds = sizeof(float) * (size_t)(C32H_LDS_WORDS + C32H_XCH_FLOATS);
  if (lds + 64 > (size_t)ctx->lds_per_cu)
    return svk_fail(ctx, SVK_ERR_UNSUPPORTED, "svk_c3d2_conv32t needs %zu bytes of LDS per workgroup (device: %d)", lds, ctx->lds_per_cu);
  SVK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t items = (int64_t)n_utt * 5;
  hipLaunchKernelGGL(kern, dim3((unsigned)std::min<int64_t>(items, ctx->num_cu)), dim3(512), lds, ctx->stream, p);
  SVK_LAUNCH_CHECK(ctx);
  return SVK_OK;
}

extern "C" int svk_c3d2_conv41(svk_ctx* ctx, const float* d_in, int32_t n_utt, const void* d_wblk, const float* d_bias,
                               const float* d_slope, int32_t flags, float* d_out) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, n_utt >= 0, "n_utt negative");
  SVK_REQUIRE(ctx, (flags & ~2) == 0, "flags: only bit 1 (slopes in [0, 1]) is defined");
  if (n_utt == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_in && d_wblk && d_bias && d_slope && d_out, "NULL buffer");
  SVK_REQUIRE(ctx, ((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_wblk) | reinterpret_cast<uintptr_t>(d_out)) & 15) == 0,
              "buffers must be 16-byte aligned");
  SVK_REQUIRE(ctx, (int64_t)n_utt + 2 * (int64_t)ctx->num_cu < ((int64_t)1 << 31), "too many cubes for one launch");
  unsigned* const queue = getenv("SVK_C3D2_STATIC_ITEMS") ? nullptr : reinterpret_cast<unsigned*>(static_cast<char*>(ctx->scratch) + 100);
  if (queue) SVK_HIP(ctx, hipMemsetAsync(queue, 0, 4, ctx->stream));
  Conv41hParams p{d_in, reinterpret_cast<const u32x4*>(d_wblk), d_bias, d_slope, d_out, n_utt, queue};
  void (*kern)(const Conv41hParams) = (flags & 2) ? c3d2_conv41h_kernel<true> : c3d2_conv41h_kernel<false>;
  const size_t lds = sizeof(unsigned) * (size_t)C41H_LDS_WORDS;
  if (lds + 64 > (size_t)ctx->lds_per_cu)
    return svk_fail(ctx, SVK_ERR_UNSUPPORTED, "svk_c3d2_conv41 needs %zu bytes of LDS per workgroup (device: %d)", lds, ctx->lds_per_cu);
  SVK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)std::min<int64_t>(n_utt, ctx->num_cu)), dim3(512), lds, ctx->stream, p);
  SVK_LAUNCH_CHECK(ctx);
  return SVK_OK;
}
