// The C3D2 embedding network's first three blocks (model.py:110-131, :141-164) on v_mfma_f32_16x16x4_f32.
//   c3d2_stage1_kernel            cube + conv1_1 + conv1_2 + pool1, direct form (described right below)
//   c3d2_stage1w_kernel<., MERGE> the same with conv1_2 through Winograd's F(2, 3) along depth; MERGE = the round-3 default
//                                 (the 4-row remainders of two depth pairs share one M tile)
//   c3d2_stage1t_kernel           experiment: input transform applied once at conv1_1's output (measured slower; DESIGN appendix)
//   c3d2_conv21_kernel / conv21w  conv2_1, direct / depth-transformed
//   c3d2_conv22_kernel / conv22w  conv2_2 + pool2, direct / depth-transformed
//   c3d2_conv31w_kernel           conv3_1, depth-transformed
//   c3d2_conv32w_kernel           conv3_2, depth-transformed, K split over the waves of a workgroup (round 2; the default is
//                                 c3d2_tail_kernel<Conv32T> of c3d2_tail.hip, where conv4_1, conv4_2 and FC5 live too)
//   bias_prelu_kernel             + bias, PReLU behind a convolution the host framework ran (the PyTorch-ROCm A/B path only)
// BatchNorm (eval mode) is folded into weights and biases by the host (model.FusedEmbedder).  The kernels that share a CU
// between workgroups (conv21w, conv22w, conv31w) draw their work items from a device-wide counter.
//
// The first block of the C3D2 embedding network as ONE gfx950 kernel:
//   feature rows + crop starts -> cube (utils.py:351-379) -> conv1_1 (1 -> 16, k(3,1,5)) + BN + PReLU
//   -> conv1_2 (16 -> 16, k(3,9,1), stride (1,2,1)) + BN + PReLU -> MaxPool3d((1,1,2))
// (/root/reference/model.py:110-117 and :141-150, eval-mode BatchNorm folded into the convolutions by the
// host).  These two layers are 46 % of the network's multiply-adds, and around them PyTorch-ROCm moved the
// network's largest tensor (conv1_1's output, 3.3 MB per cube = 3.2 GB per micro-batch) through HBM four
// times.  Here that tensor only ever exists as a 106 KB tile in LDS.
//
// Work item = (cube u, pooled output column j, half q of the output depths): conv1_2 outputs
//   d in [8q, 8q + 8), h in [0, 36), w in {2j, 2j + 1}  ->  pooled column j, 16 channels.
// A persistent workgroup of 4 waves (one per SIMD; it owns the CU's LDS) loops over items:
//   1. the 12 x 80 x 6 cube patch the item needs is fetched into registers while the previous item's
//      matrix work runs, then parked in LDS (23 KB) -- by LDS-DMA in the depth-transformed kernel below;
//   2. conv1_1 as a GEMM on v_mfma_f32_16x16x4_f32: [16 pixels] x [K = 15 taps + 1 (bias)] x [16 channels],
//      A gathered from the patch, result + PReLU written to the act1 tile in LDS:
//      10 depths x 80 rows x 2 columns x 16 channels;
//   3. conv1_2 as an implicit GEMM on the same instruction: M tile = 2 depths x 4 rows x 2 columns, N = 16
//      channels, K = 27 taps x 16 channels.  The whole weight matrix lives in 108 VGPRs per wave (B operand);
//      the A operand of tap (kd, kh) is ONE ds_read_b128 per lane at a compile-time offset from the tile's
//      base address, so the loop body is 1 LDS read per 4 MFMAs and nothing else;
//   4. bias, PReLU, max over the column pair (the two columns of a pooling window are adjacent rows of the
//      accumulator tile: no lane movement), store.
// K is permuted identically on both operands (lane (i, kk) holds channels 4 kk .. 4 kk + 3 of a 16-channel
// chunk, MFMA step e uses element e), so fragments are plain 16-byte accesses.
// act1 addressing: pixel p = (depth * 80 + row) * 2 + column lives at float 16 p + 4 (p >> 2): the 4 extra
// floats per 4 pixels spread the 16 pixels of an M tile (strides of 2 rows = 4 pixels and of 1 depth = 160
// pixels) over all 64 banks; unpadded they would share two 64-byte windows (8-way conflicts).
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
constexpr int PD = TD + 4, PW = 6;                   // cube patch: depths, columns
constexpr int OD = 16, OH = 36, OWP = 18;            // output: depths, rows, pooled columns
constexpr int ACT_FLOATS = 17 * DIN * NFRAME * 2;    // 16 p + 4 (p >> 2), p < DIN * 80 * 2
constexpr int P_FLOATS = PD * NFRAME * PW;
constexpr int N_TAPS = 27;                           // conv1_2: 3 depths x 9 rows

struct Stage1Params {
  const float* feat;
  const int32_t* crop;
  int32_t n_utt, max_frames;
  const float* w1frag;   // [4][64]: B operand of the conv1_1 GEMM, k = 4 jj + (lane >> 4): tap (k / 5, k % 5), k = 15: zero
  const float* bias1;    // [16]
  const float* slope1;   // [16]
  const f32x4* w2frag;   // [27][64]: lane (co = l & 15, kk = l >> 4), element e = W[co][4 kk + e][kd][kh], tap = 9 kd + kh
  const float* bias2;    // [16]
  const float* slope2;   // [16]
  float* out;
  int64_t s_n, s_d, s_hp, s_par, s_w;  // output strides (floats): cube, depth, row pair, row parity, pooled column
  unsigned long long* stamps;          // tuning builds only (-DSVK_TUNING): [grid][4 waves][6] summed phase cycles
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

// The item's cube patch: patch[dd][h][ww] = feat[u][crop[u][8 q + dd] + h][2 j + ww].  Thread t < 240 owns the float2
// (row h = t / 3, piece t % 3) of EVERY depth dd: nothing to decode per item, the depth's crop start is wave-uniform
// (scalar loads, one item ahead so that no feature load waits for it inside the matrix work), and the LDS address is
// a per-thread constant plus an immediate.
// (a VECTOR load by lanes 0 .. 11, not twelve scalar loads: scalar loads return out of order, so while any is in
// flight every LDS wait of the wave becomes lgkmcnt(0) -- the first gather read of the conv1_1 phase then stalled for
// the crop table's whole L2 round trip, 2 500 cycles per item by the in-kernel stamps)
__device__ __forceinline__ int fetch_starts(const Stage1Params& p, int item, int lane) {
  const int u = item / 36, rem = item - u * 36, q = rem / 18;
  const int32_t* cr = p.crop + (int64_t)u * NCROP + TD * q;
  return cr[lane < PD ? lane : 0];
}

__device__ __forceinline__ void fetch_patch(const Stage1Params& p, int item, int starts_v, int h, int piece,
                                            f32x2 (&regs)[PD]) {
  const int u = item / 36, rem = item - u * 36, j = rem % 18;
  const float* base = p.feat + (int64_t)u * p.max_frames * NCOEF + 2 * j + 2 * piece;
#pragma unroll
  for (int dd = 0; dd < PD; ++dd) {
    f32x2 v = (f32x2){0.f, 0.f};
    const int start = __builtin_amdgcn_readlane(starts_v, dd);   // wave-uniform
    if (h < NFRAME && (unsigned)start < (unsigned)p.max_frames && h < p.max_frames - start)   // (cannot overflow for any int32 start)
      v = *reinterpret_cast<const f32x2*>(base + (start + h) * NCOEF);  // < 2^31 floats per clip
    regs[dd] = v;
  }
}

__device__ __forceinline__ void park_patch(float* patch, int h, int piece, const f32x2 (&regs)[PD]) {
  if (h < NFRAME) {
    float* dst = patch + h * PW + 2 * piece;
#pragma unroll
    for (int dd = 0; dd < PD; ++dd) *reinterpret_cast<f32x2*>(dst + dd * (NFRAME * PW)) = regs[dd];
  }
}

template <bool SLOPE01>
__global__ __launch_bounds__(256) void c3d2_stage1_kernel(const Stage1Params p) {
  extern __shared__ __attribute__((aligned(16))) float smem_c3d2[];
  float* act = smem_c3d2;               // [ACT_FLOATS]
  float* patch = act + ACT_FLOATS;      // [P_FLOATS]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kk = lane >> 4;
  const int n_items = p.n_utt * 36;

  // ---- per-kernel constants in registers ----
  f32x4 w2[N_TAPS];
#pragma unroll
  for (int t = 0; t < N_TAPS; ++t) w2[t] = p.w2frag[t * 64 + lane];
  float w1[4];
  int tapoff[4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    w1[jj] = p.w1frag[jj * 64 + lane];
    const int k = 4 * jj + kk;
    tapoff[jj] = k < 15 ? (k / 5) * (NFRAME * PW) + (k % 5) : 0;
  }
  // (GEMM row k = 15 is padding: its weight is 0 and its A operand whatever patch[.. + 0] holds; the bias of
  // conv1_1 enters through the accumulators)
  const float sl1 = p.slope1[i], b1 = p.bias1[i], b2 = p.bias2[i], sl2 = p.slope2[i];
  const int pix_lane = (i >> 1) * PW + (i & 1);  // patch offset of this lane's pixel inside a conv1_1 tile (8 rows x 2 columns)

  f32x2 pre[PD];
  int starts = 0;  // lane dd < 12 holds the crop start of patch depth dd
  const int ph = threadIdx.x / 3, ppiece = threadIdx.x - 3 * ph;   // this thread's patch row (>= 80: idle) and float2 piece
  int item = blockIdx.x;
  if (item < n_items) {
    starts = fetch_starts(p, item, lane);
    fetch_patch(p, item, starts, ph, ppiece, pre);
    park_patch(patch, ph, ppiece, pre);
    if (item + (int)gridDim.x < n_items) starts = fetch_starts(p, item + gridDim.x, lane);
  }
  __syncthreads();
#ifdef SVK_TUNING
  unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0};
#endif
  for (; item < n_items; item += gridDim.x) {
    SVK_STAMP(ts0);
    const int next = item + gridDim.x;
    if (next < n_items) {
      fetch_patch(p, next, starts, ph, ppiece, pre);                            // in flight during the matrix work below
      if (next + (int)gridDim.x < n_items) starts = fetch_starts(p, next + gridDim.x, lane);  // ... and the starts of the one after
    }
    __builtin_amdgcn_sched_barrier(0);  // all of those loads are ISSUED here, not trickled into the MFMA stream
    SVK_STAMP(ts1);

    // ---- conv1_1 + PReLU: 100 tiles of 16 pixels (8 rows x 2 columns of one depth), 25 per wave, five at a
    // time: 20 gather reads in flight, then 20 MFMAs on five independent accumulators (a tile on its own is 4
    // DEPENDENT MFMAs behind one LDS round trip) ----
    // Tile tt covers pixels 16 tt .. 16 tt + 15: its patch offset (480 (tt / 10) + 48 (tt % 10) = 48 tt) and its act1
    // offset (17 x 16 tt) are LINEAR in tt, so with tt = wave + 4 m the per-lane addresses are computed once and every
    // tile is an immediate offset from them (the / 10 form cost ~25 VALU instructions per tile: issued with no MFMA
    // to hide behind, they were a tenth of the kernel).
    {
      const float* pl[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) pl[jj] = patch + 48 * wave + pix_lane + tapoff[jj];
      float* const al = act + 272 * wave + 68 * kk + i;
      // (measured, in-kernel stamps: this phase takes 7 650 cycles per item for 3 200 cycles of MFMA work -- the rest is
      // its 5 VALU per output value (accumulator read-back, PReLU as compare / multiply / select) and the LDS writes;
      // issuing the next group's reads ahead and interleaving the previous group's PReLU + writes with the MFMAs by
      // sched_group_barrier changed nothing: the VALU stream itself is the length of the phase)
#pragma unroll
      for (int g5 = 0; g5 < 5; ++g5) {
        float av[5][4];
#pragma unroll
        for (int q5 = 0; q5 < 5; ++q5) {
          const int m = 5 * g5 + q5;  // tile tt = wave + 4 m
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) av[q5][jj] = pl[jj][192 * m];
        }
        f32x4 acc1[5];  // column i of the tile = channel i: the accumulators start at its (BN-folded) bias
#pragma unroll
        for (int q5 = 0; q5 < 5; ++q5) acc1[q5] = (f32x4){b1, b1, b1, b1};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int q5 = 0; q5 < 5; ++q5) acc1[q5] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q5][jj], w1[jj], acc1[q5], 0, 0, 0);
#pragma unroll
        for (int q5 = 0; q5 < 5; ++q5) {
          // rows 4 kk + r of the tile = pixels 16 tt + 4 kk + r, column i = channel: float 17 x 16 tt + 68 kk + 16 r + i
          float* ap = al + 1088 * (5 * g5 + q5);
#pragma unroll
          for (int r = 0; r < 4; ++r) ap[16 * r] = prelu_t<SLOPE01>(acc1[q5][r], sl1);
        }
      }
    }
    SVK_STAMP(ts2);
    __syncthreads();  // act1 is complete; the patch buffer is free
    SVK_STAMP(ts3);

    // ---- conv1_2 on MFMA: this wave owns the depth pair dp = wave (output depths 2 dp, 2 dp + 1 of the item) ----
    const int u = item / 36, rem = item - u * 36, q = rem / 18, j = rem - q * 18;
    const int dl = i >> 3, hl = (i >> 1) & 3, wc = i & 1;
    const int din0 = 2 * wave + dl;
    float* const obase = p.out + (int64_t)u * p.s_n + (int64_t)(TD * q + 2 * wave + (kk >> 1)) * p.s_d + (int64_t)j * p.s_w + i;
    for (int g = 0; g < 3; ++g) {
      const float* ab[3];
      f32x4 acc[3];
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int hg = 3 * g + s;
        const int pix = (din0 * NFRAME + 8 * hg + 2 * hl) * 2 + wc;
        ab[s] = act + 16 * pix + 4 * (din0 * 40 + 4 * hg + hl) + 4 * kk;
        acc[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      // tap t's fragments are read while tap t - 1's twelve MFMAs run (the scheduler, left alone, issues the reads
      // two MFMAs before their first use: an LDS round trip exposed per tap)
      f32x4 a[3];
#pragma unroll
      for (int s = 0; s < 3; ++s) a[s] = *reinterpret_cast<const f32x4*>(ab[s]);
#pragma unroll
      for (int t = 0; t < N_TAPS; ++t) {
        f32x4 an[3];
        if (t + 1 < N_TAPS) {
          const int kd = (t + 1) / 9, kh = (t + 1) - kd * 9;
          const int off = 2720 * kd + 32 * kh + 4 * (kh >> 1);  // 16 dp + 4 d(p >> 2) for dp = 160 kd + 2 kh pixels
#pragma unroll
          for (int s = 0; s < 3; ++s) an[s] = *reinterpret_cast<const f32x4*>(ab[s] + off);
        }
        __builtin_amdgcn_sched_barrier(0);  // the next tap's three LDS reads stay in front of this tap's MFMAs
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int s = 0; s < 3; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][e], w2[t][e], acc[s], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < N_TAPS) {
#pragma unroll
          for (int s = 0; s < 3; ++s) a[s] = an[s];
        }
      }
      // rows 4 kk + r: depth dl' = kk >> 1, row hl' = 2 (kk & 1) + (r >> 1), column r & 1: pool = max over r pairs
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int hp = 2 * (3 * g + s) + (kk & 1);  // output row pair: rows 4 hg + 2 (kk & 1) + {0, 1}
        const float v0 = fmaxf(prelu_t<SLOPE01>(acc[s][0] + b2, sl2), prelu_t<SLOPE01>(acc[s][1] + b2, sl2));
        const float v1 = fmaxf(prelu_t<SLOPE01>(acc[s][2] + b2, sl2), prelu_t<SLOPE01>(acc[s][3] + b2, sl2));
        float* o = obase + (int64_t)hp * p.s_hp;
        o[0] = v0;
        o[p.s_par] = v1;
      }
    }
    SVK_STAMP(ts4);
    if (next < n_items) park_patch(patch, ph, ppiece, pre);
    SVK_STAMP(ts5);
    __syncthreads();  // the next patch is in place; act1 may be overwritten
    SVK_STAMP(ts6);
    SVK_STAMP_ADD(0, ts0, ts1);  // issue of the next patch's loads
    SVK_STAMP_ADD(1, ts1, ts2);  // conv1_1 phase
    SVK_STAMP_ADD(2, ts2, ts3);  // barrier 1
    SVK_STAMP_ADD(3, ts3, ts4);  // conv1_2 phase + epilogue
    SVK_STAMP_ADD(4, ts4, ts5);  // park
    SVK_STAMP_ADD(5, ts5, ts6);  // barrier 2
  }
#ifdef SVK_TUNING
  if (p.stamps && lane == 0)
    for (int k = 0; k < 6; ++k) p.stamps[((size_t)blockIdx.x * 4 + wave) * 6 + k] = stamp_acc[k];
#endif
}


// -----------------------------------------------------------------------------------------------------
// The same block with conv1_2 through Winograd's F(2, 3) ALONG DEPTH (every C3D2 kernel is 3 deep, stride 1):
// for an output depth pair (2 P, 2 P + 1) and act1 depths x0 .. x3 = 2 P .. 2 P + 3,
//     t0 = x0 - x2,  t1 = x1 + x2,  t2 = x2 - x1,  t3 = x1 - x3,
//     G0 = g0,  G1 = (g0 + g1 + g2) / 2,  G2 = (g0 - g1 + g2) / 2,  G3 = g2      (g = the three depth taps of a row tap),
//     a_k = sum over (row tap, channel) of t_k G_k,        y(2 P) = a0 + a1 + a2,   y(2 P + 1) = a1 - a2 - a3:
// four MFMAs where the direct form issues six.  The transform of the A operand is VALU work per fragment (16 adds
// per 16 MFMAs), and a wave does not overlap its own VALU with its own MFMAs on this chip (measured: every add between
// two MFMAs of a single resident wave costs its full issue time and more) -- so this variant runs EIGHT waves per
// workgroup, two per SIMD, each with its own M tiles: one wave's adds, LDS waits and epilogues run under the other's
// MFMAs (the conv1_1 phase, VALU-bound in the four-wave kernel, gains the same way).
//   * act1 pixel p at 16 p + 4 (p >> 2) + 16 (p >> 4): an M tile is 8 output rows x 2 columns of one depth, its 16
//     pixels 4 apart; 4 hl + 16 ((hl + s) >> 2) + 16 wc (mod 64) are 16 different multiples of 4, one conflict-free
//     ds_read_b128 per quarter wave.  The lane part depends on the row tap only through s = kh >> 1: five bases.
//   * wave = (pair P = wave & 3, part = wave >> 2): part 0 owns the tiles at rows 0, 8, 16, part 1 those at 24 and 28
//     (the last repeats rows 28 .. 31 and stores 32 .. 35): five tiles of 144 MFMAs per SIMD and item.
//   * the weights G (36 fragments = 144 VGPRs) are derived in the prologue from the same 27 fragments the direct kernel
//     takes: the C-ABI does not change.
// -----------------------------------------------------------------------------------------------------
constexpr int WPIXF = 18;                                  // average floats per act1 pixel in this layout
constexpr int WACT_FLOATS = WPIXF * DIN * NFRAME * 2;
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
__device__ __forceinline__ void dma_patch_w(const Stage1Params& p, int item, int starts_v, int pair, int lane, float* patch) {
  const int u = item / 36, rem = item - u * 36, j = rem % 18;
  const float* base = p.feat + (int64_t)u * p.max_frames * NCOEF + 2 * j;
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
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
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

// One (merged tile, k) unit of the MERGE variant below: accumulator a_k of the tile made of rows 32 .. 35 of the depth pairs
// 2 m and 2 m + 1 (lanes 0 .. 7 / 8 .. 15), k a compile-time constant: t_k needs two of the four depth planes -- two
// ds_read_b128, two packed adds and four MFMAs per row tap.
template <int K>
__device__ __forceinline__ f32x4 stage1w_merged_unit(const float* const (&pbm)[5], const f32x4 (&G)[36]) {
  constexpr int DA = K == 0 ? 0 : K == 2 ? 2 : 1, DB = K == 0 ? 2 : K == 1 ? 2 : K == 2 ? 1 : 3;   // t_K = x[DA] -/+ x[DB]
  f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;   // two chains: a dependent MFMA would wait 40 cycles for its predecessor
  f32x4 xa = *reinterpret_cast<const f32x4*>(pbm[0] + 160 * WPIXF * DA), xb = *reinterpret_cast<const f32x4*>(pbm[0] + 160 * WPIXF * DB);
#pragma unroll
  for (int kh = 0; kh < 9; ++kh) {
    const f32x2 alo = __builtin_shufflevector(xa, xa, 0, 1), ahi = __builtin_shufflevector(xa, xa, 2, 3);
    const f32x2 blo = __builtin_shufflevector(xb, xb, 0, 1), bhi = __builtin_shufflevector(xb, xb, 2, 3);
    const f32x2 tlo = K == 1 ? pk_add(alo, blo) : pk_sub(alo, blo), thi = K == 1 ? pk_add(ahi, bhi) : pk_sub(ahi, bhi);
    __builtin_amdgcn_sched_barrier(0);
    if (kh + 1 < 9) {
      const int off = 32 * (kh + 1) + 4 * ((kh + 1) >> 1);
      xa = *reinterpret_cast<const f32x4*>(pbm[(kh + 1) >> 1] + 160 * WPIXF * DA + off);
      xb = *reinterpret_cast<const f32x4*>(pbm[(kh + 1) >> 1] + 160 * WPIXF * DB + off);
    }
    __builtin_amdgcn_sched_barrier(0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(tlo[0], G[9 * K + kh][0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(tlo[1], G[9 * K + kh][1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(thi[0], G[9 * K + kh][2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(thi[1], G[9 * K + kh][3], acc1, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  return acc0 + acc1;
}

// MERGE (round 3, the default): 36 output rows are four 8-row tiles + 4 rows.  The round-2 kernel covered the remainder
// with a fifth tile that repeated rows 28 .. 31 (40 rows issued for every 36: 720 MFMAs per SIMD and item); here the
// remainders of two depth pairs make ONE tile (lanes 0 .. 7: pair 2 m, lanes 8 .. 15: pair 2 m + 1), and the two merged
// tiles of an item are cut by accumulator into eight (tile m, k) units of 36 MFMAs, one per wave: every wave issues two
// full tiles + one unit = 324 MFMAs (648 per SIMD, - 10 %), the units' accumulators meet in 8 KB of LDS and waves 0 and 4
// finish the two tiles behind the item's last barrier.  The sums are those of the round-2 kernel, in the same order.
template <bool SLOPE01, bool MERGE>
__global__ __launch_bounds__(512) void c3d2_stage1w_kernel(const Stage1Params p) {
  extern __shared__ __attribute__((aligned(16))) float smem_c3d2[];
  float* act = smem_c3d2;               // [WACT_FLOATS]
  float* patch = act + WACT_FLOATS;     // [WP_FLOATS]: [12 dd][80 h][8]
  float* const exch = patch + WP_FLOATS; // MERGE: [2 m][4 k][64 lanes] f32x4
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave: a scalar
  const int i = lane & 15, kk = lane >> 4;
  const int pair = wave & 3, part = wave >> 2;
  const int n_items = p.n_utt * 36;

  f32x4 G[36];   // [k][kh]
#pragma unroll
  for (int kh = 0; kh < 9; ++kh) {
    const f32x4 g0 = p.w2frag[kh * 64 + lane], g1 = p.w2frag[(9 + kh) * 64 + lane], g2 = p.w2frag[(18 + kh) * 64 + lane];
    G[kh] = g0;
    G[9 + kh] = 0.5f * ((g0 + g2) + g1);
    G[18 + kh] = 0.5f * ((g0 + g2) - g1);
    G[27 + kh] = g2;
  }
  float w1[4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) w1[jj] = p.w1frag[jj * 64 + lane];
  const float sl1 = p.slope1[i], b1 = p.bias1[i], b2 = p.bias2[i], sl2 = p.slope2[i];

  int starts = 0;
  int item = blockIdx.x;
  if (item < n_items) {
    starts = fetch_starts(p, item, lane);
    if (part == 0) dma_patch_w(p, item, starts, pair, lane, patch);
    if (item + (int)gridDim.x < n_items) starts = fetch_starts(p, item + gridDim.x, lane);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA has landed (the compiler emits this wait too; spelt out: the barrier relies on it)
  __syncthreads();
#ifdef SVK_TUNING
  unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0};
  // the clock the chip holds under this kernel's load: shader cycles (s_memtime) per 100 MHz tick (s_memrealtime) over the loop
  const unsigned long long clk_c0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (; item < n_items; item += gridDim.x) {
    SVK_STAMP(ts0);
    const int next = item + gridDim.x;
    SVK_STAMP(ts1);

    // ---- conv1_1 + PReLU: 100 tiles of 16 pixels, tile tt = wave + 8 m: 13 for waves 0 .. 3, 12 for the others, four at
    // a time (patch offset 64 tt and act1 offset 288 tt are linear in tt: per-lane bases + immediates) ----
    {
      const float* pl[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {   // (recomputed per item: this kernel has no registers to spare)
        // tap k = 4 jj + kk = (kd, kw) = (k / 5, k % 5) without the division: kk + (4 jj mod 5) wraps at most once
        const int t5 = kk + (4 * jj) % 5, wrap = t5 >= 5 ? 1 : 0;
        const int kw = t5 - 5 * wrap, kd = (4 * jj) / 5 + wrap;
        const bool pad = jj == 3 && kk == 3;                // k = 15: the zero row of the weights, any finite operand
        const int col = (i & 1) + (pad ? 0 : kw);           // patch column 0 .. 5 at float col + (col >= 3) of the 8-float row
        pl[jj] = patch + 8 * WPW * wave + (i >> 1) * WPW + col + (col >= 3 ? 1 : 0) + (pad ? 0 : kd * (NFRAME * WPW));
      }
      float* const al = act + 16 * WPIXF * wave + 68 * kk + i;
#pragma unroll
      for (int g4 = 0; g4 < 3; ++g4) {
        float av[4][4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) av[q4][jj] = pl[jj][64 * WPW * (4 * g4 + q4)];
        f32x4 acc1[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) acc1[q4] = (f32x4){b1, b1, b1, b1};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) acc1[q4] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q4][jj], w1[jj], acc1[q4], 0, 0, 0);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          float* ap = al + 128 * WPIXF * (4 * g4 + q4);
#pragma unroll
          for (int r = 0; r < 4; ++r) ap[16 * r] = prelu_t<SLOPE01>(acc1[q4][r], sl1);
        }
      }
      if (part == 0) {   // tile 96 + wave
        float av[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) av[jj] = pl[jj][64 * WPW * 12];
        f32x4 acc1 = (f32x4){b1, b1, b1, b1};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[jj], w1[jj], acc1, 0, 0, 0);
        float* ap = al + 128 * WPIXF * 12;
#pragma unroll
        for (int r = 0; r < 4; ++r) ap[16 * r] = prelu_t<SLOPE01>(acc1[r], sl1);
      }
    }
    SVK_STAMP(ts2);
    __syncthreads();  // act1 is complete; the patch buffer is free
    SVK_STAMP(ts3);

    // ---- conv1_2, depth-transformed ----
    {
      const int u = item / 36, rem = item - u * 36, q = rem / 18, j = rem - q * 18;
      const int hl = i >> 1, wc = i & 1;
      const float* const wbase = act + 2 * (160 * WPIXF) * pair + 68 * hl + 16 * wc + 4 * kk;
      // (wave-uniform 64-bit bases + one 32-bit lane offset: the stores need no per-store address VALU)
      float* const obase = p.out + (int64_t)u * p.s_n + (int64_t)(TD * q + 2 * pair) * p.s_d + (int64_t)j * p.s_w;
      const int olane = kk * (int)p.s_hp + i;
      // (the SIMD's two waves do not share its issue slots evenly -- the older one, part 0, gets about two in three, and
      // s_setprio changes nothing, measured -- but the younger one fills what the older leaves: the phase lasts the SUM
      // of both waves' MFMA + VALU time whichever way the five tiles are split, so 3 + 2 it is)
      // MERGE: full tiles at rows 16 part, 16 part + 8 (tl = 2 part, 2 part + 1)
      const int tl0 = MERGE ? 2 * part : (part ? 3 : 0), tl1 = MERGE ? 2 * part + 2 : (part ? 5 : 3);
#pragma unroll 1
      for (int tl = tl0; tl < tl1; ++tl) {
        // the next item's patch (LDS-DMA), by the older waves in front of their first tile (behind barrier 1: conv1_1 has
        // read the patch buffer); one piece every other row tap instead of one burst: measured 2.7 % SLOWER
        if (part == 0 && tl == tl0 && next < n_items) {
          dma_patch_w(p, next, starts, pair, lane, patch);
          if (next + (int)gridDim.x < n_items) starts = fetch_starts(p, next + gridDim.x, lane);
        }
        const int h0 = tl < 4 ? 8 * tl : 28;
        const float* pb[5];
#pragma unroll
        for (int sft = 0; sft < 5; ++sft) pb[sft] = wbase + 72 * h0 + 16 * ((hl + sft) >> 2);
        f32x4 acc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 x[4];
        f32x2 t[4][2];   // [k][element pair]
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) x[dd] = *reinterpret_cast<const f32x4*>(pb[0] + 160 * WPIXF * dd);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) wino_input_pair(x, hf, t);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) x[dd] = *reinterpret_cast<const f32x4*>(pb[0] + 160 * WPIXF * dd + 32);
        // Row tap kh: 16 MFMAs on t, then the NEXT tap's t in place (x = the next tap's fragments, read a tap earlier):
        // eight packed adds in ONE burst per tap.  (f32 MFMA and f32 VALU share the SIMD's multipliers on this chip --
        // neither the same wave nor the SIMD's other wave overlaps the two, measured -- so every add is paid for, and
        // every switch from MFMAs to adds and back costs ~10 cycles on top: packed adds, few bursts.)
#pragma unroll
        for (int kh = 0; kh < 9; ++kh) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int k = 0; k < 4; ++k)
              acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(t[k][e >> 1][e & 1], G[9 * k + kh][e], acc[k], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (kh + 1 < 9) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) wino_input_pair(x, hf, t);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (kh + 2 < 9) {
            const int off = 32 * (kh + 2) + 4 * ((kh + 2) >> 1);
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) x[dd] = *reinterpret_cast<const f32x4*>(pb[(kh + 2) >> 1] + 160 * WPIXF * dd + off);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        // rows 4 kk + r of the tile: output row h0 + 2 kk + (r >> 1), column r & 1: pool = max over r pairs
        if (MERGE || tl < 4 || kk >= 2) {
          // (packed, written out: the compiler emits scalar subtractions for the differences)
          f32x4 y0, y1;
          const f32x2 b22 = (f32x2){b2, b2};
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const f32x2 c0 = hf ? __builtin_shufflevector(acc[0], acc[0], 2, 3) : __builtin_shufflevector(acc[0], acc[0], 0, 1);
            const f32x2 c1 = hf ? __builtin_shufflevector(acc[1], acc[1], 2, 3) : __builtin_shufflevector(acc[1], acc[1], 0, 1);
            const f32x2 c2 = hf ? __builtin_shufflevector(acc[2], acc[2], 2, 3) : __builtin_shufflevector(acc[2], acc[2], 0, 1);
            const f32x2 c3 = hf ? __builtin_shufflevector(acc[3], acc[3], 2, 3) : __builtin_shufflevector(acc[3], acc[3], 0, 1);
            const f32x2 s0 = pk_add(pk_add(pk_add(c0, c1), c2), b22), s1 = pk_add(pk_sub(pk_sub(c1, c2), c3), b22);
            y0[2 * hf] = s0[0];
            y0[2 * hf + 1] = s0[1];
            y1[2 * hf] = s1[0];
            y1[2 * hf + 1] = s1[1];
          }
          float* const o00 = obase + (int64_t)(h0 / 2) * p.s_hp;
          float* const o01 = o00 + p.s_par;
          float* const o10 = o00 + p.s_d;
          float* const o11 = o10 + p.s_par;
          o00[olane] = fmaxf(prelu_t<SLOPE01>(y0[0], sl2), prelu_t<SLOPE01>(y0[1], sl2));
          o01[olane] = fmaxf(prelu_t<SLOPE01>(y0[2], sl2), prelu_t<SLOPE01>(y0[3], sl2));
          o10[olane] = fmaxf(prelu_t<SLOPE01>(y1[0], sl2), prelu_t<SLOPE01>(y1[1], sl2));
          o11[olane] = fmaxf(prelu_t<SLOPE01>(y1[2], sl2), prelu_t<SLOPE01>(y1[3], sl2));
        }
      }
    }
    if (MERGE) {
      // this wave's (merged tile m = part, k = pair) unit
      int lane_m = lane;
      asm volatile("" : "+v"(lane_m));   // (as in the finish block below)
      const int i_m = lane_m & 15;
      const int hlm = (i_m >> 1) & 3, wcm = i_m & 1;
      const float* const mbase = act + 2 * (160 * WPIXF) * (2 * part + (i_m >> 3)) + 68 * hlm + 16 * wcm + 4 * (lane_m >> 4) + 72 * 32;
      const float* pbm[5];
#pragma unroll
      for (int sft = 0; sft < 5; ++sft) pbm[sft] = mbase + 16 * ((hlm + sft) >> 2);
      f32x4 au;
      if (pair == 0) au = stage1w_merged_unit<0>(pbm, G);
      else if (pair == 1) au = stage1w_merged_unit<1>(pbm, G);
      else if (pair == 2) au = stage1w_merged_unit<2>(pbm, G);
      else au = stage1w_merged_unit<3>(pbm, G);
      *reinterpret_cast<f32x4*>(exch + ((part * 4 + pair) * 64 + lane) * 4) = au;
    }
    SVK_STAMP(ts4);
    SVK_STAMP(ts5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces have landed (as above)
    __syncthreads();  // the next patch is in place; act1 may be overwritten
    SVK_STAMP(ts6);
    if (MERGE && pair == 0) {
      // waves 0 and 4 finish merged tile m = part: rows 4 kk + r = (pair 2 m + (kk >> 1), output row 32 + 2 (kk & 1) + (r >> 1),
      // column r & 1); the exchange buffer is written again behind the next item's first barrier
      const int u = item / 36, rem = item - u * 36, q = rem / 18, j = rem - q * 18;
      int lane_f = lane;
      asm volatile("" : "+v"(lane_f));   // (keeps the address arithmetic below INSIDE the loop: hoisted, it costs registers this kernel spills)
      const int i_f = lane_f & 15, kk_f = lane_f >> 4;
      const float* xe = exch + (part * 4 * 64 + lane_f) * 4;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(xe), a1 = *reinterpret_cast<const f32x4*>(xe + 256),
                  a2 = *reinterpret_cast<const f32x4*>(xe + 512), a3 = *reinterpret_cast<const f32x4*>(xe + 768);
      const f32x4 y0 = a0 + a1 + a2 + b2, y1 = a1 - a2 - a3 + b2;
      float* const o0 = p.out + (int64_t)u * p.s_n + (int64_t)(TD * q + 2 * (2 * part + (kk_f >> 1))) * p.s_d + (int64_t)j * p.s_w +
                        (int64_t)(16 + (kk_f & 1)) * p.s_hp + i_f;
      o0[0] = fmaxf(prelu_t<SLOPE01>(y0[0], sl2), prelu_t<SLOPE01>(y0[1], sl2));
      o0[p.s_par] = fmaxf(prelu_t<SLOPE01>(y0[2], sl2), prelu_t<SLOPE01>(y0[3], sl2));
      o0[p.s_d] = fmaxf(prelu_t<SLOPE01>(y1[0], sl2), prelu_t<SLOPE01>(y1[1], sl2));
      o0[p.s_d + p.s_par] = fmaxf(prelu_t<SLOPE01>(y1[2], sl2), prelu_t<SLOPE01>(y1[3], sl2));
    }
    SVK_STAMP_ADD(0, ts0, ts1);
    SVK_STAMP_ADD(1, ts1, ts2);
    SVK_STAMP_ADD(2, ts2, ts3);
    SVK_STAMP_ADD(3, ts3, ts4);
    SVK_STAMP_ADD(4, ts4, ts5);
    SVK_STAMP_ADD(5, ts5, ts6);
  }
#ifdef SVK_TUNING
  if (p.stamps && lane == 0) {
    for (int k = 0; k < 6; ++k) p.stamps[((size_t)blockIdx.x * 8 + wave) * 6 + k] = stamp_acc[k];
    if (wave == 0) {
      unsigned long long* clk = p.stamps + (size_t)gridDim.x * 8 * 6 + (size_t)blockIdx.x * 2;
      clk[0] = __builtin_amdgcn_s_memtime() - clk_c0;
      clk[1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
    }
  }
#endif
}


// -----------------------------------------------------------------------------------------------------
// Third form of the first block (round 3, the default): the depth transform's INPUT side is applied ONCE, where act1 is
// produced, instead of at every fragment read.  c3d2_stage1w_kernel above spends 8 packed adds per 16 MFMAs turning x
// fragments into t = (x0 - x2, x1 + x2, x2 - x1, x1 - x3) in front of every tap -- and an act1 value is read by ~4.5
// taps; f32 VALU never overlaps f32 MFMA on this chip, so that was 7 % of the kernel, plus ~10 cycles per MFMA <-> VALU
// switch.  Here conv1_1's epilogue holds the six act1 depths of a pixel in registers, forms the t planes of both depth
// pairs and writes THOSE to LDS; conv1_2's loop is then ds_read_b128 + MFMA and nothing else.
//   * t planes cost 8 planes per 2 pairs where x planes cost 6, and LDS holds 8: item = (cube, pooled column j, QUARTER q
//     of the output depths) = 2 pairs; 72 items per cube.  conv1_1 recomputes the depth halo 6/4 (was 10/8): + 2.4 % MFMAs.
//   * conv1_1 with the operands SWAPPED: M = channel (A = weights), N = pixel (B = patch values), so a lane ends up
//     with FOUR CHANNELS of ONE pixel -- the six depths of that pixel are six accumulators of the same lane (the
//     transform needs no lane movement) and a t value leaves as ONE ds_write_b128 (the old form wrote four ds_write_b32
//     per tile).  K is permuted so that lane group kk < 3 reads the four contiguous column taps kw = 0 .. 3 of depth tap
//     kk with ONE ds_read_b128 (patch rows hold columns 0..3 | 1..4: both 16-byte aligned) and group 3 reads a 'side'
//     vector {kw = 4 of depth taps 0, 1, 2; 1.0} whose last element carries the bias: one ds_read_b128 per tile where the
//     old form issued four ds_read_b32.
//   * conv1_2 M tiles: 36 rows x 2 columns per pair = four 8-row tiles + 4 rows; the two pairs' 4-row remainders make
//     ONE merged tile: 9 tiles x 144 MFMAs per item where the old form issued 2 x 5 (40 rows for every 36: - 10 %).
//     Waves 0 .. 7 own the eight full tiles; the merged tile is split by k over waves 0 .. 3 (one per SIMD, 36 MFMAs each:
//     every SIMD issues 324), their accumulators meet in LDS and wave 4 finishes it behind the item's last barrier.
//   * t-plane rows are 40 floats: [column 0: 16 channels][8 pad][column 1: 16 channels].  With the true lane groups of
//     ds_read_b128 ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... MI355X_MICROARCH.md, LDS) a tile's 16 pixels must sit
//     in EVEN 16-byte slots, the lanes {0-3, 12-15} and {4-11} each covering all eight: slot = 4 hl + 6 wc (mod 16) does
//     (the old padding made all 16 slots distinct, which collides across the kk = 0 / 1 halves of a group: the 49 %
//     SQ_LDS_BANK_CONFLICT of profiles/r02_c3d2_stalls.txt).
// -----------------------------------------------------------------------------------------------------
constexpr int T_TD = 4, T_DIN = 6, T_PD = 8;              // output depths, act1 depths, patch depths per item
constexpr int T_ROW = 40;                                  // floats per t-plane row: [c = 0: 16][pad 8][c = 1: 16]
constexpr int T_PLANE = NFRAME * T_ROW;                    // 3 200 floats
constexpr int T_PLANES_FLOATS = 8 * T_PLANE;               // [pair][k]
constexpr int T_MAIN_FLOATS = T_PD * NFRAME * 8;           // [dd][h][cols 0..3 | cols 1..4]
constexpr int T_SIDE_FLOATS = T_DIN * NFRAME * 2 * 4;      // [dd'][h][c][{col 4 + c at depth taps 0, 1, 2; 1.0}]
constexpr int T_EXCH_FLOATS = 4 * 64 * 4;                  // the merged tile's four accumulators
constexpr int T_LDS_FLOATS = T_PLANES_FLOATS + T_MAIN_FLOATS + T_SIDE_FLOATS + T_EXCH_FLOATS;
constexpr int T_ITEMS = 18 * (OD / T_TD);                  // 72 items per cube
constexpr int T_PAIRS_DH = T_PD * NFRAME;                  // 640 (patch depth, row) pairs per item

struct PatchRegs {
  f32x4 lo[2], hi[2];   // the 32-byte window of feature row (crop start + h) that holds columns 2 j .. 2 j + 5
};

template <bool SLOPE01>
__global__ __launch_bounds__(512) void c3d2_stage1t_kernel(const Stage1Params p) {
  extern __shared__ __attribute__((aligned(16))) float smem_c3d2[];
  float* const tpl = smem_c3d2;                       // [8 planes][80 rows][40]
  float* const pmain = tpl + T_PLANES_FLOATS;         // [8 dd][80][8]
  float* const pside = pmain + T_MAIN_FLOATS;         // [6 dd'][80][2][4]
  float* const exch = pside + T_SIDE_FLOATS;          // [4 k][64][4]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, kk = lane >> 4;
  const int n_items = p.n_utt * T_ITEMS;

  f32x4 G[36];   // [k][kh]: conv1_2's transformed weights, as in c3d2_stage1w_kernel
#pragma unroll
  for (int kh = 0; kh < 9; ++kh) {
    const f32x4 g0 = p.w2frag[kh * 64 + lane], g1 = p.w2frag[(9 + kh) * 64 + lane], g2 = p.w2frag[(18 + kh) * 64 + lane];
    G[kh] = g0;
    G[9 + kh] = 0.5f * ((g0 + g2) + g1);
    G[18 + kh] = 0.5f * ((g0 + g2) - g1);
    G[27 + kh] = g2;
  }
  // conv1_1's A operand (weights; lane = (channel i, K group kk)): element e = tap (kd = kk, kw = e) for kk < 3, tap
  // (kd = e, kw = 4) for kk = 3, e < 3, and the (BN-folded) bias for kk = 3, e = 3 -- read from the same w1frag table the
  // other forms take ([4 jj][64]: tap 4 jj + kq of channel l & 15 in lane 16 kq + channel)
  float wA[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int t = kk < 3 ? 5 * kk + e : 5 * e + 4;
    wA[e] = (kk == 3 && e == 3) ? p.bias1[i] : p.w1frag[(t >> 2) * 64 + (t & 3) * 16 + i];
  }
  float sl1[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) sl1[r] = p.slope1[4 * kk + r];
  const float b2 = p.bias2[i], sl2 = p.slope2[i];

  // the constant 1.0 of every side vector (never overwritten) + zeroed exchange
  for (int e = threadIdx.x; e < T_DIN * NFRAME * 2; e += 512) pside[4 * e + 3] = 1.0f;

  // ---- patch staging: (patch depth dd, row h) pairs q0 = thread and thread + 512 (< 640) ----
  // (dd, h) of pair m are recomputed where they are needed: four registers this kernel does not have to spare
  auto pair_dd = [&](int m) { const int q0 = threadIdx.x + 512 * m; return q0 < T_PAIRS_DH ? q0 / NFRAME : -1; };
  auto pair_h = [&](int m) { const int q0 = threadIdx.x + 512 * m; return q0 - (q0 / NFRAME) * NFRAME; };
  auto load_starts = [&](int item, int (&st)[2]) {
    const int u = item / T_ITEMS, q = (item - u * T_ITEMS) / 18;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int dd = pair_dd(m);
      st[m] = dd >= 0 ? p.crop[(int64_t)u * NCROP + T_TD * q + dd] : -1;
    }
  };
  auto fetch_patch_t = [&](int item, const int (&st)[2], PatchRegs& pr) {
    const int u = item / T_ITEMS, rem = item - u * T_ITEMS, j = rem % 18;
    const float* base = p.feat + (int64_t)u * p.max_frames * NCOEF + ((2 * j) & ~3);   // 16-byte aligned window start
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      pr.lo[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
      pr.hi[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int start = st[m], h = pair_h(m);   // (pairs past the 640th carry start = -1)
      if ((unsigned)start < (unsigned)p.max_frames && h < p.max_frames - start) {
        const float* s = base + (int64_t)(start + h) * NCOEF;
        pr.lo[m] = *reinterpret_cast<const f32x4*>(s);
        pr.hi[m] = *reinterpret_cast<const f32x4*>(s + 4);
      }
    }
  };
  auto park_patch_t = [&](int item, const PatchRegs& pr) {
    const int u = item / T_ITEMS, rem = item - u * T_ITEMS, j = rem % 18;
    const bool odd = (j & 1) != 0;                     // columns 2 j .. 2 j + 5 start at float 2 of the window when j is odd
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int dd = pair_dd(m), h = pair_h(m);
      if (dd < 0) continue;
      const f32x4 lo = pr.lo[m], hi = pr.hi[m];
      const float v0 = odd ? lo[2] : lo[0], v1 = odd ? lo[3] : lo[1], v2 = odd ? hi[0] : lo[2], v3 = odd ? hi[1] : lo[3],
                  v4 = odd ? hi[2] : hi[0], v5 = odd ? hi[3] : hi[1];
      float* mrow = pmain + (dd * NFRAME + h) * 8;
      *reinterpret_cast<f32x4*>(mrow) = (f32x4){v0, v1, v2, v3};
      *reinterpret_cast<f32x4*>(mrow + 4) = (f32x4){v1, v2, v3, v4};
      // column 4 + c of patch depth dd is depth tap e of act1 depth dd - e: ONE base (act1 depth dd - 2) + immediates
      float* const sv = pside + ((dd - 2) * NFRAME + h) * 8;
      if (dd <= 5) {
        sv[2 * (NFRAME * 8)] = v4;
        sv[2 * (NFRAME * 8) + 4] = v5;
      }
      if (dd >= 1 && dd <= 6) {
        sv[NFRAME * 8 + 1] = v4;
        sv[NFRAME * 8 + 5] = v5;
      }
      if (dd >= 2) {
        sv[2] = v4;
        sv[6] = v5;
      }
    }
  };

  PatchRegs pre;
  int st_cur[2] = {-1, -1}, st_next[2] = {-1, -1};
  int item = blockIdx.x;
  if (item < n_items) {
    load_starts(item, st_cur);
    fetch_patch_t(item, st_cur, pre);
    __syncthreads();                                   // the side vectors' constant is in place before the first park
    park_patch_t(item, pre);
    if (item + (int)gridDim.x < n_items) load_starts(item + gridDim.x, st_next);
  }
  __syncthreads();

  const int full_pair = wave >> 2, full_h0 = 8 * (wave & 3);
#ifdef SVK_TUNING
  unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0};
#endif
  for (; item < n_items; item += gridDim.x) {
    SVK_STAMP(ts0);
    const int next = item + gridDim.x;
    const int u = item / T_ITEMS, rem = item - u * T_ITEMS, q = rem / 18, j = rem - q * 18;

    // ---- conv1_1 + PReLU + input transform: ten pixel sets (8 rows x 2 columns), set = wave, sets 8 and 9 on waves 4, 5 ----
    // per-lane address parts (recomputed per item: registers are the scarce resource here)
    const int c11_b = (kk < 3 ? (int)(pmain - smem_c3d2) + (kk * NFRAME + (i >> 1)) * 8 + 4 * (i & 1)
                              : (int)(pside - smem_c3d2) + ((i >> 1) * 2 + (i & 1)) * 4);          // conv1_1 B operand, set 0, depth 0
    const int c11_w = (i >> 1) * T_ROW + 24 * (i & 1) + 4 * kk;                                  // t-plane write, set 0
#pragma unroll 1
    for (int set = wave; set < 10; set += (wave == 4 || wave == 5) ? 4 : 16) {
      const float* bp = smem_c3d2 + c11_b + set * 64;            // 8 rows further: 64 floats in both patch arrays
      f32x4 xb[T_DIN];
#pragma unroll
      for (int d = 0; d < T_DIN; ++d) xb[d] = *reinterpret_cast<const f32x4*>(bp + d * (NFRAME * 8));
      f32x4 x[T_DIN];
#pragma unroll
      for (int d = 0; d < T_DIN; ++d) x[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int d = 0; d < T_DIN; ++d) x[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[e], xb[d][e], x[d], 0, 0, 0);
#pragma unroll
      for (int d = 0; d < T_DIN; ++d)
#pragma unroll
        for (int r = 0; r < 4; ++r) x[d][r] = prelu_t<SLOPE01>(x[d][r], sl1[r]);
      float* tw = tpl + c11_w + set * (8 * T_ROW);
#pragma unroll
      for (int pr2 = 0; pr2 < 2; ++pr2) {
        const f32x4 x0 = x[2 * pr2], x1 = x[2 * pr2 + 1], x2 = x[2 * pr2 + 2], x3 = x[2 * pr2 + 3];
        float* tp = tw + pr2 * (4 * T_PLANE);
        *reinterpret_cast<f32x4*>(tp) = x0 - x2;
        *reinterpret_cast<f32x4*>(tp + T_PLANE) = x1 + x2;
        *reinterpret_cast<f32x4*>(tp + 2 * T_PLANE) = x2 - x1;
        *reinterpret_cast<f32x4*>(tp + 3 * T_PLANE) = x1 - x3;
      }
    }
    SVK_STAMP(ts1);
    __syncthreads();   // the t planes are complete; the patch arrays are free
    SVK_STAMP(ts2);

    // ---- the next item's patch: loads issued here, parked behind this item's matrix work ----
    if (next < n_items) fetch_patch_t(next, st_next, pre);
    if (next + (int)gridDim.x < n_items) load_starts(next + gridDim.x, st_cur);   // (st_cur is dead: re-used as 'the one after')
    __builtin_amdgcn_sched_barrier(0);

    // ---- conv1_2: this wave's full tile (pair = wave >> 2, rows 8 (wave & 3) ..) ----
    float* const obase = p.out + (int64_t)u * p.s_n + (int64_t)(T_TD * q) * p.s_d + (int64_t)j * p.s_w;
    {
      const float* ab = tpl + full_pair * (4 * T_PLANE) + (2 * (full_h0 + (i >> 1))) * T_ROW + 24 * (i & 1) + 4 * kk;   // tap 0, plane 0
      f32x4 acc[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = k == 1 ? (f32x4){b2, b2, b2, b2} : (f32x4){0.f, 0.f, 0.f, 0.f};   // a1 carries the bias
      f32x4 a[4], an[2];
#pragma unroll
      for (int k = 0; k < 4; ++k) a[k] = *reinterpret_cast<const f32x4*>(ab + k * T_PLANE);
#pragma unroll
      for (int kh = 0; kh < 9; ++kh) {
        // the next tap's fragments of planes 0, 1 are read in front of this tap's MFMAs into a second register pair;
        // those of planes 2, 3 go straight into a[2], a[3] once this tap's last MFMA on them has issued (the e = 3 round
        // runs k = 2, 3, 0, 1: eight and more MFMAs = 256 cycles pass before the next tap reaches them) -- a full second
        // fragment set does not fit beside the 144 weight registers
        if (kh + 1 < 9) {
#pragma unroll
          for (int k = 0; k < 2; ++k) an[k] = *reinterpret_cast<const f32x4*>(ab + k * T_PLANE + (kh + 1) * T_ROW);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 3; ++e)
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k][e], G[9 * k + kh][e], acc[k], 0, 0, 0);
#pragma unroll
        for (int k = 2; k < 4; ++k) {
          acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k][3], G[9 * k + kh][3], acc[k], 0, 0, 0);
          if (kh + 1 < 9) {
            __builtin_amdgcn_sched_barrier(0);
            a[k] = *reinterpret_cast<const f32x4*>(ab + k * T_PLANE + (kh + 1) * T_ROW);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k][3], G[9 * k + kh][3], acc[k], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (kh + 1 < 9) {
#pragma unroll
          for (int k = 0; k < 2; ++k) a[k] = an[k];
        }
      }
      // rows 4 kk + r of the tile: output row full_h0 + 2 kk + (r >> 1), column r & 1: pool = max over the r pairs
      const f32x4 y0 = acc[0] + acc[1] + acc[2], y1 = acc[1] - acc[2] - acc[3];
      float* const o00 = obase + (int64_t)(2 * full_pair) * p.s_d + (int64_t)(full_h0 / 2) * p.s_hp;
      float* const o01 = o00 + p.s_par;
      float* const o10 = o00 + p.s_d;
      float* const o11 = o10 + p.s_par;
      const int olane = kk * (int)p.s_hp + i;
      o00[olane] = fmaxf(prelu_t<SLOPE01>(y0[0], sl2), prelu_t<SLOPE01>(y0[1], sl2));
      o01[olane] = fmaxf(prelu_t<SLOPE01>(y0[2], sl2), prelu_t<SLOPE01>(y0[3], sl2));
      o10[olane] = fmaxf(prelu_t<SLOPE01>(y1[0], sl2), prelu_t<SLOPE01>(y1[1], sl2));
      o11[olane] = fmaxf(prelu_t<SLOPE01>(y1[2], sl2), prelu_t<SLOPE01>(y1[3], sl2));
    }
    SVK_STAMP(ts3);
    // ---- the merged tile (rows 32 .. 35 of both pairs): waves 0 .. 3 compute accumulator k = wave ----
    if (wave < 4) {
      const float* ab = tpl + (i >> 3) * (4 * T_PLANE) + (2 * (32 + ((i >> 1) & 3))) * T_ROW + 24 * (i & 1) + 4 * kk + wave * T_PLANE;
      f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;   // two chains: back-to-back dependent MFMAs would wait 40 cycles each
      f32x4 a = *reinterpret_cast<const f32x4*>(ab), an;
#pragma unroll
      for (int kh = 0; kh < 9; ++kh) {
        if (kh + 1 < 9) an = *reinterpret_cast<const f32x4*>(ab + (kh + 1) * T_ROW);
        const f32x4 g = wave == 0 ? G[kh] : wave == 1 ? G[9 + kh] : wave == 2 ? G[18 + kh] : G[27 + kh];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], g[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], g[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], g[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], g[3], acc1, 0, 0, 0);
        if (kh + 1 < 9) a = an;
      }
      *reinterpret_cast<f32x4*>(exch + (wave * 64 + lane) * 4) = acc0 + acc1;
    }
    SVK_STAMP(ts4);
    if (next < n_items) park_patch_t(next, pre);
    SVK_STAMP(ts5);
    __syncthreads();   // the next patch and the merged tile's accumulators are in place; the t planes may be overwritten
    SVK_STAMP(ts6);
    SVK_STAMP_ADD(0, ts0, ts1);  // conv1_1 phase
    SVK_STAMP_ADD(1, ts1, ts2);  // barrier 1
    SVK_STAMP_ADD(2, ts2, ts3);  // patch load issue + full tile + epilogue
    SVK_STAMP_ADD(3, ts3, ts4);  // merged-tile quarter
    SVK_STAMP_ADD(4, ts4, ts5);  // park
    SVK_STAMP_ADD(5, ts5, ts6);  // barrier 2
    if (wave == 4) {
      // merged tile: rows 4 kk + r = (pair kk >> 1, output row 32 + 2 (kk & 1) + (r >> 1), column r & 1)
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(exch + lane * 4), a1 = *reinterpret_cast<const f32x4*>(exch + (64 + lane) * 4),
                  a2 = *reinterpret_cast<const f32x4*>(exch + (128 + lane) * 4), a3 = *reinterpret_cast<const f32x4*>(exch + (192 + lane) * 4);
      const f32x4 y0 = a0 + a1 + a2 + b2, y1 = a1 - a2 - a3 + b2;
      float* const o0 = obase + (int64_t)(2 * (kk >> 1)) * p.s_d + (int64_t)(16 + (kk & 1)) * p.s_hp + i;
      o0[0] = fmaxf(prelu_t<SLOPE01>(y0[0], sl2), prelu_t<SLOPE01>(y0[1], sl2));
      o0[p.s_par] = fmaxf(prelu_t<SLOPE01>(y0[2], sl2), prelu_t<SLOPE01>(y0[3], sl2));
      o0[p.s_d] = fmaxf(prelu_t<SLOPE01>(y1[0], sl2), prelu_t<SLOPE01>(y1[1], sl2));
      o0[p.s_d + p.s_par] = fmaxf(prelu_t<SLOPE01>(y1[2], sl2), prelu_t<SLOPE01>(y1[3], sl2));
    }
    // the crop starts loaded into st_cur during this item belong to the item after next: rotate
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int t = st_next[m];
      st_next[m] = st_cur[m];
      st_cur[m] = t;
    }
  }
#ifdef SVK_TUNING
  if (p.stamps && lane == 0)
    for (int k = 0; k < 6; ++k) p.stamps[((size_t)blockIdx.x * 8 + wave) * 6 + k] = stamp_acc[k];
#endif
}

}  // namespace

extern "C" {

// (of the larger variant, the depth-transformed one)
size_t svk_c3d2_stage1_lds_bytes(void) { return sizeof(float) * (size_t)std::max(WACT_FLOATS + WP_FLOATS + 2048, T_LDS_FLOATS); }

int svk_c3d2_stage1(svk_ctx* ctx, const float* d_feat, int32_t n_utt, int32_t max_frames, int32_t n_cols,
                    const int32_t* d_crop_idx, int32_t n_crops, int32_t crop_frames, const float* d_w1frag,
                    const float* d_bias1, const float* d_slope1, const float* d_w2frag, const float* d_bias2,
                    const float* d_slope2, int32_t folded, float* d_out) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  // (the slopes live on the device: whether all 32 lie in [0, 1] -- the two-instruction PReLU -- is the caller's
  // knowledge, passed in bit 1 of `folded`: 0 / 1 = layout with the general PReLU, 2 / 3 = the same with slopes in [0, 1])
  const bool slope01 = (folded & 2) != 0, wino = (folded & 4) != 0, tform = (folded & 8) != 0;   // bit 3: c3d2_stage1t_kernel
  const bool merge = (folded & 16) != 0;                                                         // bit 4: merged remainder tiles
  folded &= 1;
  SVK_REQUIRE(ctx, n_utt >= 0 && max_frames >= 1, "shape");
  if (n_cols != NCOEF || n_crops != NCROP || crop_frames != NFRAME)
    return svk_fail(ctx, SVK_ERR_UNSUPPORTED,
                    "svk_c3d2_stage1 is built for the 20 x 80 x 40 cube of utils.py:20-21 (got %d x %d x %d)", n_crops,
                    crop_frames, n_cols);
  if (n_utt == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_feat && d_crop_idx && d_w1frag && d_bias1 && d_slope1 && d_w2frag && d_bias2 && d_slope2 && d_out,
              "NULL buffer");
  SVK_REQUIRE(ctx, (reinterpret_cast<uintptr_t>(d_feat) & 7) == 0 && (reinterpret_cast<uintptr_t>(d_w2frag) & 15) == 0,
              "d_feat must be 8-byte and d_w2frag 16-byte aligned");
  SVK_REQUIRE(ctx, (int64_t)n_utt * T_ITEMS < ((int64_t)1 << 31), "too many cubes for one launch");
  SVK_REQUIRE(ctx, !tform || (reinterpret_cast<uintptr_t>(d_feat) & 15) == 0, "d_feat must be 16-byte aligned for the t-plane form");
  Stage1Params p;
  p.feat = d_feat;
  p.crop = d_crop_idx;
  p.n_utt = n_utt;
  p.max_frames = max_frames;
  p.w1frag = d_w1frag;
  p.bias1 = d_bias1;
  p.slope1 = d_slope1;
  p.w2frag = reinterpret_cast<const f32x4*>(d_w2frag);
  p.bias2 = d_bias2;
  p.slope2 = d_slope2;
  p.out = d_out;
  if (folded) {  // [n][16 d][18 row pairs][18 w][2 parity][16 c]: channels-last memory of a (n, 32, 16, 18, 18) tensor
    p.s_w = 32;
    p.s_par = 16;
    p.s_hp = (int64_t)OWP * 32;
    p.s_d = (int64_t)(OH / 2) * OWP * 32;
  } else {       // [n][16 d][36 h][18 w][16 c]: channels-last memory of a (n, 16, 16, 36, 18) tensor
    p.s_w = 16;
    p.s_par = (int64_t)OWP * 16;
    p.s_hp = 2 * (int64_t)OWP * 16;
    p.s_d = (int64_t)OH * OWP * 16;
  }
  p.s_n = (int64_t)OD * OH * OWP * 16;
  const size_t lds = sizeof(float) * (size_t)(tform ? T_LDS_FLOATS : wino ? WACT_FLOATS + WP_FLOATS + (merge ? 2048 : 0) : ACT_FLOATS + P_FLOATS);
  if (lds > (size_t)ctx->lds_per_cu)
    return svk_fail(ctx, SVK_ERR_UNSUPPORTED, "svk_c3d2_stage1 needs %zu bytes of LDS per workgroup (device: %d)", lds,
                    ctx->lds_per_cu);
  void (*kern)(const Stage1Params) = tform ? (slope01 ? c3d2_stage1t_kernel<true> : c3d2_stage1t_kernel<false>)
                                     : wino ? (merge ? (slope01 ? c3d2_stage1w_kernel<true, true> : c3d2_stage1w_kernel<false, true>)
                                                     : (slope01 ? c3d2_stage1w_kernel<true, false> : c3d2_stage1w_kernel<false, false>))
                                            : (slope01 ? c3d2_stage1_kernel<true> : c3d2_stage1_kernel<false>);
  const int n_waves = (wino || tform) ? 8 : 4;
  SVK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t items = (int64_t)n_utt * (tform ? T_ITEMS : 36);
  const unsigned grid = (unsigned)std::min<int64_t>(items, ctx->num_cu);  // one persistent workgroup per CU
  p.stamps = nullptr;
#ifdef SVK_TUNING
  const bool want_stamps = getenv("SVK_C3D2_STAMPS") != nullptr;
  const size_t stamp_bytes = ((size_t)grid * n_waves * 6 + (size_t)grid * 2) * sizeof(unsigned long long);
  if (want_stamps) {
    const int rc = svk_ensure_work(ctx, stamp_bytes);
    if (rc != SVK_OK) return rc;
    p.stamps = reinterpret_cast<unsigned long long*>(ctx->work);
    SVK_HIP(ctx, hipMemsetAsync(p.stamps, 0, stamp_bytes, ctx->stream));
  }
#endif
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * n_waves), lds, ctx->stream, p);
  SVK_LAUNCH_CHECK(ctx);
#ifdef SVK_TUNING
  if (want_stamps) {  // phase cycles (s_memtime, 100 MHz-independent shader clock), averaged over workgroups, per wave
    std::vector<unsigned long long> h((size_t)grid * n_waves * 6 + (size_t)grid * 2);
    SVK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SVK_HIP(ctx, hipMemcpy(h.data(), p.stamps, stamp_bytes, hipMemcpyDeviceToHost));
    const char* names_w[6] = {"issue next patch loads", "conv1_1 phase", "barrier 1", "conv1_2 phase + epilogue", "park", "barrier 2"};
    const char* names_t[6] = {"conv1_1 phase", "barrier 1", "patch loads + full tile + epilogue", "merged-tile quarter", "park", "barrier 2"};
    const char** names = tform ? names_t : names_w;
    const double per = (double)items / grid;
    for (int w = 0; w < n_waves; ++w) {
      fprintf(stderr, "stage1 stamps wave %d (cycles per item):", w);
      for (int k = 0; k < 6; ++k) {
        double sum = 0;
        for (unsigned b = 0; b < grid; ++b) sum += (double)h[((size_t)b * n_waves + w) * 6 + k];
        fprintf(stderr, "  %s %.0f", names[k], sum / grid / per);
      }
      fprintf(stderr, "\n");
    }
    {   // spread over workgroups of the loop's total cycles (wave 0)
      std::vector<double> tot;
      for (unsigned b = 0; b < grid; ++b) {
        double t = 0;
        for (int k = 0; k < 6; ++k) t += (double)h[((size_t)b * n_waves) * 6 + k];
        tot.push_back(t);
      }
      std::sort(tot.begin(), tot.end());
      fprintf(stderr, "stage1 loop cycles per workgroup: min %.0f  median %.0f  p90 %.0f  max %.0f\n", tot.front(), tot[tot.size() / 2],
              tot[tot.size() * 9 / 10], tot.back());
    }
    if (wino && !tform) {   // in-kernel clock: median over workgroups of shader cycles per 100 MHz reference tick
      std::vector<double> mhz;
      for (unsigned b = 0; b < grid; ++b) {
        const unsigned long long c = h[(size_t)grid * n_waves * 6 + 2 * b], r = h[(size_t)grid * n_waves * 6 + 2 * b + 1];
        if (r) mhz.push_back(100.0 * (double)c / (double)r);
      }
      if (!mhz.empty()) {
        std::sort(mhz.begin(), mhz.end());
        fprintf(stderr, "stage1 in-kernel clock: median %.0f MHz (min %.0f, max %.0f) over %zu workgroups; the 157.3 TFLOP/s peak assumes 2400\n",
                mhz[mhz.size() / 2], mhz.front(), mhz.back(), mhz.size());
      }
    }
  }
#endif
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

// ---- conv2_1: taps along w.  Item = (cube, block of C21_TH rows h): all 14 output depths x C21_TH rows x 15 columns
// (2 rows: 39 KB of LDS per workgroup, three workgroups per CU cover each other's staging; 4 rows / two per CU: 7 % slower) ----
constexpr int C21_TH = 2;
constexpr int C21_PIX = S2_D * C21_TH * S2_W;        // 1152 pixels of 16 channels, stored at 16 p + 4 (p >> 2)
constexpr int C21_LDS_FLOATS = 17 * (C21_PIX + 4);

struct Conv21Params {
  const float* in;      // [n][16][36][18][16]
  const f32x4* wfrag;   // [2 nt][12 taps][64]: lane (co = 16 nt + (l & 15), kk = l >> 4), e: W[co][4 kk + e][kd][kw], tap = 4 kd + kw
  const float* bias;    // [32]
  const float* slope;   // [32]
  float* out;           // [n][14][36][15][32]
  int32_t n_utt;
  unsigned* queue;      // work-item counter (zeroed by the host before the launch), or NULL = static round-robin
};

__global__ __launch_bounds__(256, 2) void c3d2_conv21_kernel(const Conv21Params p) {
  extern __shared__ __attribute__((aligned(16))) float smem_c21[];
  float* reg = smem_c21;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kk = lane >> 4;
  f32x4 w[2][12];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int t = 0; t < 12; ++t) w[nt][t] = p.wfrag[(nt * 12 + t) * 64 + lane];
  float b[2], sl[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    b[nt] = p.bias[16 * nt + i];
    sl[nt] = p.slope[16 * nt + i];
  }
  const int n_items = p.n_utt * (S2_H / C21_TH);
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int u = item / (S2_H / C21_TH), hb = (item - u * (S2_H / C21_TH)) * C21_TH;
    // stage: [16 d][4 rows][18 w][16 c], a 16-byte piece per thread and trip
    const float* src = p.in + (int64_t)u * (S2_D * S2_H * S2_W * 16);
    // all nine 16-byte loads of a thread first, then the LDS writes: written as a plain load-store loop the
    // compiler waited for every load before issuing the next (nine memory round trips per item; conv2_2's 18
    // were 24 k of its 97 k cycles per item by the in-kernel stamps)
    constexpr int C21_NV = 3;  // loads in flight per thread and round (more would cost the third resident workgroup its registers)
    static_assert(C21_PIX * 4 % (256 * C21_NV) == 0, "whole rounds");
#pragma unroll 1
    for (int base = threadIdx.x; base < C21_PIX * 4; base += 256 * C21_NV) {
      f32x4 sv[C21_NV];
#pragma unroll
      for (int k = 0; k < C21_NV; ++k) {
        const int e = base + 256 * k;
        const int pix = e >> 2, piece = e & 3;
        const int d = pix / (C21_TH * S2_W), rem = pix - d * (C21_TH * S2_W);  // rem = hl * 18 + w
        sv[k] = *reinterpret_cast<const f32x4*>(src + ((int64_t)(d * S2_H + hb) * S2_W + rem) * 16 + 4 * piece);
      }
#pragma unroll
      for (int k = 0; k < C21_NV; ++k) {
        const int e = base + 256 * k;
        const int pix = e >> 2, piece = e & 3;
        *reinterpret_cast<f32x4*>(reg + 16 * pix + 4 * (pix >> 2) + 4 * piece) = sv[k];
      }
    }
    __syncthreads();
    // M tiles = (output depth d', row hl): 16 pixels w' = 0 .. 15 (15 is a dummy), 14 x C21_TH / 4 per wave
    for (int tile = wave; tile < A2_D * C21_TH; tile += 4) {
      const int dp = tile / C21_TH, hl = tile - dp * C21_TH;
      const int p0 = (dp * C21_TH + hl) * S2_W + i;
      // pixel p0 + kw sits at 16 (p0 + kw) + 4 ((p0 + kw) >> 2); a depth step is 72 pixels = 18 groups of 4
      const float* ab[4];
#pragma unroll
      for (int kw = 0; kw < 4; ++kw) ab[kw] = reg + 16 * (p0 + kw) + 4 * ((p0 + kw) >> 2) + 4 * kk;
      f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
      // tap t + 1's fragment is read in front of tap t's eight MFMAs (fenced: see the stage-1 tap loop)
      f32x4 a = *reinterpret_cast<const f32x4*>(ab[0]);
#pragma unroll
      for (int t = 0; t < 12; ++t) {
        f32x4 an;
        if (t + 1 < 12) an = *reinterpret_cast<const f32x4*>(ab[(t + 1) & 3] + ((t + 1) >> 2) * (17 * C21_TH * S2_W));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], w[nt][t][e], acc[nt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < 12) a = an;
      }
      // rows 4 kk + r = output column w'; column i = channel 16 nt + i
      float* o = p.out + (((int64_t)u * A2_D + dp) * S2_H + hb + hl) * (A2_W * 32) + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int wq = 4 * kk + r;
        if (wq < A2_W) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) o[wq * 32 + 16 * nt] = prelu(acc[nt][r] + b[nt], sl[nt]);
        }
      }
    }
    __syncthreads();
  }
}

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
  float b[2], sl[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    b[nt] = p.bias[16 * nt + i];
    sl[nt] = p.slope[16 * nt + i];
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
    // M tiles = (pair P, row hl): 16 pixels w' = 0 .. 15 (15 is a dummy), tile = 4 P + hl
#pragma unroll 1
    for (int tile = wave; tile < 7 * C21W_TH; tile += 4) {
      const int P = tile >> 2, hl = tile & 3;
      const int p0 = (2 * P * C21W_TH + hl) * S2_W + i;
      const float* ab[4];
#pragma unroll
      for (int kw = 0; kw < 4; ++kw) ab[kw] = reg + 16 * (p0 + kw) + 4 * ((p0 + kw) >> 2) + 4 * kk;
      // (a1 enters both outputs with a plus sign: the bias rides in its accumulator)
      f32x4 acc[2][4];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[nt][k] = k == 1 ? (f32x4){b[nt], b[nt], b[nt], b[nt]} : (f32x4){0.f, 0.f, 0.f, 0.f};
      f32x4 x[4];
      f32x2 t[4][2];
#pragma unroll
      for (int dd = 0; dd < 4; ++dd) x[dd] = *reinterpret_cast<const f32x4*>(ab[0] + C21W_DSTEP * dd);
#pragma unroll
      for (int kw = 0; kw < 4; ++kw) {
        // this tap's transformed fragments (8 packed adds, one burst), the next tap's reads, 32 MFMAs
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) wino_input_pair(x, hf, t);
        __builtin_amdgcn_sched_barrier(0);
        if (kw + 1 < 4) {
#pragma unroll
          for (int dd = 0; dd < 4; ++dd) x[dd] = *reinterpret_cast<const f32x4*>(ab[kw + 1] + C21W_DSTEP * dd);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int k = 0; k < 4; ++k)
              acc[nt][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(t[k][e >> 1][e & 1], G[nt][k][kw][e], acc[nt][k], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // rows 4 kk + r = output column w'; column i = channel 16 nt + i; depths 2 P and 2 P + 1
      // (wave-uniform 64-bit bases + one 32-bit lane offset + immediates: the stores need no per-store address VALU)
      float* const o0 = p.out + (((int64_t)u * A2_D + 2 * P) * S2_H + hb + hl) * (A2_W * 32);
      float* const o1 = o0 + (int64_t)S2_H * (A2_W * 32);
      const int lo = 4 * kk * 32 + i;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        // y0 = (a0 + a1) + a2, y1 = (a1 - a2) - a3 as packed adds (every VALU instruction here is paid in MFMA slots)
        f32x2 y0[2], y1[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const f32x2 a0 = hf ? __builtin_shufflevector(acc[nt][0], acc[nt][0], 2, 3) : __builtin_shufflevector(acc[nt][0], acc[nt][0], 0, 1);
          const f32x2 a1 = hf ? __builtin_shufflevector(acc[nt][1], acc[nt][1], 2, 3) : __builtin_shufflevector(acc[nt][1], acc[nt][1], 0, 1);
          const f32x2 a2 = hf ? __builtin_shufflevector(acc[nt][2], acc[nt][2], 2, 3) : __builtin_shufflevector(acc[nt][2], acc[nt][2], 0, 1);
          const f32x2 a3 = hf ? __builtin_shufflevector(acc[nt][3], acc[nt][3], 2, 3) : __builtin_shufflevector(acc[nt][3], acc[nt][3], 0, 1);
          y0[hf] = pk_add(pk_add(a0, a1), a2);
          y1[hf] = pk_sub(pk_sub(a1, a2), a3);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int wq = 4 * kk + r;
          if (wq < A2_W) {
            o0[lo + r * 32 + 16 * nt] = prelu_t<SLOPE01>(y0[r >> 1][r & 1], sl[nt]);
            o1[lo + r * 32 + 16 * nt] = prelu_t<SLOPE01>(y1[r >> 1][r & 1], sl[nt]);
          }
        }
      }
    }
    __syncthreads();
    item = item_next;
  }
}

// ---- conv2_2 + pool2: taps along h (stride 2).  Item = (cube, pooled column j, half q of the output depths) ----
constexpr int C22_TD = 6;                                  // output depths per item
constexpr int C22_PIX = (C22_TD + 2) * 2 * S2_H;           // [8 d][2 w][36 h] pixels of 32 channels at 32 p + 4 (p >> 1)
constexpr int C22_LDS_FLOATS = 34 * (C22_PIX + 4);

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

__global__ __launch_bounds__(256, 2) void c3d2_conv22_kernel(const Conv22Params p) {
  extern __shared__ __attribute__((aligned(16))) float smem_c22[];
  float* reg = smem_c22;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kk = lane >> 4;
  const int nt = wave & 1, half = wave >> 1;   // output channels 16 nt .., output depths 3 half .. 3 half + 2 of the item
  f32x4 w[24][2];
#pragma unroll
  for (int t = 0; t < 24; ++t)
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) w[t][ch] = p.wfrag[((nt * 24 + t) * 2 + ch) * 64 + lane];
  const float b = p.bias[16 * nt + i], sl = p.slope[16 * nt + i];
  const int n_items = p.n_utt * (O2_W * 2);
#ifdef SVK_TUNING
  unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0};
#endif
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    SVK_STAMP(ts0);
    const int u = item / (O2_W * 2), rem = item - u * (O2_W * 2), q = rem / O2_W, j = rem - q * O2_W;
    // stage [8 d][36 h][2 w][32 c] of the input as pixels p = (dl * 2 + w) * 36 + h (h fastest)
    const float* src = p.in + ((int64_t)u * A2_D + C22_TD * q) * (S2_H * A2_W * 32) + 2 * j * 32;
    // six rounds of three 16-byte loads in flight per thread (a plain load-store loop waited for every load: 18
    // memory round trips, 24 k of the item's 97 k cycles by the in-kernel stamps; the 192 weight VGPRs leave room for
    // three -- six in flight spilled)
    constexpr int C22_NV = 3;
    static_assert(C22_PIX * 8 % (256 * C22_NV) == 0, "whole rounds");
#pragma unroll 1
    for (int base = threadIdx.x; base < C22_PIX * 8; base += 256 * C22_NV) {
      f32x4 sv[C22_NV];
#pragma unroll
      for (int k = 0; k < C22_NV; ++k) {
        const int e = base + 256 * k;
        const int piece = e & 7, wq = (e >> 3) & 1, dh = e >> 4;   // dh = dl * 36 + h
        sv[k] = *reinterpret_cast<const f32x4*>(src + (int64_t)dh * (A2_W * 32) + wq * 32 + 4 * piece);
      }
#pragma unroll
      for (int k = 0; k < C22_NV; ++k) {
        const int e = base + 256 * k;
        const int piece = e & 7, wq = (e >> 3) & 1, dh = e >> 4;
        const int dl = dh / S2_H, h = dh - dl * S2_H;
        const int pix = (dl * 2 + wq) * S2_H + h;
        *reinterpret_cast<f32x4*>(reg + 32 * pix + 4 * (pix >> 1) + 4 * piece) = sv[k];
      }
    }
    SVK_STAMP(ts1);
    __syncthreads();
    SVK_STAMP(ts2);
    for (int s = 0; s < 3; ++s) {
      const int dp = 3 * half + s;                           // output depth inside the item
      const int p0 = dp * 2 * S2_H + 2 * i;                  // column 0 of the pair; column 1 is 36 pixels on
      const float* a0 = reg + 32 * p0 + 4 * (p0 >> 1) + 4 * kk;
      const float* a1 = a0 + 34 * S2_H;
      f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
      // step = (tap, 16-channel chunk): the next step's two fragments are read in front of this step's eight MFMAs
      f32x4 x0 = *reinterpret_cast<const f32x4*>(a0), x1 = *reinterpret_cast<const f32x4*>(a1);
#pragma unroll
      for (int st = 0; st < 48; ++st) {
        f32x4 n0, n1;
        if (st + 1 < 48) {
          const int t = (st + 1) >> 1, ch = (st + 1) & 1, kd = t >> 3, kh = t & 7;
          const int off = 34 * (2 * S2_H * kd) + 32 * kh + 4 * (kh >> 1) + 16 * ch;
          n0 = *reinterpret_cast<const f32x4*>(a0 + off);
          n1 = *reinterpret_cast<const f32x4*>(a1 + off);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[e], w[st >> 1][st & 1][e], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[e], w[st >> 1][st & 1][e], acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (st + 1 < 48) {
          x0 = n0;
          x1 = n1;
        }
      }
      // rows 4 kk + r = output row h'; pool over the column pair, PReLU first (model.py:156-158)
      float* o = p.out + ((((int64_t)u * O2_D + C22_TD * q + dp) * O2_H) * O2_W + j) * 32 + 16 * nt + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int hq = 4 * kk + r;
        if (hq < O2_H) o[(int64_t)hq * (O2_W * 32)] = fmaxf(prelu(acc0[r] + b, sl), prelu(acc1[r] + b, sl));
      }
    }
    SVK_STAMP(ts3);
    __syncthreads();
    SVK_STAMP(ts4);
    SVK_STAMP_ADD(0, ts0, ts1);  // staging: global -> registers -> LDS
    SVK_STAMP_ADD(1, ts1, ts2);  // barrier 1
    SVK_STAMP_ADD(2, ts2, ts3);  // 3 x (48 steps of 8 MFMAs) + epilogues
    SVK_STAMP_ADD(3, ts3, ts4);  // barrier 2
  }
#ifdef SVK_TUNING
  if (p.stamps && lane == 0)
    for (int k = 0; k < 4; ++k) p.stamps[((size_t)blockIdx.x * 4 + wave) * 4 + k] = stamp_acc[k];
#endif
}

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
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      const int P = pass ? ch : 1 - ch;   // the partner's pair first
#pragma unroll
      for (int wq = 0; wq < 2; ++wq) {
        const float* ap = a0 + 2 * C22W_PLANE * P + C22W_COL * wq;
        f32x4 acc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = k == 1 ? (f32x4){b, b, b, b} : (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 x[4];
        f32x2 t[4][2];
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) x[dd] = *reinterpret_cast<const f32x4*>(ap + C22W_PLANE * dd);
#pragma unroll
        for (int kh = 0; kh < 8; ++kh) {
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) wino_input_pair(x, hf, t);
          __builtin_amdgcn_sched_barrier(0);
          if (kh + 1 < 8) {
            const int off = 32 * (kh + 1) + 4 * ((kh + 1) >> 1);
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) x[dd] = *reinterpret_cast<const f32x4*>(ap + C22W_PLANE * dd + off);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int k = 0; k < 4; ++k)
              acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(t[k][e >> 1][e & 1], G[k][kh][e], acc[k], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        f32x4 y0, y1;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const f32x2 c0 = hf ? __builtin_shufflevector(acc[0], acc[0], 2, 3) : __builtin_shufflevector(acc[0], acc[0], 0, 1);
          const f32x2 c1 = hf ? __builtin_shufflevector(acc[1], acc[1], 2, 3) : __builtin_shufflevector(acc[1], acc[1], 0, 1);
          const f32x2 c2 = hf ? __builtin_shufflevector(acc[2], acc[2], 2, 3) : __builtin_shufflevector(acc[2], acc[2], 0, 1);
          const f32x2 c3 = hf ? __builtin_shufflevector(acc[3], acc[3], 2, 3) : __builtin_shufflevector(acc[3], acc[3], 0, 1);
          const f32x2 s0 = pk_add(pk_add(c0, c1), c2), s1 = pk_sub(pk_sub(c1, c2), c3);
          y0[2 * hf] = s0[0];
          y0[2 * hf + 1] = s0[1];
          y1[2 * hf] = s1[0];
          y1[2 * hf + 1] = s1[1];
        }
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
    p.stamps[((size_t)blockIdx.x * 4 + wave) * 4 + 3] = wave == 0 ? __builtin_amdgcn_s_memtime() - clk_c0
                                                                    : wave == 1 ? __builtin_amdgcn_s_memrealtime() - clk_r0 : 0ull;
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
  float* out;           // [n][10][15][5][64], or chunked and column-major: [n][10][8 chunks][5 w][15 h][8] (what svk_c3d2_conv32t stages)
  int32_t n_utt;
  unsigned* queue;      // work-item counter (zeroed before the launch), or NULL
  int32_t chunked;
};

template <bool SLOPE01, bool CHUNKED>
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
  const float b = p.bias[16 * nt + i], sl = p.slope[16 * nt + i];
  // A row of this lane: position m = i -> (row hl = m / 5, column w' = m % 5); m = 15 is a dummy (reads position 14)
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
#pragma unroll 1
    for (int P = 0; P < 5; ++P) {
      const float* ap = a0 + 2 * C31_PLANE * P;
      f32x4 acc[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = k == 1 ? (f32x4){b, b, b, b} : (f32x4){0.f, 0.f, 0.f, 0.f};
      f32x4 x[4];
      f32x2 t[4][2];
#pragma unroll
      for (int dd = 0; dd < 4; ++dd) x[dd] = *reinterpret_cast<const f32x4*>(ap + C31_PLANE * dd);
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
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int k = 0; k < 4; ++k)
            acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(t[k][e >> 1][e & 1], G[k][kw][ch][e], acc[k], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // rows 4 kk + r = position m -> (row 3 rb + m / 5, column m % 5); column i = channel 16 nt + i; depths 2 P, 2 P + 1
      float* const o = CHUNKED ? p.out + (((int64_t)u * 10 + 2 * P) * 8 + 2 * nt) * (5 * 15 * 8)
                               : p.out + (((int64_t)u * 10 + 2 * P) * 15 + 3 * rb) * (5 * 64) + 16 * nt;   // wave-uniform
      const int olane = CHUNKED ? (i >> 3) * (5 * 15 * 8) + (i & 7) + 3 * rb * 8 : 4 * kk * 64 + i;
      constexpr int ostep = CHUNKED ? 8 * 5 * 15 * 8 : 15 * 5 * 64;   // one output depth further
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const f32x2 c0 = hf ? __builtin_shufflevector(acc[0], acc[0], 2, 3) : __builtin_shufflevector(acc[0], acc[0], 0, 1);
        const f32x2 c1 = hf ? __builtin_shufflevector(acc[1], acc[1], 2, 3) : __builtin_shufflevector(acc[1], acc[1], 0, 1);
        const f32x2 c2 = hf ? __builtin_shufflevector(acc[2], acc[2], 2, 3) : __builtin_shufflevector(acc[2], acc[2], 0, 1);
        const f32x2 c3 = hf ? __builtin_shufflevector(acc[3], acc[3], 2, 3) : __builtin_shufflevector(acc[3], acc[3], 0, 1);
        const f32x2 y0 = pk_add(pk_add(c0, c1), c2), y1 = pk_sub(pk_sub(c1, c2), c3);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int m = 4 * kk + 2 * hf + q;
          if (m < 15) {
            // channels last: positions are contiguous, (row, column) = m; chunked: column m % 5 outermost, then row m / 5:
            // (m % 5) * 15 + m / 5 for m = 4 kk + r as byte kk of a per-r constant (a division per store cost 10 % of the kernel)
            int kk_e = kk;
            if (CHUNKED) asm volatile("" : "+v"(kk_e));   // (keeps the eight per-lane offsets out of registers this kernel spills)
            auto cpos = [](int m2) { return (m2 % 5) * 15 + m2 / 5; };
            const unsigned tab = (unsigned)cpos(2 * hf + q) | ((unsigned)cpos(4 + 2 * hf + q) << 8) |
                                 ((unsigned)cpos(8 + 2 * hf + q) << 16) | ((unsigned)cpos(12 + 2 * hf + q < 15 ? 12 + 2 * hf + q : 0) << 24);
            const int opos = CHUNKED ? (int)((tab >> (8 * kk_e)) & 255u) * 8 : (2 * hf + q) * 64;
            o[olane + opos] = prelu_t<SLOPE01>(y0[q], sl);
            o[ostep + olane + opos] = prelu_t<SLOPE01>(y1[q], sl);
          }
        }
      }
    }
    __syncthreads();
    item = item_next;
  }
}

// ---- conv3_2 (64 -> 64, kernel (3,7,1)) + BN + PReLU (model.py:129-131, :162-164), depth-transformed.  Its transformed
// weights are 4 x 7 x 64 x 64 floats = 448 KB: sixteen (N tile, 16-channel K chunk) slabs of 112 VGPRs x 64 lanes.  A
// workgroup of EIGHT waves takes two N tiles (role = one of two N tile pairs) with their four K chunks each -- wave =
// (N tile of the pair, K chunk) -- so the four waves of an N tile produce partial sums over a quarter of K each and add
// them up through LDS; the two roles read the same input (from their XCD's L2).
// Item = (cube, depth pair): the pair's four input planes (15 x 5 pixels of 64 channels, pixel stride 68 floats = 17
// sixteen-byte slots) sit in LDS; no taps along w, so the 45 output positions (9 rows x 5 columns) are CONSECUTIVE pixels
// and row tap kh is a shift by 5 pixels: M tile t = pixels 16 t .. 16 t + 15 (three tiles, the last with 3 dummies),
// every fragment address an immediate offset from one lane base.  One workgroup per CU (131 KB of LDS), two waves per
// SIMD (with four waves -- one N tile per workgroup, half the LDS but still one workgroup per CU -- every LDS wait, the
// parking of the planes and the reduction ran with nothing beside them: 0.49 ms per 1 024 cubes); the next item's planes
// are fetched into registers during the MFMAs and parked after the barrier, like the first block's patch. ----
constexpr int C32_PIXF = 68;
constexpr int C32_PLANE = 75 * C32_PIXF;                       // floats per input plane in LDS
constexpr int C32_IN_FLOATS = 4 * C32_PLANE + 16 * C32_PIXF;   // + slack: the dummy rows of the last tile read past plane 3
constexpr int C32_XCH_FLOATS = 8 * 3 * 2 * 64 * 4;             // [wave][tile][y][lane] f32x4
constexpr int C32_LDS_FLOATS = C32_IN_FLOATS + C32_XCH_FLOATS;
constexpr int C32_NV = 10;                                     // 16-byte pieces per thread and item (4 800 / 512, rounded up)

struct Conv32Params {
  const float* in;      // [n][10][15][5][64]
  const f32x4* wfrag;   // [4 nt][21 taps][4 chunks][64]: e: W[16 nt + (l & 15)][16 chunk + 4 (l >> 4) + e][kd][kh], tap = 7 kd + kh
  const float* bias;    // [64]
  const float* slope;   // [64]
  float* out;           // [n][8][9][5][64], or chunked: [n][8][8 chunks of 8 channels][45][8] (what svk_c3d2_conv41 stages)
  int32_t n_utt;
  int32_t chunked;
};

template <bool SLOPE01>
__global__ __launch_bounds__(512) void c3d2_conv32w_kernel(const Conv32Params p) {
  extern __shared__ __attribute__((aligned(16))) float smem_c32[];
  float* reg = smem_c32;
  float* exch = reg + C32_IN_FLOATS;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, kk = lane >> 4;
  const int ch = wave & 3, half = wave >> 2;   // this wave's K chunk; which of the workgroup's two N tiles
  // role (N tile pair) and item group of this workgroup.  Workgroups go to the 8 XCDs round-robin (blockIdx % 8) and each
  // XCD has its own L2: the two roles of a group read the same planes, so they are given the same XCD when the grid allows
  int role, group;
  if ((gridDim.x & 15) == 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    role = slot & 1;
    group = xcd + 8 * (slot >> 1);
  } else {
    role = blockIdx.x & 1;
    group = blockIdx.x >> 1;
  }
  const int nt = 2 * role + half;
  f32x4 G[4][7];   // [k][kh], this wave's N tile and K chunk
#pragma unroll
  for (int kh = 0; kh < 7; ++kh) {
    const f32x4 g0 = p.wfrag[((nt * 21 + kh) * 4 + ch) * 64 + lane], g1 = p.wfrag[((nt * 21 + 7 + kh) * 4 + ch) * 64 + lane],
                g2 = p.wfrag[((nt * 21 + 14 + kh) * 4 + ch) * 64 + lane];
    G[0][kh] = g0;
    G[1][kh] = 0.5f * ((g0 + g2) + g1);
    G[2][kh] = 0.5f * ((g0 + g2) - g1);
    G[3][kh] = g2;
  }
  const float b = ch == 0 ? p.bias[16 * nt + i] : 0.f, sl = p.slope[16 * nt + i];
  const float* const a0 = reg + C32_PIXF * i + 16 * ch + 4 * kk;   // pixel i of plane 0, this wave's chunk, this lane's K piece
  const int n_items = p.n_utt * 4;                                  // (cube, pair)
  const int stride = gridDim.x >> 1;                                // workgroups per role
  f32x4 pre[C32_NV];
  auto fetch = [&](int item) {   // the pair's four planes are 19 200 contiguous floats
    const float* src = p.in + ((int64_t)(item >> 2) * 10 + 2 * (item & 3)) * (15 * 5 * 64);
#pragma unroll
    for (int k = 0; k < C32_NV; ++k) {
      const int e = threadIdx.x + 512 * k;
      if (e < 4800) pre[k] = *reinterpret_cast<const f32x4*>(src + 4 * e);
    }
  };
  auto park = [&]() {
#pragma unroll
    for (int k = 0; k < C32_NV; ++k) {
      const int e = threadIdx.x + 512 * k;
      if (e < 4800) *reinterpret_cast<f32x4*>(reg + C32_PIXF * (e >> 4) + 4 * (e & 15)) = pre[k];
    }
  };
  int item = group;
  if (item < n_items) {
    fetch(item);
    park();
  }
  __syncthreads();
  for (; item < n_items; item += stride) {
    const int next = item + stride;
    if (next < n_items) fetch(next);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int tl = 0; tl < 3; ++tl) {
      const float* ap = a0 + 16 * C32_PIXF * tl;
      f32x4 acc[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = k == 1 ? (f32x4){b, b, b, b} : (f32x4){0.f, 0.f, 0.f, 0.f};
      f32x4 x[4];
      f32x2 t[4][2];
#pragma unroll
      for (int dd = 0; dd < 4; ++dd) x[dd] = *reinterpret_cast<const f32x4*>(ap + C32_PLANE * dd);
#pragma unroll
      for (int kh = 0; kh < 7; ++kh) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) wino_input_pair(x, hf, t);
        __builtin_amdgcn_sched_barrier(0);
        if (kh + 1 < 7) {
#pragma unroll
          for (int dd = 0; dd < 4; ++dd) x[dd] = *reinterpret_cast<const f32x4*>(ap + C32_PLANE * dd + 5 * C32_PIXF * (kh + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int k = 0; k < 4; ++k)
            acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(t[k][e >> 1][e & 1], G[k][kh][e], acc[k], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      f32x4 y0, y1;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const f32x2 c0 = hf ? __builtin_shufflevector(acc[0], acc[0], 2, 3) : __builtin_shufflevector(acc[0], acc[0], 0, 1);
        const f32x2 c1 = hf ? __builtin_shufflevector(acc[1], acc[1], 2, 3) : __builtin_shufflevector(acc[1], acc[1], 0, 1);
        const f32x2 c2 = hf ? __builtin_shufflevector(acc[2], acc[2], 2, 3) : __builtin_shufflevector(acc[2], acc[2], 0, 1);
        const f32x2 c3 = hf ? __builtin_shufflevector(acc[3], acc[3], 2, 3) : __builtin_shufflevector(acc[3], acc[3], 0, 1);
        const f32x2 s0 = pk_add(pk_add(c0, c1), c2), s1 = pk_sub(pk_sub(c1, c2), c3);
        y0[2 * hf] = s0[0];
        y0[2 * hf + 1] = s0[1];
        y1[2 * hf] = s1[0];
        y1[2 * hf + 1] = s1[1];
      }
      float* xo = exch + (((wave * 3 + tl) * 2) * 64 + lane) * 4;
      *reinterpret_cast<f32x4*>(xo) = y0;
      *reinterpret_cast<f32x4*>(xo + 256) = y1;
    }
    __syncthreads();   // every wave's partial sums are in LDS; nobody reads the input planes any more
    if (next < n_items) park();
    // the six (tile, y) units of the item and N tile: wave (half, ch) adds up units ch and ch + 4 (the partial sums of its
    // half's four waves), PReLU, stores
    {
      const int u = item >> 2, P = item & 3;
#pragma unroll
      for (int rep = 0; rep < 2; ++rep) {
        const int unit = ch + 4 * rep;   // = 2 tile + y
        if (unit < 6) {
          const float* xi = exch + ((half * 4 * 6 + unit) * 64 + lane) * 4;
          f32x4 v = *reinterpret_cast<const f32x4*>(xi);
#pragma unroll
          for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const f32x4*>(xi + w * (6 * 64 * 4));
          const int tl = unit >> 1, y = unit & 1;
          // rows 4 kk + r = position 16 tl + 4 kk + r (< 45): positions are contiguous in the output, channels last -- or
          // chunked, [depth][chunk = channel / 8][position][channel % 8] (wave-uniform choice)
          float* const o = p.chunked ? p.out + ((((int64_t)u * 8 + 2 * P + y) * 8 + 2 * nt) * 45 + 16 * tl) * 8
                                     : p.out + (((int64_t)u * 8 + 2 * P + y) * 45 + 16 * tl) * 64 + 16 * nt;
          const int olane = p.chunked ? (i >> 3) * (45 * 8) + 4 * kk * 8 + (i & 7) : 4 * kk * 64 + i;
          const int rstep = p.chunked ? 8 : 64;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (16 * tl + 4 * kk + r < 45) o[olane + r * rstep] = prelu_t<SLOPE01>(v[r], sl);
        }
      }
    }
    __syncthreads();   // the next planes are in place; the exchange buffer may be overwritten
  }
}


}  // namespace

extern "C" int svk_c3d2_stage2(svk_ctx* ctx, const float* d_in, int32_t n_utt, const float* d_w21frag,
                               const float* d_bias21, const float* d_slope21, const float* d_w22frag,
                               const float* d_bias22, const float* d_slope22, int32_t flags, float* d_act2, float* d_out) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, n_utt >= 0, "n_utt negative");
  if (n_utt == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_in && d_w21frag && d_bias21 && d_slope21 && d_w22frag && d_bias22 && d_slope22 && d_act2 && d_out,
              "NULL buffer");
  SVK_REQUIRE(ctx, ((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_act2) |
                     reinterpret_cast<uintptr_t>(d_w21frag) | reinterpret_cast<uintptr_t>(d_w22frag)) & 15) == 0,
              "buffers must be 16-byte aligned");
  SVK_REQUIRE(ctx, (int64_t)n_utt * 21 < ((int64_t)1 << 31), "too many cubes for one launch");
  {
    // work-item counters of the kernels that share a CU between workgroups (slots of the handle's 256-byte scratch; svk_log_power
    // owns the first word): zeroed in stream order before the launches
    const bool static_items = getenv("SVK_C3D2_STATIC_ITEMS") != nullptr;
    unsigned* const queues = static_items ? nullptr : reinterpret_cast<unsigned*>(static_cast<char*>(ctx->scratch) + 64);
    if (queues) SVK_HIP(ctx, hipMemsetAsync(queues, 0, 16, ctx->stream));
    const bool wino = (flags & 1) != 0;   // conv2_1 through the depth transform
    Conv21Params p{d_in, reinterpret_cast<const f32x4*>(d_w21frag), d_bias21, d_slope21, d_act2, n_utt, wino ? queues : nullptr};
    void (*kern)(const Conv21Params) = !wino ? c3d2_conv21_kernel
                                       : (flags & 2) ? c3d2_conv21w_kernel<true> : c3d2_conv21w_kernel<false>;
    const size_t lds = sizeof(float) * (size_t)(wino ? C21W_LDS_FLOATS : C21_LDS_FLOATS);
    if (lds > (size_t)ctx->lds_per_cu)
      return svk_fail(ctx, SVK_ERR_UNSUPPORTED, "svk_c3d2_stage2 (conv2_1) needs %zu bytes of LDS per workgroup (device: %d)",
                      lds, ctx->lds_per_cu);
    SVK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t items = (int64_t)n_utt * (S2_H / (wino ? C21W_TH : C21_TH));
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 256, lds) != hipSuccess ||
        per_cu < 1)
      per_cu = 2;
    hipLaunchKernelGGL(kern, dim3((unsigned)std::min<int64_t>(items, (int64_t)per_cu * ctx->num_cu)), dim3(256), lds,
                       ctx->stream, p);
    SVK_LAUNCH_CHECK(ctx);
  }
  {
    const bool static_items22 = getenv("SVK_C3D2_STATIC_ITEMS") != nullptr;
    Conv22Params p{d_act2, reinterpret_cast<const f32x4*>(d_w22frag), d_bias22, d_slope22, d_out, n_utt, nullptr,
                   (flags & 4) && !static_items22 ? reinterpret_cast<unsigned*>(static_cast<char*>(ctx->scratch) + 68) : nullptr};
    if (flags & 4) {   // conv2_2 through the depth transform
      void (*kern)(const Conv22Params) = (flags & 2) ? c3d2_conv22w_kernel<true> : c3d2_conv22w_kernel<false>;
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
      return SVK_OK;
    }
    const size_t lds = sizeof(float) * (size_t)C22_LDS_FLOATS;
    if (lds > (size_t)ctx->lds_per_cu)
      return svk_fail(ctx, SVK_ERR_UNSUPPORTED, "svk_c3d2_stage2 (conv2_2) needs %zu bytes of LDS per workgroup (device: %d)",
                      lds, ctx->lds_per_cu);
    SVK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(c3d2_conv22_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t items = (int64_t)n_utt * (O2_W * 2);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(c3d2_conv22_kernel), 256, lds) !=
            hipSuccess || per_cu < 1)
      per_cu = 2;
    const unsigned grid = (unsigned)std::min<int64_t>(items, (int64_t)per_cu * ctx->num_cu);
#ifdef SVK_TUNING
    const bool want_stamps = getenv("SVK_C3D2_STAMPS") != nullptr;
    const size_t stamp_bytes = (size_t)grid * 4 * 4 * sizeof(unsigned long long);
    if (want_stamps) {
      const int rc = svk_ensure_work(ctx, stamp_bytes);
      if (rc != SVK_OK) return rc;
      p.stamps = reinterpret_cast<unsigned long long*>(ctx->work);
    }
#endif
    hipLaunchKernelGGL(c3d2_conv22_kernel, dim3(grid), dim3(256), lds, ctx->stream, p);
    SVK_LAUNCH_CHECK(ctx);
#ifdef SVK_TUNING
    if (want_stamps) {
      std::vector<unsigned long long> h((size_t)grid * 16);
      SVK_HIP(ctx, hipStreamSynchronize(ctx->stream));
      SVK_HIP(ctx, hipMemcpy(h.data(), p.stamps, stamp_bytes, hipMemcpyDeviceToHost));
      const char* names[4] = {"staging", "barrier 1", "MFMA phase + epilogues", "barrier 2"};
      const double per = (double)items / grid;
      for (int w = 0; w < 4; ++w) {
        fprintf(stderr, "conv22 stamps wave %d (cycles per item, %d workgroups per CU):", w, per_cu);
        for (int k = 0; k < 4; ++k) {
          double sum = 0;
          for (unsigned b = 0; b < grid; ++b) sum += (double)h[((size_t)b * 4 + w) * 4 + k];
          fprintf(stderr, "  %s %.0f", names[k], sum / grid / per);
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
  if (n_utt == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_in && d_wfrag && d_bias && d_slope && d_out, "NULL buffer");
  SVK_REQUIRE(ctx, ((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_wfrag)) & 15) == 0,
              "buffers must be 16-byte aligned");
  SVK_REQUIRE(ctx, (int64_t)n_utt * 5 < ((int64_t)1 << 31), "too many cubes for one launch");
  const bool static_items31 = getenv("SVK_C3D2_STATIC_ITEMS") != nullptr;
  unsigned* const queue31 = static_items31 ? nullptr : reinterpret_cast<unsigned*>(static_cast<char*>(ctx->scratch) + 80);
  if (queue31) SVK_HIP(ctx, hipMemsetAsync(queue31, 0, 4, ctx->stream));
  Conv31Params p{d_in, reinterpret_cast<const f32x4*>(d_wfrag), d_bias, d_slope, d_out, n_utt, queue31, (flags & 8) ? 1 : 0};
  void (*kern)(const Conv31Params) = (flags & 8) ? ((flags & 2) ? c3d2_conv31w_kernel<true, true> : c3d2_conv31w_kernel<false, true>)
                                                 : ((flags & 2) ? c3d2_conv31w_kernel<true, false> : c3d2_conv31w_kernel<false, false>);
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

extern "C" int svk_c3d2_conv32(svk_ctx* ctx, const float* d_in, int32_t n_utt, const float* d_wfrag, const float* d_bias,
                               const float* d_slope, int32_t flags, float* d_out) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, n_utt >= 0, "n_utt negative");
  if (n_utt == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_in && d_wfrag && d_bias && d_slope && d_out, "NULL buffer");
  SVK_REQUIRE(ctx, ((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_wfrag)) & 15) == 0,
              "buffers must be 16-byte aligned");
  SVK_REQUIRE(ctx, (int64_t)n_utt * 4 < ((int64_t)1 << 29), "too many cubes for one launch");
  Conv32Params p{d_in, reinterpret_cast<const f32x4*>(d_wfrag), d_bias, d_slope, d_out, n_utt, (flags & 8) ? 1 : 0};
  void (*kern)(const Conv32Params) = (flags & 2) ? c3d2_conv32w_kernel<true> : c3d2_conv32w_kernel<false>;
  const size_t lds = sizeof(float) * (size_t)C32_LDS_FLOATS;
  if (lds > (size_t)ctx->lds_per_cu)
    return svk_fail(ctx, SVK_ERR_UNSUPPORTED, "svk_c3d2_conv32 needs %zu bytes of LDS per workgroup (device: %d)", lds,
                    ctx->lds_per_cu);
  SVK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // two roles (N tile pairs) x up to num_cu / 2 workgroups each: one workgroup of eight waves per CU
  const int64_t items = (int64_t)n_utt * 4;
  const int64_t per_role = std::max<int64_t>(1, std::min<int64_t>(items, ctx->num_cu / 2));
  hipLaunchKernelGGL(kern, dim3((unsigned)(2 * per_role)), dim3(512), lds, ctx->stream, p);
  SVK_LAUNCH_CHECK(ctx);
  return SVK_OK;
}

// =====================================================================================================
// conv3_1 .. conv4_2 run on the host framework's convolutions (GEMM-shaped layers: K = 288 .. 2 688, N = 64 / 128);
// what follows each of them -- + bias (BatchNorm folded), PReLU (model.py:159-167) -- is ONE in-place pass here
// instead of the framework's two (a bias add and a PReLU kernel: 0.28 ms of the 6.6 ms per 1 024 cubes).
// x: [rows][channels] (channels-last activations), channels a multiple of 4.
// =====================================================================================================
namespace {

__global__ __launch_bounds__(256) void bias_prelu_kernel(float* __restrict__ x, int64_t n_vec, int c4,
                                                         const float* __restrict__ bias, const float* __restrict__ slope) {
  // thread -> 16-byte pieces; a piece's channel group is (index mod c4): the grid stride is a multiple of c4,
  // so a thread keeps ONE channel group and its bias / slope stay in registers
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int g = (int)(i % c4);
  const f32x4 b = *reinterpret_cast<const f32x4*>(bias + 4 * g);
  const f32x4 s = *reinterpret_cast<const f32x4*>(slope + 4 * g);
  f32x4* p = reinterpret_cast<f32x4*>(x);
  for (; i < n_vec; i += stride) {
    f32x4 v = p[i] + b;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = prelu(v[e], s[e]);
    p[i] = v;
  }
}

}  // namespace

extern "C" int svk_bias_prelu(svk_ctx* ctx, float* d_x, int64_t n_rows, int32_t n_channels, const float* d_bias,
                              const float* d_slope) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, n_rows >= 0 && n_channels >= 1, "shape");
  if (n_rows == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_x && d_bias && d_slope, "NULL buffer");
  if ((n_channels & 3) || n_channels > 1024 ||
      ((reinterpret_cast<uintptr_t>(d_x) | reinterpret_cast<uintptr_t>(d_bias) | reinterpret_cast<uintptr_t>(d_slope)) & 15))
    return svk_fail(ctx, SVK_ERR_UNSUPPORTED, "svk_bias_prelu: channels must be a multiple of 4 (<= 1024), buffers 16-byte aligned");
  const int c4 = n_channels / 4;
  const int64_t n_vec = n_rows * c4;
  // grid stride = blocks x 256 must be a multiple of c4: blocks = a multiple of c4 / gcd(c4, 256)
  int g = c4, r = 256;
  while (r) { const int t = g % r; g = r; r = t; }
  const int unit = c4 / g;
  int64_t blocks = std::min<int64_t>((n_vec + 255) / 256, (int64_t)ctx->num_cu * 8);
  blocks = std::max<int64_t>(unit, blocks / unit * unit);
  hipLaunchKernelGGL(bias_prelu_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_x, n_vec, c4, d_bias, d_slope);
  SVK_LAUNCH_CHECK(ctx);
  return SVK_OK;
}
