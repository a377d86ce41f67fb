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
//      matrix work runs, then parked in LDS (23 KB);
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
constexpr int P_VEC = (P_FLOATS / 2 + 255) / 256;    // float2 loads per thread for a patch

struct Stage1Params {
  const float* feat;
  const int32_t* crop;
  int32_t n_utt, max_frames;
  const float* w1frag;   // [4][64]: B operand of the conv1_1 GEMM, k = 4 jj + (lane >> 4): tap (k / 5, k % 5), k = 15: bias
  const float* slope1;   // [16]
  const f32x4* w2frag;   // [27][64]: lane (co = l & 15, kk = l >> 4), element e = W[co][4 kk + e][kd][kh], tap = 9 kd + kh
  const float* bias2;    // [16]
  const float* slope2;   // [16]
  float* out;
  int64_t s_n, s_d, s_hp, s_par, s_w;  // output strides (floats): cube, depth, row pair, row parity, pooled column
};

__device__ __forceinline__ float prelu(float v, float slope) { return v > 0.f ? v : slope * v; }

// the item's cube patch: patch[dd][h][ww] = feat[u][crop[u][8 q + dd] + h][2 j + ww]
__device__ __forceinline__ void fetch_patch(const Stage1Params& p, int item, f32x2 (&regs)[P_VEC]) {
  const int u = item / 36, rem = item - u * 36, q = rem / 18, j = rem - q * 18;
  const float* base = p.feat + (int64_t)u * p.max_frames * NCOEF + 2 * j;
  const int32_t* cr = p.crop + (int64_t)u * NCROP + TD * q;
#pragma unroll
  for (int s = 0; s < P_VEC; ++s) {
    const int e = threadIdx.x + 256 * s;  // float2 index: row = e / 3 (dd * 80 + h), piece = e % 3
    f32x2 v = (f32x2){0.f, 0.f};
    if (e < P_FLOATS / 2) {
      const int row = e / 3, piece = e - row * 3;
      const int dd = row / NFRAME, h = row - dd * NFRAME;
      const int start = cr[dd];
      if (start >= 0 && start + h < p.max_frames)
        v = *reinterpret_cast<const f32x2*>(base + (int64_t)(start + h) * NCOEF + 2 * piece);
    }
    regs[s] = v;
  }
}

__device__ __forceinline__ void park_patch(float* patch, const f32x2 (&regs)[P_VEC]) {
#pragma unroll
  for (int s = 0; s < P_VEC; ++s) {
    const int e = threadIdx.x + 256 * s;
    if (e < P_FLOATS / 2) *reinterpret_cast<f32x2*>(patch + 2 * e) = regs[s];
  }
}

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
  const bool bias_lane = kk == 3;  // k = 15 (jj = 3, kk = 3): the A operand is the constant 1 (bias row of w1frag)
  const float sl1 = p.slope1[i], b2 = p.bias2[i], sl2 = p.slope2[i];
  const int pix_lane = (i >> 1) * PW + (i & 1);  // patch offset of this lane's pixel inside a conv1_1 tile (8 rows x 2 columns)

  f32x2 pre[P_VEC];
  int item = blockIdx.x;
  if (item < n_items) {
    fetch_patch(p, item, pre);
    park_patch(patch, pre);
  }
  __syncthreads();
  for (; item < n_items; item += gridDim.x) {
    const int next = item + gridDim.x;
    if (next < n_items) fetch_patch(p, next, pre);  // in flight during the matrix work below

    // ---- conv1_1 + PReLU: 100 tiles of 16 pixels (8 rows x 2 columns of one depth), 25 per wave ----
    for (int tt = wave; tt < DIN * 10; tt += 4) {
      const int din = tt / 10, hb = (tt - din * 10) * 8;
      const float* pp = patch + din * (NFRAME * PW) + hb * PW + pix_lane;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        float a = pp[tapoff[jj]];
        if (jj == 3) a = bias_lane ? 1.0f : a;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w1[jj], acc, 0, 0, 0);
      }
      // rows 4 kk + r of the tile = pixels p0 + 4 kk + r, column i = channel; p0 = (din * 80 + hb) * 2 (a multiple of 16)
      float* ap = act + 17 * ((din * NFRAME + hb) * 2) + 68 * kk + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) ap[16 * r] = prelu(acc[r], sl1);
    }
    __syncthreads();  // act1 is complete; the patch buffer is free

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
#pragma unroll
      for (int t = 0; t < N_TAPS; ++t) {
        const int kd = t / 9, kh = t - kd * 9;
        const int off = 2720 * kd + 32 * kh + 4 * (kh >> 1);  // 16 dp + 4 d(p >> 2) for dp = 160 kd + 2 kh pixels
        f32x4 a[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) a[s] = *reinterpret_cast<const f32x4*>(ab[s] + off);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int s = 0; s < 3; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][e], w2[t][e], acc[s], 0, 0, 0);
      }
      // rows 4 kk + r: depth dl' = kk >> 1, row hl' = 2 (kk & 1) + (r >> 1), column r & 1: pool = max over r pairs
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int hp = 2 * (3 * g + s) + (kk & 1);  // output row pair: rows 4 hg + 2 (kk & 1) + {0, 1}
        const float v0 = fmaxf(prelu(acc[s][0] + b2, sl2), prelu(acc[s][1] + b2, sl2));
        const float v1 = fmaxf(prelu(acc[s][2] + b2, sl2), prelu(acc[s][3] + b2, sl2));
        float* o = obase + (int64_t)hp * p.s_hp;
        o[0] = v0;
        o[p.s_par] = v1;
      }
    }
    if (next < n_items) park_patch(patch, pre);
    __syncthreads();  // the next patch is in place; act1 may be overwritten
  }
}

}  // namespace

extern "C" {

size_t svk_c3d2_stage1_lds_bytes(void) { return sizeof(float) * (size_t)(ACT_FLOATS + P_FLOATS); }

int svk_c3d2_stage1(svk_ctx* ctx, const float* d_feat, int32_t n_utt, int32_t max_frames, int32_t n_cols,
                    const int32_t* d_crop_idx, int32_t n_crops, int32_t crop_frames, const float* d_w1frag,
                    const float* d_slope1, const float* d_w2frag, const float* d_bias2, const float* d_slope2,
                    int32_t folded, float* d_out) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, n_utt >= 0 && max_frames >= 1, "shape");
  if (n_cols != NCOEF || n_crops != NCROP || crop_frames != NFRAME)
    return svk_fail(ctx, SVK_ERR_UNSUPPORTED,
                    "svk_c3d2_stage1 is built for the 20 x 80 x 40 cube of utils.py:20-21 (got %d x %d x %d)", n_crops,
                    crop_frames, n_cols);
  if (n_utt == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_feat && d_crop_idx && d_w1frag && d_slope1 && d_w2frag && d_bias2 && d_slope2 && d_out, "NULL buffer");
  SVK_REQUIRE(ctx, (reinterpret_cast<uintptr_t>(d_feat) & 7) == 0 && (reinterpret_cast<uintptr_t>(d_w2frag) & 15) == 0,
              "d_feat must be 8-byte and d_w2frag 16-byte aligned");
  SVK_REQUIRE(ctx, (int64_t)n_utt * 36 < ((int64_t)1 << 31), "too many cubes for one launch");
  Stage1Params p;
  p.feat = d_feat;
  p.crop = d_crop_idx;
  p.n_utt = n_utt;
  p.max_frames = max_frames;
  p.w1frag = d_w1frag;
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
  const size_t lds = svk_c3d2_stage1_lds_bytes();
  if (lds > (size_t)ctx->lds_per_cu)
    return svk_fail(ctx, SVK_ERR_UNSUPPORTED, "svk_c3d2_stage1 needs %zu bytes of LDS per workgroup (device: %d)", lds,
                    ctx->lds_per_cu);
  SVK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(c3d2_stage1_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t items = (int64_t)n_utt * 36;
  const unsigned grid = (unsigned)std::min<int64_t>(items, ctx->num_cu);  // one persistent workgroup per CU
  hipLaunchKernelGGL(c3d2_stage1_kernel, dim3(grid), dim3(256), lds, ctx->stream, p);
  SVK_LAUNCH_CHECK(ctx);
  return SVK_OK;
}

}  // extern "C"
