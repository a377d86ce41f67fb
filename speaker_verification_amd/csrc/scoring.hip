// Scoring kernels.
//   svk_cosine_scores <- /root/reference/evaluation.py:67-84: the reference loops over
//       (utterance, speaker) pairs calling sklearn's cosine_similarity on two (1,128)
//       float32 rows (Q18); here the whole [n_test x n_enroll] matrix is one launch on
//       v_mfma_f32_16x16x4_f32 (exact f32 products, f32 accumulation).
//   svk_l2_dist       <- /root/reference/siamese.py:29-30.
#include <algorithm>

#include "svk_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// One wave = one 16-row tile of the test matrix; it walks all 16-row tiles of the enrolled
// matrix.  A/B fragments come straight from HBM/L2 as 16-byte loads: lane (i = l & 15,
// kk = l >> 4) reads row i, floats [16 u + 4 kk, +4) of chunk u, and MFMA step (u, e) uses
// element e of both operands, so the K order is permuted identically on both sides.
// The same registers give the squared row norms (reduced over kk with two shuffles).
__device__ __forceinline__ f32x4 load4(const float* row, int col, int dim, bool row_ok, bool vec_ok) {
  f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (!row_ok) return v;
  if (vec_ok && col + 4 <= dim) return *reinterpret_cast<const f32x4*>(row + col);
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (col + e < dim) v[e] = row[col + e];
  return v;
}

__global__ __launch_bounds__(256) void cosine_kernel(const float* __restrict__ test, const float* __restrict__ enroll,
                                                     int nt, int ns, int dim, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kk = lane >> 4;
  const int nchunk = (dim + 15) >> 4;
  const bool vec_ok = (dim & 3) == 0 && ((reinterpret_cast<uintptr_t>(test) | reinterpret_cast<uintptr_t>(enroll)) & 15) == 0;
  const int n_ttiles = (nt + 15) >> 4, n_stiles = (ns + 15) >> 4;
  for (int tt = blockIdx.x * 4 + wave; tt < n_ttiles; tt += gridDim.x * 4) {
    const int trow = tt * 16 + i;
    const bool t_ok = trow < nt;
    const float* tp = test + (int64_t)trow * dim;
    // squared norm of this lane's test row
    float tn = 0.f;
    for (int u = 0; u < nchunk; ++u) {
      const f32x4 a = load4(tp, 16 * u + 4 * kk, dim, t_ok, vec_ok);
      tn += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
    }
    tn += __shfl_xor(tn, 16, 64);
    tn += __shfl_xor(tn, 32, 64);
    float tnorm = sqrtf(tn);
    tnorm = tnorm == 0.f ? 1.f : tnorm;  // sklearn normalize(): a zero norm divides by 1
    for (int st = 0; st < n_stiles; ++st) {
      const int srow = st * 16 + i;
      const bool s_ok = srow < ns;
      const float* sp = enroll + (int64_t)srow * dim;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      float sn = 0.f;
      for (int u = 0; u < nchunk; ++u) {
        const f32x4 a = load4(tp, 16 * u + 4 * kk, dim, t_ok, vec_ok);
        const f32x4 b = load4(sp, 16 * u + 4 * kk, dim, s_ok, vec_ok);
        sn += b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc, 0, 0, 0);
      }
      sn += __shfl_xor(sn, 16, 64);
      sn += __shfl_xor(sn, 32, 64);
      float snorm = sqrtf(sn);
      snorm = snorm == 0.f ? 1.f : snorm;
      // acc[r] = dot(test row 16 tt + 4 kk + r, enroll row 16 st + i)
      const int col = st * 16 + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = tt * 16 + 4 * kk + r;
        const float rn = __shfl(tnorm, 4 * kk + r, 64);  // lane (4 kk + r) holds that test row's norm
        if (row < nt && col < ns) out[(int64_t)row * ns + col] = acc[r] / (rn * snorm);
      }
    }
  }
}

// Large problems (dev-set scale: 148 642 x 1 211): a register / LDS tiled version of the same product.
// A workgroup owns 128 test rows (each of its 4 waves two 16-row tiles, whose fragments stay in
// registers for the whole kernel when dim <= 128) and walks the enrolled matrix 32 rows at a time; the
// 32 x 128 enrolled block is staged in LDS once per workgroup (rows padded to 136 floats: conflict-free
// ds_read_b128) and double-buffered: the next block's global loads are issued before the 128 MFMAs of
// the current one and written to the other buffer after them.  16 MFMAs per ds_read_b128 pair instead
// of 4 per pair of 16-byte global loads.
// The 1 / norm of the enrolled rows comes from a one-wave-per-row pre-pass (inv_norm_kernel): summing
// squares inside the main loop cost 64 VALU instructions per 128 MFMAs, and on gfx950 VALU and MFMA
// issue never overlap.  Blocks that lie wholly inside both matrices (all but the last row / column
// block) take wave-uniform fast paths for the loads and the stores: no per-element predicates.
constexpr int CT_BM = 128, CT_BN = 32, CT_KB = 128, CT_LD = CT_KB + 8;

// out[row] = 1 / ||x[row]||, a zero norm divides by 1 (sklearn normalize()).  One wave per row.
__global__ __launch_bounds__(256) void inv_norm_kernel(const float* __restrict__ x, int n, int dim,
                                                       float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < n; row += gridDim.x * 4) {
    const float* p = x + (int64_t)row * dim;
    float s = 0.f;
    for (int k = lane; k < dim; k += 64) s = fmaf(p[k], p[k], s);
    s = wave_sum(s);
    if (lane == 0) out[row] = s == 0.f ? 1.f : 1.0f / sqrtf(s);
  }
}

template <bool HOIST>
__global__ __launch_bounds__(256) void cosine_tiled_kernel(const float* __restrict__ test, const float* __restrict__ enroll,
                                                           const float* __restrict__ einv, int nt, int ns, int dim,
                                                           float* __restrict__ out, int gy, int n_full, int tail_split) {
  __shared__ __attribute__((aligned(16))) float bs[2][CT_BN * CT_LD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kk = lane >> 4;
  const int nkb = (dim + CT_KB - 1) / CT_KB;
  const bool vec_ok = (dim & 3) == 0 && ((reinterpret_cast<uintptr_t>(test) | reinterpret_cast<uintptr_t>(enroll)) & 15) == 0;
  // Work items = (128-row block, 1 / gy of the enrolled range): gy > 1 lets a few tall row blocks
  // still fill the chip.  The items of the last, partly filled round of workgroups are cut into
  // `tail_split` pieces each, so that round costs 1 / tail_split of a full one (at the dev-set
  // shape 1 162 items on 768 resident workgroups: 2 rounds -> 1.5).
  int item = blockIdx.x, part = 0, parts = 1;
  if (item >= n_full) {
    const int t = item - n_full;
    item = n_full + t / tail_split;
    part = t % tail_split;
    parts = tail_split;
  }
  const int m0 = (item / gy) * CT_BM + wave * 32;
  const int all_stiles = (ns + CT_BN - 1) / CT_BN;
  const int per_y = (all_stiles + gy - 1) / gy;
  int st_begin = (item % gy) * per_y;
  int n_stiles = min(all_stiles, st_begin + per_y);
  if (parts > 1) {
    const int sub = (max(n_stiles - st_begin, 0) + parts - 1) / parts;
    st_begin += part * sub;
    n_stiles = min(n_stiles, st_begin + sub);
  }

  // staging assignment: thread t moves 4 float4 of the 32 x 128 block: row = (t >> 5) + 8 j, float4 column = t & 31
  const int srow = threadIdx.x >> 5, scol = (threadIdx.x & 31) * 4;
  auto fetch = [&](int st, int kb, f32x4 (&regs)[4]) {
    if (vec_ok && (st + 1) * CT_BN <= ns && (kb + 1) * CT_KB <= dim) {  // workgroup-uniform: whole block inside
      const float* base = enroll + ((int64_t)st * CT_BN + srow) * dim + kb * CT_KB + scol;
#pragma unroll
      for (int j = 0; j < 4; ++j) regs[j] = *reinterpret_cast<const f32x4*>(base + (int64_t)(8 * j) * dim);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = st * CT_BN + srow + 8 * j;
        regs[j] = load4(enroll + (int64_t)r * dim, kb * CT_KB + scol, dim, r < ns, vec_ok);
      }
    }
  };
  auto stash = [&](float* buf, const f32x4 (&regs)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(buf + (srow + 8 * j) * CT_LD + scol) = regs[j];
  };

  // test-row fragments and norms
  f32x4 a[2][8];
  float tn[2] = {0.f, 0.f};
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int row = m0 + 16 * rt + i;
    const float* tp = test + (int64_t)row * dim;
    for (int kb = 0; kb < nkb; ++kb) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const f32x4 v = load4(tp, kb * CT_KB + 16 * u + 4 * kk, dim, row < nt, vec_ok);
        tn[rt] += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        if (HOIST) a[rt][u] = v;
      }
    }
    tn[rt] += __shfl_xor(tn[rt], 16, 64);
    tn[rt] += __shfl_xor(tn[rt], 32, 64);
    tn[rt] = sqrtf(tn[rt]);
    tn[rt] = tn[rt] == 0.f ? 1.f : tn[rt];
  }
  // 1 / norm of the four output rows this lane writes per tile (row 4 kk + r lives in lane 4 kk + r)
  float rinv[2][4];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) rinv[rt][r] = 1.0f / __shfl(tn[rt], 4 * kk + r, 64);
  const bool rows_in = m0 + 32 <= nt;  // wave-uniform
  // element offsets of this lane's first output of each (row tile, r): row (m0 + 16 rt + 4 kk + r), column i
  float* const orow = out + (int64_t)(m0 + 4 * kk) * ns + i;

  f32x4 pre[4];
  if (st_begin >= n_stiles) return;  // uniform per workgroup
  fetch(st_begin, 0, pre);
  stash(bs[0], pre);
  __syncthreads();
  int cur = 0;
  for (int st = st_begin; st < n_stiles; ++st) {
    f32x4 acc[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // this lane's two output columns: their 1 / norm is needed only after the MFMAs, load it now
    float sinv[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int col = st * CT_BN + 16 * ct + i;
      sinv[ct] = einv[col < ns ? col : ns - 1];
    }
    for (int kb = 0; kb < nkb; ++kb) {
      // prefetch the next (enroll block, K block) while this one is multiplied
      const bool last_kb = kb + 1 == nkb;
      const int nst = last_kb ? st + 1 : st, nkb_i = last_kb ? 0 : kb + 1;
      const bool more = nst < n_stiles;
      if (more) fetch(nst, nkb_i, pre);
      const float* b = bs[cur];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        f32x4 av[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          if (HOIST) {
            av[rt] = a[rt][u];
          } else {
            const int row = m0 + 16 * rt + i;
            av[rt] = load4(test + (int64_t)row * dim, kb * CT_KB + 16 * u + 4 * kk, dim, row < nt, vec_ok);
          }
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(b + (16 * ct + i) * CT_LD + 16 * u + 4 * kk);
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) {
            acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][0], bv[0], acc[rt][ct], 0, 0, 0);
            acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][1], bv[1], acc[rt][ct], 0, 0, 0);
            acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][2], bv[2], acc[rt][ct], 0, 0, 0);
            acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][3], bv[3], acc[rt][ct], 0, 0, 0);
          }
        }
      }
      if (more) stash(bs[cur ^ 1], pre);
      __syncthreads();  // everyone is done with bs[cur]; bs[cur ^ 1] is complete
      cur ^= 1;
    }
    float* const o = orow + st * CT_BN;
    if (rows_in && (st + 1) * CT_BN <= ns) {  // wave-uniform: the whole 32 x 32 block is inside: 16 plain stores
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
            o[(int64_t)(16 * rt + r) * ns + 16 * ct] = acc[rt][ct][r] * rinv[rt][r] * sinv[ct];
    } else {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const int col = st * CT_BN + 16 * ct + i;
        if (col < ns) {
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = m0 + 16 * rt + 4 * kk + r;
              if (row < nt) out[(int64_t)row * ns + col] = acc[rt][ct][r] * rinv[rt][r] * sinv[ct];
            }
          }
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void l2_dist_kernel(const float* __restrict__ a, const float* __restrict__ b, int n,
                                                      int dim, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < n; row += gridDim.x * 4) {
    const float* pa = a + (int64_t)row * dim;
    const float* pb = b + (int64_t)row * dim;
    float s = 0.f;
    for (int k = lane; k < dim; k += 64) {
      const float d = pa[k] - pb[k];
      s += d * d;
    }
    s = wave_sum(s);
    if (lane == 0) out[row] = sqrtf(s);
  }
}

}  // namespace

extern "C" {

int svk_cosine_scores(svk_ctx* ctx, const float* d_test, const float* d_enroll, int32_t n_test, int32_t n_enroll,
                      int32_t dim, float* d_out) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, n_test >= 0 && n_enroll >= 0 && dim >= 1, "negative shape");
  if (dim > 4096) return svk_fail(ctx, SVK_ERR_UNSUPPORTED, "embedding dim %d > 4096", dim);
  if (n_test == 0 || n_enroll == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_test && d_enroll && d_out, "NULL buffer");
  if ((int64_t)n_test * n_enroll >= ((int64_t)1 << 22) && n_enroll >= 64) {
    // enough work to fill the chip with 128-row workgroups: tiled kernel
    const unsigned gx = (unsigned)((n_test + CT_BM - 1) / CT_BM);
    const unsigned stiles = (unsigned)((n_enroll + CT_BN - 1) / CT_BN);
    const unsigned gy = std::max(1u, std::min(stiles, (unsigned)(2 * ctx->num_cu + gx - 1) / gx));
    const int rc = svk_ensure_work(ctx, sizeof(float) * (size_t)n_enroll);
    if (rc != SVK_OK) return rc;
    float* einv = static_cast<float*>(ctx->work);
    hipLaunchKernelGGL(inv_norm_kernel, dim3((unsigned)std::min((n_enroll + 3) / 4, ctx->num_cu * 8)), dim3(256), 0,
                       ctx->stream, d_enroll, n_enroll, dim, einv);
    auto kern = dim <= CT_KB ? cosine_tiled_kernel<true> : cosine_tiled_kernel<false>;
    // resident workgroups = one "round"; the items of the last partial round are split (see the kernel)
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 256, 0) != hipSuccess ||
        per_cu < 1)
      per_cu = 2;
    const long long slots = (long long)per_cu * ctx->num_cu, items = (long long)gx * gy;
    const long long n_full = (items / slots) * slots, rem = items - n_full;
    const unsigned per_item_tiles = (stiles + gy - 1) / gy;
    // pieces per tail item: minimise (rounds of pieces) x (column blocks per piece + ~2 blocks' worth of
    // per-piece set-up: the test-row fragments and norms are loaded again)
    long long split = 1, best = -1;
    for (long long f = 1; rem > 0 && f <= std::max<long long>(1, per_item_tiles / 2); ++f) {
      const long long cost = ((rem * f + slots - 1) / slots) * ((per_item_tiles + f - 1) / f + 2);
      if (best < 0 || cost < best) best = cost, split = f;
    }
    const long long blocks = n_full + rem * split;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_test, d_enroll, einv, n_test, n_enroll,
                       dim, d_out, (int)gy, (int)n_full, (int)split);
  } else {
    const int tiles = (n_test + 15) / 16;
    const unsigned grid = (unsigned)std::max(1, std::min((tiles + 3) / 4, ctx->num_cu * 8));
    hipLaunchKernelGGL(cosine_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_test, d_enroll, n_test, n_enroll, dim,
                       d_out);
  }
  SVK_LAUNCH_CHECK(ctx);
  return SVK_OK;
}

int svk_l2_dist(svk_ctx* ctx, const float* d_a, const float* d_b, int32_t n, int32_t dim, float* d_out) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, n >= 0 && dim >= 0, "negative shape");
  if (n == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_a && d_b && d_out, "NULL buffer");
  const unsigned grid = (unsigned)std::max(1, std::min((n + 3) / 4, ctx->num_cu * 8));
  hipLaunchKernelGGL(l2_dist_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_a, d_b, n, dim, d_out);
  SVK_LAUNCH_CHECK(ctx);
  return SVK_OK;
}

}  // extern "C"
