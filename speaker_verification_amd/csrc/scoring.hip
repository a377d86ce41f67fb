// Scoring kernels.
//   svk_cosine_scores <- /root/reference/evaluation.py:67-84: the reference loops over
//       (utterance, speaker) pairs calling sklearn's cosine_similarity on two (1,128)
//       float32 rows (Q18); here the whole [n_test x n_enroll] matrix is one launch on
//       v_mfma_f32_16x16x4_f32 (exact f32 products, f32 accumulation).
//   svk_l2_dist       <- /root/reference/siamese.py:29-30.
#include <algorithm>
#include <cstdlib>

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
// Epilogue staging: a wave parks its 32 x 32 score block in LDS (row stride 36 floats: the accumulator
// layout writes conflict-free) and writes it out as whole 128-byte row segments, 8 rows per store
// instruction (4 store instructions per block instead of 16 that each touch four 64-byte pieces of four
// rows 4 844 bytes apart).
constexpr int CT_SLD = 36;
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte access on a 4-byte boundary (row pitch 4 ns)

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
__device__ __forceinline__ void cosine_tiled_body(const float* __restrict__ test, const float* __restrict__ enroll,
                                                  const float* __restrict__ einv, int nt, int ns, int dim,
                                                  float* __restrict__ out, long long total_units, float (&bs)[2][CT_BN * CT_LD],
                                                  float* __restrict__ stage_all) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* const stage = stage_all + wave * 32 * CT_SLD;  // private to this wave
  const int i = lane & 15, kk = lane >> 4;
  const int nkb = (dim + CT_KB - 1) / CT_KB;
  const bool vec_ok = (dim & 3) == 0 && ((reinterpret_cast<uintptr_t>(test) | reinterpret_cast<uintptr_t>(enroll)) & 15) == 0;
  // Work unit = (128-row block, one 32-row block of the enrolled matrix), numbered row-block-major.
  // The grid is one workgroup per resident slot and workgroup b takes the contiguous unit range
  // [b U / G, (b + 1) U / G): every workgroup gets the same number of units (+-1) whatever the shape,
  // and its range touches at most a few row blocks, so the test-row fragments are (re)loaded a few
  // times per workgroup.  No reduction is needed: the split is over output columns, never over K.
  // (Whole row blocks per workgroup left the last round of workgroups ragged: 1 162 blocks on 768
  // slots kept waves resident only 73 % of the launch.)
  const int all_stiles = (ns + CT_BN - 1) / CT_BN;
  const long long u_begin = total_units * blockIdx.x / gridDim.x, u_end = total_units * (blockIdx.x + 1) / gridDim.x;

  // staging assignment: thread t moves 4 float4 of the 32 x 128 block: row = (t >> 5) + 8 j, float4 column = t & 31
  const int srow = threadIdx.x >> 5, scol = (threadIdx.x & 31) * 4;
  auto fetch = [&](int st, int kb, f32x4 (&regs)[4]) {
    if (vec_ok && (kb + 1) * CT_KB <= dim) {  // workgroup-uniform: whole K block inside
      // rows past the enrolled matrix (ragged last block) re-read its last row: those columns are never stored
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = min(st * CT_BN + srow + 8 * j, ns - 1);
        regs[j] = *reinterpret_cast<const f32x4*>(enroll + (int64_t)r * dim + kb * CT_KB + scol);
      }
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

  for (long long u = u_begin; u < u_end;) {
  const int rb = (int)(u / all_stiles), st_begin = (int)(u - (long long)rb * all_stiles);
  const int n_stiles = (int)min((long long)all_stiles, st_begin + (u_end - u));
  u += n_stiles - st_begin;
  const int m0 = rb * CT_BM + wave * 32;
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

  // Epilogue of one 32 x 32 block, through LDS for every block, ragged ones included: scale by the two norms, park in
  // this wave's staging tile, write whole 128-byte row segments.  One wave's DS instructions execute in order: fences only.
  auto epilogue = [&](const f32x4 (&accv)[2][2], int st_e, const float (&sinv_e)[2]) __attribute__((always_inline)) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          stage[(16 * rt + 4 * kk + r) * CT_SLD + 16 * ct + i] = accv[rt][ct][r] * rinv[rt][r] * sinv_e[ct];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int c4 = 4 * (lane & 7);
    float* const ob = out + (int64_t)m0 * ns + st_e * CT_BN + c4;
    const bool cols_in = (st_e + 1) * CT_BN <= ns;  // wave-uniform
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (lane >> 3) + 8 * j;
      const f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * CT_SLD + c4);
      if (rows_in || m0 + row < nt) {
        float* const dst = ob + (int64_t)row * ns;
        if (cols_in) {
#ifdef SVK_COS_NOSTORE
          if (v[0] == 123456.0f)
#endif
          *reinterpret_cast<f32x4_u*>(dst) = v;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (st_e * CT_BN + c4 + e < ns) dst[e] = v[e];
        }
      }
    }
  };
  // The epilogue of block st - 1 is issued in the MIDDLE of block st's 128 MFMAs (its ~60 VALU / LDS / store
  // instructions fit the issue slots the matrix pipe leaves free) instead of after its own: the accumulators of
  // the finished block wait in `pacc`.
  f32x4 pacc[2][2];
  float psinv[2] = {0.f, 0.f};
  int pst = 0;
  bool pending = false;

  f32x4 pre[4];
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
        if (u == 1 && kb == 0 && pending) {  // wave-uniform
          epilogue(pacc, pst, psinv);
          pending = false;
        }
      }
      if (more) stash(bs[cur ^ 1], pre);
      __syncthreads();  // everyone is done with bs[cur]; bs[cur ^ 1] is complete
      cur ^= 1;
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) pacc[rt][ct] = acc[rt][ct];
    psinv[0] = sinv[0];
    psinv[1] = sinv[1];
    pst = st;
    pending = true;
  }
  if (pending) {  // the segment's last block (the row block, and with it rinv / the output rows, changes next)
    epilogue(pacc, pst, psinv);
    pending = false;
  }
  }  // next segment of this workgroup's unit range (the st loop ended on a __syncthreads: bs[] is free)
}

// Two register budgets of the same body.  Left alone the compiler puts the accumulators in AGPRs on
// top of 158 VGPRs (176 in all: two workgroups per CU); hinted to 3 waves per SIMD it fits 168 with
// the accumulators in VGPRs.  Which one is faster depends on the shape (SVK_COS_WAVES=2|3 picks, tuning).
template <bool HOIST>
__global__ __launch_bounds__(256) void cosine_tiled_kernel(const float* __restrict__ test, const float* __restrict__ enroll,
                                                           const float* __restrict__ einv, int nt, int ns, int dim,
                                                           float* __restrict__ out, long long total_units) {
  __shared__ __attribute__((aligned(16))) float bs[2][CT_BN * CT_LD];
  __shared__ __attribute__((aligned(16))) float stage[4 * 32 * CT_SLD];
  cosine_tiled_body<HOIST>(test, enroll, einv, nt, ns, dim, out, total_units, bs, stage);
}
template <bool HOIST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void cosine_tiled_kernel_w3(
    const float* __restrict__ test, const float* __restrict__ enroll, const float* __restrict__ einv, int nt, int ns,
    int dim, float* __restrict__ out, long long total_units) {
  __shared__ __attribute__((aligned(16))) float bs[2][CT_BN * CT_LD];
  __shared__ __attribute__((aligned(16))) float stage[4 * 32 * CT_SLD];
  cosine_tiled_body<HOIST>(test, enroll, einv, nt, ns, dim, out, total_units, bs, stage);
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
    const int rc = svk_ensure_work(ctx, sizeof(float) * (size_t)n_enroll);
    if (rc != SVK_OK) return rc;
    float* einv = static_cast<float*>(ctx->work);
    hipLaunchKernelGGL(inv_norm_kernel, dim3((unsigned)std::min((n_enroll + 3) / 4, ctx->num_cu * 8)), dim3(256), 0,
                       ctx->stream, d_enroll, n_enroll, dim, einv);
    const char* wenv = getenv("SVK_COS_WAVES");
    const bool w3 = wenv ? atoi(wenv) == 3 : false;
    auto kern = dim <= CT_KB ? (w3 ? cosine_tiled_kernel_w3<true> : cosine_tiled_kernel<true>)
                             : (w3 ? cosine_tiled_kernel_w3<false> : cosine_tiled_kernel<false>);
    // one workgroup per resident slot (fewer when there is less work than that)
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 256, 0) != hipSuccess ||
        per_cu < 1)
      per_cu = 2;
    const long long units = (long long)gx * stiles;
    const long long blocks = std::min<long long>((long long)per_cu * ctx->num_cu, std::max<long long>(1, units / 4));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_test, d_enroll, einv, n_test, n_enroll,
                       dim, d_out, units);
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
