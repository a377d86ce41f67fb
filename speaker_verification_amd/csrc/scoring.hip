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
  const int tiles = (n_test + 15) / 16;
  const unsigned grid = (unsigned)std::max(1, std::min((tiles + 3) / 4, ctx->num_cu * 8));
  hipLaunchKernelGGL(cosine_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_test, d_enroll, n_test, n_enroll, dim,
                     d_out);
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
