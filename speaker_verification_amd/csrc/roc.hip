// ROC / EER / AUC on the device: what /root/reference/evaluation.py:47-52 gets from
// sklearn.roc_curve + roc_auc_score + brentq(interp1d) -- one sort of all (label, score) pairs --
// for score sets too large to bring back to the host (148 642 x 1 211 = 1.8e8 pairs).
//
//   1. stable radix sort of the scores, descending, labels riding along   (rocPRIM via hipCUB)
//   2. inclusive scan of the labels            -> tps[i] = positives among the i+1 best scores
//   3. distinct-score boundaries               -> the ROC's threshold points (roc_curve keeps one
//      point per distinct score; its drop_intermediate only removes collinear points)
//   4. one pass over the points: trapezoid area (AUC) and the segment where 1 - fpr - tpr
//      changes sign, solved linearly (the root brentq finds on the linear interpolant).
// Sort and scan are library primitives; steps 3-4 are the kernels below.  Bit-level equality with
// sklearn is not expected (float64 accumulation order), |d| ~ 1e-15.
#include <hipcub/hipcub.hpp>

#include "svk_internal.h"

namespace {

struct ToU32 {
  __host__ __device__ __forceinline__ unsigned operator()(const uint8_t& v) const { return v ? 1u : 0u; }
};

__global__ __launch_bounds__(256) void boundary_kernel(const float* __restrict__ keys, int64_t n,
                                                       uint8_t* __restrict__ flags) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    flags[i] = (i == n - 1 || keys[i] != keys[i + 1]) ? 1 : 0;
}

// out[0] = eer, out[1] = auc (accumulated), out[2] = positives, out[3] = number of ROC points
__global__ __launch_bounds__(256) void roc_points_kernel(const unsigned* __restrict__ idx, const unsigned* __restrict__ m_ptr,
                                                         const unsigned* __restrict__ tps, int64_t n,
                                                         double* __restrict__ out) {
  __shared__ double red[4];
  const unsigned m = *m_ptr;
  const double P = (double)tps[n - 1], N = (double)n - P;
  double area = 0.0;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < (int64_t)m; j += (int64_t)gridDim.x * blockDim.x) {
    const double t1 = (double)tps[idx[j]], f1 = (double)idx[j] + 1.0 - t1;
    double t0 = 0.0, f0 = 0.0;  // roc_curve prepends the point (0, 0)
    if (j > 0) {
      t0 = (double)tps[idx[j - 1]];
      f0 = (double)idx[j - 1] + 1.0 - t0;
    }
    const double x0 = f0 / N, x1 = f1 / N, y0 = t0 / P, y1 = t1 / P;
    area += (x1 - x0) * (y0 + y1) * 0.5;
    const double g0 = 1.0 - x0 - y0, g1 = 1.0 - x1 - y1;
    if (g0 > 0.0 && g1 <= 0.0) out[0] = x0 + (x1 - x0) * g0 / (g0 - g1);  // exactly one segment qualifies
  }
  area = wave_sum(area);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = area;
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&out[1], red[0] + red[1] + red[2] + red[3]);
    if (blockIdx.x == 0) {
      out[2] = P;
      out[3] = (double)m;
    }
  }
}

struct RocLayout {
  size_t keys, vals, tps, flags, idx, misc, cub, total, cub_bytes;
};

RocLayout roc_layout(int64_t n) {
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  RocLayout l;
  size_t sort_b = 0, scan_b = 0, sel_b = 0;
  (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, sort_b, (const float*)nullptr, (float*)nullptr,
                                               (const uint8_t*)nullptr, (uint8_t*)nullptr, n);
  hipcub::TransformInputIterator<unsigned, ToU32, const uint8_t*> it((const uint8_t*)nullptr, ToU32());
  (void)hipcub::DeviceScan::InclusiveSum(nullptr, scan_b, it, (unsigned*)nullptr, n);
  hipcub::CountingInputIterator<unsigned> cnt(0);
  (void)hipcub::DeviceSelect::Flagged(nullptr, sel_b, cnt, (const uint8_t*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr, n);
  l.cub_bytes = std::max(sort_b, std::max(scan_b, sel_b));
  size_t o = 0;
  l.keys = o;  o += up((size_t)n * 4);
  l.vals = o;  o += up((size_t)n);
  l.tps = o;   o += up((size_t)n * 4);
  l.flags = o; o += up((size_t)n);
  l.idx = o;   o += up((size_t)n * 4);
  l.misc = o;  o += 256;  // [0..3] doubles out, then the selected-count word
  l.cub = o;   o += up(l.cub_bytes);
  l.total = o;
  return l;
}

}  // namespace

extern "C" {

size_t svk_roc_workspace_bytes(int64_t n) { return n > 0 ? roc_layout(n).total : 0; }

int svk_roc_eer(svk_ctx* ctx, const float* d_scores, const uint8_t* d_labels, int64_t n, void* d_workspace,
                size_t workspace_bytes, double* h_out) {
  if (!ctx || !h_out) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, n >= 2, "need at least two (label, score) pairs");
  SVK_REQUIRE(ctx, n < ((int64_t)1 << 32), "at most 2^32 - 1 pairs");
  SVK_REQUIRE(ctx, d_scores && d_labels && d_workspace, "NULL buffer");
  const RocLayout l = roc_layout(n);
  SVK_REQUIRE(ctx, workspace_bytes >= l.total, "workspace smaller than svk_roc_workspace_bytes(n)");
  char* w = reinterpret_cast<char*>(d_workspace);
  float* keys = reinterpret_cast<float*>(w + l.keys);
  uint8_t* vals = reinterpret_cast<uint8_t*>(w + l.vals);
  unsigned* tps = reinterpret_cast<unsigned*>(w + l.tps);
  uint8_t* flags = reinterpret_cast<uint8_t*>(w + l.flags);
  unsigned* idx = reinterpret_cast<unsigned*>(w + l.idx);
  double* out = reinterpret_cast<double*>(w + l.misc);
  unsigned* m_ptr = reinterpret_cast<unsigned*>(w + l.misc + 64);
  void* cub = w + l.cub;
  size_t cub_bytes = l.cub_bytes;
  hipStream_t st = ctx->stream;

  SVK_HIP(ctx, hipMemsetAsync(out, 0, 128, st));
  SVK_HIP(ctx, hipcub::DeviceRadixSort::SortPairsDescending(cub, cub_bytes, d_scores, keys, d_labels, vals, n, 0, 32, st));
  hipcub::TransformInputIterator<unsigned, ToU32, const uint8_t*> it(vals, ToU32());
  cub_bytes = l.cub_bytes;
  SVK_HIP(ctx, hipcub::DeviceScan::InclusiveSum(cub, cub_bytes, it, tps, n, st));
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, (int64_t)ctx->num_cu * 8));
  hipLaunchKernelGGL(boundary_kernel, dim3(grid), dim3(256), 0, st, keys, n, flags);
  SVK_LAUNCH_CHECK(ctx);
  hipcub::CountingInputIterator<unsigned> cnt(0);
  cub_bytes = l.cub_bytes;
  SVK_HIP(ctx, hipcub::DeviceSelect::Flagged(cub, cub_bytes, cnt, flags, idx, m_ptr, n, st));
  hipLaunchKernelGGL(roc_points_kernel, dim3(grid), dim3(256), 0, st, idx, m_ptr, tps, n, out);
  SVK_LAUNCH_CHECK(ctx);
  SVK_HIP(ctx, hipMemcpyAsync(h_out, out, 4 * sizeof(double), hipMemcpyDeviceToHost, st));
  SVK_HIP(ctx, hipStreamSynchronize(st));
  if (h_out[2] <= 0.0 || h_out[2] >= (double)n)
    return svk_fail(ctx, SVK_ERR_BAD_ARG, "ROC needs both classes: %.0f positives of %lld", h_out[2], (long long)n);
  return SVK_OK;
}

}  // extern "C"
