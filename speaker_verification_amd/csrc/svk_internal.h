// Internal declarations shared by the translation units of libsvk.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/svk.h"

struct svk_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int num_cu = 256;
  int clock_khz = 0;
  int lds_per_cu = 160 * 1024;
  void* scratch = nullptr;  // 256 bytes of device memory for tiny reductions (svk_log_power)
  void* work = nullptr;     // grow-only device workspace owned by the handle (row norms of svk_cosine_scores)
  size_t work_bytes = 0;
  void* comm = nullptr;     // RCCL communicator (svk_comm_init), or NULL
  int comm_ranks = 0, comm_rank = 0;
  char err[512] = {0};
};

inline int svk_fail(svk_ctx* ctx, int code, const char* fmt, ...) {
  if (ctx) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
    va_end(ap);
  }
  return code;
}

// Grow-only workspace.  Kernels using it are ordered on ctx->stream; growing frees the old block
// only after the stream has drained.
inline int svk_ensure_work(svk_ctx* ctx, size_t bytes) {
  if (ctx->work_bytes >= bytes) return SVK_OK;
  if (ctx->work) {
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return svk_fail(ctx, SVK_ERR_HIP, "stream sync before workspace growth failed");
    (void)hipFree(ctx->work);
    ctx->work = nullptr;
    ctx->work_bytes = 0;
  }
  const size_t want = (bytes + 4095) & ~(size_t)4095;
  if (hipMalloc(&ctx->work, want) != hipSuccess) return svk_fail(ctx, SVK_ERR_OOM, "workspace of %zu bytes", want);
  ctx->work_bytes = want;
  return SVK_OK;
}

#define SVK_HIP(ctx, call)                                                                      \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return svk_fail((ctx), SVK_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                      __FILE__, __LINE__);                                                      \
  } while (0)

#define SVK_REQUIRE(ctx, cond, msg)                                                   \
  do {                                                                                \
    if (!(cond)) return svk_fail((ctx), SVK_ERR_BAD_ARG, "bad argument: %s", (msg));  \
  } while (0)

// Every launch is followed by this: catches bad grids / missing code objects at once.
#define SVK_LAUNCH_CHECK(ctx)                                                                  \
  do {                                                                                         \
    hipError_t e_ = hipGetLastError();                                                         \
    if (e_ != hipSuccess)                                                                      \
      return svk_fail((ctx), SVK_ERR_HIP, "kernel launch failed: %s (%s:%d)",                  \
                      hipGetErrorString(e_), __FILE__, __LINE__);                              \
  } while (0)

// 64-lane wave sum; every lane gets the total.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ long long wave_sum(long long v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
