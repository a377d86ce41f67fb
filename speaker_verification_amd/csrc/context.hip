// Context, stream and memory entry points of the C-ABI (include/svk.h).
#include "svk_internal.h"

extern "C" {

int svk_version(void) { return SVK_VERSION; }

int svk_create(int device_id, svk_ctx** out) {
  if (!out) return SVK_ERR_BAD_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return SVK_ERR_NO_DEVICE;
  if (device_id < 0 || device_id >= count) return SVK_ERR_BAD_ARG;
  if (hipSetDevice(device_id) != hipSuccess) return SVK_ERR_HIP;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return SVK_ERR_HIP;
  // The code objects in this library are gfx950 only; refuse anything else loudly.
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return SVK_ERR_NO_DEVICE;
  svk_ctx* ctx = new (std::nothrow) svk_ctx();
  if (!ctx) return SVK_ERR_OOM;
  ctx->device = device_id;
  ctx->num_cu = prop.multiProcessorCount;
  ctx->clock_khz = prop.clockRate;
  ctx->lds_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
  if (hipMalloc(&ctx->scratch, 256) != hipSuccess) {
    delete ctx;
    return SVK_ERR_OOM;
  }
  *out = ctx;
  return SVK_OK;
}

void svk_destroy(svk_ctx* ctx) {
  if (!ctx) return;
  if (ctx->comm) (void)svk_comm_destroy(ctx);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->work) (void)hipFree(ctx->work);
  delete ctx;
}

const char* svk_last_error(const svk_ctx* ctx) { return ctx ? ctx->err : "null context"; }

int svk_set_stream(svk_ctx* ctx, void* hip_stream) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  ctx->stream = (hipStream_t)hip_stream;
  return SVK_OK;
}

int svk_sync(svk_ctx* ctx) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SVK_OK;
}

int svk_malloc(svk_ctx* ctx, size_t bytes, void** d_out) {
  if (!ctx || !d_out) return SVK_ERR_BAD_ARG;
  *d_out = nullptr;
  if (bytes == 0) return SVK_OK;
  SVK_HIP(ctx, hipSetDevice(ctx->device));
  hipError_t e = hipMalloc(d_out, bytes);
  if (e == hipErrorOutOfMemory) return svk_fail(ctx, SVK_ERR_OOM, "hipMalloc(%zu) out of memory", bytes);
  SVK_HIP(ctx, e);
  return SVK_OK;
}

int svk_free(svk_ctx* ctx, void* d_ptr) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  if (d_ptr) SVK_HIP(ctx, hipFree(d_ptr));
  return SVK_OK;
}

int svk_memcpy_h2d(svk_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
  if (!ctx || (bytes && (!d_dst || !h_src))) return SVK_ERR_BAD_ARG;
  if (bytes == 0) return SVK_OK;
  SVK_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
  SVK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SVK_OK;
}

int svk_memcpy_d2h(svk_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
  if (!ctx || (bytes && (!h_dst || !d_src))) return SVK_ERR_BAD_ARG;
  if (bytes == 0) return SVK_OK;
  SVK_HIP(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  SVK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SVK_OK;
}

int svk_memset(svk_ctx* ctx, void* d_dst, int value, size_t bytes) {
  if (!ctx || (bytes && !d_dst)) return SVK_ERR_BAD_ARG;
  if (bytes == 0) return SVK_OK;
  SVK_HIP(ctx, hipMemsetAsync(d_dst, value, bytes, ctx->stream));
  return SVK_OK;
}

int svk_device_info(svk_ctx* ctx, int64_t out[4]) {
  if (!ctx || !out) return SVK_ERR_BAD_ARG;
  out[0] = ctx->num_cu;
  out[1] = ctx->clock_khz;
  out[2] = ctx->lds_per_cu;
  out[3] = 64;
  return SVK_OK;
}

}  // extern "C"
