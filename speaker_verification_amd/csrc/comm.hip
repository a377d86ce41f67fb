// Multi-GPU exchange step of the path: ONE all-gather of the per-rank embedding shards before
// scoring, over RCCL (xGMI).  The reference has no collective at all (single-process
// torch.nn.DataParallel in its trainers, /root/reference/train.py:40-41); a torch host uses
// torch.distributed (speaker_verification_amd/distributed.py), a torch-less host these thin wrappers.
//
// RCCL is bound at run time (dlopen of librccl.so.1, the copy already mapped into the process if there is
// one -- PyTorch ships its own): libsvk.so carries no link-time dependency on it, and a single-GPU host
// that never calls svk_comm_* never loads it.  Only the few entry points used are declared here, with the
// ABI of rccl.h (ncclUniqueId = 128 bytes, ncclFloat = 7, ncclSuccess = 0).
#include <dlfcn.h>

#include <cstdio>
#include <mutex>

#include "svk_internal.h"

namespace {

struct nccl_unique_id {
  char internal[128];
};
typedef void* nccl_comm_t;
typedef int (*get_unique_id_fn)(nccl_unique_id*);
typedef int (*comm_init_rank_fn)(nccl_comm_t*, int, nccl_unique_id, int);
typedef int (*all_gather_fn)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t);
typedef int (*comm_destroy_fn)(nccl_comm_t);
typedef const char* (*get_error_string_fn)(int);

struct Rccl {
  void* handle = nullptr;
  get_unique_id_fn get_unique_id = nullptr;
  comm_init_rank_fn comm_init_rank = nullptr;
  all_gather_fn all_gather = nullptr;
  comm_destroy_fn comm_destroy = nullptr;
  get_error_string_fn get_error_string = nullptr;
  char load_error[256] = "";   // dlerror() text of the failed bind, captured once (dlerror() clears itself on read)
};
Rccl g_rccl;
std::once_flag g_rccl_once;

const Rccl* rccl(svk_ctx* ctx) {
  std::call_once(g_rccl_once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      g_rccl.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (g_rccl.handle) break;
    }
    if (g_rccl.handle) {
      g_rccl.get_unique_id = (get_unique_id_fn)dlsym(g_rccl.handle, "ncclGetUniqueId");
      g_rccl.comm_init_rank = (comm_init_rank_fn)dlsym(g_rccl.handle, "ncclCommInitRank");
      g_rccl.all_gather = (all_gather_fn)dlsym(g_rccl.handle, "ncclAllGather");
      g_rccl.comm_destroy = (comm_destroy_fn)dlsym(g_rccl.handle, "ncclCommDestroy");
      g_rccl.get_error_string = (get_error_string_fn)dlsym(g_rccl.handle, "ncclGetErrorString");
      if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.all_gather || !g_rccl.comm_destroy)
        snprintf(g_rccl.load_error, sizeof(g_rccl.load_error), "a required nccl* symbol is missing");
    } else {
      const char* why = dlerror();
      snprintf(g_rccl.load_error, sizeof(g_rccl.load_error), "%s", why ? why : "dlopen failed");
    }
  });
  if (g_rccl.load_error[0]) {
    svk_fail(ctx, SVK_ERR_RCCL, "librccl.so.1 could not be loaded (%s)", g_rccl.load_error);
    return nullptr;
  }
  return &g_rccl;
}

int rccl_fail(svk_ctx* ctx, const Rccl* r, const char* what, int code) {
  return svk_fail(ctx, SVK_ERR_RCCL, "%s failed: %s (rccl code %d)", what,
                  r->get_error_string ? r->get_error_string(code) : "?", code);
}

}  // namespace

extern "C" {

int svk_comm_unique_id(svk_ctx* ctx, char out[128]) {
  if (!ctx || !out) return SVK_ERR_BAD_ARG;
  const Rccl* r = rccl(ctx);
  if (!r) return SVK_ERR_RCCL;
  nccl_unique_id id;
  const int rc = r->get_unique_id(&id);
  if (rc != 0) return rccl_fail(ctx, r, "ncclGetUniqueId", rc);
  memcpy(out, id.internal, 128);
  return SVK_OK;
}

int svk_comm_init(svk_ctx* ctx, const char id[128], int32_t n_ranks, int32_t rank) {
  if (!ctx || !id) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, n_ranks >= 1 && rank >= 0 && rank < n_ranks, "0 <= rank < n_ranks");
  SVK_REQUIRE(ctx, ctx->comm == nullptr, "the context already has a communicator (svk_comm_destroy first)");
  const Rccl* r = rccl(ctx);
  if (!r) return SVK_ERR_RCCL;
  SVK_HIP(ctx, hipSetDevice(ctx->device));
  nccl_unique_id uid;
  memcpy(uid.internal, id, 128);
  nccl_comm_t comm = nullptr;
  const int rc = r->comm_init_rank(&comm, n_ranks, uid, rank);
  if (rc != 0) return rccl_fail(ctx, r, "ncclCommInitRank", rc);
  ctx->comm = comm;
  ctx->comm_ranks = n_ranks;
  ctx->comm_rank = rank;
  return SVK_OK;
}

int svk_allgather_f32(svk_ctx* ctx, const float* d_send, float* d_recv, size_t count_per_rank) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, ctx->comm != nullptr, "no communicator: call svk_comm_init first");
  if (count_per_rank == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_send && d_recv, "NULL buffer");
  const Rccl* r = rccl(ctx);
  if (!r) return SVK_ERR_RCCL;
  const int rc = r->all_gather(d_send, d_recv, count_per_rank, /* ncclFloat */ 7, ctx->comm, ctx->stream);
  if (rc != 0) return rccl_fail(ctx, r, "ncclAllGather", rc);
  return SVK_OK;
}

int svk_comm_destroy(svk_ctx* ctx) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  if (!ctx->comm) return SVK_OK;
  const Rccl* r = rccl(ctx);
  if (!r) return SVK_ERR_RCCL;
  (void)hipStreamSynchronize(ctx->stream);
  const int rc = r->comm_destroy(ctx->comm);
  ctx->comm = nullptr;
  ctx->comm_ranks = 0;
  ctx->comm_rank = 0;
  if (rc != 0) return rccl_fail(ctx, r, "ncclCommDestroy", rc);
  return SVK_OK;
}

int svk_comm_info(const svk_ctx* ctx, int32_t out[2]) {
  if (!ctx || !out) return SVK_ERR_BAD_ARG;
  out[0] = ctx->comm ? ctx->comm_ranks : 0;
  out[1] = ctx->comm ? ctx->comm_rank : 0;
  return SVK_OK;
}

}  // extern "C"
