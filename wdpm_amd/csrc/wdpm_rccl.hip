/*
 * wdpm_rccl.hip — halo rows between GPUs by RCCL send/recv over xGMI (include/wdpm.h: wdpm_comm_*).
 *
 * The reference is single-device: its device set-up is create_device() picking one OpenCL device
 * (src/WDPMCL.c:80-121, :598-638).  Here a context can join an RCCL communicator — one rank per GPU,
 * either one process per rank (ncclCommInitRank with an id every rank was handed) or all ranks in
 * one process (ncclCommInitAll) — and the row-block driver (wdpm_rowblock.c) refreshes the water
 * halos with grouped ncclSend/ncclRecv issued on the context's own stream: ordered on the device
 * behind the kernels that produced the rows and ahead of those that consume them, no host round trip.
 *
 * RCCL is bound at run time (dlopen of librccl.so.1): a single-GPU run needs no RCCL at all, and
 * inside a PyTorch process the copy PyTorch already mapped is the one that is used (same SONAME),
 * so there is one RCCL and one HIP runtime per process.
 */
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>

#include "wdpm_ctx.h"

namespace {

struct RcclApi {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int *) = nullptr;
  char why[256] = "";
};

RcclApi g_api;
std::once_flag g_once;

void load_rccl() {
  const char *names[] = {getenv("WDPM_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *n : names) {
    if (!n || !*n) continue;
    g_api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (g_api.handle) break;
    snprintf(g_api.why, sizeof g_api.why, "%s", dlerror());
  }
  if (!g_api.handle) return;
#define SYM(field, name)                                                                    \
  do {                                                                                      \
    *(void **)(&g_api.field) = dlsym(g_api.handle, name);                                   \
    if (!g_api.field) {                                                                     \
      snprintf(g_api.why, sizeof g_api.why, "%s is missing from the RCCL library", name);   \
      g_api.handle = nullptr;                                                               \
      return;                                                                               \
    }                                                                                       \
  } while (0)
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommInitAll, "ncclCommInitAll");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(CommCount, "ncclCommCount");
  SYM(GroupStart, "ncclGroupStart");
  SYM(GroupEnd, "ncclGroupEnd");
  SYM(Send, "ncclSend");
  SYM(Recv, "ncclRecv");
  SYM(AllGather, "ncclAllGather");
  SYM(GetErrorString, "ncclGetErrorString");
  SYM(GetVersion, "ncclGetVersion");
#undef SYM
}

int need_rccl() {
  std::call_once(g_once, load_rccl);
  if (!g_api.handle) return wdpm_fail("RCCL is not available: %s", g_api.why[0] ? g_api.why : "librccl.so.1 not found");
  return 0;
}

}  // namespace

struct wdpm_comm {
  ncclComm_t comm;
  int rank, nranks;
  double *d_mine, *d_all;   /* wdpm_comm_allgather staging: kGatherMax doubles, nranks * kGatherMax doubles */
  double *h_all;            /* pinned */
};
constexpr int kGatherMax = 8;

#define NCCL_TRY(expr)                                                                                 \
  do {                                                                                                 \
    ncclResult_t r_ = (expr);                                                                          \
    if (r_ != ncclSuccess)                                                                             \
      return wdpm_fail("%s failed: %s (%s:%d)", #expr, g_api.GetErrorString(r_), __FILE__, __LINE__);  \
  } while (0)
#define HIP_TRY(expr)                                                                                  \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) return wdpm_fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

static int attach(wdpm_ctx *x, ncclComm_t comm, int rank, int nranks) {
  wdpm_comm *c = new wdpm_comm{comm, rank, nranks, nullptr, nullptr, nullptr};
  x->comm = c;
  HIP_TRY(hipSetDevice(x->p.device));
  HIP_TRY(hipMalloc(&c->d_mine, kGatherMax * sizeof(double)));
  HIP_TRY(hipMalloc(&c->d_all, (size_t)nranks * kGatherMax * sizeof(double)));
  HIP_TRY(hipHostMalloc(&c->h_all, (size_t)(nranks + 1) * kGatherMax * sizeof(double)));
  return 0;
}

void wdpm_comm_release(wdpm_ctx *x) {
  if (!x || !x->comm) return;
  wdpm_comm *c = x->comm;
  x->comm = nullptr;
  (void)hipSetDevice(x->p.device);
  if (c->comm && g_api.handle) (void)g_api.CommDestroy(c->comm);
  (void)hipFree(c->d_mine);
  (void)hipFree(c->d_all);
  if (c->h_all) (void)hipHostFree(c->h_all);
  delete c;
}

extern "C" {

int wdpm_comm_available(void) {
  std::call_once(g_once, load_rccl);
  return g_api.handle ? 1 : 0;
}

int wdpm_comm_unique_id(void *id128) {
  if (!id128) return wdpm_fail("wdpm_comm_unique_id: null argument");
  if (need_rccl()) return 1;
  static_assert(sizeof(ncclUniqueId) == WDPM_COMM_ID_BYTES, "include/wdpm.h: WDPM_COMM_ID_BYTES");
  ncclUniqueId id;
  NCCL_TRY(g_api.GetUniqueId(&id));
  memcpy(id128, &id, sizeof id);
  return 0;
}

int wdpm_comm_init_rank(wdpm_ctx *x, int32_t nranks, int32_t rank, const void *id128) {
  if (!x || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return wdpm_fail("wdpm_comm_init_rank: bad argument");
  if (x->comm) return wdpm_fail("wdpm_comm_init_rank: the context already has a communicator");
  if (need_rccl()) return 1;
  HIP_TRY(hipSetDevice(x->p.device));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  ncclComm_t comm;
  NCCL_TRY(g_api.CommInitRank(&comm, nranks, id, rank));
  return attach(x, comm, rank, nranks);
}

int wdpm_comm_init_all(wdpm_ctx **ctxs, int32_t n) {
  if (!ctxs || n < 1 || n > 64) return wdpm_fail("wdpm_comm_init_all: bad argument");
  if (need_rccl()) return 1;
  int dev[64];
  for (int i = 0; i < n; i++) {
    if (!ctxs[i] || ctxs[i]->comm) return wdpm_fail("wdpm_comm_init_all: bad context %d", i);
    dev[i] = ctxs[i]->p.device;
    for (int j = 0; j < i; j++)
      if (dev[j] == dev[i] && !getenv("WDPM_RCCL_SHARED_DEVICE_OK"))   /* tests with a stand-in RCCL library set it */
        return wdpm_fail("wdpm_comm_init_all: device %d named twice (RCCL wants one rank per GPU)", dev[i]);
  }
  ncclComm_t comm[64];
  NCCL_TRY(g_api.CommInitAll(comm, n, dev));
  for (int i = 0; i < n; i++)
    if (attach(ctxs[i], comm[i], i, n)) return 1;
  return 0;
}

int wdpm_comm_size(wdpm_ctx *x, int32_t *nranks, int32_t *rank) {
  if (!x || !x->comm) return wdpm_fail("wdpm_comm_size: the context has no communicator");
  int n = 0;
  NCCL_TRY(g_api.CommCount(x->comm->comm, &n));   /* what RCCL itself says, not what we were told */
  if (nranks) *nranks = n;
  if (rank) *rank = x->comm->rank;
  return 0;
}

/* Rows of the CURRENT water raster to and from neighbouring ranks, one RCCL group on the context's
 * stream.  Deliberately does not join a pending overlapped interior launch (wdpm_iterate_overlapped):
 * the rows sent were produced by the boundary launches on this stream, the rows received are halo rows
 * the interior launch does not touch - so the transfer runs beside it.  The next use of the raster by
 * the library joins as usual. */
int wdpm_comm_exchange(wdpm_ctx *x, int32_t nsend, const wdpm_halo_op *sends, int32_t nrecv, const wdpm_halo_op *recvs) {
  if (!x || !x->comm) return wdpm_fail("wdpm_comm_exchange: the context has no communicator");
  if (nsend < 0 || nrecv < 0 || (nsend && !sends) || (nrecv && !recvs)) return wdpm_fail("wdpm_comm_exchange: bad argument");
  wdpm_comm *c = x->comm;
  for (int pass = 0; pass < 2; pass++) {
    const wdpm_halo_op *ops = pass ? recvs : sends;
    for (int i = 0; i < (pass ? nrecv : nsend); i++)
      if (ops[i].peer < 0 || ops[i].peer >= c->nranks || ops[i].row < 0 || ops[i].nrows < 0 ||
          ops[i].row + ops[i].nrows > x->g.rows)
        return wdpm_fail("wdpm_comm_exchange: bad row range or peer");
  }
  HIP_TRY(hipSetDevice(x->p.device));
  if (nsend > 0 && wdpm_apply_owed_drain(x)) return 1;   /* rows that leave carry the last iteration's drain() (see wdpm_copy_rows) */
  for (int i = 0; i < nrecv; i++)
    if (wdpm_tiles_touch(x, recvs[i].row, recvs[i].nrows)) return 1;   /* received rows: those tiles are no longer known dry */
  double *w = x->d_w[x->cur];
  const size_t ncp = (size_t)x->g.ncp;
  NCCL_TRY(g_api.GroupStart());
  for (int i = 0; i < nsend; i++)
    if (sends[i].nrows > 0)
      NCCL_TRY(g_api.Send(w + (size_t)sends[i].row * ncp, (size_t)sends[i].nrows * ncp, ncclDouble, sends[i].peer, c->comm, x->stream));
  for (int i = 0; i < nrecv; i++)
    if (recvs[i].nrows > 0)
      NCCL_TRY(g_api.Recv(w + (size_t)recvs[i].row * ncp, (size_t)recvs[i].nrows * ncp, ncclDouble, recvs[i].peer, c->comm, x->stream));
  NCCL_TRY(g_api.GroupEnd());
  return 0;
}

/* all[r * n + i] = mine[i] of rank r (n <= 8), on every rank; synchronous (block-loop scalars:
 * max_diff, totaldrain, the chained volume sum - once per 1000 iterations) */
int wdpm_comm_allgather(wdpm_ctx *x, const double *mine, int32_t n, double *all) {
  if (!x || !x->comm) return wdpm_fail("wdpm_comm_allgather: the context has no communicator");
  if (!mine || !all || n < 1 || n > kGatherMax) return wdpm_fail("wdpm_comm_allgather: bad argument");
  wdpm_comm *c = x->comm;
  HIP_TRY(hipSetDevice(x->p.device));
  double *h_mine = c->h_all + (size_t)c->nranks * kGatherMax;
  memcpy(h_mine, mine, n * sizeof(double));
  HIP_TRY(hipMemcpyAsync(c->d_mine, h_mine, n * sizeof(double), hipMemcpyHostToDevice, x->stream));
  NCCL_TRY(g_api.AllGather(c->d_mine, c->d_all, (size_t)n, ncclDouble, c->comm, x->stream));
  HIP_TRY(hipMemcpyAsync(c->h_all, c->d_all, (size_t)c->nranks * n * sizeof(double), hipMemcpyDeviceToHost, x->stream));
  HIP_TRY(hipStreamSynchronize(x->stream));
  memcpy(all, c->h_all, (size_t)c->nranks * n * sizeof(double));
  return 0;
}

const char *wdpm_comm_version(void) {
  static char v[64] = "";
  if (need_rccl()) return "unavailable";
  int code = 0;
  if (g_api.GetVersion(&code) == ncclSuccess) snprintf(v, sizeof v, "RCCL %d.%d.%d", code / 10000, (code / 100) % 100, code % 100);
  return v;
}

} /* extern "C" */
