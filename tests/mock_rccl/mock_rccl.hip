/*
 * mock_rccl.hip — TEST INFRASTRUCTURE.  A stand-in for librccl.so.1 that lets the library's RCCL halo path
 * (wdpm_amd/csrc/wdpm_rccl.hip) run with SEVERAL ranks on the ONE GPU of a test box, which the real RCCL
 * refuses ("duplicate GPU").  It implements just the entry points wdpm_rccl.hip binds, for the ranks of one
 * process (ncclCommInitAll; one host thread per rank, as wdpm_group drives them):
 *   ncclSend / ncclRecv inside ncclGroupStart / ncclGroupEnd = device-to-device copies ordered by events:
 *   the receiver's stream waits for an event the sender recorded on ITS stream when it posted the send, then
 *   copies; the sender's stream waits for the receiver's "copied" event before anything queued later.
 * So the data path is the product's own (op lists, row offsets, streams, tile-flag and max-diff bookkeeping
 * around a refresh, the overlapped last iteration with a communicator attached); only the wire is faked.
 * Loaded through WDPM_RCCL_LIB by tests/test_mock_rccl.py; never part of the product.
 */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstring>
#include <mutex>
#include <vector>

namespace {
struct Post { const void *buf; size_t bytes; hipEvent_t ready, copied; bool posted, taken; };
struct World {
  int n;
  std::mutex mu;
  std::condition_variable cv;
  std::vector<Post> box;                 // [src * n + dst]
};
struct Op { bool send; void *buf; size_t bytes; int peer; hipStream_t stream; };
thread_local std::vector<Op> t_ops;
thread_local int t_depth = 0;
thread_local ncclComm *t_comm = nullptr;
}  // namespace

struct ncclComm { World *w; int rank; int dev; };

static size_t type_size(ncclDataType_t t) { return t == ncclDouble || t == ncclInt64 || t == ncclUint64 ? 8 : (t == ncclFloat || t == ncclInt32 || t == ncclUint32 ? 4 : 1); }

extern "C" {
ncclResult_t ncclGetVersion(int *v) { *v = 29999; return ncclSuccess; }
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "mock RCCL error"; }
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) { memset(id, 0x5a, sizeof *id); return ncclSuccess; }
ncclResult_t ncclCommInitRank(ncclComm_t *, int, ncclUniqueId, int) { return ncclInvalidUsage; }   // one process only
ncclResult_t ncclCommInitAll(ncclComm_t *comm, int n, const int *devs) {
  World *w = new World;
  w->n = n;
  w->box.assign((size_t)n * n, Post{nullptr, 0, nullptr, nullptr, false, false});
  for (int i = 0; i < n * n; i++) {
    if (hipEventCreateWithFlags(&w->box[i].ready, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
    if (hipEventCreateWithFlags(&w->box[i].copied, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
  }
  for (int i = 0; i < n; i++) comm[i] = new ncclComm{w, i, devs ? devs[i] : i};
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { delete c; return ncclSuccess; }   // the World leaks: a test process
ncclResult_t ncclCommCount(const ncclComm_t c, int *n) { *n = c->w->n; return ncclSuccess; }
ncclResult_t ncclGroupStart() { t_depth++; return ncclSuccess; }
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) {
  if (!t_depth) return ncclInvalidUsage;
  t_comm = c;
  t_ops.push_back(Op{true, const_cast<void *>(buf), count * type_size(t), peer, s});
  return ncclSuccess;
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) {
  if (!t_depth) return ncclInvalidUsage;
  t_comm = c;
  t_ops.push_back(Op{false, buf, count * type_size(t), peer, s});
  return ncclSuccess;
}
ncclResult_t ncclGroupEnd() {
  if (--t_depth > 0) return ncclSuccess;
  if (t_ops.empty()) return ncclSuccess;
  ncclComm *c = t_comm;
  World *w = c->w;
  ncclResult_t rc = ncclSuccess;
  // post every send: "my rows are produced once this event fires"
  for (const Op &o : t_ops)
    if (o.send) {
      std::unique_lock<std::mutex> lk(w->mu);
      Post &p = w->box[(size_t)c->rank * w->n + o.peer];
      w->cv.wait(lk, [&] { return !p.posted; });                 // the previous message of this pair was taken
      if (hipEventRecord(p.ready, o.stream) != hipSuccess) rc = ncclUnhandledCudaError;
      p.buf = o.buf; p.bytes = o.bytes; p.posted = true; p.taken = false;
      w->cv.notify_all();
    }
  // take every receive: copy on my stream behind the sender's event
  for (const Op &o : t_ops)
    if (!o.send) {
      std::unique_lock<std::mutex> lk(w->mu);
      Post &p = w->box[(size_t)o.peer * w->n + c->rank];
      w->cv.wait(lk, [&] { return p.posted && !p.taken; });
      if (p.bytes != o.bytes) rc = ncclInvalidArgument;          // send / recv sizes must match, as with the real thing
      if (hipStreamWaitEvent(o.stream, p.ready, 0) != hipSuccess) rc = ncclUnhandledCudaError;
      if (hipMemcpyAsync(o.buf, p.buf, o.bytes < p.bytes ? o.bytes : p.bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess)
        rc = ncclUnhandledCudaError;
      if (hipEventRecord(p.copied, o.stream) != hipSuccess) rc = ncclUnhandledCudaError;
      p.taken = true;
      w->cv.notify_all();
    }
  // my sends: nothing I queue later may overwrite the rows before the receiver has copied them
  for (const Op &o : t_ops)
    if (o.send) {
      std::unique_lock<std::mutex> lk(w->mu);
      Post &p = w->box[(size_t)c->rank * w->n + o.peer];
      w->cv.wait(lk, [&] { return p.taken; });
      if (hipStreamWaitEvent(o.stream, p.copied, 0) != hipSuccess) rc = ncclUnhandledCudaError;
      p.posted = false;
      w->cv.notify_all();
    }
  t_ops.clear();
  return rc;
}
ncclResult_t ncclAllGather(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) { return ncclInvalidUsage; }
}
