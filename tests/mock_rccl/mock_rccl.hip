/*
 * mock_rccl.hip — TEST INFRASTRUCTURE.  A stand-in for librccl.so.1 that lets the library's RCCL halo path
 * (wdpm_amd/csrc/wdpm_rccl.hip) run with SEVERAL ranks on the ONE GPU of a test box, which the real RCCL
 * refuses ("duplicate GPU").  It implements just the entry points wdpm_rccl.hip binds, two ways:
 *
 *  - the ranks of ONE process (ncclCommInitAll; one host thread per rank, as wdpm_group drives them):
 *    ncclSend / ncclRecv inside ncclGroupStart / ncclGroupEnd = device-to-device copies ordered by events:
 *    the receiver's stream waits for an event the sender recorded on ITS stream when it posted the send, then
 *    copies; the sender's stream waits for the receiver's "copied" event before anything queued later.
 *  - one PROCESS per rank (ncclCommInitRank with the id every rank was handed; what bench.py's ranks and
 *    wdpm_rank_create do): the ranks meet in a POSIX shared-memory segment named after the id.  A send posts an
 *    IPC handle of the allocation that holds the rows (hipIpcGetMemHandle: needs HSA_ENABLE_IPC_MODE_LEGACY=0 on
 *    this pool, as the real RCCL does) + offset + size once its stream has produced them; the receiver maps the
 *    handle (hipIpcOpenMemHandle, cached), copies device to device on its own stream and acknowledges; the sender
 *    returns from ncclGroupEnd when its rows have been read.  ncclAllGather goes through the same segment.  If the
 *    platform refuses IPC handles the rows are staged through a shared-memory file instead (MOCK_RCCL_WIRE=shm
 *    forces that); stderr says which wire ran.  Every wait is bounded (MOCK_RCCL_TIMEOUT_S, default 60): a rank
 *    that never shows up is an error return, not a hang.
 *
 * So the data path is the product's own (communicator set-up from an id, op lists, row offsets, streams,
 * tile-flag and max-diff bookkeeping around a refresh, the overlapped last iteration with a communicator attached,
 * the block scalars through ncclAllGather); only the wire is faked.
 * Fault injection (tests/test_mock_rccl.py) - a rank of a thread group that dies between collectives: MOCK_RCCL_FAIL_RANK=<r>
 * with MOCK_RCCL_FAIL_AFTER=<n> makes that rank's (n+1)-th transfer fail before anything is posted; the product must then end
 * every rank's communicator (ncclCommAbort wakes whoever waits here) and fail, not hang.  For the product's deadlines: MOCK_RCCL_HANG_INIT_RANK=<r> makes that
 * rank's ncclCommInitRank sleep MOCK_RCCL_HANG_S seconds (default 30) before failing; MOCK_RCCL_BLOCK_FIRST_S=<s> keeps the first
 * transfer of a thread group's rank MOCK_RCCL_BLOCK_RANK inside the call for s seconds; MOCK_RCCL_STALL_RECV_S=<s>
 * makes the FIRST receive of every rank return at once with the stream blocked for s seconds by a host function
 * (a transfer that never completes, as seen from the host), bounded so that nothing can hang the box.
 * Loaded through WDPM_RCCL_LIB by the tests; never part of the product.
 */
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {
constexpr int kGatherBytes = 1024;       // per rank and all-gather round
// ---- ranks of one process ------------------------------------------------------------------------------------
struct Post { const void *buf; size_t bytes; hipEvent_t ready, copied; bool posted, taken; };
struct World {
  int n;
  std::mutex mu;
  std::condition_variable cv;
  std::vector<Post> box;                 // [src * n + dst]
  bool dead = false;                     // ncclCommAbort on any rank: everybody's waits end with an error
  int group_ends = 0;                    // fault injection: MOCK_RCCL_FAIL_AFTER counts rank MOCK_RCCL_FAIL_RANK's groups
  bool blocked = false;                  // fault injection: MOCK_RCCL_BLOCK_FIRST_S has had its one call
  std::vector<uint64_t> gather_seq;      // ncclAllGather between the ranks of one process: round each rank has posted
  std::vector<unsigned char> gather[2];  // two rounds of n x kGatherBytes (a rank is at most one round ahead of the slowest)
};

// ---- one process per rank ------------------------------------------------------------------------------------
constexpr int kMaxRanks = 64;
struct Mail {                            // one per ordered pair (src, dst), in shared memory
  std::atomic<uint64_t> posted, taken;   // sequence numbers: messages posted by src / read by dst
  hipIpcMemHandle_t handle;              // allocation holding the rows (wire 0)
  uint64_t offset, bytes;
  int wire;                              // 0 = IPC-mapped device memory, 1 = staged through a shared-memory file
};
struct Shared {
  std::atomic<int> arrived, dead;
  std::atomic<uint64_t> gather_seq[kMaxRanks];
  unsigned char gather[2][kMaxRanks][kGatherBytes];
  Mail mail[kMaxRanks * kMaxRanks];
};
struct Mapped { hipIpcMemHandle_t h; void *p; };
struct Proc {
  Shared *sh = nullptr;
  std::string name;
  uint64_t sent[kMaxRanks] = {}, got[kMaxRanks] = {}, gather_round = 0;
  std::vector<Mapped> mapped;
  int wire = 0;
  bool said = false, stalled = false;
};

struct Op { bool send; void *buf; size_t bytes; int peer; hipStream_t stream; };
thread_local std::vector<Op> t_ops;
thread_local int t_depth = 0;
thread_local ncclComm *t_comm = nullptr;

double env_num(const char *name, double dflt) {
  const char *e = getenv(name);
  return e && *e ? atof(e) : dflt;
}
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// spin (politely) until pred() or the deadline; false on timeout or when another rank has aborted
template <class F> bool wait_for(Shared *sh, F pred) {
  const double limit = now_s() + env_num("MOCK_RCCL_TIMEOUT_S", 60.0);
  for (int spins = 0; !pred(); spins++) {
    if (sh->dead.load(std::memory_order_acquire)) return false;
    if (now_s() > limit) return false;
    if (spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
    else std::this_thread::yield();
  }
  return true;
}
}  // namespace

struct ncclComm { World *w; int rank; int dev; int n; Proc *proc; uint64_t gather_round = 0; };

static size_t type_size(ncclDataType_t t) { return t == ncclDouble || t == ncclInt64 || t == ncclUint64 ? 8 : (t == ncclFloat || t == ncclInt32 || t == ncclUint32 ? 4 : 1); }

static std::string stage_name(const Proc *p, int src, int dst) { return p->name + "_" + std::to_string(src) + "_" + std::to_string(dst); }

// the rows of one message through a shared-memory file (the wire when IPC handles cannot be had)
static bool stage_put(const Proc *p, int src, int dst, const void *dev, size_t bytes) {
  const std::string nm = stage_name(p, src, dst);
  const int fd = shm_open(nm.c_str(), O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) { if (fd >= 0) close(fd); return false; }
  void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) return false;
  const bool ok = hipMemcpy(m, dev, bytes, hipMemcpyDeviceToHost) == hipSuccess;
  munmap(m, bytes);
  return ok;
}
static bool stage_get(const Proc *p, int src, int dst, void *dev, size_t bytes, hipStream_t s) {
  const std::string nm = stage_name(p, src, dst);
  const int fd = shm_open(nm.c_str(), O_RDWR, 0600);
  if (fd < 0) return false;
  void *m = mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) return false;
  bool ok = hipMemcpyAsync(dev, m, bytes, hipMemcpyHostToDevice, s) == hipSuccess;
  ok = ok && hipStreamSynchronize(s) == hipSuccess;
  munmap(m, bytes);
  shm_unlink(nm.c_str());
  return ok;
}

static void *map_handle(Proc *p, const hipIpcMemHandle_t &h) {
  for (const Mapped &m : p->mapped)
    if (!memcmp(&m.h, &h, sizeof h)) return m.p;
  void *ptr = nullptr;
  if (hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  p->mapped.push_back(Mapped{h, ptr});
  return ptr;
}

static ncclResult_t group_end_threads(ncclComm *c) {
  World *w = c->w;
  ncclResult_t rc = ncclSuccess;
  {
    // fault injection: this rank's (n+1)-th transfer fails before anything is posted - a rank that dies between collectives
    const int fail_rank = (int)env_num("MOCK_RCCL_FAIL_RANK", -1.0), fail_after = (int)env_num("MOCK_RCCL_FAIL_AFTER", -1.0);
    std::unique_lock<std::mutex> lk(w->mu);
    if (w->dead) return ncclSystemError;
    if (c->rank == fail_rank && fail_after >= 0 && w->group_ends++ >= fail_after) return ncclSystemError;
    // fault injection: the FIRST transfer of rank MOCK_RCCL_BLOCK_RANK (default 0) does not come back for MOCK_RCCL_BLOCK_FIRST_S
    // seconds (a fabric that never answers): the product's deadline for a communicator's first transfer must end the run
    const double block = env_num("MOCK_RCCL_BLOCK_FIRST_S", 0.0);
    if (block > 0 && c->rank == (int)env_num("MOCK_RCCL_BLOCK_RANK", 0.0) && !w->blocked) {
      w->blocked = true;
      lk.unlock();
      std::this_thread::sleep_for(std::chrono::milliseconds((long)(block * 1000.0)));
      return ncclSystemError;
    }
  }
  // post every send: "my rows are produced once this event fires"
  for (const Op &o : t_ops)
    if (o.send) {
      std::unique_lock<std::mutex> lk(w->mu);
      Post &p = w->box[(size_t)c->rank * w->n + o.peer];
      w->cv.wait(lk, [&] { return !p.posted || w->dead; });      // the previous message of this pair was taken
      if (w->dead) return ncclSystemError;
      if (hipEventRecord(p.ready, o.stream) != hipSuccess) rc = ncclUnhandledCudaError;
      p.buf = o.buf; p.bytes = o.bytes; p.posted = true; p.taken = false;
      w->cv.notify_all();
    }
  // take every receive: copy on my stream behind the sender's event
  for (const Op &o : t_ops)
    if (!o.send) {
      std::unique_lock<std::mutex> lk(w->mu);
      Post &p = w->box[(size_t)o.peer * w->n + c->rank];
      w->cv.wait(lk, [&] { return (p.posted && !p.taken) || w->dead; });
      if (w->dead) return ncclSystemError;
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
      w->cv.wait(lk, [&] { return p.taken || w->dead; });
      if (w->dead) return ncclSystemError;
      if (hipStreamWaitEvent(o.stream, p.copied, 0) != hipSuccess) rc = ncclUnhandledCudaError;
      p.posted = false;
      w->cv.notify_all();
    }
  return rc;
}

static void stall_host_fn(void *arg) { std::this_thread::sleep_for(std::chrono::milliseconds((long)(size_t)arg)); }

static ncclResult_t group_end_processes(ncclComm *c) {
  Proc *p = c->proc;
  Shared *sh = p->sh;
  const int n = c->n;
  const double stall = env_num("MOCK_RCCL_STALL_RECV_S", 0.0);
  if (stall > 0 && !p->stalled) {
    // fault injection: "the transfer was queued and never completes" - the call returns, the stream does not move
    p->stalled = true;
    for (const Op &o : t_ops)
      if (!o.send) { (void)hipLaunchHostFunc(o.stream, stall_host_fn, (void *)(size_t)(stall * 1000.0)); break; }
    return ncclSuccess;
  }
  // post every send once its rows exist
  for (const Op &o : t_ops)
    if (o.send) {
      if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
      Mail &m = sh->mail[(size_t)c->rank * n + o.peer];
      const uint64_t seq = p->sent[o.peer];
      if (!wait_for(sh, [&] { return m.taken.load(std::memory_order_acquire) == seq; })) return ncclSystemError;
      m.bytes = o.bytes;
      m.wire = p->wire;
      if (p->wire == 0) {
        void *base = nullptr;
        size_t size = 0;
        if (hipMemGetAddressRange((hipDeviceptr_t *)&base, &size, (hipDeviceptr_t)o.buf) != hipSuccess ||
            hipIpcGetMemHandle(&m.handle, base) != hipSuccess) {
          (void)hipGetLastError();
          p->wire = m.wire = 1;          // this platform hands out no IPC handles: stage through shared memory from now on
        } else {
          m.offset = (uint64_t)((char *)o.buf - (char *)base);
        }
      }
      if (m.wire == 1 && !stage_put(p, c->rank, o.peer, o.buf, o.bytes)) return ncclSystemError;
      if (!p->said) {
        p->said = true;
        fprintf(stderr, "mock RCCL: rank %d of %d (pid %d) sends over %s\n", c->rank, n, (int)getpid(),
                m.wire == 0 ? "IPC-mapped device memory" : "a shared-memory file");
      }
      p->sent[o.peer] = seq + 1;
      m.posted.store(seq + 1, std::memory_order_release);
    }
  // take every receive
  ncclResult_t rc = ncclSuccess;
  for (const Op &o : t_ops)
    if (!o.send) {
      Mail &m = sh->mail[(size_t)o.peer * n + c->rank];
      const uint64_t seq = p->got[o.peer] + 1;
      if (!wait_for(sh, [&] { return m.posted.load(std::memory_order_acquire) >= seq; })) return ncclSystemError;
      if (m.bytes != o.bytes) rc = ncclInvalidArgument;          // send / recv sizes must match, as with the real thing
      const size_t bytes = o.bytes < m.bytes ? o.bytes : (size_t)m.bytes;
      if (m.wire == 0) {
        char *src = (char *)map_handle(p, m.handle);
        if (!src) return ncclUnhandledCudaError;
        if (hipMemcpyAsync(o.buf, src + m.offset, bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess ||
            hipStreamSynchronize(o.stream) != hipSuccess)
          rc = ncclUnhandledCudaError;
      } else if (!stage_get(p, o.peer, c->rank, o.buf, bytes, o.stream)) {
        rc = ncclSystemError;
      }
      p->got[o.peer] = seq;
      m.taken.store(seq, std::memory_order_release);
    }
  // my sends: the receiver has read the rows before I may queue anything that overwrites them
  for (const Op &o : t_ops)
    if (o.send) {
      Mail &m = sh->mail[(size_t)c->rank * n + o.peer];
      const uint64_t seq = p->sent[o.peer];
      if (!wait_for(sh, [&] { return m.taken.load(std::memory_order_acquire) == seq; })) return ncclSystemError;
    }
  return rc;
}

extern "C" {
ncclResult_t ncclGetVersion(int *v) { *v = 29999; return ncclSuccess; }
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "mock RCCL error"; }
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  memset(id, 0x5a, sizeof *id);
  const uint64_t a = (uint64_t)getpid(), b = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
  memcpy(id->internal, &a, sizeof a);
  memcpy(id->internal + 8, &b, sizeof b);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int n, ncclUniqueId id, int rank) {
  if (n < 1 || n > kMaxRanks || rank < 0 || rank >= n) return ncclInvalidArgument;
  const int hang_rank = (int)env_num("MOCK_RCCL_HANG_INIT_RANK", -1.0);
  if (hang_rank == rank) {               // fault injection: a communicator set-up that does not come back (bounded)
    std::this_thread::sleep_for(std::chrono::milliseconds((long)(env_num("MOCK_RCCL_HANG_S", 30.0) * 1000.0)));
    return ncclSystemError;
  }
  uint64_t a, b;
  memcpy(&a, id.internal, sizeof a);
  memcpy(&b, id.internal + 8, sizeof b);
  char name[96];
  snprintf(name, sizeof name, "/mock_rccl_%llx_%llx", (unsigned long long)a, (unsigned long long)b);
  // whoever comes first creates the segment (a fresh one is all zeros); the others find it
  int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd >= 0) {
    if (ftruncate(fd, sizeof(Shared)) != 0) { close(fd); shm_unlink(name); return ncclSystemError; }
  } else {
    const double limit = now_s() + env_num("MOCK_RCCL_TIMEOUT_S", 60.0);
    struct stat st;
    for (;;) {
      fd = shm_open(name, O_RDWR, 0600);
      if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= sizeof(Shared)) break;
      if (fd >= 0) close(fd);
      if (now_s() > limit) return ncclSystemError;
      std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
  }
  void *m = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) return ncclSystemError;
  Proc *p = new Proc;
  p->sh = (Shared *)m;
  p->name = name;
  const char *wire = getenv("MOCK_RCCL_WIRE");
  p->wire = wire && !strcmp(wire, "shm") ? 1 : 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const int mine = p->sh->arrived.fetch_add(1) + 1;
  if (!wait_for(p->sh, [&] { return p->sh->arrived.load(std::memory_order_acquire) >= n; })) {
    p->sh->dead.store(1);
    shm_unlink(name);
    return ncclSystemError;
  }
  if (mine == n) shm_unlink(name);       // everybody has it mapped: the name can go
  *comm = new ncclComm{nullptr, rank, dev, n, p};
  return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *comm, int n, const int *devs) {
  World *w = new World;
  w->n = n;
  w->box.assign((size_t)n * n, Post{nullptr, 0, nullptr, nullptr, false, false});
  for (int i = 0; i < n * n; i++) {
    if (hipEventCreateWithFlags(&w->box[i].ready, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
    if (hipEventCreateWithFlags(&w->box[i].copied, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
  }
  w->gather_seq.assign((size_t)n, 0);
  w->gather[0].assign((size_t)n * kGatherBytes, 0);
  w->gather[1].assign((size_t)n * kGatherBytes, 0);
  for (int i = 0; i < n; i++) comm[i] = new ncclComm{w, i, devs ? devs[i] : i, n, nullptr};
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) {   // the World leaks: a test process
  if (c->proc) {
    for (const Mapped &m : c->proc->mapped) (void)hipIpcCloseMemHandle(m.p);
    munmap(c->proc->sh, sizeof(Shared));
    delete c->proc;
  }
  delete c;
  return ncclSuccess;
}
ncclResult_t ncclCommAbort(ncclComm_t c) {
  if (c->proc) c->proc->sh->dead.store(1, std::memory_order_release);   // everybody's bounded waits end now
  if (c->w) {                                                           // ranks of one process: wake whoever waits for a post
    std::unique_lock<std::mutex> lk(c->w->mu);
    c->w->dead = true;
    c->w->cv.notify_all();
    return ncclSuccess;                  // (the communicator object stays: another rank thread may be inside a call with it)
  }
  return ncclCommDestroy(c);
}
ncclResult_t ncclCommCount(const ncclComm_t c, int *n) { *n = c->n; return ncclSuccess; }
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
  const ncclResult_t rc = c->proc ? group_end_processes(c) : group_end_threads(c);
  t_ops.clear();
  return rc;
}
ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t s) {
  if (!c->proc) {
    // ranks of one process (ncclCommInitAll), one host thread each - wdpm_group itself never asks (its threads share memory),
    // tests/test_multi_gpu.py does, as a caller of the real RCCL may
    World *w = c->w;
    const size_t nbytes = count * type_size(t);
    if (nbytes > (size_t)kGatherBytes) return ncclInvalidArgument;
    std::vector<unsigned char> mine(nbytes), all((size_t)c->n * nbytes);
    if (hipMemcpyAsync(mine.data(), send, nbytes, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
      return ncclUnhandledCudaError;
    {
      std::unique_lock<std::mutex> lk(w->mu);
      const uint64_t round = ++c->gather_round;
      std::vector<unsigned char> &slot = w->gather[round & 1];
      memcpy(slot.data() + (size_t)c->rank * kGatherBytes, mine.data(), nbytes);
      w->gather_seq[(size_t)c->rank] = round;
      w->cv.notify_all();
      const bool ok = w->cv.wait_for(lk, std::chrono::duration<double>(env_num("MOCK_RCCL_TIMEOUT_S", 60.0)), [&] {
        if (w->dead) return true;
        for (int q = 0; q < c->n; q++)
          if (w->gather_seq[(size_t)q] < round) return false;
        return true;
      });
      if (!ok || w->dead) return ncclSystemError;
      for (int q = 0; q < c->n; q++) memcpy(all.data() + (size_t)q * nbytes, slot.data() + (size_t)q * kGatherBytes, nbytes);
    }
    if (hipMemcpyAsync(recv, all.data(), all.size(), hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
      return ncclUnhandledCudaError;
    return ncclSuccess;
  }
  Proc *p = c->proc;
  Shared *sh = p->sh;
  const size_t bytes = count * type_size(t);
  if (bytes > (size_t)kGatherBytes) return ncclInvalidArgument;
  const uint64_t round = ++p->gather_round;
  unsigned char(*slot)[kGatherBytes] = sh->gather[round & 1];
  if (hipMemcpyAsync(slot[c->rank], send, bytes, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
    return ncclUnhandledCudaError;
  sh->gather_seq[c->rank].store(round, std::memory_order_release);
  std::vector<unsigned char> all((size_t)c->n * bytes);
  for (int q = 0; q < c->n; q++) {
    // a rank can be at most one round ahead of the slowest (it waits here for everybody), so two slots do
    if (!wait_for(sh, [&] { return sh->gather_seq[q].load(std::memory_order_acquire) >= round; })) return ncclSystemError;
    memcpy(all.data() + (size_t)q * bytes, slot[q], bytes);
  }
  if (hipMemcpyAsync(recv, all.data(), all.size(), hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
    return ncclUnhandledCudaError;
  return ncclSuccess;
}
}
