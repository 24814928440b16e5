/*
 * wdpm_capi.hip — the C ABI of include/wdpm.h implemented on HIP for gfx950.
 *
 * One context = one raster slab resident in one GPU's HBM for the context's lifetime:
 *   dem      rows x (C+2) fp64   read-only after upload; NODATA cells are rewritten to +inf on the device
 *   water[2] rows x (C+2) fp64   ping-pong pair (the fused kernel reads one, writes the other)
 *   old      rows x (C+2) fp64   snapshot for the convergence test
 *   scal     {totaldrain, olddrain} fp64 + one uint64 reduction cell
 * No CPU fallback: every compute entry point launches a HIP kernel or fails.
 * Replaces the reference's OpenCL host path (src/WDPMCL.c:598-638, :1126-1236).
 */
#include <hip/hip_runtime.h>

#include <time.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <vector>

#include "wdpm_ctx.h"

/* chunk height of the iteration kernel once most tiles are dry (24 ... 384 rows measured in round 2: profiles/r02/sparse.txt) */
constexpr int kSparseChunkRows = 96;

static thread_local char g_err[512] = "";

int wdpm_fail(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return 1;
}
#define fail wdpm_fail

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

struct wdpm_ctx;
static int tiles_touch(wdpm_ctx *x, int slot, int row, int nrows);
int wdpm_tiles_touch(wdpm_ctx *x, int row, int nrows) { return tiles_touch(x, x->cur, row, nrows); }
static int ensure_drained(wdpm_ctx *x);
int wdpm_apply_owed_drain(wdpm_ctx *x) { return ensure_drained(x); }
static int ensure_flushed(wdpm_ctx *x);
int wdpm_apply_owed_flush(wdpm_ctx *x) { return ensure_flushed(x); }
static void flushed_whole(wdpm_ctx *x, double thres);

int wdpm_stream_sync(wdpm_ctx *x, hipStream_t s) {
  if (!x->comm) {
    const hipError_t e = hipStreamSynchronize(s);
    return e == hipSuccess ? 0 : wdpm_fail("hipStreamSynchronize failed: %s", hipGetErrorString(e));
  }
  static std::atomic<int> limit_ms{0};
  if (!limit_ms) { const char *t = getenv("WDPM_SYNC_TIMEOUT_S"); const double v = t ? atof(t) : 0.0; limit_ms = (int)((v > 0 ? v : 600.0) * 1000.0); }
  timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (;;) {
    const hipError_t e = hipStreamQuery(s);
    if (e == hipSuccess) return 0;
    if (e != hipErrorNotReady) return wdpm_fail("hipStreamQuery failed: %s", hipGetErrorString(e));
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    const double ms = (double)(t.tv_sec - t0.tv_sec) * 1e3 + (double)(t.tv_nsec - t0.tv_nsec) * 1e-6;
    if (ms > (double)limit_ms) break;
    if (ms > 5.0) { const timespec nap{0, 100000}; nanosleep(&nap, nullptr); }   /* short waits spin, long ones nap 0.1 ms */
  }
  /* a halo transfer whose peer never sent (or died): end the communicator so that the queue drains, and say so */
  x->leak = true;
  (void)wdpm_comm_abort(x);
  return wdpm_fail("the GPU did not finish within %.0f s (WDPM_SYNC_TIMEOUT_S): a halo transfer did not complete; "
                   "the communicator was aborted", limit_ms / 1000.0);
}

extern "C" {

const char *wdpm_last_error(void) { return g_err; }
void wdpm_set_last_error(const char *msg) { snprintf(g_err, sizeof g_err, "%s", msg ? msg : ""); }
const char *wdpm_backend_name(void) { return "hip-gfx950"; }
int wdpm_abi_version(void) { return WDPM_ABI_VERSION; }

static int bind(wdpm_ctx *x) {
  HIP_TRY(hipSetDevice(x->p.device));
  if (x->pending_join) {   /* the interior launch of the last overlapped iteration: order it before anything new */
    HIP_TRY(hipStreamWaitEvent(x->stream, x->ev_join, 0));
    x->pending_join = false;
  }
  return 0;
}

/* The snapshot of a block is not a copy: wdpm_begin_block just declares the current raster to BE the
 * snapshot (old = cur) and owes it the threshold flush; the next iteration launch reads it - flushing
 * while it loads - and writes the other two rasters in turn, so the snapshot stays untouched until
 * wdpm_max_diff reads it (applying the same flush on the fly).  At 16384^2 that is 1.3 ms per block
 * that no longer exists.  Whoever needs the current raster flushed in memory, or wants to write
 * into it, asks for that first: */
/* rows [row, row + nrows) of d_w[slot] are about to be (or have been) written by somebody other than the
 * iteration kernel: the tiles they touch are no longer known to be dry */
static int tiles_touch(wdpm_ctx *x, int slot, int row, int nrows) {
  if (slot == x->cur && nrows > 0 && row < x->md_hi && row + nrows > x->md_lo) x->md_valid = false;   /* rows of the folded max diff */
  if (!x->zero_valid[slot] || nrows <= 0) return 0;
  if (x->tile_H < 6 || x->tile_nchunks < 1) { x->zero_valid[slot] = false; return 0; }
  /* chunk i's block is rows [H*i + 2 (0 for i = 0), H*(i+1) + 1] */
  int c0 = (row - 2) / x->tile_H, c1 = (row + nrows - 1 - 2) / x->tile_H;
  if (row < 2) c0 = 0;
  if (c0 < 0) c0 = 0;
  if (c1 > x->tile_nchunks - 1) c1 = x->tile_nchunks - 1;
  if (c1 < c0) return 0;
  const size_t pitch = (size_t)x->tile_nstrips + 2;
  HIP_TRY(hipMemset2DAsync(x->d_zero[slot] + (size_t)(c0 + 1) * pitch + 1, pitch, 0, (size_t)x->tile_nstrips, (size_t)(c1 - c0 + 1),
                           x->stream));
  return 0;
}

static int free_slot(const wdpm_ctx *x) {            /* a raster that is neither current nor the snapshot */
  for (int i = 0; i < 3; i++)
    if (i != x->cur && i != x->old) return i;
  return 0;
}

/* Drain module: drain() after each iteration (WDPMCL.c:1089) has no launch of its own either - the next
 * iteration launch applies it while loading (wdpm_fused.hip: drain_owed).  Anybody else who looks at the
 * raster or at totaldrain gets it applied first.  (cur != old whenever it is owed: an iteration has run.) */
static int ensure_drained(wdpm_ctx *x) {
  if (!x->drain_owed) return 0;
  x->drain_owed = false;
  HIP_TRY(wdpm_launch_drain_outlet(x->d_w[x->cur], x->d_dem, x->g, x->d_scal, x->stream));
  return 0;
}

/* readers of d_w[cur]: the owed flush is applied in place (cur == old: the snapshot gets it too, idempotently) */
static int ensure_flushed(wdpm_ctx *x) {
  if (ensure_drained(x)) return 1;
  if (!x->flush_pending) return 0;
  HIP_TRY(wdpm_launch_flush_snapshot(x->d_w[x->cur], x->d_w[x->cur], x->cells, x->flush_thres, x->stream));
  x->flush_pending = false;
  flushed_whole(x, x->flush_thres);
  return 0;
}

/* in-place writers of d_w[cur] (pass kernels, drain(), partial uploads): the snapshot must survive */
static int ensure_private(wdpm_ctx *x) {
  x->md_valid = false;             /* whoever asks is about to write into the raster */
  if (ensure_drained(x)) return 1;
  if (x->cur != x->old) return 0;
  const int t = free_slot(x);
  if (x->flush_pending) {
    HIP_TRY(wdpm_launch_flush_snapshot(x->d_w[x->cur], x->d_w[t], x->cells, x->flush_thres, x->stream));
    x->flush_pending = false;
    flushed_whole(x, x->flush_thres);      /* (the copy that becomes current is the flushed one) */
  } else {
    HIP_TRY(hipMemcpyAsync(x->d_w[t], x->d_w[x->cur], x->cells * sizeof(double), hipMemcpyDeviceToDevice, x->stream));
  }
  x->zero_valid[t] = false;      /* a copy: its flags could be copied too, but whoever asked is about to write into it */
  x->cur = t;
  return 0;
}

/* a whole new raster is about to be written: no copy needed, just do not overwrite the snapshot */
static int ensure_fresh_slot(wdpm_ctx *x) {
  if (ensure_flushed(x)) return 1;
  if (x->cur == x->old) x->cur = free_slot(x);
  x->zero_valid[x->cur] = false;
  x->md_valid = false;
  return 0;
}

/* Guard bands (debugging aid, WDPM_GUARD_KB=<n> in the environment when the context is made): the big device buffers - DEM,
 * the three water rasters, the DEM codes, the tile flags - get n KiB of 0xA5 bytes in front and behind, and
 * wdpm_get_option(WDPM_OPT_GUARD_BAD) counts guard bytes that no longer hold 0xA5: a kernel writing outside its raster shows up
 * there even when the stray bytes land in nobody's data.  The GPU pool has no address sanitizer; the fuzz tests run with this on. */
static size_t guard_bytes() {
  static std::atomic<long> kb{-1};
  if (kb < 0) { const char *e = getenv("WDPM_GUARD_KB"); kb = e ? atol(e) : 0; if (kb < 0) kb = 0; }
  return (size_t)kb * 1024;
}
static hipError_t guarded_malloc(wdpm_ctx *x, void **p, size_t bytes) {
  const size_t g = guard_bytes();
  char *base = nullptr;
  hipError_t e = hipMalloc(&base, bytes + 2 * g);
  if (e != hipSuccess) return e;
  if (g) {
    e = hipMemset(base, 0xA5, g);
    if (e == hipSuccess) e = hipMemset(base + g + bytes, 0xA5, g);
    if (e != hipSuccess) { (void)hipFree(base); return e; }
    x->guards.push_back({base, bytes});
  }
  *p = base + g;
  return hipSuccess;
}
static void guarded_free(void *p) {
  if (p) (void)hipFree(static_cast<char *>(p) - guard_bytes());
}
static int guard_damage(wdpm_ctx *x, int64_t *bad) {
  *bad = 0;
  const size_t g = guard_bytes();
  if (!g || x->guards.empty()) return 0;
  if (bind(x)) return 1;
  if (wdpm_stream_sync(x, x->stream)) return 1;
  if (x->side && wdpm_stream_sync(x, x->side)) return 1;
  std::vector<unsigned char> h(g);
  for (const auto &gb : x->guards)
    for (int side = 0; side < 2; side++) {
      HIP_TRY(hipMemcpy(h.data(), gb.base + (side ? g + gb.bytes : 0), g, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < g; i++) *bad += h[i] != 0xA5;
    }
  return 0;
}

int wdpm_create(wdpm_ctx **out, const wdpm_params *p) {
  if (!out || !p) return fail("wdpm_create: null argument");
  if (p->nrows < 1 || p->ncols < 1) return fail("wdpm_create: bad raster size %d x %d", p->nrows, p->ncols);
  if (p->module < WDPM_ADD || p->module > WDPM_DRAIN) return fail("wdpm_create: bad module %d", p->module);
  if (p->kernel < WDPM_KERNEL_AUTO || p->kernel > WDPM_KERNEL_FUSED) return fail("wdpm_create: bad kernel selector %d", p->kernel);
  if (p->slab_row0 < 0 || p->slab_row0 % 3 != 0) return fail("wdpm_create: slab_row0 must be a non-negative multiple of 3");
  const int rows = p->slab_rows > 0 ? p->slab_rows : p->nrows + 2;
  if (p->slab_row0 + rows > p->nrows + 2) return fail("wdpm_create: slab exceeds raster");
  if (rows < 3) return fail("wdpm_create: a slab needs at least 3 rows");
  if ((double)rows * (double)(p->ncols + 2) > 2.0e9) return fail("wdpm_create: raster too large: a row block takes at most 2e9 cells (%d rows x %d here; use more row blocks)", rows, p->ncols + 2);
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (ndev < 1) return fail("wdpm_create: no HIP device (this library has no CPU fallback)");
  if (p->device < 0 || p->device >= ndev) return fail("wdpm_create: device %d out of range (%d devices)", p->device, ndev);
  HIP_TRY(hipSetDevice(p->device));

  wdpm_ctx *x = new wdpm_ctx();
  x->p = *p;
  x->g.rows = rows;
  x->g.ncp = p->ncols + 2;
  x->g.row0 = p->slab_row0;
  x->g.R = p->nrows;
  x->g.C = p->ncols;
  x->g.dr = p->drainrow - p->slab_row0;
  x->g.dc = p->draincol;
  x->g.miss = p->missingvalue;
  if (p->module != WDPM_DRAIN) { x->g.dr = -1000000; x->g.dc = -1000000; }
  x->cells = (size_t)rows * x->g.ncp;
  x->kernel = p->kernel;
  if (x->kernel == WDPM_KERNEL_AUTO) {
    /* The one-iteration kernel; WDPM_KERNEL=pass|fused selects one explicitly.  (Round 1 also carried two
     * two-iterations-per-launch kernels: -0.5 % .. +8 % against this one depending on slab shape and box,
     * VALU-bound; they were taken out of the product in round 2 - DESIGN.md §7b.) */
    x->kernel = WDPM_KERNEL_FUSED;
    if (const char *e = getenv("WDPM_KERNEL")) {
      if (!strcmp(e, "pass")) x->kernel = WDPM_KERNEL_PASS;
      else if (!strcmp(e, "fused")) x->kernel = WDPM_KERNEL_FUSED;
    }
  }
  x->cur = 0;
  x->signed_zero_safe = false;
  x->w_negative = false;
  x->w_odd = true;                /* nothing known about the rasters before the first upload */
  x->launches = 0;
  x->ms = 0.0;
  x->steady_launches = 0;
  x->steady_ms = 0.0;
  x->timing = false; x->xch_count = 0; x->xch_ms = 0.0;
  x->comm = nullptr;
  x->leak = false;
  x->d_dem = x->d_w[0] = x->d_w[1] = x->d_w[2] = nullptr;
  x->old = 2;
  for (int i = 0; i < 3; i++) { x->d_zero[i] = nullptr; x->zero_valid[i] = false; }
  x->tile_cap = x->tile_nstrips = x->tile_H = x->tile_nchunks = 0;
  x->d_active = nullptr; x->h_active = nullptr;
  x->tiles_launched = 0; x->sparse = false; x->stat_tiles = x->stat_active = 0;
  x->wide_tri_ok = false; x->blocks_unprobed = 0;
  memset(&x->bal, 0, sizeof x->bal);
  { const char *e = getenv("WDPM_TILES"); x->tiles_mode = e ? atoi(e) : 1; }
  x->flush_pending = false;
  x->d_md = nullptr; x->md_hint = x->md_valid = false; x->md_lo = x->md_hi = 0;
  x->drain_owed = false;
  x->flush_thres = -__builtin_inf();
  x->d_scal = nullptr; x->d_bits = nullptr; x->h_pin = nullptr; x->d_stat = nullptr;
  x->d_dem32 = nullptr; x->code = DemCode{nullptr, 0.0, 1.0, 1.0, 0, nullptr, nullptr, 0}; x->dem32_encodable = false; x->dem_bounded = false;
  x->d_dem16 = nullptr; x->d_gbase = nullptr; x->dem16_encodable = false; x->dem16_wanted = true;
  x->graph_mode = -1; x->graph_launches = 0;
  x->d_sum_approx = nullptr; x->d_sum_i = nullptr; x->d_sum_k = nullptr; x->d_sum_flag = nullptr;
  x->own_stream = true;
  x->side = nullptr; x->ev_fork = nullptr; x->ev_join = nullptr; x->pending_join = false; x->ev_copy[0] = x->ev_copy[1] = nullptr;
  const size_t bytes = x->cells * sizeof(double);
  hipError_t e = hipStreamCreateWithFlags(&x->stream, hipStreamNonBlocking);
  /* the side stream carries the interior launch of wdpm_iterate_overlapped: lowest priority, so that
   * the boundary launches on the main stream (whose rows a neighbour is waiting for) get the CUs first */
  int prio_least = 0, prio_greatest = 0;
  if (e == hipSuccess) e = hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  if (e == hipSuccess) e = hipStreamCreateWithPriority(&x->side, hipStreamNonBlocking, prio_least);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&x->ev_fork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&x->ev_join, hipEventDisableTiming);
  if (e == hipSuccess) e = guarded_malloc(x, (void **)&x->d_dem, bytes + 192 * sizeof(double));   /* spare cells: see the water rasters */
  /* + 192 doubles behind each raster: the fused kernel's dump area for masked-out stores (a strip's width), and what an edge
   * wave's three-column loads may read past the end of the slab's last row (two cells; the DEM and its codes have spare cells too) */
  if (e == hipSuccess) e = guarded_malloc(x, (void **)&x->d_w[0], bytes + 192 * sizeof(double));
  if (e == hipSuccess) e = guarded_malloc(x, (void **)&x->d_w[1], bytes + 192 * sizeof(double));
  if (e == hipSuccess) e = guarded_malloc(x, (void **)&x->d_w[2], bytes + 192 * sizeof(double));
  if (e == hipSuccess) e = guarded_malloc(x, (void **)&x->d_dem32, x->cells * sizeof(int) + 64);
  if (e == hipSuccess) e = guarded_malloc(x, (void **)&x->d_dem16, x->cells * sizeof(unsigned short) + 64);
  if (e == hipSuccess) e = guarded_malloc(x, (void **)&x->d_gbase, (size_t)x->g.rows * ((x->g.ncp + kDemGroup - 1) / kDemGroup) * sizeof(int) + 64);
  if (e == hipSuccess) e = hipMalloc(&x->d_scal, 2 * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&x->d_bits, 2 * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMalloc(&x->d_md, sizeof(unsigned long long));
  if (e == hipSuccess) {
    /* chunk heights by what each XCD delivers (wdpm_kernels.h::XcdBalance).  WDPM_BALANCE=0: equal heights always; 2: the table on
     * launches of any size, starting from deliberately skewed weights (the parity suites under it) */
    const char *be = getenv("WDPM_BALANCE");
    x->bal.mode = be ? atoi(be) : 1;
    if (x->bal.mode < 0 || x->bal.mode > 2) x->bal.mode = 1;
    x->bal.capacity = 16384;
    if (x->bal.mode) {
      static const float even[kBalClasses] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 0.95f, 1.f};
      static const float skew[kBalClasses] = {0.8f, 1.2f, 0.9f, 1.1f, 1.25f, 0.75f, 1.0f, 1.0f, 0.8f, 1.15f};
      e = hipMalloc(&x->bal.table, (size_t)x->bal.capacity * sizeof(int));
      if (e == hipSuccess) e = hipMalloc(&x->bal.acc, (2 * kBalClasses + 2) * sizeof(unsigned long long));
      if (e == hipSuccess) e = hipMalloc(&x->bal.weight, kBalClasses * sizeof(float));
      if (e == hipSuccess) e = hipMemsetAsync(x->bal.acc, 0, (2 * kBalClasses + 2) * sizeof(unsigned long long), x->stream);
      if (e == hipSuccess) e = hipMemcpyAsync(x->bal.weight, x->bal.mode == 2 ? skew : even, kBalClasses * sizeof(float), hipMemcpyHostToDevice, x->stream);
    }
  }
  {
    /* tile flags: strips of 171 columns x chunks of >= 6 rows */
    x->tile_cap = (x->g.ncp / 171 + 4) * (rows / 6 + 4);
    for (int i = 0; i < 3 && e == hipSuccess; i++) e = guarded_malloc(x, (void **)&x->d_zero[i], (size_t)x->tile_cap);
    if (e == hipSuccess) e = hipMalloc(&x->d_active, sizeof(unsigned));
    if (e == hipSuccess) e = hipHostMalloc(&x->h_active, sizeof(unsigned));
    if (e == hipSuccess) e = hipMemsetAsync(x->d_active, 0, sizeof(unsigned), x->stream);
  }
  if (e == hipSuccess) e = hipHostMalloc(&x->h_pin, 4 * sizeof(double));
  if (e == hipSuccess) e = hipMemsetAsync(x->d_scal, 0, 2 * sizeof(double), x->stream);
  if (e == hipSuccess) e = hipMemsetAsync(x->d_w[2], 0, bytes, x->stream);   /* the snapshot before any block: zeros */
  if (e == hipSuccess) e = hipStreamSynchronize(x->stream);
  if (e != hipSuccess) {
    fail("wdpm_create: device allocation failed: %s", hipGetErrorString(e));
    wdpm_destroy(x);
    return 1;
  }
  *out = x;
  return 0;
}

void wdpm_destroy(wdpm_ctx *x) {
  if (!x) return;
  (void)hipSetDevice(x->p.device);
  if (!x->leak) (void)wdpm_stream_sync(x, x->stream);
  wdpm_comm_release(x);
  if (x->leak) return;           /* a stuck transfer or helper thread may still touch the stream and the buffers: nothing is freed */
  if (x->side) { (void)hipStreamSynchronize(x->side); (void)hipStreamDestroy(x->side); }
  if (x->ev_fork) (void)hipEventDestroy(x->ev_fork);
  if (x->ev_join) (void)hipEventDestroy(x->ev_join);
  for (int i = 0; i < 2; i++)
    if (x->ev_copy[i]) (void)hipEventDestroy(x->ev_copy[i]);
  for (auto &ge : x->graphs) (void)hipGraphExecDestroy(ge.exec);
  for (auto &ep : x->pending) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
  for (auto &ep : x->pending_steady) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
  for (auto &ep : x->pool) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
  guarded_free(x->d_dem); guarded_free(x->d_w[0]); guarded_free(x->d_w[1]); guarded_free(x->d_w[2]);
  (void)hipFree(x->d_scal); (void)hipFree(x->d_bits); guarded_free(x->d_dem32); (void)hipFree(x->d_stat);
  guarded_free(x->d_dem16); guarded_free(x->d_gbase);
  for (int i = 0; i < 3; i++) guarded_free(x->d_zero[i]);
  (void)hipFree(x->d_active);
  (void)hipFree(x->d_md);
  (void)hipFree(x->bal.table); (void)hipFree(x->bal.acc); (void)hipFree(x->bal.weight);
  if (x->h_active) (void)hipHostFree(x->h_active);
  if (x->h_pin) (void)hipHostFree(x->h_pin);
  (void)hipFree(x->d_sum_approx); (void)hipFree(x->d_sum_i); (void)hipFree(x->d_sum_k); (void)hipFree(x->d_sum_flag);
  if (x->own_stream && x->stream) (void)hipStreamDestroy(x->stream);
  delete x;
}

int wdpm_set_stream(wdpm_ctx *x, void *hip_stream) {
  if (bind(x)) return 1;   /* also joins a pending interior launch onto the old stream */
  if (wdpm_stream_sync(x, x->stream)) return 1;
  if (x->own_stream && x->stream) HIP_TRY(hipStreamDestroy(x->stream));
  x->stream = (hipStream_t)hip_stream;
  x->own_stream = false;
  return 0;
}

int wdpm_synchronize(wdpm_ctx *x) {
  if (bind(x)) return 1;
  if (wdpm_stream_sync(x, x->stream)) return 1;
  return 0;
}

/* ---- data movement ---------------------------------------------------------------------- */
/* after water has been written to rows [row, row+nrows): what kinds of depth does it hold (wdpm_launch_scan_water)? */
static int note_negzero(wdpm_ctx *x, int row, int nrows) {
  const size_t off = (size_t)row * x->g.ncp;
  HIP_TRY(hipMemsetAsync(x->d_bits, 0, sizeof(unsigned long long), x->stream));
  HIP_TRY(wdpm_launch_scan_water(x->d_w[x->cur] + off, x->d_dem + off, (size_t)nrows * x->g.ncp, x->d_bits, x->stream));
  HIP_TRY(hipMemcpyAsync(x->h_pin, x->d_bits, sizeof(double), hipMemcpyDeviceToHost, x->stream));
  if (wdpm_stream_sync(x, x->stream)) return 1;
  unsigned long long bits;
  memcpy(&bits, x->h_pin, sizeof bits);
  if (bits & WDPM_WATER_NEGZERO) x->signed_zero_safe = true;
  if (bits & WDPM_WATER_NEGATIVE) x->w_negative = true;
  if (bits & WDPM_WATER_ODD) x->w_odd = true;
  return 0;
}

/* the block's threshold flush has just been applied to the whole current raster: with thres >= 0 no negative depth is left */
static void flushed_whole(wdpm_ctx *x, double thres) {
  if (thres >= 0.0) x->w_negative = false;
}

/* wdpm_kernels.h: WDPM_LAUNCH_PLAIN | WDPM_LAUNCH_CLAMP_OK for the iteration launches of this context */
static bool plain_water(const wdpm_ctx *x);
static int launch_flags(const wdpm_ctx *x) {
  static std::atomic<int> env{-1};
  if (env < 0) { const char *e = getenv("WDPM_CLAMP"); env = e ? atoi(e) : 1; }      /* WDPM_CLAMP=0: the unclamped neighbour step everywhere (A/B, tests) */
  return (plain_water(x) ? WDPM_LAUNCH_PLAIN : 0) | (env != 0 && x->dem_bounded ? WDPM_LAUNCH_CLAMP_OK : 0);
}

static bool plain_water(const wdpm_ctx *x) {
  static std::atomic<int> env{-1};
  if (env < 0) { const char *e = getenv("WDPM_PLAIN"); env = e ? atoi(e) : 1; }      /* WDPM_PLAIN=0: gated variants only (A/B, tests) */
  return env != 0 && !x->signed_zero_safe && !x->w_negative && !x->w_odd;
}

/* captured launches carry what they were given by value (wdpm_ctx::GraphEntry): a new DEM, other codes or another outlet end them */
static void drop_graphs(wdpm_ctx *x) {
  if (x->graphs.empty()) return;
  (void)hipStreamSynchronize(x->stream);             /* a replay may still be running */
  for (auto &ge : x->graphs) (void)hipGraphExecDestroy(ge.exec);
  x->graphs.clear();
}

/* Try to express the DEM just uploaded as 32-bit codes k with dem == (k + k0) / 10^e bit for bit
 * (e = 0..6, the smallest that works; real DEMs are decimal text).  The device checks every cell
 * with the decoder the iteration kernel uses; any miss leaves the fp64 DEM in charge. */
static int encode_dem(wdpm_ctx *x) {
  drop_graphs(x);
  /* every level starts from "not encodable": a DEM uploaded over an earlier one inherits nothing (a smooth DEM, then a
   * rough one that passes only the 32-bit check, left dem16_encodable set and d_dem16 holding truncated offsets) */
  x->dem32_encodable = false; x->dem16_encodable = false; x->dem_bounded = false; x->dem16_wanted = true;
  x->code = DemCode{nullptr, 0.0, 1.0, 1.0, 0, nullptr, nullptr, 0};
  HIP_TRY(wdpm_launch_dem_min(x->d_dem, x->cells, x->d_bits, x->stream));
  HIP_TRY(hipMemcpyAsync(x->h_pin, x->d_bits, 2 * sizeof(double), hipMemcpyDeviceToHost, x->stream));
  if (wdpm_stream_sync(x, x->stream)) return 1;
  unsigned long long key, amax;
  memcpy(&key, x->h_pin, sizeof key);
  memcpy(&amax, x->h_pin + 1, sizeof amax);
  x->dem_bounded = amax < 0x41d0000000000000ull;               /* |dem| < 2^30 on every valid cell (none at all: 0) */
  const char *env = getenv("WDPM_DEM32");
  if (env && atoi(env) == 0) return 0;
  if (key == ~0ull) return 0;                                  /* no valid cell at all */
  const double vmin = wdpm_dem_key_to_double(key);
  static const double p10[7] = {1.0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6};
  for (int e = 0; e <= 6; e++) {
    const double D = p10[e], rD = 1.0 / D, k0 = rint(vmin * D);
    if (!(fabs(k0) < 4.0e15)) break;                           /* k = q + k0 must stay an exact integer */
    HIP_TRY(wdpm_launch_dem_encode(x->d_dem, x->cells, k0, D, rD, x->d_dem32, x->d_bits, x->stream));
    HIP_TRY(hipMemcpyAsync(x->h_pin, x->d_bits, sizeof(double), hipMemcpyDeviceToHost, x->stream));
    if (wdpm_stream_sync(x, x->stream)) return 1;
    unsigned long long bad;
    memcpy(&bad, x->h_pin, sizeof bad);
    if (!bad) {
      x->dem32_encodable = true;
      x->code = DemCode{x->d_dem32, k0, D, rD, (env && atoi(env) == 2) ? 1 : 0, nullptr, nullptr, 0};   /* WDPM_DEM32=2: on launches of any size */
      break;
    }
  }
  /* second level: the codes as 16-bit offsets from one base per 48 columns of a row - where the terrain allows it (WDPM_DEM16=0: never) */
  const char *env16 = getenv("WDPM_DEM16");
  if (x->dem32_encodable && !(env16 && atoi(env16) == 0)) {
    const int ngroups = (x->g.ncp + kDemGroup - 1) / kDemGroup;
    HIP_TRY(wdpm_launch_dem16_encode(x->d_dem32, x->g.rows, x->g.ncp, ngroups, x->d_dem16, x->d_gbase, x->d_bits, x->stream));
    HIP_TRY(hipMemcpyAsync(x->h_pin, x->d_bits, sizeof(double), hipMemcpyDeviceToHost, x->stream));
    if (wdpm_stream_sync(x, x->stream)) return 1;
    unsigned long long bad;
    memcpy(&bad, x->h_pin, sizeof bad);
    if (!bad) {
      x->dem16_encodable = true;
      x->code.h = x->d_dem16;
      x->code.gb = x->d_gbase;
      x->code.ngroups = ngroups;
    }
  }
  return 0;
}

int wdpm_upload(wdpm_ctx *x, const double *bigdem, const double *bigwater) {
  if (!bigdem || !bigwater) return fail("wdpm_upload: null array");
  if (bind(x)) return 1;
  const size_t bytes = x->cells * sizeof(double);
  if (ensure_fresh_slot(x)) return 1;
  HIP_TRY(hipMemcpyAsync(x->d_dem, bigdem, bytes, hipMemcpyHostToDevice, x->stream));
  HIP_TRY(hipMemcpyAsync(x->d_w[x->cur], bigwater, bytes, hipMemcpyHostToDevice, x->stream));
  HIP_TRY(wdpm_launch_mark_nodata(x->d_dem, x->cells, x->g.miss, x->stream));
  x->signed_zero_safe = x->w_negative = x->w_odd = false;
  if (note_negzero(x, 0, x->g.rows)) return 1;
  if (encode_dem(x)) return 1;
  return 0;
}

/* Set-up on the device: the slab's padded rasters straight from the UNPADDED file rasters.
 * The two water rasters that are not current serve as staging for the host-to-device copies. */
int wdpm_upload_unpadded(wdpm_ctx *x, const double *dem, const double *water, const wdpm_setup *su) {
  if (!dem || !su) return fail("wdpm_upload_unpadded: null argument");
  if (su->op < 0 || su->op > 2) return fail("wdpm_upload_unpadded: bad water operation %d", su->op);
  if (bind(x)) return 1;
  if (ensure_fresh_slot(x)) return 1;
  const int a = (x->cur + 1) % 3, b = (x->cur + 2) % 3;
  x->zero_valid[a] = x->zero_valid[b] = false;
  const int R = x->g.R, C = x->g.C;
  int f0 = x->g.row0 - 1, f1 = x->g.row0 + x->g.rows - 2;     /* file rows behind the slab's padded rows */
  const int first = f0 < 0 ? 0 : f0, last = f1 > R - 1 ? R - 1 : f1;
  if (last >= first) {
    const size_t off = (size_t)(first - f0) * C, n = (size_t)(last - first + 1) * C;
    HIP_TRY(hipMemcpyAsync(x->d_w[a] + off, dem + (size_t)first * C, n * sizeof(double), hipMemcpyHostToDevice, x->stream));
    if (water)
      HIP_TRY(hipMemcpyAsync(x->d_w[b] + off, water + (size_t)first * C, n * sizeof(double), hipMemcpyHostToDevice, x->stream));
  }
  HIP_TRY(wdpm_launch_pad_setup(x->d_w[a], water ? x->d_w[b] : nullptr, x->d_dem, x->d_w[x->cur], x->g, su->op, su->add,
                                su->rof, su->sub, x->stream));
  /* the snapshot before any block is all zeros (as after wdpm_create) */
  x->old = a;
  x->flush_thres = -__builtin_inf();
  HIP_TRY(hipMemsetAsync(x->d_w[a], 0, x->cells * sizeof(double), x->stream));
  x->signed_zero_safe = x->w_negative = x->w_odd = false;
  if (note_negzero(x, 0, x->g.rows)) return 1;
  return encode_dem(x);
}

int wdpm_set_drain(wdpm_ctx *x, int32_t drainrow, int32_t draincol) {
  if (x->p.module != WDPM_DRAIN) return fail("wdpm_set_drain: not a drain context");
  if (bind(x) || ensure_drained(x)) return 1;
  x->p.drainrow = drainrow;
  x->p.draincol = draincol;
  x->g.dr = drainrow - x->p.slab_row0;
  x->g.dc = draincol;
  return 0;
}

int wdpm_count_stats(wdpm_ctx *x, int32_t row_lo, int32_t row_hi, int64_t *valid, int64_t *wet, double *maxv) {
  if (row_lo < 0 || row_hi > x->g.rows || row_lo > row_hi) return fail("wdpm_count_stats: bad row range");
  if (bind(x) || ensure_flushed(x)) return 1;
  if (!x->d_stat) HIP_TRY(hipMalloc(&x->d_stat, 4 * sizeof(unsigned long long)));
  HIP_TRY(wdpm_launch_count_stats(x->d_w[x->cur], x->d_dem, (size_t)row_lo * x->g.ncp, (size_t)row_hi * x->g.ncp, x->g.miss,
                                  x->d_stat, x->stream));
  HIP_TRY(hipMemcpyAsync(x->h_pin, x->d_stat, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, x->stream));
  if (wdpm_stream_sync(x, x->stream)) return 1;
  unsigned long long v[3];
  memcpy(v, x->h_pin, sizeof v);
  if (valid) *valid = (int64_t)v[0];
  if (wet) *wet = (int64_t)v[1];
  if (maxv) *maxv = v[2] ? wdpm_dem_key_to_double(v[2]) : -__builtin_inf();
  return 0;
}

int wdpm_find_drain(wdpm_ctx *x, int32_t row_lo, int32_t row_hi, double *mindem, int32_t *row, int32_t *col) {
  if (row_lo < 0 || row_hi > x->g.rows || row_lo > row_hi || !mindem || !row || !col) return fail("wdpm_find_drain: bad argument");
  if (bind(x)) return 1;
  if (!x->d_stat) HIP_TRY(hipMalloc(&x->d_stat, 4 * sizeof(unsigned long long)));
  HIP_TRY(wdpm_launch_find_drain(x->d_dem, (size_t)row_lo * x->g.ncp, (size_t)row_hi * x->g.ncp, x->d_stat, x->stream));
  HIP_TRY(hipMemcpyAsync(x->h_pin, x->d_stat, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, x->stream));
  if (wdpm_stream_sync(x, x->stream)) return 1;
  unsigned long long v[2];
  memcpy(v, x->h_pin, sizeof v);
  if (v[0] == ~0ull || v[1] == ~0ull) { *row = -1; *col = -1; *mindem = __builtin_inf(); return 0; }
  *mindem = wdpm_dem_key_to_double(v[0]);
  *row = (int32_t)(v[1] / x->g.ncp);
  *col = (int32_t)(v[1] % x->g.ncp);
  return 0;
}

int wdpm_get_cell(wdpm_ctx *x, int32_t row, int32_t col, double *water, double *dem) {
  if (row < 0 || row >= x->g.rows || col < 0 || col >= x->g.ncp) return fail("wdpm_get_cell: cell outside the slab");
  if (bind(x) || ensure_flushed(x)) return 1;
  const size_t k = (size_t)row * x->g.ncp + col;
  HIP_TRY(hipMemcpyAsync(x->h_pin, x->d_w[x->cur] + k, sizeof(double), hipMemcpyDeviceToHost, x->stream));
  HIP_TRY(hipMemcpyAsync(x->h_pin + 1, x->d_dem + k, sizeof(double), hipMemcpyDeviceToHost, x->stream));
  if (wdpm_stream_sync(x, x->stream)) return 1;
  if (water) *water = x->h_pin[0];
  if (dem) *dem = x->h_pin[1] < __builtin_inf() ? x->h_pin[1] : x->g.miss;     /* the device marks NODATA as +inf */
  return 0;
}

int wdpm_download_unpadded(wdpm_ctx *x, int32_t file_row, int32_t nrows, int32_t mask_missing, double *dst) {
  const int p0 = file_row + 1 - x->g.row0;                      /* slab-local padded row of the first file row */
  if (nrows < 0 || file_row < 0 || file_row + nrows > x->g.R || p0 < 0 || p0 + nrows > x->g.rows || !dst)
    return fail("wdpm_download_unpadded: rows outside the slab");
  if (nrows == 0) return 0;
  if (bind(x) || ensure_flushed(x)) return 1;
  const int st = free_slot(x);
  double *stage = x->d_w[st];
  x->zero_valid[st] = false;
  HIP_TRY(wdpm_launch_unpad(x->d_w[x->cur], x->d_dem, x->g, file_row, nrows, mask_missing, stage, x->stream));
  HIP_TRY(hipMemcpyAsync(dst, stage, (size_t)nrows * x->g.C * sizeof(double), hipMemcpyDeviceToHost, x->stream));
  if (wdpm_stream_sync(x, x->stream)) return 1;
  return 0;
}

int wdpm_upload_water(wdpm_ctx *x, const double *bigwater) {
  if (!bigwater) return fail("wdpm_upload_water: null array");
  if (bind(x)) return 1;
  if (ensure_fresh_slot(x)) return 1;
  HIP_TRY(hipMemcpyAsync(x->d_w[x->cur], bigwater, x->cells * sizeof(double), hipMemcpyHostToDevice, x->stream));
  x->signed_zero_safe = x->w_negative = x->w_odd = false;
  return note_negzero(x, 0, x->g.rows);
}

int wdpm_download_water(wdpm_ctx *x, double *bigwater) {
  if (!bigwater) return fail("wdpm_download_water: null array");
  if (bind(x)) return 1;
  if (ensure_flushed(x)) return 1;
  HIP_TRY(hipMemcpyAsync(bigwater, x->d_w[x->cur], x->cells * sizeof(double), hipMemcpyDeviceToHost, x->stream));
  if (wdpm_stream_sync(x, x->stream)) return 1;
  return 0;
}

int wdpm_download_rows(wdpm_ctx *x, int32_t row, int32_t nrows, double *dst) {
  if (row < 0 || nrows < 0 || row + nrows > x->g.rows || !dst) return fail("wdpm_download_rows: bad row range");
  if (bind(x)) return 1;
  if (ensure_flushed(x)) return 1;
  HIP_TRY(hipMemcpyAsync(dst, x->d_w[x->cur] + (size_t)row * x->g.ncp, (size_t)nrows * x->g.ncp * sizeof(double),
                         hipMemcpyDeviceToHost, x->stream));
  if (wdpm_stream_sync(x, x->stream)) return 1;
  return 0;
}

int wdpm_upload_rows(wdpm_ctx *x, int32_t row, int32_t nrows, const double *src) {
  if (row < 0 || nrows < 0 || row + nrows > x->g.rows || !src) return fail("wdpm_upload_rows: bad row range");
  if (bind(x)) return 1;
  if (ensure_private(x)) return 1;
  x->zero_valid[x->cur] = false;
  HIP_TRY(hipMemcpyAsync(x->d_w[x->cur] + (size_t)row * x->g.ncp, src, (size_t)nrows * x->g.ncp * sizeof(double),
                         hipMemcpyHostToDevice, x->stream));
  return note_negzero(x, row, nrows);
}

/* Direct peer access between two devices, asked for once per ordered pair: without it
 * hipMemcpyPeerAsync silently stages through host memory on ROCm.  A refusal (no xGMI/PCIe path) is
 * remembered and leaves the staged copy in charge - slower, same bits. */
static int enable_peer(int dev, int peer) {
  static signed char state[64][64];   /* 0 unknown, 1 enabled, -1 refused */
  if (dev == peer || dev < 0 || peer < 0 || dev >= 64 || peer >= 64 || state[dev][peer]) return 0;
  int can = 0;
  HIP_TRY(hipSetDevice(dev));
  HIP_TRY(hipDeviceCanAccessPeer(&can, dev, peer));
  if (can) {
    const hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) can = 0;
    (void)hipGetLastError();
  }
  state[dev][peer] = can ? 1 : -1;
  if (!can) fprintf(stderr, "wdpm: no direct peer access from device %d to device %d; halo copies are staged\n", dev, peer);
  return 0;
}

int wdpm_enable_peer_access(wdpm_ctx *a, wdpm_ctx *b) {
  if (!a || !b) return fail("wdpm_enable_peer_access: null context");
  if (enable_peer(a->p.device, b->p.device)) return 1;
  return enable_peer(b->p.device, a->p.device);
}

int wdpm_copy_rows(wdpm_ctx *dst, int32_t dst_row, wdpm_ctx *src, int32_t src_row, int32_t nrows) {
  if (!dst || !src || dst->g.ncp != src->g.ncp || nrows < 0 || dst_row < 0 || src_row < 0 ||
      dst_row + nrows > dst->g.rows || src_row + nrows > src->g.rows)
    return fail("wdpm_copy_rows: bad row range");
  if (nrows == 0) return 0;
  if (src->p.device != dst->p.device && wdpm_enable_peer_access(dst, src)) return 1;
  /* Ordered on the device, never on the host: the copy runs on the destination's stream behind
   * everything queued there (its kernels are done with the rows being overwritten), after an event
   * that marks the source's queue (its kernels have produced the rows); the source's later work in
   * turn waits for the copy to have read them.  With N devices in a chain the halo refresh is then
   * 2(N-1) enqueues and no host round trip. */
  if (bind(src)) return 1;                       /* also joins a pending overlapped interior launch */
  /* Rows that leave a context carry the last iteration's drain(): the receiver applies its own owed drain() only where the
   * outlet lies strictly inside ITS slab - not when the outlet's row is the slab's first or last row, and then the nine
   * cells must arrive zeroed (tests/test_rowblock.py::test_hip_outlet_on_the_last_row_of_the_neighbours_halo). */
  if (ensure_flushed(src)) return 1;   /* ... and the block's threshold flush, should the source still owe it to its raster */
  /* ... and what kinds of depth the source may hold travel with the rows: a caller who peer-copies rows with a -0.0, a
   * negative or an odd depth into a context whose own scan was clean must not leave it on the gate-free kernel variants */
  dst->signed_zero_safe = dst->signed_zero_safe || src->signed_zero_safe;
  dst->w_negative = dst->w_negative || src->w_negative;
  dst->w_odd = dst->w_odd || src->w_odd;
  if (!src->ev_copy[0]) HIP_TRY(hipEventCreateWithFlags(&src->ev_copy[0], hipEventDisableTiming));
  HIP_TRY(hipEventRecord(src->ev_copy[0], src->stream));
  if (bind(dst)) return 1;
  if (!dst->ev_copy[1]) HIP_TRY(hipEventCreateWithFlags(&dst->ev_copy[1], hipEventDisableTiming));
  HIP_TRY(hipStreamWaitEvent(dst->stream, src->ev_copy[0], 0));
  if (tiles_touch(dst, dst->cur, dst_row, nrows)) return 1;       /* queued on dst's stream, ahead of the copy */
  EventPair xep;
  if (wdpm_xch_timing_begin(dst, &xep)) return 1;
  const size_t bytes = (size_t)nrows * src->g.ncp * sizeof(double);
  double *d = dst->d_w[dst->cur] + (size_t)dst_row * dst->g.ncp;
  const double *s = src->d_w[src->cur] + (size_t)src_row * src->g.ncp;
  if (src->p.device == dst->p.device) HIP_TRY(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, dst->stream));
  else HIP_TRY(hipMemcpyPeerAsync(d, dst->p.device, s, src->p.device, bytes, dst->stream));
  if (wdpm_xch_timing_end(dst, &xep)) return 1;
  HIP_TRY(hipEventRecord(dst->ev_copy[1], dst->stream));
  HIP_TRY(hipSetDevice(src->p.device));
  HIP_TRY(hipStreamWaitEvent(src->stream, dst->ev_copy[1], 0));
  return 0;
}

int wdpm_get_option(wdpm_ctx *x, int32_t key, int64_t *value) {
  if (!value) return fail("wdpm_get_option: null argument");
  if (key == WDPM_OPT_SIGNED_ZERO_SAFE) *value = x->signed_zero_safe ? 1 : 0;
  else if (key == WDPM_OPT_WATER_KINDS) *value = (x->signed_zero_safe ? WDPM_WATER_NEGZERO : 0) | (x->w_negative ? WDPM_WATER_NEGATIVE : 0) | (x->w_odd ? WDPM_WATER_ODD : 0);
  else if (key == WDPM_OPT_PLAIN_WATER) *value = plain_water(x) ? 1 : 0;
  else if (key == WDPM_OPT_DEM32) *value = x->code.q ? 1 : 0;
  else if (key == WDPM_OPT_DEM16) *value = x->code.q && x->code.h ? (wdpm_dem16_pays((long long)x->cells, x->code.force) ? 1 : 2) : 0;   /* 2: available, but a whole-slab launch of this size keeps the 32-bit codes */
  else if (key == WDPM_OPT_TILES) *value = x->tiles_mode;
  else if (key == WDPM_OPT_TILES_SEEN) *value = x->stat_tiles;
  else if (key == WDPM_OPT_TILES_WORKED) *value = x->stat_active;
  else if (key == WDPM_OPT_SPARSE) *value = x->sparse ? 1 : 0;
  else if (key == WDPM_OPT_GRAPH_LAUNCHES) *value = x->graph_launches;
  else if (key == WDPM_OPT_GUARD_BAD) return guard_damage(x, value);
  else return fail("wdpm_get_option: unknown option %d", key);
  return 0;
}

int wdpm_set_option(wdpm_ctx *x, int32_t key, int64_t value) {
  drop_graphs(x);
  if (key == WDPM_OPT_SIGNED_ZERO_SAFE) {
    x->signed_zero_safe = value != 0;
  } else if (key == WDPM_OPT_WATER_KINDS) {        /* OR-ed in: what a multi-GPU driver found on the other ranks */
    if (value & WDPM_WATER_NEGZERO) x->signed_zero_safe = true;
    if (value & WDPM_WATER_NEGATIVE) x->w_negative = true;
    if (value & WDPM_WATER_ODD) x->w_odd = true;
  } else if (key == WDPM_OPT_DEM32 || key == WDPM_OPT_DEM16) {
    /* switching the codes on is honoured only for a DEM that passed the device's bit-for-bit check; the 16-bit offsets are
     * offsets of the 32-bit codes, so they follow them, and offsets, group bases and their pitch go on and off together */
    if (key == WDPM_OPT_DEM32) {
      x->code.q = (value != 0 && x->dem32_encodable) ? x->d_dem32 : nullptr;
      x->code.force = value == 2;   /* 2: also on launches too small for the codes to pay (tests) */
    } else {
      x->dem16_wanted = value != 0;
    }
    const bool on16 = x->dem16_wanted && x->dem16_encodable && x->code.q != nullptr;
    x->code.h = on16 ? x->d_dem16 : nullptr;
    x->code.gb = on16 ? x->d_gbase : nullptr;
    x->code.ngroups = on16 ? (x->g.ncp + kDemGroup - 1) / kDemGroup : 0;
  } else if (key == WDPM_OPT_TILES) {
    x->tiles_mode = value != 0;
    x->zero_valid[0] = x->zero_valid[1] = x->zero_valid[2] = false;
  } else if (key == WDPM_OPT_SPARSE) {
    x->sparse = value != 0;
  } else {
    return fail("wdpm_set_option: unknown option %d", key);
  }
  return 0;
}

int wdpm_set_totaldrain(wdpm_ctx *x, double v) {
  if (bind(x) || ensure_drained(x)) return 1;
  x->h_pin[0] = v;
  HIP_TRY(hipMemcpyAsync(x->d_scal, x->h_pin, sizeof(double), hipMemcpyHostToDevice, x->stream));
  if (wdpm_stream_sync(x, x->stream)) return 1;
  return 0;
}

int wdpm_get_totaldrain(wdpm_ctx *x, double *v) {
  if (bind(x) || ensure_drained(x)) return 1;
  HIP_TRY(hipMemcpyAsync(x->h_pin, x->d_scal, sizeof(double), hipMemcpyDeviceToHost, x->stream));
  if (wdpm_stream_sync(x, x->stream)) return 1;
  *v = x->h_pin[0];
  return 0;
}

int wdpm_water_ptr(wdpm_ctx *x, void **ptr) {
  if (bind(x) || ensure_private(x)) return 1;   /* the caller may write through it */
  x->zero_valid[x->cur] = false;
  x->w_odd = true;                               /* ... anything at all: the gated kernel variants from here on */
  *ptr = x->d_w[x->cur];
  return 0;
}
int wdpm_dem_ptr(wdpm_ctx *x, void **ptr) {
  x->dem_bounded = false;                        /* the caller may write anything through it */
  *ptr = x->d_dem;
  return 0;
}

/* ---- block loop ------------------------------------------------------------------------- */
int wdpm_begin_block(wdpm_ctx *x, double thres) {
  if (bind(x)) return 1;
  x->md_valid = x->md_hint = false;
  if (ensure_flushed(x)) return 1;                       /* a flush still owed from a block that never iterated */
  if (x->kernel == WDPM_KERNEL_FUSED && !x->signed_zero_safe) {
    /* no pass over the raster: the current raster becomes the snapshot, the flush rides on the next launch's loads */
    x->old = x->cur;
    x->flush_pending = true;
  } else {
    const int t = x->cur == x->old ? free_slot(x) : x->old;
    HIP_TRY(wdpm_launch_flush_snapshot(x->d_w[x->cur], x->d_w[t], x->cells, thres, x->stream));
    x->zero_valid[t] = false;
    x->old = t;
    flushed_whole(x, thres);
  }
  x->flush_thres = thres;
  HIP_TRY(hipMemcpyAsync(x->d_scal + 1, x->d_scal, sizeof(double), hipMemcpyDeviceToDevice, x->stream)); /* olddrain */
  return 0;
}

static int fold_timing(wdpm_ctx *x) {
  for (auto &ep : x->pending) {
    HIP_TRY(hipEventSynchronize(ep.b));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ep.a, ep.b));
    x->ms += ms;
    x->pool.push_back(ep);
  }
  x->pending.clear();
  for (auto &ep : x->pending_steady) {
    HIP_TRY(hipEventSynchronize(ep.b));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ep.a, ep.b));
    x->steady_ms += ms;
    x->pool.push_back(ep);
  }
  x->pending_steady.clear();
  for (auto &ep : x->pending_xch) {
    HIP_TRY(hipEventSynchronize(ep.b));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ep.a, ep.b));
    x->xch_ms += ms;
    x->xch_count += 1;
    x->pool.push_back(ep);
  }
  x->pending_xch.clear();
  return 0;
}

int wdpm_xch_timing_begin(wdpm_ctx *x, EventPair *ep) {
  ep->a = ep->b = nullptr;
  if (!x->timing) return 0;
  if (x->pending_xch.size() >= 256 && fold_timing(x)) return 1;
  if (!x->pool.empty()) { *ep = x->pool.back(); x->pool.pop_back(); }
  else { HIP_TRY(hipEventCreate(&ep->a)); HIP_TRY(hipEventCreate(&ep->b)); }
  HIP_TRY(hipEventRecord(ep->a, x->stream));
  return 0;
}

int wdpm_xch_timing_end(wdpm_ctx *x, const EventPair *ep) {
  if (!ep->a) return 0;
  HIP_TRY(hipEventRecord(ep->b, x->stream));
  x->pending_xch.push_back(*ep);
  return 0;
}

static int one_pass(wdpm_ctx *x, int oi, int oj) {
  if (ensure_private(x)) return 1;
  x->zero_valid[x->cur] = false;
  HIP_TRY(wdpm_launch_pass(x->p.module, x->d_w[x->cur], x->d_dem, x->g, oi, oj, x->d_scal, x->stream));
  return 0;
}

int wdpm_pass(wdpm_ctx *x, int32_t oi, int32_t oj) {
  if (oi < 1 || oi > 3 || oj < 1 || oj > 3) return fail("wdpm_pass: oi,oj must be in 1..3");
  if (bind(x)) return 1;
  return one_pass(x, oi, oj);
}

int wdpm_drain_outlet(wdpm_ctx *x) {
  if (x->p.module != WDPM_DRAIN) return 0;
  if (bind(x) || ensure_private(x)) return 1;
  HIP_TRY(wdpm_launch_drain_outlet(x->d_w[x->cur], x->d_dem, x->g, x->d_scal, x->stream));
  return 0;
}

int wdpm_iterate(wdpm_ctx *x, int32_t n_iter) {
  if (n_iter < 0) return fail("wdpm_iterate: negative iteration count");
  if (n_iter == 0) return 0;
  if (bind(x)) return 1;
  const bool fold = x->md_hint && x->kernel == WDPM_KERNEL_FUSED && x->p.module != WDPM_DRAIN && !x->signed_zero_safe;
  x->md_hint = x->md_valid = false;
  /* stencil timing is opt-in (wdpm_timing_reset switches it on): two event records per call are a
   * measurable share of a small raster's iteration */
  EventPair ep{nullptr, nullptr};
  if (x->timing) {
    if (x->pending.size() >= 256 && fold_timing(x)) return 1;
    if (!x->pool.empty()) { ep = x->pool.back(); x->pool.pop_back(); }
    else { HIP_TRY(hipEventCreate(&ep.a)); HIP_TRY(hipEventCreate(&ep.b)); }
    HIP_TRY(hipEventRecord(ep.a, x->stream));
  }
  EventPair st{nullptr, nullptr};
  const bool steady = x->timing && n_iter >= 3 && x->kernel == WDPM_KERNEL_FUSED;
  if (steady) {
    if (!x->pool.empty()) { st = x->pool.back(); x->pool.pop_back(); }
    else { HIP_TRY(hipEventCreate(&st.a)); HIP_TRY(hipEventCreate(&st.b)); }
  }
  auto one_iteration = [&](const int it) -> int {
    if (steady && it == 1) HIP_TRY(hipEventRecord(st.a, x->stream));
    if (steady && it == n_iter - 1) HIP_TRY(hipEventRecord(st.b, x->stream));
    if (x->kernel == WDPM_KERNEL_FUSED) {
      if (x->flush_pending && x->signed_zero_safe && ensure_flushed(x)) return 1;   /* no flush-on-load variant of that kernel */
      const int t = free_slot(x);
      TilePlan tp{x->d_zero[x->cur], x->d_zero[t], x->zero_valid[x->cur] ? 1 : 0, x->zero_valid[t] ? 1 : 0, x->d_active,
                  x->tile_cap, x->tile_nstrips, x->tile_H, x->tile_nchunks, 0, x->wide_tri_ok ? 1 : 0};
      /* A mostly wet raster (the last block that kept flags found most tiles working) drops the dry-tile flags for chunk heights
       * that follow the XCDs' speeds (wdpm_kernels.h::XcdBalance): a tiling of its own.  The flags are looked at again every 16 blocks. */
      const bool balanced = x->bal.mode == 2 || (x->bal.mode == 1 && x->wide_tri_ok);
      const bool track = x->tiles_mode != 0 && !x->signed_zero_safe && !balanced;
      /* sparse rasters march short chunks: the launch takes as long as its wettest tile */
      const int chunk_rows = x->p.chunk_rows >= 3 ? x->p.chunk_rows : (track && x->sparse ? kSparseChunkRows : 0);
      /* the block's last iteration also reduces max |w - oldw| over the rows the caller announced */
      MaxDiffArgs md{nullptr, x->flush_thres, x->md_lo, x->md_hi, x->d_md};
      if (fold && it == n_iter - 1) {
        md.old = x->d_w[x->old];
        HIP_TRY(hipMemsetAsync(x->d_md, 0, sizeof(unsigned long long), x->stream));
      }
      HIP_TRY(wdpm_launch_fused(x->p.module, x->d_w[x->cur], x->d_w[t], x->d_dem, x->code, x->g, chunk_rows,
                                x->signed_zero_safe ? 1 : 0, x->flush_pending ? &x->flush_thres : nullptr,
                                x->drain_owed ? 1 : 0, x->d_scal, x->stream, track ? &tp : nullptr, md.old ? &md : nullptr,
                                launch_flags(x), x->bal.mode ? &x->bal : nullptr));
      if (x->flush_pending) flushed_whole(x, x->flush_thres);   /* the launch flushed every value it loaded, and it loaded them all */
      if (md.old) x->md_valid = true;
      if (track && tp.maintained) {
        if (tp.nstrips != x->tile_nstrips || tp.H != x->tile_H || tp.nchunks != x->tile_nchunks) {
          /* another tiling from here on: nothing known about any raster, and the flag arrays get their border of
           * 1s for the new pitch (queued behind the launch, which wrote t's interior flags: only the other two) */
          x->zero_valid[0] = x->zero_valid[1] = x->zero_valid[2] = false;
          x->tile_nstrips = tp.nstrips; x->tile_H = tp.H; x->tile_nchunks = tp.nchunks;
          for (int i = 0; i < 3; i++)
            if (i != t) HIP_TRY(hipMemsetAsync(x->d_zero[i], 1, (size_t)x->tile_cap, x->stream));
        }
        x->zero_valid[t] = true;
        x->tiles_launched += (int64_t)tp.nstrips * tp.nchunks;
      } else {
        x->zero_valid[t] = false;
      }
      x->cur = t;
      x->flush_pending = false;
      x->drain_owed = x->p.module == WDPM_DRAIN;     /* this iteration's drain(): owed to the next launch or reader */
      x->launches += 1;
      return 0;
    } else {
      for (int oi = 1; oi <= 3; oi++)
        for (int oj = 1; oj <= 3; oj++)
          if (one_pass(x, oi, oj)) return 1;
      x->launches += 9;
    }
    if (x->p.module == WDPM_DRAIN)
      HIP_TRY(wdpm_launch_drain_outlet(x->d_w[x->cur], x->d_dem, x->g, x->d_scal, x->stream));
    return 0;
  };
  int it = 0;
  /* Small rasters are launch-bound (482 x 471: 5.2 us of kernel, 6.3 us from launch to launch when the host queues them one by one):
   * the iterations between a block's first (threshold flush on load) and last (max diff) are replayed as HIP graphs of kGraphIters
   * launches each, captured from this very loop (round 5; `tools/graph_probe.py` measured +21 % for add at that size, +3 % for drain,
   * nothing from 2048^2 up).  Only where a launch keeps no state on the host (the relay / triangle kernels: no tile flags, no balance
   * table) and nobody times the launches; the rasters ping-pong between two buffers within a block, so an even count returns to
   * the state it was captured in.  WDPM_GRAPH=0: never. */
  constexpr int kGraphIters = 32;
  if (x->kernel == WDPM_KERNEL_FUSED && !x->timing && n_iter >= kGraphIters + 2 && x->graph_mode != 0) {
    if (x->graph_mode < 0) { const char *e = getenv("WDPM_GRAPH"); x->graph_mode = (e && atoi(e) == 0) ? 0 : 1; }
    const bool balanced = x->bal.mode == 2 || (x->bal.mode == 1 && x->wide_tri_ok);
    const bool track = x->tiles_mode != 0 && !x->signed_zero_safe && !balanced;
    const int chunk_rows = x->p.chunk_rows >= 3 ? x->p.chunk_rows : (track && x->sparse ? kSparseChunkRows : 0);
    TilePlan tq{nullptr, nullptr, 0, 0, nullptr, 0, 0, 0, 0, 0, x->wide_tri_ok ? 1 : 0};
    if (x->graph_mode == 1 && wdpm_small_rows_take(x->p.module, x->g, chunk_rows, x->signed_zero_safe ? 1 : 0, track ? &tq : nullptr)) {
      if (one_iteration(it++)) return 1;                   /* the block's first launch: as ever */
      while (n_iter - 1 - it >= kGraphIters && x->graph_mode == 1) {
        wdpm_ctx::GraphEntry key{x->cur, x->old, launch_flags(x), x->drain_owed ? 1 : 0, chunk_rows, x->g.dr, x->g.dc, x->wide_tri_ok ? 1 : 0,
                                 x->code.force, x->code.q, x->code.h, nullptr};
        hipGraphExec_t exec = nullptr;
        for (const auto &ge : x->graphs)
          if (ge.cur == key.cur && ge.old == key.old && ge.flags == key.flags && ge.drain_owed == key.drain_owed && ge.chunk_rows == key.chunk_rows &&
              ge.dr == key.dr && ge.dc == key.dc && ge.wide == key.wide && ge.force == key.force && ge.q == key.q && ge.h == key.h) { exec = ge.exec; break; }
        if (!exec) {
          /* capture kGraphIters launches of this loop (nothing runs yet; the host's bookkeeping moves on as if they had) */
          hipGraph_t graph = nullptr;
          if (hipStreamBeginCapture(x->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); x->graph_mode = 0; break; }
          int bad = 0;
          for (int k = 0; k < kGraphIters && !bad; k++) bad = one_iteration(it + k);
          const hipError_t ec = hipStreamEndCapture(x->stream, &graph);
          if (bad || ec != hipSuccess || !graph || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
            if (graph) (void)hipGraphDestroy(graph);
            (void)hipGetLastError();
            x->graph_mode = 0;
            return fail("wdpm_iterate: capturing %d iterations as a HIP graph failed (set WDPM_GRAPH=0 to launch them one by one)", kGraphIters);
          }
          (void)hipGraphDestroy(graph);
          if (x->graphs.size() >= 12) drop_graphs(x);
          key.exec = exec;
          x->graphs.push_back(key);
          HIP_TRY(hipGraphLaunch(exec, x->stream));
          x->graph_launches++;
        } else {
          HIP_TRY(hipGraphLaunch(exec, x->stream));
          x->graph_launches++;
          /* what kGraphIters passes through one_iteration() leave on the host: the same current raster (an even count), nothing known
           * about dry tiles of the two rasters written, drain() owed by the last launch */
          x->zero_valid[x->cur] = x->zero_valid[free_slot(x)] = false;
          x->drain_owed = x->p.module == WDPM_DRAIN;
          x->launches += kGraphIters;
        }
        it += kGraphIters;
      }
    }
  }
  for (; it < n_iter; it++)
    if (one_iteration(it)) return 1;
  if (x->timing) {
    HIP_TRY(hipEventRecord(ep.b, x->stream));
    x->pending.push_back(ep);
  }
  if (steady) {
    x->pending_steady.push_back(st);
    x->steady_launches += n_iter - 2;
  }
  return 0;
}

int wdpm_iterate_overlapped(wdpm_ctx *x, int32_t n_iter, int32_t top_rows, int32_t bottom_rows) {
  if (n_iter < 0 || top_rows < 0 || bottom_rows < 0) return fail("wdpm_iterate_overlapped: negative argument");
  const int rows = x->g.rows;
  /* window boundaries: the top launch produces rows [0, t_last] with (t_last + 1) % 3 == 2, the bottom
   * launch rows [b_first, rows-1] with b_first % 3 == 2, the interior launch the rows between */
  int t_last = -1, b_first = rows;
  if (top_rows > 0) { t_last = top_rows - 1; while ((t_last + 1) % 3 != 2) t_last++; }
  if (bottom_rows > 0) { b_first = rows - bottom_rows; while (b_first % 3 != 2) b_first--; }
  const bool usable = x->kernel == WDPM_KERNEL_FUSED && x->p.module != WDPM_DRAIN && n_iter > 0 &&
                      (top_rows > 0 || bottom_rows > 0) && b_first - (t_last + 1) >= 24 && t_last < rows - 1 &&
                      b_first >= 2;
  if (!usable) return wdpm_iterate(x, n_iter);
  const bool hint = x->md_hint;
  x->md_hint = false;
  if (n_iter > 1 && wdpm_iterate(x, n_iter - 1)) return 1;
  x->md_hint = hint;
  if (bind(x)) return 1;
  if (x->flush_pending && x->signed_zero_safe && ensure_flushed(x)) return 1;
  const int t_slot = free_slot(x);
  const double *w_in = x->d_w[x->cur];
  double *w_out = x->d_w[t_slot];
  const int szs = x->signed_zero_safe ? 1 : 0;
  const double *flush = x->flush_pending ? &x->flush_thres : nullptr;
  const int plain = launch_flags(x);
  const bool fold = x->md_hint && !x->signed_zero_safe;     /* (usable: fused kernel, not the drain module) */
  x->md_hint = x->md_valid = false;
  MaxDiffArgs md{fold ? x->d_w[x->old] : nullptr, x->flush_thres, x->md_lo, x->md_hi, x->d_md};
  const MaxDiffArgs *mdp = fold ? &md : nullptr;
  if (fold) HIP_TRY(hipMemsetAsync(x->d_md, 0, sizeof(unsigned long long), x->stream));
  /* stencil timing of this iteration: from here on the main stream to the end of the interior launch
   * on the side stream (the longest of the three) */
  EventPair ep{nullptr, nullptr};
  if (x->timing) {
    if (x->pending.size() >= 256 && fold_timing(x)) return 1;
    if (!x->pool.empty()) { ep = x->pool.back(); x->pool.pop_back(); }
    else { HIP_TRY(hipEventCreate(&ep.a)); HIP_TRY(hipEventCreate(&ep.b)); }
    HIP_TRY(hipEventRecord(ep.a, x->stream));
  }
  HIP_TRY(hipEventRecord(x->ev_fork, x->stream));               /* w_in is complete here */
  if (t_last >= 0)
    HIP_TRY(wdpm_launch_fused_rows(x->p.module, w_in, w_out, x->d_dem, x->code, x->g, 0, t_last, x->p.chunk_rows, szs, flush,
                                   0, x->d_scal, x->stream, nullptr, mdp, 0, plain));
  if (b_first < rows)
    HIP_TRY(wdpm_launch_fused_rows(x->p.module, w_in, w_out, x->d_dem, x->code, x->g, b_first - 2, rows - 1, x->p.chunk_rows,
                                   szs, flush, 0, x->d_scal, x->stream, nullptr, mdp, 0, plain));
  HIP_TRY(hipStreamWaitEvent(x->side, x->ev_fork, 0));
  HIP_TRY(wdpm_launch_fused_rows(x->p.module, w_in, w_out, x->d_dem, x->code, x->g, t_last >= 0 ? t_last - 1 : 0,
                                 b_first < rows ? b_first - 1 : rows - 1, x->p.chunk_rows, szs, flush, 0, x->d_scal, x->side, nullptr, mdp,
                                 x->comm ? 8 : 0, plain));   /* 8 of 256 CUs stay free for the RCCL kernels of the refresh that follows */
  if (x->timing) {
    HIP_TRY(hipEventRecord(ep.b, x->side));
    x->pending.push_back(ep);
  }
  HIP_TRY(hipEventRecord(x->ev_join, x->side));
  x->pending_join = true;
  x->zero_valid[t_slot] = false;   /* three windows, not the tiling the flags are kept for */
  if (fold) x->md_valid = true;
  x->cur = t_slot;
  if (x->flush_pending) flushed_whole(x, x->flush_thres);
  x->flush_pending = false;
  x->launches += 3;
  return 0;
}

int wdpm_expect_max_diff(wdpm_ctx *x, int32_t row_lo, int32_t row_hi) {
  if (row_lo < 0 || row_hi > x->g.rows || row_lo > row_hi) return fail("wdpm_expect_max_diff: bad row range");
  /* a result already folded by the last launch stays good for ITS rows only (an expectation for other rows, with no
   * iteration after it, must not be answered from it: tests/test_hip_parity.py::test_random_call_sequences) */
  if (x->md_valid && (x->md_lo != row_lo || x->md_hi != row_hi)) x->md_valid = false;
  x->md_hint = true;
  x->md_lo = row_lo;
  x->md_hi = row_hi;
  return 0;
}

int wdpm_max_diff(wdpm_ctx *x, int32_t row_lo, int32_t row_hi, double *out) {
  if (row_lo < 0 || row_hi > x->g.rows || row_lo > row_hi || !out) return fail("wdpm_max_diff: bad row range");
  if (bind(x)) return 1;
  if (ensure_flushed(x)) return 1;
  const bool folded = x->md_valid && x->md_lo == row_lo && x->md_hi == row_hi;   /* the last iteration launch already reduced it */
  /* the snapshot may hold the raster as it was BEFORE the block's flush: the kernel applies the flush as it reads */
  if (folded) HIP_TRY(hipMemcpyAsync(x->d_bits, x->d_md, sizeof(unsigned long long), hipMemcpyDeviceToDevice, x->stream));
  else HIP_TRY(wdpm_launch_max_diff(x->d_w[x->cur], x->d_w[x->old], x->flush_thres, x->d_dem, x->g, row_lo, row_hi, x->d_bits, x->stream));
  HIP_TRY(hipMemcpyAsync(x->h_pin, x->d_bits, sizeof(double), hipMemcpyDeviceToHost, x->stream));
  const bool look = x->tiles_launched > 0;
  if (look) {
    HIP_TRY(hipMemcpyAsync(x->h_active, x->d_active, sizeof(unsigned), hipMemcpyDeviceToHost, x->stream));
    HIP_TRY(hipMemsetAsync(x->d_active, 0, sizeof(unsigned), x->stream));
  }
  if (wdpm_stream_sync(x, x->stream)) return 1;
  *out = x->h_pin[0];
  if (look) {
    /* once per block: how many tiles worked?  Mostly dry -> short chunks from the next block on (and back) */
    const double frac = (double)*x->h_active / (double)x->tiles_launched;
    x->stat_tiles += x->tiles_launched;
    x->stat_active += *x->h_active;
    x->tiles_launched = 0;
    if (!x->sparse && frac < 0.30 && x->g.rows >= 4 * kSparseChunkRows) x->sparse = true;
    else if (x->sparse && frac > 0.60) x->sparse = false;
    /* mostly wet: a raster of a few rounds of triangle waves goes to that kernel from the next block on (no flags there, so
     * the marching kernel looks again every 16 blocks) */
    x->wide_tri_ok = !x->sparse && frac > 0.60;
    x->blocks_unprobed = 0;
  } else if (x->wide_tri_ok && ++x->blocks_unprobed >= 16) {
    x->wide_tri_ok = false;
  }
  return 0;
}

/* WDPMCL.c:1257-1268.  final_sum is the reference's sequential row-major sum (an ordinary parallel
 * sum would round differently): wdpm_volume_partial evaluates exactly that sum on the device. */
int wdpm_drain_stats(wdpm_ctx *x, double *diffdrain, double *final_sum) {
  if (bind(x) || ensure_drained(x)) return 1;
  if (diffdrain) {
    HIP_TRY(hipMemcpyAsync(x->h_pin, x->d_scal, 2 * sizeof(double), hipMemcpyDeviceToHost, x->stream));
    if (wdpm_stream_sync(x, x->stream)) return 1;
    *diffdrain = fabs(x->h_pin[0] - x->h_pin[1]);
  }
  if (final_sum) return wdpm_volume_partial(x, 0, x->g.rows, 0.0, final_sum);
  return 0;
}

/* The reference's sequential row-major sum over the valid cells of rows [row_lo, row_hi), continued
 * from `start` - evaluated in parallel on the device, bit for bit the left-to-right fp64 sum
 * (tests/seqsum_model.py explains why that is possible and models every step below).  Only chunks
 * whose running sum comes too close to a power of two, or that hold an exact rounding tie or a
 * negative / non-finite depth, are fetched and summed term by term. */
int wdpm_volume_partial(wdpm_ctx *x, int32_t row_lo, int32_t row_hi, double start, double *sum) {
  if (row_lo < 0 || row_hi > x->g.rows || row_lo > row_hi || !sum) return fail("wdpm_volume_partial: bad row range");
  if (bind(x)) return 1;
  if (ensure_flushed(x)) return 1;
  const size_t first = (size_t)row_lo * x->g.ncp, n = (size_t)(row_hi - row_lo) * x->g.ncp;
  const size_t nchunks = (n + kSeqSumChunk - 1) / kSeqSumChunk;
  if (nchunks == 0) { *sum = start; return 0; }
  const size_t cap = x->cells / kSeqSumChunk + 2;
  if (!x->d_sum_approx) {
    HIP_TRY(hipMalloc(&x->d_sum_approx, cap * sizeof(double)));
    HIP_TRY(hipMalloc(&x->d_sum_i, cap * sizeof(long long)));
    HIP_TRY(hipMalloc(&x->d_sum_k, cap * sizeof(int)));
    HIP_TRY(hipMalloc(&x->d_sum_flag, cap * sizeof(unsigned)));
  }
  const double *w = x->d_w[x->cur] + first, *dem = x->d_dem + first;
  std::vector<double> approx(nchunks), tw, td;
  std::vector<unsigned> flag(nchunks);
  std::vector<int> kexp(nchunks);
  std::vector<long long> isum(nchunks);

  /* pass A: ordinary per-chunk sums -> which binade each chunk's running sum will live in */
  HIP_TRY(wdpm_launch_seqsum_a(w, dem, n, x->d_sum_approx, x->d_sum_flag, x->stream));
  HIP_TRY(hipMemcpyAsync(approx.data(), x->d_sum_approx, nchunks * sizeof(double), hipMemcpyDeviceToHost, x->stream));
  HIP_TRY(hipMemcpyAsync(flag.data(), x->d_sum_flag, nchunks * sizeof(unsigned), hipMemcpyDeviceToHost, x->stream));
  if (wdpm_stream_sync(x, x->stream)) return 1;
  const bool start_ok = start >= 0.0 && start < __builtin_inf();
  double prefix = start_ok ? start : 0.0;
  size_t eligible = 0;
  for (size_t c = 0; c < nchunks; c++) {
    const double lo = prefix, hi = prefix + approx[c];
    prefix = hi;
    kexp[c] = (int)0x80000000;
    if (start_ok && !flag[c] && lo > 1e-290 && hi < 1e300) {
      int e_lo, e_hi;
      (void)frexp(lo * (1.0 - 1e-6), &e_lo);
      (void)frexp(hi * (1.0 + 1e-6), &e_hi);
      if (e_lo == e_hi) { kexp[c] = e_lo - 1; eligible++; }          /* 2^k <= running sum < 2^(k+1) all through the chunk */
    }
  }
  /* pass B: exact integer sums in units of 2^(k-52) */
  if (eligible) {
    HIP_TRY(hipMemcpyAsync(x->d_sum_k, kexp.data(), nchunks * sizeof(int), hipMemcpyHostToDevice, x->stream));
    HIP_TRY(wdpm_launch_seqsum_b(w, dem, n, x->d_sum_k, x->d_sum_i, x->d_sum_flag, x->stream));
    HIP_TRY(hipMemcpyAsync(isum.data(), x->d_sum_i, nchunks * sizeof(long long), hipMemcpyDeviceToHost, x->stream));
    HIP_TRY(hipMemcpyAsync(flag.data(), x->d_sum_flag, nchunks * sizeof(unsigned), hipMemcpyDeviceToHost, x->stream));
    if (wdpm_stream_sync(x, x->stream)) return 1;
  }
  /* chain: a chunk's integer sum is used only if its assumption holds for the ACTUAL running sum */
  double s = start;
  for (size_t c = 0; c < nchunks; c++) {
    const int k = kexp[c];
    if (k != (int)0x80000000 && !flag[c] && s >= ldexp(1.0, k)) {
      const double s_new = s + ldexp((double)isum[c], k - 52);        /* exact: both are multiples of 2^(k-52) */
      if (s_new < ldexp(1.0, k + 1)) { s = s_new; continue; }
    }
    const size_t off = c * kSeqSumChunk, len = n - off < (size_t)kSeqSumChunk ? n - off : (size_t)kSeqSumChunk;
    tw.resize(len); td.resize(len);
    HIP_TRY(hipMemcpyAsync(tw.data(), w + off, len * sizeof(double), hipMemcpyDeviceToHost, x->stream));
    HIP_TRY(hipMemcpyAsync(td.data(), dem + off, len * sizeof(double), hipMemcpyDeviceToHost, x->stream));
    if (wdpm_stream_sync(x, x->stream)) return 1;
    for (size_t i = 0; i < len; i++)
      if (td[i] < __builtin_inf()) s += tw[i];                        /* the device DEM holds +inf for NODATA */
  }
  *sum = s;
  return 0;
}

int wdpm_host_alloc(size_t bytes, void **ptr) {
  if (!ptr) return fail("wdpm_host_alloc: null argument");
  *ptr = nullptr;
  HIP_TRY(hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocPortable));
  return 0;
}

void wdpm_host_free(void *ptr) {
  if (ptr) (void)hipHostFree(ptr);
}

int wdpm_run_block(wdpm_ctx *x, int32_t n_iter, double thres, double *max_diff) {
  if (wdpm_begin_block(x, thres)) return 1;
  if (wdpm_expect_max_diff(x, 0, x->g.rows)) return 1;
  if (wdpm_iterate(x, n_iter)) return 1;
  return wdpm_max_diff(x, 0, x->g.rows, max_diff);
}

int wdpm_timing_reset(wdpm_ctx *x) {
  if (bind(x)) return 1;
  if (fold_timing(x)) return 1;
  x->launches = 0;
  x->ms = 0.0;
  x->steady_launches = 0;
  x->steady_ms = 0.0;
  x->xch_count = 0;
  x->xch_ms = 0.0;
  x->timing = true;
  return 0;
}

int wdpm_device_info(wdpm_ctx *x, int32_t *ordinal, char *pci_bus_id, int32_t len) {
  if (ordinal) *ordinal = x->p.device;
  if (pci_bus_id) {
    if (len < 16) return fail("wdpm_device_info: the PCI bus id needs 16 bytes, got %d", (int)len);
    HIP_TRY(hipDeviceGetPCIBusId(pci_bus_id, len, x->p.device));
  }
  return 0;
}

int wdpm_balance_info(wdpm_ctx *x, int32_t *updates, double *weights9) {
  if (updates) *updates = x->bal.mode ? x->bal.updates : 0;
  if (!weights9) return 0;
  for (int i = 0; i < 9; i++) weights9[i] = i < 8 ? 1.0 : 0.95;
  if (!x->bal.mode || !x->bal.weight) return 0;
  if (bind(x)) return 1;
  float w[9];
  HIP_TRY(hipMemcpyAsync(w, x->bal.weight, sizeof w, hipMemcpyDeviceToHost, x->stream));
  if (wdpm_stream_sync(x, x->stream)) return 1;
  for (int i = 0; i < 9; i++) weights9[i] = (double)w[i];
  return 0;
}

int wdpm_timing_get_exchange(wdpm_ctx *x, int64_t *refreshes, double *ms) {
  if (bind(x)) return 1;
  if (fold_timing(x)) return 1;
  if (refreshes) *refreshes = x->xch_count;
  if (ms) *ms = x->xch_ms;
  return 0;
}

int wdpm_timing_get_steady(wdpm_ctx *x, int64_t *launches, double *ms) {
  if (bind(x)) return 1;
  if (fold_timing(x)) return 1;
  if (launches) *launches = x->steady_launches;
  if (ms) *ms = x->steady_ms;
  return 0;
}

int wdpm_timing_get(wdpm_ctx *x, int64_t *launches, double *ms) {
  if (bind(x)) return 1;
  if (fold_timing(x)) return 1;
  if (launches) *launches = x->launches;
  if (ms) *ms = x->ms;
  return 0;
}

} /* extern "C" */
