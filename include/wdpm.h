/*
 * wdpm.h — C ABI of the MI355X-native WDPM water-redistribution ("smoothing") path.
 *
 * This is the drop-in boundary for the hot path of CentreForHydrology/WDPM's WDPMCL:
 * the 9-colour 3x3 water-transfer sweep and its max-change convergence test.  The
 * reference has no library API for this path: its seam is three C
 * functions on globals plus the OpenCL kernel argument lists.  Every entry point below
 * cites the reference interface it replaces (file:line into the reference tree's
 * src/WDPMCL.c and src/runoff.cl).
 *
 * Conventions
 *   - plain C types only; every function returns 0 on success, non-zero on failure, and
 *     wdpm_last_error() then returns a message.  Nothing throws across this boundary.
 *   - rasters are PADDED, ROW-MAJOR double arrays: (nrows+2) x (ncols+2); interior cell
 *     (i,j) of the ArcASCII file is element [(i+1)*(ncols+2) + (j+1)]  (ref "bigdem",
 *     "bigwater", WDPMCL.c:796-807).  The reference OpenCL path is column-major
 *     (WDPMCL.c:1129-1134); this ABI is not.
 *   - the host owns host memory; the library owns device memory for the context's life.
 *   - a context may hold a SLAB (contiguous block of padded rows) of a larger raster for
 *     row-block domain decomposition across GPUs; slab_row0 must be a multiple of 3 so the
 *     colour alignment of the passes is the same on every slab.
 *
 * Two libraries export this same ABI:
 *   wdpm_amd/csrc  -> libwdpm_hip.so   the product (HIP kernels for gfx950; no CPU fallback)
 *   oracle/        -> libwdpm_oracle.so the CPU restatement used ONLY by tests/bench baseline
 */
#ifndef WDPM_H
#define WDPM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WDPM_ABI_VERSION 1

/* module selector: argv[1] of WDPMCL ("add" | "subtract" | "drain"), WDPMCL.c:308-355 */
enum { WDPM_ADD = 0, WDPM_SUBTRACT = 1, WDPM_DRAIN = 2 };

/* which stencil kernel implementation wdpm_iterate uses (product library only).
 * AUTO picks FUSED (WDPM_KERNEL=pass|fused in the environment overrides AUTO).  PASS = one launch per
 * colour pass (9 per iteration), the direct analogue of the reference's 9 clEnqueueNDRangeKernel calls
 * (WDPMCL.c:1184-1206); FUSED = one launch per iteration, register-resident marching window. */
enum { WDPM_KERNEL_AUTO = 0, WDPM_KERNEL_PASS = 1, WDPM_KERNEL_FUSED = 2 };

typedef struct wdpm_ctx wdpm_ctx; /* opaque */

typedef struct wdpm_params {
  int32_t module;       /* WDPM_ADD | WDPM_SUBTRACT | WDPM_DRAIN */
  int32_t nrows;        /* NROWS of the whole raster (file rows R), WDPMCL.c:553 */
  int32_t ncols;        /* NCOLS of the whole raster (file cols C), WDPMCL.c:552 */
  int32_t drainrow;     /* drain cell, padded coords of the WHOLE raster (WDPMCL.c:1005-1017); drain only */
  int32_t draincol;
  int32_t slab_row0;    /* first padded row of the whole raster held by this context (multiple of 3) */
  int32_t slab_rows;    /* padded rows held (0 => all nrows+2 rows, slab_row0 must then be 0) */
  int32_t device;       /* HIP device ordinal (replaces create_device(), WDPMCL.c:80-121) */
  int32_t kernel;       /* WDPM_KERNEL_* */
  int32_t chunk_rows;   /* fused kernel: rows per marching chunk (multiple of 3); 0 = automatic */
  double  missingvalue; /* NODATA_VALUE, WDPMCL.c:554 */
} wdpm_params;

/* -- lifetime ------------------------------------------------------------------------------
 * replaces OpenCL context/program/queue/kernel creation (WDPMCL.c:598-638) and the per-block
 * clCreateBuffer/clReleaseMemObject churn (WDPMCL.c:1138-1141,1223-1227): buffers persist.
 * One context takes at most 2e9 padded cells ((nrows + 2) * (ncols + 2)); a larger raster goes into several row blocks
 * (wdpm_group_create may name one device more than once). */
int  wdpm_create(wdpm_ctx **out, const wdpm_params *p);
void wdpm_destroy(wdpm_ctx *ctx);                      /* WDPMCL.c:1475-1483 */
const char *wdpm_last_error(void);                     /* replaces exitOnFail(), WDPMCL.c:225-232; per host thread */
void wdpm_set_last_error(const char *msg);             /* drivers above the ABI hand a worker thread's message to the caller's */
const char *wdpm_backend_name(void);                   /* "hip-gfx950" or "oracle-cpu" */
int  wdpm_abi_version(void);

/* -- data movement --------------------------------------------------------------------------
 * upload: replaces the two blocking clEnqueueWriteBuffer calls (WDPMCL.c:1143-1153).
 * Arrays hold slab_rows x (ncols+2) doubles (the context's slab only) and carry the reference's 1-cell border: the first and last
 * column - and, where the slab holds them, the raster's first and last row - have bigdem <= missingvalue and no water, as the
 * reference builds them (WDPMCL.c:796-807).  The serial loops never make a border cell a centre (rows 1..R, columns 1..C,
 * :1079-1080,1097-1098) and reach it as a neighbour only to find it NODATA (:1944,:1976); both back-ends rest on exactly that. */
int wdpm_upload(wdpm_ctx *ctx, const double *bigdem, const double *bigwater);
int wdpm_upload_water(wdpm_ctx *ctx, const double *bigwater);
/* download: replaces clEnqueueReadBuffer of bigwater (WDPMCL.c:1217-1221) */
int wdpm_download_water(wdpm_ctx *ctx, double *bigwater);
/* copy `nrows` slab rows starting at slab-local row `row` to/from host (halo refresh, tests) */
int wdpm_download_rows(wdpm_ctx *ctx, int32_t row, int32_t nrows, double *dst);
int wdpm_upload_rows(wdpm_ctx *ctx, int32_t row, int32_t nrows, const double *src);

/* totaldrain: the reference global (WDPMCL.c:236,1029), device-resident here (replaces
 * d_totaldrain, WDPMCL.c:1168-1178,1208-1213). */
int wdpm_set_totaldrain(wdpm_ctx *ctx, double v);
int wdpm_get_totaldrain(wdpm_ctx *ctx, double *v);

/* -- the block loop, step by step (WDPMCL.c:1049-1377 minus printing/termination) ------------
 * begin_block: threshold flush over the whole padded slab, border included (WDPMCL.c:1055-1065),
 *              olddrain = totaldrain (:1066-1068), snapshot oldwater = bigwater (:1069-1073). */
int wdpm_begin_block(wdpm_ctx *ctx, double thres);
/* iterate: n_iter iterations, each = 9 colour passes in the order oi=1..3 outer, oj=1..3 inner
 *          (WDPMCL.c:1094-1106; kernels add/subtract runoff.cl:137-164), stencil runoffs()
 *          (WDPMCL.c:1934-1964) for add and subtract; for drain runoffd() (:1967-2006) with the
 *          drain-centre gate (:1081-1082) and drain() (:1859-1897) after each iteration (:1089). */
int wdpm_iterate(wdpm_ctx *ctx, int32_t n_iter);
/* iterate, with the LAST iteration split so that a halo exchange can overlap it: the rows a
 * neighbour needs — the first `top_rows` and the last `bottom_rows` rows of the slab (0 = none) — are
 * produced first by two short launches on the context's stream; the remaining rows are produced by a
 * third launch on an internal side stream.  Work the caller queues on the context's stream right
 * after this call (the RCCL send/recv of the boundary rows) therefore starts as soon as the boundary
 * rows exist; the library joins the side stream before its own next use of the raster.
 * Results are identical to wdpm_iterate.  Add / subtract with the fused kernel; otherwise it simply
 * calls wdpm_iterate. */
int wdpm_iterate_overlapped(wdpm_ctx *ctx, int32_t n_iter, int32_t top_rows, int32_t bottom_rows);
/* one colour pass (oi,oj in 1..3) — the unit the reference launches (WDPMCL.c:1187-1204);
 * exposed for golden-vector tests of single passes. */
int wdpm_pass(wdpm_ctx *ctx, int32_t oi, int32_t oj);
/* drain() alone (WDPMCL.c:1859-1897): totaldrain += sum over outlet 3x3; zero the 9 cells */
int wdpm_drain_outlet(wdpm_ctx *ctx);
/* max_diff over slab-local rows [row_lo,row_hi) of |bigwater-oldwater| where bigdem>missing,
 * seeded with diff[0][0] of the slab's first row when row_lo==0 (WDPMCL.c:1239-1254).
 * row_lo=0,row_hi=slab_rows gives the reference value for a whole-raster context. */
int wdpm_max_diff(wdpm_ctx *ctx, int32_t row_lo, int32_t row_hi, double *out);
/* a hint: the caller is about to run the LAST wdpm_iterate / wdpm_iterate_overlapped of a block and will then ask
 * wdpm_max_diff for exactly these rows.  The library may then reduce max|bigwater-oldwater| inside that call's last
 * iteration launch, whose waves hold the final values anyway, instead of in a pass of its own over three rasters;
 * wdpm_max_diff returns that value if nothing has written those rows in between.  Never changes a result. */
int wdpm_expect_max_diff(wdpm_ctx *ctx, int32_t row_lo, int32_t row_hi);
/* drain bookkeeping (WDPMCL.c:1257-1268): diffdrain = |totaldrain-olddrain| (NOT yet times
 * cellarea); final_sum = sum of bigwater over bigdem>missing in row-major order (NOT yet times
 * cellarea), bit-identical to the reference's sequential summation. */
int wdpm_drain_stats(wdpm_ctx *ctx, double *diffdrain, double *final_sum);
/* the same row-major sum restricted to slab-local rows [row_lo,row_hi) and continued from `start`:
 * rank g of a row-block decomposition passes the sum of ranks 0..g-1, so the chain reproduces
 * the single-raster summation order exactly. */
int wdpm_volume_partial(wdpm_ctx *ctx, int32_t row_lo, int32_t row_hi, double start, double *sum);
/* convenience: begin_block + iterate(n_iter) + max_diff over the whole slab */
int wdpm_run_block(wdpm_ctx *ctx, int32_t n_iter, double thres, double *max_diff);

/* -- set-up and final statistics next to the rasters ---------------------------------
 * What the reference does in host loops around the block loop, here on the device (CPU restatement: host
 * loops).  A context holding a slab does its slab's share; wdpm_group_* below combines the shares. */
typedef struct wdpm_setup {
  int32_t op;             /* 0: water as given; 1: add (WDPMCL.c:727-740: wet cells += add, then cells <= 0 := add*rof);
                             2: subtract (:879-885: max(water - sub, 0)); valid cells only */
  double add, rof, sub;   /* metres, fraction, metres */
} wdpm_setup;
/* instead of wdpm_upload: the slab's padded rasters (border = missing / 0, WDPMCL.c:796-807) built from the
 * UNPADDED file rasters (nrows x ncols of the whole raster; water may be NULL = zeros), the module's water
 * adjustment applied on the way */
int wdpm_upload_unpadded(wdpm_ctx *ctx, const double *dem, const double *water, const wdpm_setup *setup);
/* over slab-local padded rows [row_lo,row_hi): cells with bigdem > missing (basincount, WDPMCL.c:643-650),
 * those with water > 0.001 (:1397-1404), and max over ALL cells of (valid ? water : missing) decided by `>`
 * from -inf (:1448-1457 after the masking of :1386-1391; the caller folds in the reference's seed cell) */
int wdpm_count_stats(wdpm_ctx *ctx, int32_t row_lo, int32_t row_hi, int64_t *valid, int64_t *wet, double *maxv);
/* the drain cell (WDPMCL.c:1005-1017) among rows [row_lo,row_hi): smallest bigdem > 0, first in row-major
 * order; slab-local row (-1: no such cell) */
int wdpm_find_drain(wdpm_ctx *ctx, int32_t row_lo, int32_t row_hi, double *mindem, int32_t *row, int32_t *col);
/* a drain context created before the drain cell was known (drainrow < 0) learns it; whole-raster coordinates */
int wdpm_set_drain(wdpm_ctx *ctx, int32_t drainrow, int32_t draincol);
int wdpm_get_cell(wdpm_ctx *ctx, int32_t row, int32_t col, double *water, double *dem);   /* slab-local row */
/* rows [file_row, file_row+nrows) of the FILE raster (they must lie in the slab), un-padded, NODATA cells as
 * missingvalue when mask_missing (WDPMCL.c:1379-1392; scratch :1336-1344) */
int wdpm_download_unpadded(wdpm_ctx *ctx, int32_t file_row, int32_t nrows, int32_t mask_missing, double *dst);

/* -- plumbing for multi-GPU drivers (product library; the oracle returns host pointers) ------ */
/* device pointer of the CURRENT water raster (changes after every wdpm_iterate/wdpm_pass) */
int wdpm_water_ptr(wdpm_ctx *ctx, void **ptr);
int wdpm_dem_ptr(wdpm_ctx *ctx, void **ptr);   /* device copy: NODATA cells hold +inf */
/* run all work of this context on the caller's hipStream_t (e.g. torch's current stream) */
int wdpm_set_stream(wdpm_ctx *ctx, void *hip_stream);
int wdpm_synchronize(wdpm_ctx *ctx);
/* device-side timing of the stencil launches since the last reset (HIP events on the
 * context's stream): number of stencil launches and their summed duration in ms. */
int wdpm_timing_reset(wdpm_ctx *ctx);
int wdpm_timing_get(wdpm_ctx *ctx, int64_t *launches, double *ms);
/* the same restricted to the launches of each wdpm_iterate call between its first and its last (calls of >= 3
 * iterations): the first launch of a block may be the flush-on-load variant of the kernel and the last the max-diff
 * variant, so this is the kernel a profiler lists as the dominant one.  0 launches if there were none. */
int wdpm_timing_get_steady(wdpm_ctx *ctx, int64_t *launches, double *ms);
/* ... and of the halo refreshes INTO this context since the last reset: how many, and the time from the point of the
 * context's stream where a transfer (RCCL group or peer copy) was queued to the point where its rows had arrived - next
 * to the kernel time, what an N-GPU run needs to explain its scaling (bench.py reports both per rank).  No counterpart in
 * the reference (one OpenCL device, WDPMCL.c:598-638). */
int wdpm_timing_get_exchange(wdpm_ctx *ctx, int64_t *refreshes, double *ms);
/* chunk heights that follow what each XCD delivers (no counterpart in the reference): how often the weights have been
 * rebalanced so far, and the nine weights - relative chunk heights of the work items of XCD 0 .. 7 (logical: blockIdx % 8), mean 1, and
 * the factor on a strip's last chunk.  All 1 (0.95) and 0 updates while the balance has not engaged (small or mostly dry rasters, WDPM_BALANCE=0).
 * bench.py prints them: which XCDs of a box are slow, and by how much, is part of what explains a number. */
int wdpm_balance_info(wdpm_ctx *ctx, int32_t *updates, double *weights9);
/* what this library was built from: "kernels=<first 16 hex digits of the sha256 of the kernel sources> arch=... sched=..." - bench.py
 * quotes counter evidence collected on another run (profiles/traffic.json) only for a library that says the same */
const char *wdpm_build_info(void);
/* which physical GPU a context lives on: the HIP ordinal as this process sees it (HIP_VISIBLE_DEVICES renumbers) and the PCI bus id
 * ("0000:c1:00.0", hipDeviceGetPCIBusId: one per GPU of a node whatever the ordinal; `len` >= 16 bytes).  bench.py gathers one per
 * rank, so that an N-rank line shows N distinct GPUs - or says that it is a rehearsal on fewer.  Takes the place of
 * the reference's device report (create_device() prints CL_DEVICE_NAME of the ONE device it picks, WDPMCL.c:80-121).
 * The CPU restatement: ordinal -1, "host". */
int wdpm_device_info(wdpm_ctx *ctx, int32_t *ordinal, char *pci_bus_id, int32_t len);

/* copy `nrows` rows of the CURRENT water raster from slab-local row `src_row` of `src` to row
 * `dst_row` of `dst` (same raster width).  Device to device on the HIP back-end (peer copy over
 * xGMI when the contexts live on different GPUs; direct peer access is enabled on first use),
 * ordered after everything queued on either context and before anything queued on either
 * afterwards — on the devices (stream events); the call does not wait for the copy.  The halo
 * refresh primitive of the in-process peer-copy transport (WDPM_HALO_PEER).  One host thread at a
 * time may use a given context. */
int wdpm_copy_rows(wdpm_ctx *dst, int32_t dst_row, wdpm_ctx *src, int32_t src_row, int32_t nrows);
/* hipDeviceCanAccessPeer / hipDeviceEnablePeerAccess in both directions between the contexts' devices
 * (a refusal is reported on stderr and leaves staged copies in charge; the CPU restatement: no-op) */
int wdpm_enable_peer_access(wdpm_ctx *a, wdpm_ctx *b);

/* -- RCCL halos (product library; the CPU restatement has none and fails these calls) ------------
 * Replaces the reference's device set-up, which picks ONE OpenCL device (create_device(),
 * WDPMCL.c:80-121, :598-638): a context joins an RCCL communicator, one rank per GPU. */
#define WDPM_COMM_ID_BYTES 128
typedef struct wdpm_halo_op { int32_t peer, row, nrows; } wdpm_halo_op;   /* rank, slab-local first row, row count */
int wdpm_comm_available(void);                         /* 1 when the RCCL library could be bound */
const char *wdpm_comm_version(void);
int wdpm_comm_unique_id(void *id128);                  /* ncclGetUniqueId: rank 0 makes it, every rank gets a copy */
/* one process per rank: collective over all ranks (ncclCommInitRank on the context's device) */
int wdpm_comm_init_rank(wdpm_ctx *ctx, int32_t nranks, int32_t rank, const void *id128);
/* all ranks in one process: ncclCommInitAll over the contexts' (distinct) devices, rank = index */
int wdpm_comm_init_all(wdpm_ctx **ctxs, int32_t n);
int wdpm_comm_size(wdpm_ctx *ctx, int32_t *nranks, int32_t *rank);   /* as RCCL reports it (ncclCommCount) */
/* grouped ncclSend / ncclRecv of rows of the CURRENT water raster on the context's stream; returns
 * once queued.  Every rank must call it with matching ops. */
int wdpm_comm_exchange(wdpm_ctx *ctx, int32_t nsend, const wdpm_halo_op *sends, int32_t nrecv, const wdpm_halo_op *recvs);
/* all[r*n + i] = mine[i] of rank r on every rank (n <= 8); synchronous */
int wdpm_comm_allgather(wdpm_ctx *ctx, const double *mine, int32_t n, double *all);
/* Deadlines: communicator set-up and the first transfer / all-gather of a communicator run with a deadline
 * (WDPM_RCCL_TIMEOUT_S, default 90 s), and so does every wait for the stream of a context that has a communicator
 * (WDPM_SYNC_TIMEOUT_S, default 600 s): past it the call returns 1 and the communicator is aborted.
 * wdpm_comm_abort ends a context's communicator at once (ncclCommAbort; no-op without one): its queued transfers end,
 * the peers' matching ones fail instead of waiting for rows that never come.  Callable from another thread than the one
 * driving the context - a rank of a group that fails aborts every rank's (the reference exits on any device error,
 * WDPMCL.c:92-118,225-232; so does WDPMCL here, non-zero, instead of hanging). */
int wdpm_comm_abort(wdpm_ctx *ctx);

/* -- row-block decomposition: the raster over several GPUs (wdpm_rowblock.c, both libraries) ------
 * New work relative to the single-device reference.  Rank g owns padded rows
 * [own_lo, own_hi] (own_lo % 3 == 2) and holds 3k-1 halo rows above / 6k-2 below, runs
 * k = exchange_every iterations without communication, then refreshes the halos from its
 * neighbours' owned rows; the last iteration before a refresh is split so that the transfer
 * overlaps its interior rows (wdpm_iterate_overlapped).  Results are bit-identical to one context.
 * For the drain module the partition keeps the outlet's row at least three rows inside its owner's
 * rows, so that everything totaldrain is summed from is exact in the owner. */
typedef struct wdpm_slab {
  int32_t own_lo, own_hi;   /* owned padded rows of the whole raster, inclusive */
  int32_t row0, rows;       /* slab held: first padded row (multiple of 3), row count */
  int32_t up, down;         /* halo rows held above own_lo / below own_hi */
} wdpm_slab;
/* slabs[0..nranks): 0 on success, 1 when some slab would be shorter than the halo it must serve.
 * drainrow < 0 (or a module other than WDPM_DRAIN): no outlet to keep clear of. */
int wdpm_partition(int32_t nrows, int32_t nranks, int32_t exchange_every, int32_t module, int32_t drainrow,
                   wdpm_slab *slabs);

/* how halo rows travel between ranks */
enum {
  WDPM_HALO_AUTO = 0,   /* RCCL where it can be had, else the next that applies */
  WDPM_HALO_RCCL = 1,   /* ncclSend/ncclRecv on the context's stream (wdpm_comm_*) */
  WDPM_HALO_PEER = 2,   /* ranks of ONE process: hipMemcpyPeerAsync between the contexts (wdpm_copy_rows) */
  WDPM_HALO_HOST = 3    /* staged through host memory and moved by the caller's functions (e.g. gloo): several
                           ranks sharing one GPU in tests, or a platform that refuses the device paths */
};
/* WDPM_HALO_HOST: the caller moves the bytes.  Both functions are collective and return 0 on success. */
typedef struct wdpm_host_transport {
  void *user;
  /* n_ops buffers: is_send[i] ? send count[i] doubles at buf[i] to rank peer[i] : receive them from it */
  int (*exchange)(void *user, int32_t n_ops, const int32_t *is_send, const int32_t *peer, double *const *buf,
                  const int64_t *count);
  /* all[r*n + i] = mine[i] of rank r */
  int (*allgather)(void *user, const double *mine, int32_t n, double *all);
} wdpm_host_transport;

/* One rank of a decomposed run: its slab context plus its end of the halo transport.  Every function
 * marked (collective) must be called by all ranks.  `whole` describes the whole raster; its device field
 * names this rank's GPU.  rccl_id: WDPM_COMM_ID_BYTES from wdpm_comm_unique_id of rank 0 (RCCL), host: the
 * caller's transport (HOST); AUTO takes RCCL when rccl_id is given, else HOST. */
typedef struct wdpm_rank wdpm_rank;
int  wdpm_rank_create(wdpm_rank **out, const wdpm_params *whole, int32_t rank, int32_t nranks, int32_t exchange_every,
                      int32_t halo, const void *rccl_id, const wdpm_host_transport *host);   /* (collective) */
void wdpm_rank_destroy(wdpm_rank *r);
wdpm_ctx *wdpm_rank_ctx(wdpm_rank *r);                                   /* the slab context (timing, options, rows) */
int  wdpm_rank_slab(wdpm_rank *r, int32_t of_rank, wdpm_slab *out);      /* of_rank < 0: this rank's */
int  wdpm_rank_info(wdpm_rank *r, int32_t *halo, int32_t *exchange_every, int32_t *drain_owner);
int  wdpm_rank_upload(wdpm_rank *r, const double *slab_dem, const double *slab_water);      /* (collective) this rank's slab rows */
int  wdpm_rank_upload_global(wdpm_rank *r, const double *bigdem, const double *bigwater);   /* (collective) whole padded rasters */
int  wdpm_rank_set_totaldrain(wdpm_rank *r, double v);
int  wdpm_rank_get_totaldrain(wdpm_rank *r, double *v);                  /* (collective) the outlet owner's value */
int  wdpm_rank_begin_block(wdpm_rank *r, double thres);
int  wdpm_rank_iterate(wdpm_rank *r, int32_t n_iter);                    /* (collective) halo refresh every k iterations */
int  wdpm_rank_exchange(wdpm_rank *r);                                   /* (collective) refresh the halos now */
int  wdpm_rank_max_diff(wdpm_rank *r, double *max_diff);                 /* (collective) over all ranks' owned rows */
int  wdpm_rank_run_block(wdpm_rank *r, int32_t n_iter, double thres, double *max_diff);     /* (collective) */
int  wdpm_rank_drain_stats(wdpm_rank *r, double *diffdrain, double *final_sum);             /* (collective) */
int  wdpm_rank_download_owned(wdpm_rank *r, double *dst);                /* own_hi-own_lo+1 rows */

/* -- a raster spread over several GPUs of ONE process (what WDPMCL uses when WDPM_GPUS > 1) ------
 * The same ranks, each driven by its own host thread; halos by RCCL (ncclCommInitAll) when the
 * devices are distinct and RCCL can be bound, else by peer copies (WDPM_HALO=rccl|peer in the
 * environment decides otherwise).  `p` describes the WHOLE raster (slab/device fields ignored);
 * devices[] may name the same device several times (testing on one GPU: peer-copy transport). */
typedef struct wdpm_group wdpm_group;
int  wdpm_group_create(wdpm_group **out, const wdpm_params *p, int32_t ndev, const int32_t *devices,
                       int32_t exchange_every);
void wdpm_group_destroy(wdpm_group *grp);
int  wdpm_group_size(wdpm_group *grp);                      /* devices actually used */
int  wdpm_group_halo(wdpm_group *grp);                      /* WDPM_HALO_* in use (0 for one device) */
wdpm_rank *wdpm_group_rank(wdpm_group *grp, int32_t i);
int  wdpm_group_upload(wdpm_group *grp, const double *bigdem, const double *bigwater);
int  wdpm_group_download_water(wdpm_group *grp, double *bigwater);
int  wdpm_group_set_totaldrain(wdpm_group *grp, double v);
int  wdpm_group_get_totaldrain(wdpm_group *grp, double *v);
int  wdpm_group_run_block(wdpm_group *grp, int32_t n_iter, double thres, double *max_diff);
int  wdpm_group_drain_stats(wdpm_group *grp, double *diffdrain, double *final_sum);
/* set-up and final statistics over the whole raster (see wdpm_upload_unpadded etc.):
 * upload_unpadded replaces wdpm_group_upload; find_drain returns whole-raster padded coordinates;
 * set_drain returns 2 (and changes nothing) when the outlet lies too close to a slab boundary of the current
 * partition - the caller then creates the group again with the now known drain cell;
 * count_stats: maxv still wants the reference's seed folded in (water[0][0] of the masked file raster) */
int  wdpm_group_upload_unpadded(wdpm_group *grp, const double *dem, const double *water, const wdpm_setup *setup);
int  wdpm_group_count_stats(wdpm_group *grp, int64_t *valid, int64_t *wet, double *maxv);
int  wdpm_group_find_drain(wdpm_group *grp, double *mindem, int32_t *drainrow, int32_t *draincol);
int  wdpm_group_set_drain(wdpm_group *grp, int32_t drainrow, int32_t draincol);
int  wdpm_group_get_cell(wdpm_group *grp, int32_t row, int32_t col, double *water, double *dem);
int  wdpm_group_download_unpadded(wdpm_group *grp, int32_t mask_missing, double *water);
/* host time the group's threads spent since creation queueing iteration launches and, separately, in halo
 * refreshes (both summed over ranks, seconds; the launch calls block when the device's queue is full, the
 * peer transport's refresh includes waiting for the other threads), and the iterations ONE rank queued */
int  wdpm_group_enqueue_stats(wdpm_group *grp, double *seconds, double *exchange_seconds, int64_t *iterations);

/* -- options -----------------------------------------------------------------------------------
 * WDPM_OPT_SIGNED_ZERO_SAFE (get/set): 1 = the add/subtract stencil must preserve the sign of
 *   zero-depth cells.  The library sets it by itself when an uploaded water raster contains -0.0
 *   (it then runs the kernel variant that keeps the reference's conditional updates); otherwise the
 *   faster variant, which is bit-identical whenever no -0.0 is present (none can be created), runs.
 *   A multi-GPU driver whose transport writes halo rows straight into device memory must OR the
 *   flag over all ranks and set it (wdpm_amd/rowblock.py does).
 * WDPM_OPT_DEM32 (get/set): 1 = the iteration kernel streams the static DEM as 32-bit codes
 *   (4 bytes per cell-update of HBM traffic instead of 8).  Switched on by wdpm_upload when the device
 *   has verified, cell by cell and bit for bit, that the DEM is k / 10^e with 32-bit k (DEMs read
 *   from decimal text are); results are identical either way.  Setting 0 forces the fp64 DEM,
 *   setting 1 is honoured only for a DEM that passed the check; the kernel then uses the codes on
 *   launches large enough for them to pay (add / subtract >= 4096^2 or so; drain, from round 4, on launches of two waves per SIMD),
 *   setting 2 on launches of any size.  WDPM_DEM32=0 in the environment
 *   disables the encoding altogether, WDPM_DEM32=2 makes 2 the default.  The CPU restatement reports 0.
 * WDPM_OPT_DEM16 (get/set; round 4): 1 = the marching kernel streams those codes as 16-bit offsets from one 32-bit base per 48
 *   columns of a row (2.08 B per cell: 18.1 B of HBM traffic per cell-update) - an exact integer identity with the verified 32-bit
 *   codes, possible where no such group spans more than 65 534 quanta; checked at upload, results identical either way.  The get returns
 *   1 when whole-slab launches of this context use them (from 10^8 cells on, where they pay), 2 when they are available but the
 *   context is too small for that, 0 when they are off or the terrain does not allow them.
 * WDPM_OPT_WATER_KINDS (get/set; round 3): what the library's scan of every uploaded water raster found, as a bit mask -
 *   1: a -0.0 depth (= WDPM_OPT_SIGNED_ZERO_SAFE), 2: a negative depth (forgotten again once a threshold flush with
 *   thres >= 0 has been applied), 4: NaN, a depth above 1e290 or water on a NODATA cell.  With none of them every cell that
 *   may not give water (dry, NODATA) holds +0.0 exactly, a state no operation of the loop can leave, and the iteration kernels
 *   then run variants that spend no instructions on the reference's centre test (WDPMCL.c:1099: `bigwater > 0 &&
 *   bigdem > missingvalue`) - such a centre's transfers come out as zero by themselves; WDPM_OPT_PLAIN_WATER (get) says
 *   whether they run now.  Setting ORs bits in: a multi-GPU driver whose transport writes halo rows straight into device
 *   memory must OR the mask over all ranks after every upload and set it on each (wdpm_rowblock.c does).  WDPM_PLAIN=0 in the
 *   environment keeps the gated variants.  Results are identical either way.
 * WDPM_OPT_TILES (get/set): 1 (default; WDPM_TILES=0 in the environment: 0) = dry-tile skipping.  The reference
 *   skips dry centres cell by cell (WDPMCL.c:1099); the iteration kernel keeps, per water raster, one flag per
 *   tile (the output block of one wave) saying "all +0.0", and a wave whose tile and eight neighbours are flagged
 *   neither loads nor - when the output raster's block is known to hold zeros too - stores anything.  Results
 *   are identical either way.  WDPM_OPT_TILES_SEEN / _WORKED (get): tiles of flag-keeping launches and those
 *   among them that did work, summed at every wdpm_max_diff.  WDPM_OPT_SPARSE (get/set): the kernel marches
 *   short chunks because most tiles were dry in the last block (the library switches by itself).
 * WDPM_OPT_GRAPH_LAUNCHES (get; round 5): HIP graphs launched by wdpm_iterate so far.  Small rasters are launch-bound (basin5: 5.2 us of
 *   kernel, 6.3 us from launch to launch queued one by one), so the iterations between a block's first and last launch are replayed as
 *   graphs of 32 launches captured from the library's own loop - where a launch keeps no state on the host (the small-raster kernels)
 *   and nobody times the launches.  WDPM_GRAPH=0 in the environment: never.  Results are identical either way.  The reference's
 *   counterpart is nine blocking clEnqueueNDRangeKernel + clFinish pairs per iteration (WDPMCL.c:1155-1210).
 * WDPM_OPT_GUARD_BAD (get): with WDPM_GUARD_KB=<n> in the environment when the context was made, the big device buffers
 *   carry n KiB guard bands; this counts guard bytes that were overwritten (0 = no kernel wrote outside its buffer; always 0
 *   without the variable).  A debugging aid: the GPU pool has no address sanitizer. */
enum { WDPM_OPT_SIGNED_ZERO_SAFE = 1, WDPM_OPT_DEM32 = 2, WDPM_OPT_TILES = 3, WDPM_OPT_TILES_SEEN = 4,
       WDPM_OPT_TILES_WORKED = 5, WDPM_OPT_SPARSE = 6, WDPM_OPT_GUARD_BAD = 7, WDPM_OPT_WATER_KINDS = 8,
       WDPM_OPT_PLAIN_WATER = 9, WDPM_OPT_DEM16 = 10, WDPM_OPT_GRAPH_LAUNCHES = 11 };
int wdpm_get_option(wdpm_ctx *ctx, int32_t key, int64_t *value);
int wdpm_set_option(wdpm_ctx *ctx, int32_t key, int64_t value);

/* -- host staging memory for whole rasters.  The HIP back-end returns page-locked memory, so that
 * wdpm_upload / wdpm_download_water / the scratch checkpoint move at PCIe rate instead of through a
 * pageable bounce buffer (the reference maps its host arrays with CL_MEM_USE_HOST_PTR,
 * src/WDPMCL.c:1138-1141); the CPU restatement returns malloc memory.  Release with wdpm_host_free. */
int wdpm_host_alloc(size_t bytes, void **ptr);
void wdpm_host_free(void *ptr);

/* -- synthetic DEM generator: integer-seeded, identical on every
 * host.  Writes an n x n UNPADDED row-major raster. */
int wdpm_synth_dem(int32_t n, uint64_t seed, double *dem);

#ifdef __cplusplus
}
#endif
#endif /* WDPM_H */
