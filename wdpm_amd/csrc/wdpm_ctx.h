/*
 * wdpm_ctx.h — the context object behind the opaque wdpm_ctx of include/wdpm.h (HIP back-end).
 * Internal to the product library: shared by wdpm_capi.hip (block loop) and wdpm_rccl.hip (RCCL halos).
 */
#ifndef WDPM_CTX_H
#define WDPM_CTX_H

#include <hip/hip_runtime.h>

#include <vector>

#include "../../include/wdpm.h"
#include "wdpm_kernels.h"

struct wdpm_comm;   /* RCCL communicator state of a context (wdpm_rccl.hip) */

struct EventPair { hipEvent_t a, b; };

struct wdpm_ctx {
  wdpm_params p;
  SlabGeom g;
  size_t cells;
  hipStream_t stream;
  bool own_stream;
  double *d_dem, *d_w[3];       /* three water rasters: the current one, the iteration kernel's write target, the snapshot */
  int cur;                      /* d_w[cur] is bigwater */
  int old;                      /* d_w[old] is oldwater (WDPMCL.c:1069-1073) - possibly still UNFLUSHED, see flush_pending */
  bool flush_pending;           /* wdpm_begin_block's threshold flush has not been applied yet: cur == old, and the next
                                   iteration launch applies it while loading (no pass over the raster of its own) */
  /* dry-tile skipping (wdpm_kernels.h::TileFlags): one flag array per water raster, valid for one tiling */
  unsigned char *d_zero[3];
  bool zero_valid[3];           /* d_zero[i] describes d_w[i]'s current content (for tiling tile_*) */
  int tile_cap, tile_nstrips, tile_H, tile_nchunks;
  unsigned *d_active;           /* waves that did not skip since the last look */
  unsigned *h_active;           /* pinned */
  int64_t tiles_launched;       /* tiles of the flag-maintaining launches since the last look */
  bool sparse;                  /* most tiles are dry: short chunks (96 rows), so that a wet tile is a short march */
  bool wide_tri_ok;             /* the last block that kept tile flags found > 60 % of the tiles working: mid-size rasters may go to the
                                 * triangle kernel (no flags); probed again with the marching kernel every 16 blocks */
  int blocks_unprobed;
  int tiles_mode;               /* 1 on (default), 0 off (WDPM_TILES=0 / WDPM_OPT_TILES) */
  int64_t stat_tiles, stat_active;   /* running totals for wdpm_get_option */
  /* max |w - oldw| folded into the last iteration launch of a block (wdpm_expect_max_diff) */
  unsigned long long *d_md;     /* its reduction cell */
  bool md_hint, md_valid;       /* the next wdpm_iterate may fold it / d_md holds the value for rows [md_lo, md_hi) */
  int md_lo, md_hi;
  bool drain_owed;              /* drain module: the last iteration's drain() (WDPMCL.c:1089) has not been applied to d_w[cur];
                                   the next iteration launch does it as it loads, anybody else asks ensure_drained() first */
  double flush_thres;           /* the threshold of the current block: the flush still owed to d_w[cur] (flush_pending)
                                   and to the snapshot when wdpm_max_diff reads it; -inf = none */
  double *d_scal;               /* [0] totaldrain, [1] olddrain */
  unsigned long long *d_bits;   /* max-diff reduction cell */
  double *h_pin;                /* pinned staging: 4 doubles */
  unsigned long long *d_stat;   /* wdpm_count_stats / wdpm_find_drain: 4 reduction cells */
  double *d_sum_approx;         /* wdpm_volume_partial: per-chunk approximate sums, integer sums, binades, flags */
  long long *d_sum_i;
  int *d_sum_k;
  unsigned *d_sum_flag;
  int kernel;                   /* resolved WDPM_KERNEL_* */
  bool signed_zero_safe;        /* a -0.0 depth was uploaded (or the caller asked): exact-zero stencil variant */
  bool w_negative, w_odd;       /* the current raster may hold negative depths (until the next threshold flush with thres >= 0) /
                                   NaN, absurd depths or water on NODATA cells (for good): wdpm_kernels.h::wdpm_launch_scan_water.
                                   With neither, and no -0.0, the gate-free kernel variants run (wdpm_fused.hip: PLAIN) */
  struct GuardedBuffer { char *base; size_t bytes; };   /* base = start of the front guard band; bytes = payload between the bands */
  std::vector<GuardedBuffer> guards;   /* WDPM_GUARD_KB: the buffers wdpm_get_option(WDPM_OPT_GUARD_BAD) inspects */
  int *d_dem32;                 /* the DEM as verified-lossless 32-bit codes (wdpm_kernels.h::DemCode) */
  DemCode code;                 /* code.q == d_dem32 while the uploaded DEM is encodable and the option is on */
  bool dem32_encodable;
  unsigned short *d_dem16;      /* the codes once more as 16-bit offsets from d_gbase (DemCode::h, ::gb); code.h is set while they are in use */
  int *d_gbase;
  bool dem16_encodable;
  bool dem16_wanted;          /* WDPM_OPT_DEM16 as last set (default on); what runs is decided with dem16_encodable and code.q */
  XcdBalance bal;               /* chunk heights by what each XCD delivers (wdpm_kernels.h); bal.mode == 0: off */
  bool dem_bounded;             /* every valid elevation of the uploaded DEM is below 2^30 m in magnitude (scan_dem): the clamped
                                   neighbour step may run where the depths allow it (wdpm_kernels.h: WDPM_LAUNCH_CLAMP_OK) */
  /* wdpm_iterate_overlapped: side stream for the interior launch and the event that joins it */
  hipStream_t side;
  hipEvent_t ev_fork, ev_join;
  bool pending_join;
  hipEvent_t ev_copy[2];        /* wdpm_copy_rows: [0] "my rows are produced" as source, [1] "the copy has read them" as destination */
  /* stencil timing */
  /* HIP graphs of steady small-raster iterations (wdpm_capi.hip: wdpm_iterate).  A launch of the relay / triangle kernels keeps no state
   * on the host, and within a block the water rasters ping-pong between two buffers: an even number of iterations captured once from
   * the context's own launches replays any number of times from the same pair.  What a captured launch was given is the key. */
  struct GraphEntry {
    int cur, old, flags, drain_owed, chunk_rows, dr, dc, wide, force;
    const void *q, *h;
    hipGraphExec_t exec;
  };
  std::vector<GraphEntry> graphs;
  int graph_mode;               /* -1 unknown, 0 off (WDPM_GRAPH=0, or a capture failed on this context), 1 on */
  int64_t graph_launches;       /* graphs launched so far (WDPM_OPT_GRAPH_LAUNCHES) */
  std::vector<EventPair> pending;
  std::vector<EventPair> pool;
  /* the same for the launches of a call between its first and its last (those two may be the flush-on-load and the
   * max-diff variants of the kernel): what rocprofv3 lists as the dominant kernel */
  std::vector<EventPair> pending_steady;
  /* halo refreshes (wdpm_comm_exchange, wdpm_copy_rows into this context): from the point of the stream where the transfer is
   * queued to the point where the rows have arrived - what an N-GPU bench line needs to explain its scaling */
  std::vector<EventPair> pending_xch;
  int64_t xch_count;
  double xch_ms;
  int64_t steady_launches;
  double steady_ms;
  int64_t launches;
  double ms;
  bool timing;                  /* record the event pairs at all (off until wdpm_timing_reset asks) */
  wdpm_comm *comm;              /* wdpm_comm_init_rank / wdpm_comm_init_all, or nullptr */
  bool leak;                    /* a guarded RCCL call or a stream wait ran past its deadline: somebody may still be using this
                                   context's stream and buffers, so wdpm_destroy frees none of them */
};

/* sets wdpm_last_error() of the calling thread and returns 1 */
int wdpm_fail(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
void wdpm_comm_release(wdpm_ctx *x);   /* wdpm_destroy's hook */
int wdpm_tiles_touch(wdpm_ctx *x, int row, int nrows);   /* rows of the current raster written from outside: dry-tile flags */
int wdpm_apply_owed_drain(wdpm_ctx *x);   /* drain module: the last iteration's drain() if it has not been applied yet (before rows leave the context) */
int wdpm_apply_owed_flush(wdpm_ctx *x);   /* that, and the block's threshold flush if the current raster is still owed it */
/* hipStreamSynchronize - with a deadline when the context has a communicator (a transfer whose peer has died never completes):
 * WDPM_SYNC_TIMEOUT_S, default 600 s; past it the communicator is aborted, the context is marked `leak` and 1 is returned */
int wdpm_stream_sync(wdpm_ctx *x, hipStream_t s);
/* halo-refresh timing (only while wdpm_timing_reset has switched timing on): begin records an event on the context's stream and
 * returns the pair, end records its second event */
extern "C" int wdpm_xch_timing_begin(wdpm_ctx *x, EventPair *ep);
extern "C" int wdpm_xch_timing_end(wdpm_ctx *x, const EventPair *ep);

#endif
