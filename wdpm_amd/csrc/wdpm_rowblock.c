/*
 * wdpm_rowblock.c — the raster over several GPUs: row-block decomposition with deep halos.
 *
 * The ONE multi-GPU driver of this code base.  Back-end agnostic C written against the per-context
 * ABI of include/wdpm.h (linked into the HIP library and, for CPU tests of the very same logic, into
 * the oracle library).  The reference is single-device (its block loop is
 * src/WDPMCL.c:1049-1377); this is what lets the drop-in use the GPUs of a node.
 *
 * A `wdpm_rank` is one rank of a decomposed run: its slab context plus its end of the halo
 * transport.  Ranks are driven either by one process each (bench.py under torch.distributed.run:
 * wdpm_rank_create with an RCCL id) or by one host thread each inside one process (`wdpm_group`, what
 * WDPMCL uses with WDPM_GPUS=N) — the same code runs in both.
 *
 * Exactness.  Rows outside a slab act as NODATA, so the rows next to a slab edge go wrong and the
 * error creeps inward at the rate at which a cell can depend on other cells: per iteration an error at
 * the lower slab edge climbs 4 rows in the first iteration and 6 in each further one, an error at the
 * upper edge descends 2 rows, then 3 per iteration.  k iterations therefore need 3k-1 halo rows above
 * and 6k-2 below when the slab boundary L satisfies L % 3 == 2 (tests/test_rowblock.py re-derives both
 * with a cell-level dependency simulation of the pass order and shows on rasters that the rule is
 * sufficient).  A rank that owns rows [L, H] and holds that halo runs k iterations with no
 * communication and its owned rows stay bit-identical to the single-device result; the halos are then
 * refreshed from the neighbours' owned rows.  The slab's first row L-(3k-1) is a multiple of 3, which
 * keeps the colour alignment of every slab equal to the whole raster's.
 * Drain: totaldrain is summed from the outlet's 3x3 neighbourhood (runoffd's outlet branch,
 * WDPMCL.c:1980-1985, and drain(), :1859-1897), so those three rows must be OWNED rows of the rank
 * whose totaldrain is reported: wdpm_partition moves a boundary that comes closer than three rows.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/wdpm.h"

#define MAXR 64
#define GATHER_MAX 8

static int rb_fail(const char *msg) {
  wdpm_set_last_error(msg);
  return 1;
}

static double now_s(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

/* ---- partition ------------------------------------------------------------------------------ */
int wdpm_partition(int32_t nrows, int32_t n, int32_t k, int32_t module, int32_t drainrow, wdpm_slab *s) {
  if (nrows < 1 || n < 1 || n > MAXR || k < 1 || !s) return 1;
  const int P = nrows + 2, up = 3 * k - 1, down = 6 * k - 2;
  int bounds[MAXR + 1];
  bounds[0] = 0;
  for (int g = 1; g < n; g++) {
    int b = (int)((long long)P * g / n);
    b -= ((b - 2) % 3 + 3) % 3;                       /* boundaries = 2 (mod 3) */
    if (module == WDPM_DRAIN && drainrow >= 0)
      while (b > drainrow - 3 && b < drainrow + 4) b -= 3;   /* outlet >= 3 rows inside its owner's rows */
    bounds[g] = b;
  }
  bounds[n] = P;
  for (int g = 0; g < n; g++) {
    const int lo = bounds[g], hi = bounds[g + 1] - 1;
    if (hi < lo || lo < 0) return 1;
    if (n > 1 && hi - lo + 1 < (up > down ? up : down)) return 1;   /* it must serve its neighbours' halos from owned rows */
    const int r0 = g > 0 ? (lo - up > 0 ? lo - up : 0) : 0;
    const int r1 = g < n - 1 ? (hi + down < P - 1 ? hi + down : P - 1) : P - 1;
    s[g].own_lo = lo; s[g].own_hi = hi; s[g].row0 = r0; s[g].rows = r1 - r0 + 1;
    s[g].up = lo - r0; s[g].down = r1 - hi;
  }
  return 0;
}

/* ---- one rank -------------------------------------------------------------------------------- */
struct wdpm_group;
struct wdpm_rank {
  wdpm_params p;                /* the whole raster; .device = this rank's GPU */
  int rank, n, k, since, ncp;
  wdpm_slab s[MAXR];            /* everybody's slab */
  wdpm_ctx *c;
  int halo;                     /* WDPM_HALO_* in use (0 when n == 1) */
  int drain_owner;
  int overlap;                  /* split the iteration before an exchange (wdpm_iterate_overlapped) */
  wdpm_host_transport host;
  struct wdpm_group *grp;       /* the in-process group this rank is a thread of, or NULL */
  int nsend, nrecv;
  wdpm_halo_op sends[2], recvs[2];
  double *stage[4];             /* HOST: staging for sends[0..1], recvs[0..1] */
  int top_rows, bottom_rows;    /* rows a neighbour needs from the top / bottom of the slab */
  double enqueue_s;             /* host time spent queueing iteration launches */
  double exchange_s;            /* host time spent in halo refreshes (peer transport: includes waiting for the other threads) */
  int64_t iters;
};

struct wdpm_group {
  int n, halo;
  wdpm_rank *r[MAXR];
  pthread_t th[MAXR];
  int nthreads;
  pthread_mutex_t mu;
  pthread_cond_t cv_cmd, cv_done, cv_bar;
  int cmd, quit, ndone;
  unsigned seq;
  /* arguments of the current command */
  const double *a_dem, *a_water;
  double *a_out;
  int a_iter, a_want_diff, a_want_sum, a_mask, a_row, a_col;
  double a_thres, a_value;
  const wdpm_setup *a_setup;
  /* results, one per rank (all ranks agree on the collective ones; rank 0's is returned) */
  double res_a[MAXR], res_b[MAXR], res_c[MAXR];
  int64_t res_i[MAXR], res_j[MAXR];
  /* barrier + scalar exchange between the rank threads */
  int bar_count;
  unsigned bar_gen;
  double gather[MAXR * GATHER_MAX];
  int failed;
  char err[512];
};

static void grp_fail(struct wdpm_group *g, const char *msg) {
  pthread_mutex_lock(&g->mu);
  const int first = !g->failed;
  if (first) {
    g->failed = 1;
    snprintf(g->err, sizeof g->err, "%s", msg && *msg ? msg : "a rank of the group failed");
  }
  pthread_cond_broadcast(&g->cv_bar);
  pthread_mutex_unlock(&g->mu);
  /* the other ranks may have halo receives queued whose sender - this rank - will never post: end every communicator,
   * so that their streams drain and their waits return instead of hanging (the host barriers were released above) */
  if (first && g->halo == WDPM_HALO_RCCL)
    for (int i = 0; i < g->n; i++)
      if (g->r[i]) wdpm_comm_abort(g->r[i]->c);
}

/* all rank threads meet; released early (returning 1) once any rank has failed */
static int grp_barrier(struct wdpm_group *g) {
  pthread_mutex_lock(&g->mu);
  if (!g->failed) {
    const unsigned gen = g->bar_gen;
    if (++g->bar_count == g->n) {
      g->bar_count = 0;
      g->bar_gen++;
      pthread_cond_broadcast(&g->cv_bar);
    } else {
      while (gen == g->bar_gen && !g->failed) pthread_cond_wait(&g->cv_bar, &g->mu);
    }
  }
  const int f = g->failed;
  pthread_mutex_unlock(&g->mu);
  if (f) wdpm_set_last_error(g->err);
  return f;
}

/* all[q*n + i] = mine[i] of rank q */
static int rank_allgather(wdpm_rank *r, const double *mine, int n, double *all) {
  if (r->n == 1) { memcpy(all, mine, (size_t)n * sizeof(double)); return 0; }
  if (r->grp) {                                         /* threads of one process: shared memory */
    struct wdpm_group *g = r->grp;
    memcpy(g->gather + (size_t)r->rank * GATHER_MAX, mine, (size_t)n * sizeof(double));
    if (grp_barrier(g)) return 1;
    for (int q = 0; q < r->n; q++) memcpy(all + (size_t)q * n, g->gather + (size_t)q * GATHER_MAX, (size_t)n * sizeof(double));
    return grp_barrier(g);                              /* nobody overwrites its cell before all have read */
  }
  if (r->halo == WDPM_HALO_RCCL) return wdpm_comm_allgather(r->c, mine, n, all);
  if (r->host.allgather(r->host.user, mine, n, all)) return rb_fail("the caller's allgather failed");
  return 0;
}

static void plan_exchange(wdpm_rank *r) {
  const wdpm_slab *me = &r->s[r->rank];
  const int lo = me->own_lo - me->row0, hi = me->own_hi - me->row0 + 1;   /* slab-local owned rows [lo, hi) */
  r->nsend = r->nrecv = 0;
  r->top_rows = r->bottom_rows = 0;
  if (r->rank > 0) {
    const wdpm_slab *above = &r->s[r->rank - 1];
    r->sends[r->nsend++] = (wdpm_halo_op){r->rank - 1, lo, above->down};          /* my first rows are its lower halo */
    r->recvs[r->nrecv++] = (wdpm_halo_op){r->rank - 1, 0, me->up};
    r->top_rows = lo + above->down;
  }
  if (r->rank < r->n - 1) {
    const wdpm_slab *below = &r->s[r->rank + 1];
    r->sends[r->nsend++] = (wdpm_halo_op){r->rank + 1, hi - below->up, below->up}; /* my last rows are its upper halo */
    r->recvs[r->nrecv++] = (wdpm_halo_op){r->rank + 1, hi, me->down};
    r->bottom_rows = me->rows - (hi - below->up);
  }
}

/* rank object without a transport yet */
static int rank_new(wdpm_rank **out, const wdpm_params *whole, int rank, int n, int k, int device) {
  wdpm_rank *r = (wdpm_rank *)calloc(1, sizeof *r);
  if (!r) return rb_fail("wdpm_rank: out of memory");
  r->p = *whole;
  r->p.device = device;
  r->p.slab_row0 = 0;
  r->p.slab_rows = 0;
  r->rank = rank; r->n = n; r->k = k; r->ncp = whole->ncols + 2;
  if (wdpm_partition(whole->nrows, n, k, whole->module, whole->module == WDPM_DRAIN ? whole->drainrow : -1, r->s)) {
    free(r);
    return rb_fail("wdpm_rank: the raster is too short for this many ranks and this exchange interval");
  }
  r->drain_owner = 0;
  for (int i = 0; i < n; i++)
    if (whole->drainrow >= r->s[i].own_lo && whole->drainrow <= r->s[i].own_hi) r->drain_owner = i;
  const char *ov = getenv("WDPM_OVERLAP");
  r->overlap = n > 1 && !(ov && atoi(ov) == 0);
  wdpm_params q = r->p;
  if (n > 1) { q.slab_row0 = r->s[rank].row0; q.slab_rows = r->s[rank].rows; }
  if (wdpm_create(&r->c, &q)) { free(r); return 1; }
  plan_exchange(r);
  *out = r;
  return 0;
}

int wdpm_rank_create(wdpm_rank **out, const wdpm_params *whole, int32_t rank, int32_t nranks, int32_t exchange_every,
                     int32_t halo, const void *rccl_id, const wdpm_host_transport *host) {
  if (!out || !whole || nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) return rb_fail("wdpm_rank_create: bad argument");
  int k = exchange_every < 1 ? 1 : exchange_every;
  wdpm_slab probe[MAXR];
  /* every rank shrinks the exchange interval the same way until every slab can serve its halos */
  while (k > 1 && wdpm_partition(whole->nrows, nranks, k, whole->module, whole->module == WDPM_DRAIN ? whole->drainrow : -1, probe)) k--;
  wdpm_rank *r = NULL;
  if (rank_new(&r, whole, rank, nranks, k, whole->device)) return 1;
  if (nranks > 1) {
    if (halo == WDPM_HALO_AUTO) halo = rccl_id ? WDPM_HALO_RCCL : WDPM_HALO_HOST;
    if (halo == WDPM_HALO_RCCL) {
      if (!rccl_id) { wdpm_rank_destroy(r); return rb_fail("wdpm_rank_create: RCCL halos need the id of wdpm_comm_unique_id"); }
      if (wdpm_comm_init_rank(r->c, nranks, rank, rccl_id)) { wdpm_rank_destroy(r); return 1; }
    } else if (halo == WDPM_HALO_HOST) {
      if (!host || !host->exchange || !host->allgather) { wdpm_rank_destroy(r); return rb_fail("wdpm_rank_create: host halos need the caller's transport"); }
      r->host = *host;
      for (int i = 0; i < 4; i++) {
        const wdpm_halo_op *op = i < 2 ? (i < r->nsend ? &r->sends[i] : NULL) : (i - 2 < r->nrecv ? &r->recvs[i - 2] : NULL);
        if (!op) continue;
        void *m = NULL;
        if (wdpm_host_alloc((size_t)(op->nrows > 0 ? op->nrows : 1) * r->ncp * sizeof(double), &m)) { wdpm_rank_destroy(r); return 1; }
        r->stage[i] = (double *)m;
      }
    } else {
      wdpm_rank_destroy(r);
      return rb_fail("wdpm_rank_create: ranks in separate processes exchange halos by RCCL or through the host");
    }
    r->halo = halo;
  }
  *out = r;
  return 0;
}

void wdpm_rank_destroy(wdpm_rank *r) {
  if (!r) return;
  for (int i = 0; i < 4; i++)
    if (r->stage[i]) wdpm_host_free(r->stage[i]);
  wdpm_destroy(r->c);
  free(r);
}

wdpm_ctx *wdpm_rank_ctx(wdpm_rank *r) { return r ? r->c : NULL; }

int wdpm_rank_slab(wdpm_rank *r, int32_t of_rank, wdpm_slab *out) {
  if (!r || !out || of_rank >= r->n) return rb_fail("wdpm_rank_slab: bad argument");
  *out = r->s[of_rank < 0 ? r->rank : of_rank];
  return 0;
}

int wdpm_rank_info(wdpm_rank *r, int32_t *halo, int32_t *exchange_every, int32_t *drain_owner) {
  if (!r) return rb_fail("wdpm_rank_info: null rank");
  if (halo) *halo = r->halo;
  if (exchange_every) *exchange_every = r->k;
  if (drain_owner) *drain_owner = r->drain_owner;
  return 0;
}

/* What kinds of depth the raster holds (a -0.0 anywhere, negative depths, odd values: WDPM_OPT_WATER_KINDS) decides which
 * stencil variant runs, and halo rows are written straight into device memory by the transport, past the library's own
 * upload scan: every rank takes the OR over all ranks */
static int agree_on_options(wdpm_rank *r) {
  int64_t v = 0;
  if (wdpm_get_option(r->c, WDPM_OPT_WATER_KINDS, &v)) return 1;
  double mine = (double)v, all[MAXR];
  if (rank_allgather(r, &mine, 1, all)) return 1;
  for (int q = 0; q < r->n; q++) v |= (int64_t)all[q];
  return wdpm_set_option(r->c, WDPM_OPT_WATER_KINDS, v);
}

int wdpm_rank_upload(wdpm_rank *r, const double *slab_dem, const double *slab_water) {
  if (wdpm_upload(r->c, slab_dem, slab_water)) return 1;
  r->since = 0;
  return agree_on_options(r);
}

int wdpm_rank_upload_global(wdpm_rank *r, const double *bigdem, const double *bigwater) {
  if (!bigdem || !bigwater) return rb_fail("wdpm_rank_upload_global: null array");
  const size_t off = r->n > 1 ? (size_t)r->s[r->rank].row0 * r->ncp : 0;
  return wdpm_rank_upload(r, bigdem + off, bigwater + off);
}

int wdpm_rank_set_totaldrain(wdpm_rank *r, double v) { return wdpm_set_totaldrain(r->c, v); }

int wdpm_rank_get_totaldrain(wdpm_rank *r, double *v) {
  if (r->n == 1) return wdpm_get_totaldrain(r->c, v);
  double mine = 0.0, all[MAXR];
  if (r->rank == r->drain_owner && wdpm_get_totaldrain(r->c, &mine)) return 1;   /* the rank that owns the outlet's rows */
  if (rank_allgather(r, &mine, 1, all)) return 1;
  *v = all[r->drain_owner];
  return 0;
}

/* ---- halo refresh ---------------------------------------------------------------------------- */
static int exchange_peer_all(struct wdpm_group *g) {
  for (int i = 0; i + 1 < g->n; i++) {
    const wdpm_slab *a = &g->r[i]->s[i], *b = &g->r[i]->s[i + 1];
    /* a's lower halo <- b's first owned rows */
    if (wdpm_copy_rows(g->r[i]->c, a->own_hi + 1 - a->row0, g->r[i + 1]->c, b->own_lo - b->row0, a->down)) return 1;
    /* b's upper halo <- a's last owned rows */
    if (wdpm_copy_rows(g->r[i + 1]->c, 0, g->r[i]->c, a->own_hi + 1 - b->up - a->row0, b->up)) return 1;
  }
  return 0;
}

static int exchange_now(wdpm_rank *r);

int wdpm_rank_exchange(wdpm_rank *r) {
  const double t0 = now_s();
  const int rc = exchange_now(r);
  r->exchange_s += now_s() - t0;
  return rc;
}

static int exchange_now(wdpm_rank *r) {
  r->since = 0;
  if (r->n == 1) return 0;
  if (r->halo == WDPM_HALO_RCCL) return wdpm_comm_exchange(r->c, r->nsend, r->sends, r->nrecv, r->recvs);
  if (r->halo == WDPM_HALO_PEER) {
    /* every rank thread has queued its iterations; one of them chains the copies between the
     * contexts' streams with events (wdpm_copy_rows), the others wait for it to have done so */
    struct wdpm_group *g = r->grp;
    if (grp_barrier(g)) return 1;
    if (r->rank == 0 && exchange_peer_all(g)) grp_fail(g, wdpm_last_error());
    return grp_barrier(g);
  }
  /* HOST: rows down to staging, the caller's transport, rows up again */
  int32_t is_send[4], peer[4];
  double *buf[4];
  int64_t count[4];
  int m = 0;
  for (int i = 0; i < r->nsend; i++) {
    if (wdpm_download_rows(r->c, r->sends[i].row, r->sends[i].nrows, r->stage[i])) return 1;
    is_send[m] = 1; peer[m] = r->sends[i].peer; buf[m] = r->stage[i]; count[m] = (int64_t)r->sends[i].nrows * r->ncp; m++;
  }
  for (int i = 0; i < r->nrecv; i++) {
    is_send[m] = 0; peer[m] = r->recvs[i].peer; buf[m] = r->stage[2 + i]; count[m] = (int64_t)r->recvs[i].nrows * r->ncp; m++;
  }
  if (r->host.exchange(r->host.user, m, is_send, peer, buf, count)) return rb_fail("the caller's halo exchange failed");
  for (int i = 0; i < r->nrecv; i++)
    if (r->recvs[i].nrows > 0 && wdpm_upload_rows(r->c, r->recvs[i].row, r->recvs[i].nrows, r->stage[2 + i])) return 1;
  return 0;
}

/* ---- the block loop pieces (WDPMCL.c:1055-1125, 1239-1268) ------------------------------------- */
int wdpm_rank_begin_block(wdpm_rank *r, double thres) { return wdpm_begin_block(r->c, thres); }

static int rank_iterate(wdpm_rank *r, int32_t n_iter, int block_end);
static void owned_rows(const wdpm_rank *r, int *lo, int *hi);

int wdpm_rank_iterate(wdpm_rank *r, int32_t n_iter) { return rank_iterate(r, n_iter, 0); }

/* block_end: a wdpm_rank_max_diff follows - the last launch call may fold the reduction over the owned rows */
static int rank_iterate(wdpm_rank *r, int32_t n_iter, int block_end) {
  if (n_iter < 0) return rb_fail("wdpm_rank_iterate: negative iteration count");
  int done = 0;
  while (done < n_iter) {
    int room = r->k - r->since;
    if (r->n > 1 && room <= 0) {
      if (wdpm_rank_exchange(r)) return 1;
      room = r->k;
    }
    const int step = r->n == 1 ? n_iter - done : (room < n_iter - done ? room : n_iter - done);
    const double t0 = now_s();
    if (block_end && done + step == n_iter) {
      int lo, hi;
      owned_rows(r, &lo, &hi);
      if (wdpm_expect_max_diff(r->c, lo, hi)) return 1;
    }
    if (r->overlap && (step == room || (block_end && done + step == n_iter))) {
      /* this step ends a group of k - or the block, whose max diff starts from refreshed halos - and an exchange
       * follows: produce the rows the neighbours need first, so that the transfer overlaps the interior rows of
       * the last iteration */
      if (wdpm_iterate_overlapped(r->c, step, r->top_rows, r->bottom_rows)) return 1;
    } else if (wdpm_iterate(r->c, step)) {
      return 1;
    }
    r->enqueue_s += now_s() - t0;
    done += step;
    r->since += step;
  }
  r->iters += n_iter;
  return 0;
}

int wdpm_rank_max_diff(wdpm_rank *r, double *max_diff) {
  if (!max_diff) return rb_fail("wdpm_rank_max_diff: null argument");
  if (r->n > 1 && r->since && wdpm_rank_exchange(r)) return 1;     /* blocks start from exact halos (the flush sees them) */
  const wdpm_slab *me = &r->s[r->rank];
  const int lo = r->n > 1 ? me->own_lo - me->row0 : 0;
  const int hi = r->n > 1 ? me->own_hi - me->row0 + 1 : r->p.nrows + 2;
  double mine = 0.0, all[MAXR];
  if (wdpm_max_diff(r->c, lo, hi, &mine)) return 1;
  if (rank_allgather(r, &mine, 1, all)) return 1;
  double m = all[0];
  for (int q = 1; q < r->n; q++)
    if (all[q] > m) m = all[q];                                    /* WDPMCL.c:1250: `>` decides, a NaN never wins */
  *max_diff = m;
  return 0;
}

int wdpm_rank_run_block(wdpm_rank *r, int32_t n_iter, double thres, double *max_diff) {
  if (wdpm_rank_begin_block(r, thres)) return 1;
  if (rank_iterate(r, n_iter, 1)) return 1;
  return wdpm_rank_max_diff(r, max_diff);
}

/* WDPMCL.c:1257-1268 across ranks: |totaldrain - olddrain| of the outlet's owner; the row-major volume
 * sum chained rank to rank, so that its rounding equals the single-raster sum */
int wdpm_rank_drain_stats(wdpm_rank *r, double *diffdrain, double *final_sum) {
  if (r->n == 1) return wdpm_drain_stats(r->c, diffdrain, final_sum);
  double all[MAXR];
  if (diffdrain) {
    double mine = 0.0;
    if (r->rank == r->drain_owner && wdpm_drain_stats(r->c, &mine, NULL)) return 1;
    if (rank_allgather(r, &mine, 1, all)) return 1;
    *diffdrain = all[r->drain_owner];
  }
  if (final_sum) {
    const wdpm_slab *me = &r->s[r->rank];
    double run = 0.0;
    for (int q = 0; q < r->n; q++) {
      double part = 0.0;
      if (q == r->rank && wdpm_volume_partial(r->c, me->own_lo - me->row0, me->own_hi - me->row0 + 1, run, &part)) return 1;
      if (rank_allgather(r, &part, 1, all)) return 1;
      run = all[q];
    }
    *final_sum = run;
  }
  return 0;
}

int wdpm_rank_download_owned(wdpm_rank *r, double *dst) {
  if (r->n == 1) return wdpm_download_water(r->c, dst);
  const wdpm_slab *me = &r->s[r->rank];
  return wdpm_download_rows(r->c, me->own_lo - me->row0, me->own_hi - me->own_lo + 1, dst);
}

/* ---- set-up and final statistics across ranks ---------------------------------- */
static void owned_rows(const wdpm_rank *r, int *lo, int *hi) {     /* slab-local [lo, hi) */
  const wdpm_slab *me = &r->s[r->rank];
  *lo = r->n > 1 ? me->own_lo - me->row0 : 0;
  *hi = r->n > 1 ? me->own_hi - me->row0 + 1 : r->p.nrows + 2;
}

static int rank_upload_unpadded(wdpm_rank *r, const double *dem, const double *water, const wdpm_setup *su) {
  if (wdpm_upload_unpadded(r->c, dem, water, su)) return 1;
  r->since = 0;
  return agree_on_options(r);
}

static int rank_count_stats(wdpm_rank *r, int64_t *valid, int64_t *wet, double *maxv) {
  int lo, hi;
  owned_rows(r, &lo, &hi);
  int64_t nv = 0, nw = 0;
  double m = 0.0;
  if (wdpm_count_stats(r->c, lo, hi, &nv, &nw, &m)) return 1;
  double mine[3] = {(double)nv, (double)nw, m}, all[MAXR * 3];      /* counts < 2^53: exact in a double */
  if (rank_allgather(r, mine, 3, all)) return 1;
  nv = nw = 0;
  m = all[2];
  for (int q = 0; q < r->n; q++) {
    nv += (int64_t)all[3 * q];
    nw += (int64_t)all[3 * q + 1];
    if (all[3 * q + 2] > m) m = all[3 * q + 2];
  }
  if (valid) *valid = nv;
  if (wet) *wet = nw;
  if (maxv) *maxv = m;
  return 0;
}

/* WDPMCL.c:1005-1017 over the whole raster: the smallest value wins, among equal values the lowest rank
 * (= the first rows), within a rank the first cell in row-major order */
static int rank_find_drain(wdpm_rank *r, double *mindem, int32_t *drainrow, int32_t *draincol) {
  int lo, hi;
  owned_rows(r, &lo, &hi);
  double md = 0.0;
  int32_t row = -1, col = -1;
  if (wdpm_find_drain(r->c, lo, hi, &md, &row, &col)) return 1;
  const wdpm_slab *me = &r->s[r->rank];
  double mine[3] = {md, row < 0 ? -1.0 : (double)(row + (r->n > 1 ? me->row0 : 0)), (double)col}, all[MAXR * 3];
  if (rank_allgather(r, mine, 3, all)) return 1;
  int best = -1;
  for (int q = 0; q < r->n; q++)
    if (all[3 * q + 1] >= 0 && (best < 0 || all[3 * q] < all[3 * best])) best = q;
  if (best < 0) { *mindem = 0.0; *drainrow = 0; *draincol = 0; return 0; }   /* no cell with dem > 0: the reference keeps 0,0 */
  *mindem = all[3 * best];
  *drainrow = (int32_t)all[3 * best + 1];
  *draincol = (int32_t)all[3 * best + 2];
  return 0;
}

/* 0: set; 2: the outlet is closer than three rows to a boundary of this partition (nothing changed) */
static int rank_set_drain(wdpm_rank *r, int32_t drainrow, int32_t draincol) {
  for (int i = 1; i < r->n; i++) {
    const int b = r->s[i].own_lo;
    if (b > drainrow - 3 && b < drainrow + 4) return 2;
  }
  r->p.drainrow = drainrow;
  r->p.draincol = draincol;
  r->drain_owner = 0;
  for (int i = 0; i < r->n; i++)
    if (drainrow >= r->s[i].own_lo && drainrow <= r->s[i].own_hi) r->drain_owner = i;
  return wdpm_set_drain(r->c, drainrow, draincol);
}

static int rank_get_cell(wdpm_rank *r, int row, int col, double *water, double *dem) {
  int owner = 0;
  for (int i = 0; i < r->n; i++)
    if (row >= r->s[i].own_lo && row <= r->s[i].own_hi) owner = i;
  double mine[2] = {0.0, 0.0}, all[MAXR * 2];
  if (r->rank == owner && wdpm_get_cell(r->c, row - (r->n > 1 ? r->s[owner].row0 : 0), col, &mine[0], &mine[1])) return 1;
  if (rank_allgather(r, mine, 2, all)) return 1;
  if (water) *water = all[2 * owner];
  if (dem) *dem = all[2 * owner + 1];
  return 0;
}

/* this rank's owned FILE rows into the whole un-padded raster at `water` */
static int rank_download_unpadded(wdpm_rank *r, int mask, double *water) {
  const wdpm_slab *me = &r->s[r->rank];
  int f0 = (r->n > 1 ? me->own_lo : 0) - 1, f1 = (r->n > 1 ? me->own_hi : r->p.nrows + 1) - 1;   /* file rows of the owned padded rows */
  if (f0 < 0) f0 = 0;
  if (f1 > r->p.nrows - 1) f1 = r->p.nrows - 1;
  if (f1 < f0) return 0;
  return wdpm_download_unpadded(r->c, f0, f1 - f0 + 1, mask, water + (size_t)f0 * r->p.ncols);
}

/* ---- the ranks of one process, one host thread each ------------------------------------------- */
enum { CMD_UPLOAD = 1, CMD_RUN_BLOCK, CMD_DRAIN_STATS, CMD_GET_TD, CMD_SET_TD, CMD_DOWNLOAD, CMD_UPLOAD_UNPADDED,
       CMD_COUNT_STATS, CMD_FIND_DRAIN, CMD_SET_DRAIN, CMD_GET_CELL, CMD_DOWNLOAD_UNPADDED };

static int group_execute(struct wdpm_group *g, int i) {
  wdpm_rank *r = g->r[i];
  switch (g->cmd) {
    case CMD_UPLOAD: return wdpm_rank_upload_global(r, g->a_dem, g->a_water);
    case CMD_RUN_BLOCK: return wdpm_rank_run_block(r, g->a_iter, g->a_thres, &g->res_a[i]);
    case CMD_DRAIN_STATS:
      return wdpm_rank_drain_stats(r, g->a_want_diff ? &g->res_a[i] : NULL, g->a_want_sum ? &g->res_b[i] : NULL);
    case CMD_GET_TD: return wdpm_rank_get_totaldrain(r, &g->res_a[i]);
    case CMD_SET_TD: return wdpm_rank_set_totaldrain(r, g->a_value);
    case CMD_DOWNLOAD:
      return wdpm_rank_download_owned(r, g->a_out + (g->n > 1 ? (size_t)r->s[i].own_lo * r->ncp : 0));
    case CMD_UPLOAD_UNPADDED: return rank_upload_unpadded(r, g->a_dem, g->a_water, g->a_setup);
    case CMD_COUNT_STATS: return rank_count_stats(r, &g->res_i[i], &g->res_j[i], &g->res_a[i]);
    case CMD_FIND_DRAIN: {
      int32_t dr = 0, dc = 0;
      if (rank_find_drain(r, &g->res_a[i], &dr, &dc)) return 1;
      g->res_i[i] = dr; g->res_j[i] = dc;
      return 0;
    }
    case CMD_SET_DRAIN: {
      const int rc = rank_set_drain(r, g->a_row, g->a_col);
      g->res_i[i] = rc;
      return rc == 2 ? 0 : rc;
    }
    case CMD_GET_CELL: return rank_get_cell(r, g->a_row, g->a_col, &g->res_a[i], &g->res_b[i]);
    case CMD_DOWNLOAD_UNPADDED: return rank_download_unpadded(r, g->a_mask, g->a_out);
  }
  return rb_fail("wdpm_group: unknown command");
}

typedef struct { struct wdpm_group *g; int i; } worker_arg;

static void *worker_main(void *arg) {
  worker_arg *wa = (worker_arg *)arg;
  struct wdpm_group *g = wa->g;
  const int i = wa->i;
  free(wa);
  unsigned seen = 0;
  pthread_mutex_lock(&g->mu);
  for (;;) {
    while (g->seq == seen && !g->quit) pthread_cond_wait(&g->cv_cmd, &g->mu);
    if (g->quit) break;
    seen = g->seq;
    pthread_mutex_unlock(&g->mu);
    if (group_execute(g, i)) grp_fail(g, wdpm_last_error());
    pthread_mutex_lock(&g->mu);
    g->ndone++;
    pthread_cond_broadcast(&g->cv_done);
  }
  pthread_mutex_unlock(&g->mu);
  return NULL;
}

/* run the command on every rank (each on its own thread) and wait for all of them */
static int group_dispatch(struct wdpm_group *g, int cmd) {
  if (g->failed) { wdpm_set_last_error(g->err); return 1; }
  g->cmd = cmd;
  if (g->n == 1) return group_execute(g, 0);
  pthread_mutex_lock(&g->mu);
  g->ndone = 0;
  g->seq++;
  pthread_cond_broadcast(&g->cv_cmd);
  while (g->ndone < g->n) pthread_cond_wait(&g->cv_done, &g->mu);
  const int f = g->failed;
  pthread_mutex_unlock(&g->mu);
  if (f) wdpm_set_last_error(g->err);
  return f;
}

int wdpm_group_create(wdpm_group **out, const wdpm_params *p, int32_t ndev, const int32_t *devices,
                      int32_t exchange_every) {
  if (!out || !p || ndev < 1 || ndev > MAXR || !devices) return rb_fail("wdpm_group_create: bad argument");
  struct wdpm_group *g = (struct wdpm_group *)calloc(1, sizeof *g);
  if (!g) return rb_fail("wdpm_group_create: out of memory");
  pthread_mutex_init(&g->mu, NULL);
  pthread_cond_init(&g->cv_cmd, NULL);
  pthread_cond_init(&g->cv_done, NULL);
  pthread_cond_init(&g->cv_bar, NULL);
  int n = ndev, k = exchange_every < 1 ? 1 : exchange_every;
  wdpm_slab probe[MAXR];
  const int dr = p->module == WDPM_DRAIN ? p->drainrow : -1;
  /* shrink the exchange interval, then the device count, until every slab can serve its halos */
  while (wdpm_partition(p->nrows, n, k, p->module, dr, probe) != 0) {
    if (k > 1) k--;
    else if (n > 1) { n--; k = exchange_every < 1 ? 1 : exchange_every; }
    else { wdpm_group_destroy(g); return rb_fail("wdpm_group_create: cannot partition the raster"); }
  }
  for (int i = 0; i < n; i++) {
    if (rank_new(&g->r[i], p, i, n, k, devices[i])) { wdpm_group_destroy(g); return 1; }
    g->n = i + 1;
    g->r[i]->grp = g;
  }
  g->n = n;
  if (n > 1) {
    /* halo transport: RCCL over the devices (ncclCommInitAll) unless the environment or the device list says otherwise */
    const char *e = getenv("WDPM_HALO");
    int want = WDPM_HALO_AUTO;
    if (e && !strcmp(e, "rccl")) want = WDPM_HALO_RCCL;
    else if (e && !strcmp(e, "peer")) want = WDPM_HALO_PEER;
    int distinct = 1;
    for (int i = 0; i < n; i++)
      for (int j = 0; j < i; j++) distinct &= devices[i] != devices[j];
    int halo = WDPM_HALO_PEER;
    if (want == WDPM_HALO_RCCL || (want == WDPM_HALO_AUTO && distinct && wdpm_comm_available())) {
      wdpm_ctx *ctxs[MAXR];
      for (int i = 0; i < n; i++) ctxs[i] = g->r[i]->c;
      if (wdpm_comm_init_all(ctxs, n) == 0) {
        halo = WDPM_HALO_RCCL;
      } else if (want == WDPM_HALO_RCCL) {
        wdpm_group_destroy(g);
        return 1;
      } else {
        fprintf(stderr, "wdpm: RCCL halos unavailable (%s); using peer copies\n", wdpm_last_error());
      }
    }
    if (halo == WDPM_HALO_PEER)
      for (int i = 0; i + 1 < n; i++)
        if (wdpm_enable_peer_access(g->r[i]->c, g->r[i + 1]->c)) { wdpm_group_destroy(g); return 1; }
    g->halo = halo;
    for (int i = 0; i < n; i++) g->r[i]->halo = halo;
    for (int i = 0; i < n; i++) {
      worker_arg *wa = (worker_arg *)malloc(sizeof *wa);
      if (!wa) { wdpm_group_destroy(g); return rb_fail("wdpm_group_create: out of memory"); }
      wa->g = g; wa->i = i;
      if (pthread_create(&g->th[i], NULL, worker_main, wa) != 0) { free(wa); wdpm_group_destroy(g); return rb_fail("wdpm_group_create: cannot start a rank thread"); }
      g->nthreads = i + 1;
    }
  }
  *out = g;
  return 0;
}

void wdpm_group_destroy(wdpm_group *g) {
  if (!g) return;
  pthread_mutex_lock(&g->mu);
  g->quit = 1;
  pthread_cond_broadcast(&g->cv_cmd);
  pthread_mutex_unlock(&g->mu);
  for (int i = 0; i < g->nthreads; i++) pthread_join(g->th[i], NULL);
  for (int i = 0; i < MAXR; i++)
    if (g->r[i]) wdpm_rank_destroy(g->r[i]);
  pthread_cond_destroy(&g->cv_cmd);
  pthread_cond_destroy(&g->cv_done);
  pthread_cond_destroy(&g->cv_bar);
  pthread_mutex_destroy(&g->mu);
  free(g);
}

int wdpm_group_size(wdpm_group *g) { return g->n; }
int wdpm_group_halo(wdpm_group *g) { return g->halo; }
wdpm_rank *wdpm_group_rank(wdpm_group *g, int32_t i) { return g && i >= 0 && i < g->n ? g->r[i] : NULL; }

int wdpm_group_upload(wdpm_group *g, const double *bigdem, const double *bigwater) {
  g->a_dem = bigdem; g->a_water = bigwater;
  return group_dispatch(g, CMD_UPLOAD);
}

int wdpm_group_run_block(wdpm_group *g, int32_t n_iter, double thres, double *max_diff) {
  g->a_iter = n_iter; g->a_thres = thres;
  if (group_dispatch(g, CMD_RUN_BLOCK)) return 1;
  if (max_diff) *max_diff = g->res_a[0];
  return 0;
}

int wdpm_group_download_water(wdpm_group *g, double *bigwater) {
  if (!bigwater) return rb_fail("wdpm_group_download_water: null array");
  g->a_out = bigwater;
  return group_dispatch(g, CMD_DOWNLOAD);
}

int wdpm_group_set_totaldrain(wdpm_group *g, double v) {
  g->a_value = v;
  return group_dispatch(g, CMD_SET_TD);
}

int wdpm_group_get_totaldrain(wdpm_group *g, double *v) {
  if (group_dispatch(g, CMD_GET_TD)) return 1;
  *v = g->res_a[0];
  return 0;
}

int wdpm_group_drain_stats(wdpm_group *g, double *diffdrain, double *final_sum) {
  g->a_want_diff = diffdrain != NULL; g->a_want_sum = final_sum != NULL;
  if (group_dispatch(g, CMD_DRAIN_STATS)) return 1;
  if (diffdrain) *diffdrain = g->res_a[0];
  if (final_sum) *final_sum = g->res_b[0];
  return 0;
}

int wdpm_group_enqueue_stats(wdpm_group *g, double *seconds, double *exchange_seconds, int64_t *iterations) {
  double s = 0.0, e = 0.0;
  for (int i = 0; i < g->n; i++) { s += g->r[i]->enqueue_s; e += g->r[i]->exchange_s; }
  if (seconds) *seconds = s;
  if (exchange_seconds) *exchange_seconds = e;
  if (iterations) *iterations = g->r[0]->iters;
  return 0;
}

int wdpm_group_upload_unpadded(wdpm_group *g, const double *dem, const double *water, const wdpm_setup *setup) {
  if (!dem || !setup) return rb_fail("wdpm_group_upload_unpadded: null argument");
  g->a_dem = dem; g->a_water = water; g->a_setup = setup;
  return group_dispatch(g, CMD_UPLOAD_UNPADDED);
}

int wdpm_group_count_stats(wdpm_group *g, int64_t *valid, int64_t *wet, double *maxv) {
  if (group_dispatch(g, CMD_COUNT_STATS)) return 1;
  if (valid) *valid = g->res_i[0];
  if (wet) *wet = g->res_j[0];
  if (maxv) *maxv = g->res_a[0];
  return 0;
}

int wdpm_group_find_drain(wdpm_group *g, double *mindem, int32_t *drainrow, int32_t *draincol) {
  if (group_dispatch(g, CMD_FIND_DRAIN)) return 1;
  if (mindem) *mindem = g->res_a[0];
  if (drainrow) *drainrow = (int32_t)g->res_i[0];
  if (draincol) *draincol = (int32_t)g->res_j[0];
  return 0;
}

int wdpm_group_set_drain(wdpm_group *g, int32_t drainrow, int32_t draincol) {
  g->a_row = drainrow; g->a_col = draincol;
  if (group_dispatch(g, CMD_SET_DRAIN)) return 1;
  return (int)g->res_i[0];
}

int wdpm_group_get_cell(wdpm_group *g, int32_t row, int32_t col, double *water, double *dem) {
  g->a_row = row; g->a_col = col;
  if (group_dispatch(g, CMD_GET_CELL)) return 1;
  if (water) *water = g->res_a[0];
  if (dem) *dem = g->res_b[0];
  return 0;
}

int wdpm_group_download_unpadded(wdpm_group *g, int32_t mask_missing, double *water) {
  if (!water) return rb_fail("wdpm_group_download_unpadded: null array");
  g->a_mask = mask_missing; g->a_out = water;
  return group_dispatch(g, CMD_DOWNLOAD_UNPADDED);
}
