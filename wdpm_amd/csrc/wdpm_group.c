/*
 * wdpm_group.c — one raster spread over several contexts (GPUs) of one process.
 *
 * Back-end agnostic C written against the per-context ABI of include/wdpm.h (it is linked into the
 * HIP library and, for CPU tests of the WDPMCL plumbing, into the oracle library).  Same
 * decomposition and halo rule as wdpm_amd/rowblock.py, whose docstring has the derivation:
 * boundaries at rows = 2 (mod 3), 3k-1 halo rows above and 6k-2 below buy k iterations without
 * communication; tests/test_rowblock.py proves the rule with a dependency simulation.
 * The reference is single-device; this is what lets the WDPMCL drop-in use the
 * GPUs of a node (WDPM_GPUS=N).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/wdpm.h"

#define MAXDEV 64

typedef struct {
  int own_lo, own_hi;   /* owned padded rows (global, inclusive) */
  int row0, rows;       /* slab held: first global row, row count */
  int up, down;         /* halo rows actually held above / below */
} slab_t;

struct wdpm_group {
  wdpm_params p;
  int n, k, since;
  int ncp;
  slab_t s[MAXDEV];
  wdpm_ctx *c[MAXDEV];
  int drain_owner;
};

/* returns 0 when every slab is at least as tall as the halos it must serve */
static int partition(int nrows, int n, int k, slab_t *s) {
  const int P = nrows + 2, up = 3 * k - 1, down = 6 * k - 2;
  int bounds[MAXDEV + 1];
  bounds[0] = 0;
  for (int g = 1; g < n; g++) {
    int b = (int)((long long)P * g / n);
    b -= ((b - 2) % 3 + 3) % 3;           /* boundaries = 2 (mod 3) */
    bounds[g] = b;
  }
  bounds[n] = P;
  for (int g = 0; g < n; g++) {
    const int lo = bounds[g], hi = bounds[g + 1] - 1;
    if (hi < lo) return 1;
    if (n > 1 && hi - lo + 1 < (up > down ? up : down)) return 1;
    const int r0 = g > 0 ? (lo - up > 0 ? lo - up : 0) : 0;
    const int r1 = g < n - 1 ? (hi + down < P - 1 ? hi + down : P - 1) : P - 1;
    s[g].own_lo = lo; s[g].own_hi = hi; s[g].row0 = r0; s[g].rows = r1 - r0 + 1;
    s[g].up = lo - r0; s[g].down = r1 - hi;
  }
  return 0;
}

int wdpm_group_create(wdpm_group **out, const wdpm_params *p, int32_t ndev, const int32_t *devices,
                      int32_t exchange_every) {
  if (!out || !p || ndev < 1 || ndev > MAXDEV || !devices) return 1;
  wdpm_group *g = (wdpm_group *)calloc(1, sizeof *g);
  if (!g) return 1;
  g->p = *p;
  g->ncp = p->ncols + 2;
  int n = ndev, k = exchange_every < 1 ? 1 : exchange_every;
  /* shrink the exchange interval, then the device count, until every slab can serve its halos */
  while (partition(p->nrows, n, k, g->s) != 0) {
    if (k > 1) k--;
    else if (n > 1) { n--; k = exchange_every < 1 ? 1 : exchange_every; }
    else { free(g); return 1; }
  }
  g->n = n; g->k = k; g->since = 0; g->drain_owner = 0;
  for (int i = 0; i < n; i++) {
    wdpm_params q = *p;
    q.device = devices[i];
    q.slab_row0 = n > 1 ? g->s[i].row0 : 0;
    q.slab_rows = n > 1 ? g->s[i].rows : 0;
    if (wdpm_create(&g->c[i], &q) != 0) { wdpm_group_destroy(g); return 1; }
    if (p->drainrow >= g->s[i].own_lo && p->drainrow <= g->s[i].own_hi) g->drain_owner = i;
  }
  *out = g;
  return 0;
}

void wdpm_group_destroy(wdpm_group *g) {
  if (!g) return;
  for (int i = 0; i < g->n; i++) wdpm_destroy(g->c[i]);
  free(g);
}

int wdpm_group_size(wdpm_group *g) { return g->n; }

int wdpm_group_upload(wdpm_group *g, const double *bigdem, const double *bigwater) {
  int64_t any = 0;
  for (int i = 0; i < g->n; i++) {
    const size_t off = (size_t)g->s[i].row0 * g->ncp;
    if (wdpm_upload(g->c[i], bigdem + off, bigwater + off)) return 1;
    int64_t v = 0;
    if (wdpm_get_option(g->c[i], WDPM_OPT_SIGNED_ZERO_SAFE, &v)) return 1;
    any |= v;
  }
  for (int i = 0; i < g->n; i++)   /* a -0.0 depth anywhere: every device keeps the sign of zero */
    if (wdpm_set_option(g->c[i], WDPM_OPT_SIGNED_ZERO_SAFE, any)) return 1;
  g->since = 0;
  return 0;
}

/* refresh every halo from the neighbour that owns those rows */
static int exchange(wdpm_group *g) {
  for (int i = 0; i + 1 < g->n; i++) {
    const slab_t *a = &g->s[i], *b = &g->s[i + 1];
    /* a's lower halo <- b's first owned rows */
    if (wdpm_copy_rows(g->c[i], a->own_hi + 1 - a->row0, g->c[i + 1], b->own_lo - b->row0, a->down)) return 1;
    /* b's upper halo <- a's last owned rows */
    if (wdpm_copy_rows(g->c[i + 1], 0, g->c[i], a->own_hi + 1 - b->up - a->row0, b->up)) return 1;
  }
  g->since = 0;
  return 0;
}

static int iterate(wdpm_group *g, int n_iter) {
  int done = 0;
  while (done < n_iter) {
    int room = g->k - g->since;
    if (room <= 0) {
      if (g->n > 1 && exchange(g)) return 1;
      g->since = 0;
      room = g->k;
    }
    const int step = room < n_iter - done ? room : n_iter - done;
    for (int i = 0; i < g->n; i++)          /* asynchronous on each device's stream */
      if (wdpm_iterate(g->c[i], step)) return 1;
    done += step;
    g->since += step;
  }
  return 0;
}

int wdpm_group_run_block(wdpm_group *g, int32_t n_iter, double thres, double *max_diff) {
  for (int i = 0; i < g->n; i++)
    if (wdpm_begin_block(g->c[i], thres)) return 1;
  if (iterate(g, n_iter)) return 1;
  if (g->n > 1 && g->since && exchange(g)) return 1;
  double m = 0.0;
  for (int i = 0; i < g->n; i++) {
    const slab_t *s = &g->s[i];
    double v;
    const int lo = g->n > 1 ? s->own_lo - s->row0 : 0;
    const int hi = g->n > 1 ? s->own_hi - s->row0 + 1 : g->p.nrows + 2;
    if (wdpm_max_diff(g->c[i], lo, hi, &v)) return 1;
    if (v > m) m = v;
  }
  *max_diff = m;
  return 0;
}

int wdpm_group_download_water(wdpm_group *g, double *bigwater) {
  if (g->n == 1) return wdpm_download_water(g->c[0], bigwater);
  for (int i = 0; i < g->n; i++) {
    const slab_t *s = &g->s[i];
    if (wdpm_download_rows(g->c[i], s->own_lo - s->row0, s->own_hi - s->own_lo + 1,
                           bigwater + (size_t)s->own_lo * g->ncp)) return 1;
  }
  return 0;
}

int wdpm_group_set_totaldrain(wdpm_group *g, double v) {
  for (int i = 0; i < g->n; i++)
    if (wdpm_set_totaldrain(g->c[i], v)) return 1;
  return 0;
}

int wdpm_group_get_totaldrain(wdpm_group *g, double *v) {
  return wdpm_get_totaldrain(g->c[g->drain_owner], v);   /* the device that owns the outlet row */
}

int wdpm_group_drain_stats(wdpm_group *g, double *diffdrain, double *final_sum) {
  if (g->n == 1) return wdpm_drain_stats(g->c[0], diffdrain, final_sum);
  if (diffdrain && wdpm_drain_stats(g->c[g->drain_owner], diffdrain, NULL)) return 1;
  if (final_sum) {
    double run = 0.0;   /* chained so the rounding equals the single-raster row-major sum */
    for (int i = 0; i < g->n; i++) {
      const slab_t *s = &g->s[i];
      if (wdpm_volume_partial(g->c[i], s->own_lo - s->row0, s->own_hi - s->row0 + 1, run, &run)) return 1;
    }
    *final_sum = run;
  }
  return 0;
}
