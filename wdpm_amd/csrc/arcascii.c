/*
 * arcascii.c — ArcASCII grid I/O for the WDPMCL drop-in (see arcascii.h).
 *
 * Reading: the whole file is pulled into memory and tokenised with strtod, which is what glibc's
 * fscanf("%lf") uses underneath, so every value parses to the same double as in the reference.
 * Writing: header lines and "%f " cells formatted exactly as the reference's write_gis
 * (src/WDPMCL.c:1538-1551) into a large buffer, flushed with few write calls.
 */
#include "arcascii.h"

#include <ctype.h>
#include <pthread.h>
#include <unistd.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- exact fast paths ------------------------------------------------------------------------
 * asc_format_f6: what snprintf(buf, n, "%f", x) writes, without the trailing NUL: the exact binary
 * value rounded to 6 decimals, ties to even (glibc in the default rounding mode).  Pure integer
 * arithmetic on the IEEE-754 fields; anything unusual (NaN, inf, |x| >= 2^63) goes to snprintf.
 * asc_parse_double: strtod for the plain decimal form [+-]digits[.digits][e[+-]digits] when the
 * result is exact by Clinger's fast path (<= 15 significant digits, |power of ten| <= 22: one
 * correctly rounded multiply or divide of two exact doubles); everything else goes to strtod. */
int asc_format_f6(double x, char *out) {
  uint64_t bits;
  memcpy(&bits, &x, sizeof bits);
  const int neg = (int)(bits >> 63);
  const int e = (int)((bits >> 52) & 0x7FF);
  uint64_t m = bits & 0xFFFFFFFFFFFFFull;
  if (e == 0x7FF || e >= 1023 + 63) return sprintf(out, "%f", x);
  uint64_t ip, q;
  if (e == 0 && m == 0) {
    ip = 0; q = 0;
  } else {
    int sh;                         /* value = m * 2^-sh */
    if (e == 0) sh = 1074; else { m |= 1ull << 52; sh = 1075 - e; }
    if (sh <= 0) { ip = m << (-sh); q = 0; }
    else if (sh >= 75) { ip = 0; q = 0; }                 /* < 2^-22: rounds to 0.000000 */
    else {
      uint64_t fr;
      if (sh >= 64) { ip = 0; fr = m; } else { ip = m >> sh; fr = m & ((1ull << sh) - 1); }
      const unsigned __int128 num = (unsigned __int128)fr * 1000000u;
      const unsigned __int128 one = (unsigned __int128)1 << sh;
      q = (uint64_t)(num >> sh);
      const unsigned __int128 rem = num & (one - 1), half = one >> 1;
      if (rem > half || (rem == half && (q & 1))) q++;
      if (q == 1000000u) { q = 0; ip++; }
    }
  }
  int n = 0;
  if (neg) out[n++] = '-';
  if (ip < 10) {
    out[n++] = (char)('0' + ip);
  } else if (ip < 10000) {               /* the usual case: depths and elevations in metres */
    const unsigned v = (unsigned)ip;
    if (v >= 1000) out[n++] = (char)('0' + v / 1000);
    if (v >= 100) out[n++] = (char)('0' + v / 100 % 10);
    out[n++] = (char)('0' + v / 10 % 10);
    out[n++] = (char)('0' + v % 10);
  } else {
    char tmp[24];
    int k = 0;
    do { tmp[k++] = (char)('0' + ip % 10); ip /= 10; } while (ip);
    while (k) out[n++] = tmp[--k];
  }
  out[n++] = '.';
  const unsigned qq = (unsigned)q, hi = qq / 1000, lo = qq % 1000;
  out[n + 0] = (char)('0' + hi / 100); out[n + 1] = (char)('0' + hi / 10 % 10); out[n + 2] = (char)('0' + hi % 10);
  out[n + 3] = (char)('0' + lo / 100); out[n + 4] = (char)('0' + lo / 10 % 10); out[n + 5] = (char)('0' + lo % 10);
  return n + 6;
}

static const double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                  1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

double asc_parse_double(const char *p, char **end) {
  const char *s = p;
  while (*s == ' ' || *s == '\n' || *s == '\t' || *s == '\r' || *s == '\f' || *s == '\v') s++;
  const char *start = s;
  int neg = 0;
  if (*s == '-') { neg = 1; s++; } else if (*s == '+') s++;
  uint64_t w = 0;
  int nd = 0, dropped = 0, exp10 = 0, any = 0;
  for (; *s >= '0' && *s <= '9'; s++) {
    any = 1;
    if (nd < 19) { if (w || *s != '0') { w = w * 10 + (uint64_t)(*s - '0'); nd++; } }
    else dropped = 1, exp10++;
  }
  if (*s == '.') {
    s++;
    for (; *s >= '0' && *s <= '9'; s++) {
      any = 1;
      if (nd < 19) { if (w || *s != '0') { w = w * 10 + (uint64_t)(*s - '0'); nd++; } exp10--; }
      else dropped = 1;
    }
  }
  if (!any) return strtod(p, end);                  /* "nan", "inf", junk, empty: strtod decides */
  if (*s == 'e' || *s == 'E') {
    const char *t = s + 1;
    int eneg = 0, ev = 0, ed = 0;
    if (*t == '-') { eneg = 1; t++; } else if (*t == '+') t++;
    for (; *t >= '0' && *t <= '9' && ev < 10000; t++) { ev = ev * 10 + (*t - '0'); ed = 1; }
    if (ed) { while (*t >= '0' && *t <= '9') t++; exp10 += eneg ? -ev : ev; s = t; }
  } else if (*s == 'x' || *s == 'X' || *s == 'p' || *s == 'P') {
    return strtod(p, end);                          /* hex float */
  }
  if (dropped || nd > 15 || exp10 < -22 || exp10 > 22) return strtod(start, end);
  double v = (double)w;                             /* exact: w < 10^15 < 2^53 */
  v = exp10 < 0 ? v / kPow10[-exp10] : v * kPow10[exp10];
  *end = (char *)s;
  return neg ? -v : v;
}

static char *slurp(const char *path, size_t *len) {
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  if (fseek(f, 0, SEEK_END) != 0) { fclose(f); return NULL; }
  long n = ftell(f);
  if (n < 0) { fclose(f); return NULL; }
  rewind(f);
  char *buf = (char *)malloc((size_t)n + 1);
  if (!buf) { fclose(f); return NULL; }
  size_t got = fread(buf, 1, (size_t)n, f);
  fclose(f);
  buf[got] = '\0';
  *len = got;
  return buf;
}

static const char *skip_ws(const char *p) {
  while (*p && isspace((unsigned char)*p)) p++;
  return p;
}

/* one "%30s %lf" pair; returns the position after it, or NULL when no keyword is left */
static const char *header_pair(const char *p, char *name, double *value) {
  p = skip_ws(p);
  if (!*p) return NULL;
  int n = 0;
  while (*p && !isspace((unsigned char)*p) && n < 30) name[n++] = *p++;
  name[n] = '\0';
  char *end;
  double v = strtod(p, &end);
  if (end != p) { *value = v; p = end; }
  return p;
}

int asc_read_header(const char *path, asc_header *h) {
  /* the six pairs sit in the first few hundred bytes: read 64 KiB, not the (multi-GB) file */
  FILE *f = fopen(path, "rb");
  if (!f) return 1;
  char *buf = (char *)malloc(65536 + 1);
  if (!buf) { fclose(f); return 1; }
  const size_t got = fread(buf, 1, 65536, f);
  fclose(f);
  buf[got] = '\0';
  memset(h, 0, sizeof *h);
  const char *p = buf;
  for (int i = 0; i < 6 && p; i++) p = header_pair(p, h->name[i], &h->value[i]);
  free(buf);
  return 0;
}

/* ---- threads: big rasters are parsed / formatted by several host threads (WDPM_IO_THREADS, default
 * = online cores up to 16; small rasters stay on one thread) ------------------------------------ */
static int io_threads(size_t cells) {
  const char *pm = getenv("WDPM_HOST_PAR_MIN");            /* cells below which the I/O stays on one thread (tests lower it) */
  if (cells < (pm ? (size_t)atoll(pm) : (size_t)1 << 20)) return 1;
  const char *e = getenv("WDPM_IO_THREADS");
  long n = e ? atol(e) : sysconf(_SC_NPROCESSORS_ONLN);
  if (n > 16) n = 16;
  return n < 1 ? 1 : (int)n;
}

static int is_ws(char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\f' || c == '\v'; }

typedef struct {
  const char *begin, *end;   /* text slice, begins at a token start (or whitespace) and ends at a token end */
  size_t ntok, first;        /* tokens in the slice; index of its first token in the grid */
  double *dst;
  size_t total;
  int stop;                  /* a token that is not a number was met: the reference stops reading there */
} parse_job;

static void *count_tokens(void *arg) {
  parse_job *j = (parse_job *)arg;
  size_t n = 0;
  int in = 0;
  for (const char *p = j->begin; p < j->end; p++) {
    const int w = is_ws(*p);
    if (!w && !in) n++;
    in = !w;
  }
  j->ntok = n;
  return NULL;
}

static void *parse_slice(void *arg) {
  parse_job *j = (parse_job *)arg;
  const char *p = j->begin;
  size_t k = j->first;
  while (k < j->total) {
    while (p < j->end && is_ws(*p)) p++;
    if (p >= j->end) break;
    char *e;
    const double v = asc_parse_double(p, &e);
    if (e == p) { j->stop = 1; break; }
    j->dst[k++] = v;
    p = e;
  }
  return NULL;
}

/* the text of a (possibly multi-GB) grid file: mapped read-only when the bytes after its end are
 * readable zeros (file size not a multiple of the page size: the parsers rely on a terminating
 * NUL), else read into memory */
typedef struct { char *p; size_t len, maplen; } text_t;
static int text_open(const char *path, text_t *t) {
  t->p = NULL; t->len = t->maplen = 0;
  const int fd = open(path, O_RDONLY);
  if (fd >= 0) {
    struct stat st;
    const long page = sysconf(_SC_PAGESIZE);
    if (fstat(fd, &st) == 0 && st.st_size > (off_t)(1 << 20) && page > 0 && st.st_size % page != 0) {
      void *m = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
      if (m != MAP_FAILED) {
        (void)madvise(m, (size_t)st.st_size, MADV_WILLNEED);
        t->p = (char *)m; t->len = t->maplen = (size_t)st.st_size;
      }
    }
    close(fd);
  }
  if (!t->p) t->p = slurp(path, &t->len);
  return t->p ? 0 : 1;
}
static void text_close(text_t *t) {
  if (t->maplen) munmap(t->p, t->maplen); else free(t->p);
}

int asc_read_grid(const char *path, int nrows, int ncols, double *dst) {
  text_t text;
  if (text_open(path, &text)) return 1;
  const size_t len = text.len;
  char *buf = text.p;
  const char *p = buf;
  char name[32];
  double dummy;
  for (int i = 0; i < 6 && p; i++) p = header_pair(p, name, &dummy);
  const size_t total = (size_t)nrows * ncols;
  if (!p) { text_close(&text); return 0; }
  const char *const textend = buf + len;
  int T = io_threads(total);
  parse_job jobs[16];
  pthread_t th[16];
  /* cut the text into T slices at whitespace */
  const char *cut = p;
  for (int t = 0; t < T; t++) {
    jobs[t].begin = cut;
    const char *e = t == T - 1 ? textend : p + (size_t)(textend - p) / T * (t + 1);
    if (e < cut) e = cut;
    while (e < textend && !is_ws(*e)) e++;
    jobs[t].end = e;
    jobs[t].dst = dst; jobs[t].total = total; jobs[t].stop = 0; jobs[t].ntok = 0; jobs[t].first = 0;
    cut = e;
  }
  if (T > 1) {
    for (int t = 0; t < T; t++) pthread_create(&th[t], NULL, count_tokens, &jobs[t]);
    for (int t = 0; t < T; t++) pthread_join(th[t], NULL);
    size_t acc = 0;
    for (int t = 0; t < T; t++) { jobs[t].first = acc; acc += jobs[t].ntok; }
    for (int t = 0; t < T; t++) pthread_create(&th[t], NULL, parse_slice, &jobs[t]);
    for (int t = 0; t < T; t++) pthread_join(th[t], NULL);
    /* a non-numeric token ends the reference's fscanf loop: nothing after it may have been stored.
     * Re-do the rare malformed file serially so the cells after the bad token keep their values. */
    int bad = 0;
    for (int t = 0; t < T; t++) bad |= jobs[t].stop;
    if (bad) {
      /* cells were pre-filled by the caller; we cannot restore them, so parse serially into place:
       * identical result for every cell before the bad token, and the caller-visible difference
       * (cells after it) is limited to files the reference itself reads as garbage. */
      T = 1;
    }
  }
  if (T == 1) {
    parse_job j = {p, textend, 0, 0, dst, total, 0};
    parse_slice(&j);
  }
  text_close(&text);
  return 0;
}

typedef struct {
  const double *src;
  int ncols, row0, row1;
  char *buf;
  size_t cap, len;
} fmt_job;

static void *format_rows(void *arg) {
  fmt_job *j = (fmt_job *)arg;
  /* worst case per cell: "%f" of a double below 2^63 is at most 28 characters + the blank; the
   * snprintf fallback for huge / non-finite values is bounded by 320 */
  size_t n = 0;
  for (int r = j->row0; r < j->row1; r++) {
    const double *row = j->src + (size_t)r * j->ncols;
    for (int c = 0; c < j->ncols; c++) {
      if (n + 340 > j->cap) {
        const size_t ncap = j->cap ? j->cap * 2 : (size_t)1 << 20;
        char *nb = (char *)realloc(j->buf, ncap);
        if (!nb) { j->len = (size_t)-1; return NULL; }
        j->buf = nb; j->cap = ncap;
      }
      n += (size_t)asc_format_f6(row[c], j->buf + n);
      j->buf[n++] = ' ';
    }
    j->buf[n++] = '\n';
  }
  j->len = n;
  return NULL;
}

typedef struct { FILE *f; fmt_job *jobs; int T, rc; } band_write;
static void *write_band(void *arg) {
  band_write *b = (band_write *)arg;
  for (int t = 0; t < b->T && !b->rc; t++) {
    if (b->jobs[t].len == (size_t)-1) b->rc = 1;
    else if (b->jobs[t].len && fwrite(b->jobs[t].buf, 1, b->jobs[t].len, b->f) != b->jobs[t].len) b->rc = 1;
  }
  return NULL;
}

int asc_write_grid(const char *path, const asc_header *h, int nrows, int ncols, const double *src) {
  FILE *f = fopen(path, "w");
  if (!f) return 1;
  static const size_t kBuf = 1u << 22;
  char *buf = (char *)malloc(kBuf + 512);
  if (!buf) { fclose(f); return 1; }
  size_t n = 0;
  n += (size_t)sprintf(buf + n, "%s %d\n", h->name[0], (int)h->value[0]);
  n += (size_t)sprintf(buf + n, "%s %d\n", h->name[1], (int)h->value[1]);
  n += (size_t)sprintf(buf + n, "%s %14.6f\n", h->name[2], h->value[2]);
  n += (size_t)sprintf(buf + n, "%s %14.6f\n", h->name[3], h->value[3]);
  n += (size_t)sprintf(buf + n, "%s %9.6f\n", h->name[4], h->value[4]);
  n += (size_t)sprintf(buf + n, "%s %14.6f\n", h->name[5], h->value[5]);
  int rc = 0;
  if (fwrite(buf, 1, n, f) != n) rc = 1;
  free(buf);
  const int T = io_threads((size_t)nrows * ncols);
  /* rows are formatted in bands; within a band each thread formats a contiguous run of rows into
   * its own buffer and the buffers are written out in order - by a writer thread, while the
   * formatters are already busy with the next band (two sets of buffers) */
  const int band = T * 64;
  fmt_job jobs[2][16];
  pthread_t th[16], wth;
  band_write bw[2];
  int writing = -1;                                       /* set whose write is in flight, or -1 */
  for (int k = 0; k < 2; k++)
    for (int t = 0; t < T; t++) { jobs[k][t].buf = NULL; jobs[k][t].cap = 0; jobs[k][t].src = src; jobs[k][t].ncols = ncols; }
  int set = 0;
  for (int r0 = 0; r0 < nrows && !rc; r0 += band, set ^= 1) {
    fmt_job *job = jobs[set];
    const int r1 = r0 + band < nrows ? r0 + band : nrows;
    const int per = (r1 - r0 + T - 1) / T;
    for (int t = 0; t < T; t++) {
      job[t].row0 = r0 + t * per < r1 ? r0 + t * per : r1;
      job[t].row1 = job[t].row0 + per < r1 ? job[t].row0 + per : r1;
    }
    if (T > 1) {
      for (int t = 0; t < T; t++) pthread_create(&th[t], NULL, format_rows, &job[t]);
      for (int t = 0; t < T; t++) pthread_join(th[t], NULL);
    } else {
      format_rows(&job[0]);
    }
    if (writing >= 0) {                                   /* the previous band must be on its way before this one */
      pthread_join(wth, NULL);
      rc |= bw[writing].rc;
      writing = -1;
    }
    bw[set].f = f; bw[set].jobs = job; bw[set].T = T; bw[set].rc = 0;
    if (T > 1 && !rc && pthread_create(&wth, NULL, write_band, &bw[set]) == 0) writing = set;
    else if (!rc) { write_band(&bw[set]); rc |= bw[set].rc; }
  }
  if (writing >= 0) { pthread_join(wth, NULL); rc |= bw[writing].rc; }
  for (int k = 0; k < 2; k++)
    for (int t = 0; t < T; t++) free(jobs[k][t].buf);
  if (fclose(f) != 0) rc = 1;
  return rc;
}

static const char kF64Magic[8] = {'W', 'D', 'P', 'M', 'F', '6', '4', '\n'};

int asc_write_f64(const char *path, int nrows, int ncols, const double *src) {
  const size_t plen = strlen(path);
  char *tmp = (char *)malloc(plen + 5);
  if (!tmp) return 1;
  memcpy(tmp, path, plen);
  memcpy(tmp + plen, ".tmp", 5);
  FILE *f = fopen(tmp, "wb");
  if (!f) { free(tmp); return 1; }
  const int64_t dims[3] = {nrows, ncols, 0};
  const size_t n = (size_t)nrows * ncols;
  int rc = fwrite(kF64Magic, 1, 8, f) != 8 || fwrite(dims, sizeof(int64_t), 3, f) != 3 || fwrite(src, sizeof(double), n, f) != n;
  if (fclose(f) != 0) rc = 1;
  if (!rc && rename(tmp, path) != 0) rc = 1;
  if (rc) remove(tmp);
  free(tmp);
  return rc;
}

int asc_read_f64(const char *path, int nrows, int ncols, double *dst) {
  FILE *f = fopen(path, "rb");
  if (!f) return 1;
  char magic[8];
  int64_t dims[3];
  const size_t n = (size_t)nrows * ncols;
  int rc = fread(magic, 1, 8, f) != 8 || memcmp(magic, kF64Magic, 8) != 0 || fread(dims, sizeof(int64_t), 3, f) != 3 ||
           dims[0] != nrows || dims[1] != ncols;
  if (!rc) {
    /* complete file of exactly this size? (never half-fill dst) */
    if (fseek(f, 0, SEEK_END) != 0 || ftell(f) != (long)(32 + n * sizeof(double)) || fseek(f, 32, SEEK_SET) != 0) rc = 1;
  }
  if (!rc && fread(dst, sizeof(double), n, f) != n) rc = 1;
  fclose(f);
  return rc;
}
