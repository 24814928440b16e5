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
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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
  size_t len;
  char *buf = slurp(path, &len);
  if (!buf) return 1;
  memset(h, 0, sizeof *h);
  const char *p = buf;
  for (int i = 0; i < 6 && p; i++) p = header_pair(p, h->name[i], &h->value[i]);
  free(buf);
  return 0;
}

int asc_read_grid(const char *path, int nrows, int ncols, double *dst) {
  size_t len;
  char *buf = slurp(path, &len);
  if (!buf) return 1;
  const char *p = buf;
  char name[32];
  double dummy;
  for (int i = 0; i < 6 && p; i++) p = header_pair(p, name, &dummy);
  const size_t total = (size_t)nrows * ncols;
  for (size_t k = 0; k < total && p; k++) {
    char *end;
    double v = strtod(p, &end);
    if (end == p) break;        /* end of data or junk: the remaining cells keep their values */
    dst[k] = v;
    p = end;
  }
  free(buf);
  return 0;
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
  for (int r = 0; r < nrows && !rc; r++) {
    const double *row = src + (size_t)r * ncols;
    for (int c = 0; c < ncols; c++) {
      n += (size_t)snprintf(buf + n, 400, "%f ", row[c]);
      if (n >= kBuf) {
        if (fwrite(buf, 1, n, f) != n) { rc = 1; break; }
        n = 0;
      }
    }
    buf[n++] = '\n';
  }
  if (!rc && n && fwrite(buf, 1, n, f) != n) rc = 1;
  free(buf);
  if (fclose(f) != 0) rc = 1;
  return rc;
}
