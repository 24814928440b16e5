/*
 * arcascii.h — ESRI ArcASCII grid reader/writer with the exact textual behaviour of the reference
 * (src/WDPMCL.c: header :534-556, read_dem_array/read_water_array :1556-1599, write_gis :1533-1554).
 * Host code of the WDPMCL drop-in; not on the timed path.
 */
#ifndef WDPM_ARCASCII_H
#define WDPM_ARCASCII_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  char name[6][32];   /* header keywords as they appear in the file (max 30 chars, like "%30s") */
  double value[6];    /* NCOLS NROWS XLLCORNER YLLCORNER CELLSIZE NODATA_VALUE, by position */
} asc_header;

/* read the six "%30s %lf" header pairs.  returns 0 on success */
int asc_read_header(const char *path, asc_header *h);

/* read nrows*ncols whitespace-separated values that follow the six header pairs into dst
 * (row-major).  Cells for which the file has no (parsable) value keep their current content,
 * as with the reference's fscanf loop.  returns 0 on success, non-zero if the file cannot be read */
int asc_read_grid(const char *path, int nrows, int ncols, double *dst);

/* write header + grid: "%s %d" / "%s %14.6f" / "%s %9.6f" lines and "%f " per cell */
int asc_write_grid(const char *path, const asc_header *h, int nrows, int ncols, const double *src);

/* lossless side-format of a checkpoint (the ASCII scratch keeps 1e-6 m): a 32-byte
 * header {"WDPMF64\n", int64 nrows, int64 ncols, int64 0} followed by nrows*ncols host-order doubles.
 * Written to "<path>.tmp" and renamed, so a reader never sees a torn file.  asc_read_f64 fills dst
 * and returns 0 only if the file exists, is complete and has exactly these dimensions. */
int asc_write_f64(const char *path, int nrows, int ncols, const double *src);
int asc_read_f64(const char *path, int nrows, int ncols, double *dst);

/* exact fast paths behind the reader/writer (exposed for tests): the characters of
 * printf("%f", x) without NUL (returns their count; out needs >= 340 bytes), and strtod for plain
 * decimal numbers */
int asc_format_f6(double x, char *out);
double asc_parse_double(const char *p, char **end);

#ifdef __cplusplus
}
#endif
#endif
