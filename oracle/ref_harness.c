/*
 * ref_harness.c — TEST INFRASTRUCTURE.  Builds the UNMODIFIED reference translation unit
 * (src/WDPMCL.c, included from where it lies under /root/reference; nothing is copied) into
 * oracle/_ref/libwdpm_ref.so and exposes its serial stencil functions runoffs(), runoffd(),
 * drain() on the reference's own globals, driven in the loop order of WDPMCL.c:1074-1123.
 * Used to (1) generate tests/golden/ fixtures (tests/golden/make_golden.py) and (2) validate the
 * CPU restatement oracle/wdpm_oracle.c bit-for-bit in this container.  /root/reference does not
 * exist on the GPU box; the prebuilt .so travels, the reference sources never do.
 *
 * Build (oracle/Makefile):  gcc -O2 -shared -fPIC -DREF_SRC='"/root/reference/src/WDPMCL.c"' ...
 */
#ifndef REF_SRC
#error "define REF_SRC to the path of the reference's src/WDPMCL.c"
#endif

#define main wdpmcl_reference_main
#include REF_SRC
#undef main

static double **alloc2(int r, int c) {
  double **a = malloc(r * sizeof(double *));
  for (int i = 0; i < r; i++) a[i] = malloc(c * sizeof(double));
  return a;
}
static void free2(double **a, int r) {
  if (!a) return;
  for (int i = 0; i < r; i++) free(a[i]);
  free(a);
}

static int h_rows = 0;

/* set the reference globals (WDPMCL.c:235-239) from flat padded row-major arrays */
void ref_setup(int R, int C, double missing, const double *bdem, const double *bwater,
               double tdrain, int drow, int dcol) {
  free2(bigdem, h_rows);
  free2(bigwater, h_rows);
  numrows = R; numcols = C; missingvalue = missing;
  totaldrain = tdrain; drainrow = drow; draincol = dcol;
  h_rows = R + 2;
  bigdem = alloc2(R + 2, C + 2);
  bigwater = alloc2(R + 2, C + 2);
  for (int i = 0; i < R + 2; i++)
    for (int j = 0; j < C + 2; j++) {
      bigdem[i][j] = bdem[(size_t)i * (C + 2) + j];
      bigwater[i][j] = bwater[(size_t)i * (C + 2) + j];
    }
}

/* one colour pass exactly as the serial loops visit it (WDPMCL.c:1079-1086, :1097-1103) */
void ref_pass(int module, int oi, int oj) {
  for (int row = oi; row <= numrows; row += 3)
    for (int col = oj; col <= numcols; col += 3) {
      if (module == 2) {
        if (bigwater[row][col] > 0.0 && (bigdem[row][col] > missingvalue) &&
            (row != drainrow || col != draincol))
          runoffd(row, col, drainrow, draincol, missingvalue);
      } else {
        if (bigwater[row][col] > 0.0 && (bigdem[row][col] > missingvalue))
          runoffs(row, col, missingvalue);
      }
    }
}

void ref_drain_outlet(void) { totaldrain = totaldrain + drain(drainrow, draincol); }

void ref_iterate(int module, int n) {
  for (int i = 0; i < n; i++) {
    for (int oi = 1; oi < 4; oi++)
      for (int oj = 1; oj < 4; oj++) ref_pass(module, oi, oj);
    if (module == 2) ref_drain_outlet();
  }
}

void ref_get_water(double *out) {
  for (int i = 0; i < numrows + 2; i++)
    for (int j = 0; j < numcols + 2; j++) out[(size_t)i * (numcols + 2) + j] = bigwater[i][j];
}
double ref_get_totaldrain(void) { return totaldrain; }
