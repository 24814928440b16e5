/* writes the synthetic n x n DEM of SURVEY.md §8d (wdpm_synth_dem, seed = n) as an ArcASCII grid:
 *   synth_asc n out.asc        (tooling for end-to-end timing of the WDPMCL drop-in at large sizes) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../include/wdpm.h"
#include "../wdpm_amd/csrc/arcascii.h"

int main(int argc, char **argv) {
  if (argc != 3) { fprintf(stderr, "usage: synth_asc n out.asc\n"); return 2; }
  const int n = atoi(argv[1]);
  double *dem = (double *)malloc((size_t)n * n * sizeof(double));
  if (!dem || wdpm_synth_dem(n, (uint64_t)n, dem)) { fprintf(stderr, "synth failed\n"); return 1; }
  asc_header h;
  memset(&h, 0, sizeof h);
  const char *names[6] = {"ncols", "nrows", "xllcorner", "yllcorner", "cellsize", "NODATA_value"};
  const double values[6] = {n, n, 0, 0, 10, -99999};
  for (int i = 0; i < 6; i++) { strcpy(h.name[i], names[i]); h.value[i] = values[i]; }
  return asc_write_grid(argv[2], &h, n, n, dem);
}
