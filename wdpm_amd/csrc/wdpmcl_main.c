/*
 * wdpmcl_main.c — the WDPMCL command line on top of the wdpm C ABI (include/wdpm.h).
 *
 * Drop-in for the reference executable (src/WDPMCL.c main(), :266-1506): same positional
 * arguments (argc 13 for add, 12 for subtract/drain, or one parameter-file argument), same units,
 * "NULL" handling, report text on stdout, ArcASCII water/scratch/output files and exit codes
 * (42 for usage errors and for a missing drain water file).  The redistribution loop itself —
 * flush, snapshot, 1000 iterations of the 9-colour sweep, max-change test — runs on the GPU through
 * wdpm_run_block(); this file is host plumbing written in C like the reference's.
 *
 * Differences, all deliberate:
 *   - the "0 serial / 1 OpenCL" and "0 CPU / 1 GPU" slots are accepted and echoed in the report as
 *     the reference does, but the computation always runs on the HIP device (there is no CPU
 *     path in this build); the device in use is reported on stderr so stdout stays identical.
 *   - runoff.cl is not needed in the working directory.
 *   - malformed invocations that make the reference read uninitialised memory (unknown module with
 *     12/13 arguments, short parameter file) print the usage text and exit 42 instead.
 *   - WDPM_REPORT_BACKEND=1 adds one line to the report (after the "Using ... for Computation" lines) naming the
 *     back-end, the devices and the decomposition actually used; off by default so that stdout stays the reference's.
 *   - WDPM_COLOR_RELIEF=<colour map file> hands the output raster to `gdaldem color-relief` once it is written,
 *     as the reference's src/cmap_black.sh does for the GUI's PNG button (<output>.png; the .aux.xml is removed).
 *   - WDPM_DEVICE=<n> selects the HIP device (default 0).  WDPM_GPUS=<N> spreads the raster over
 *     devices 0..N-1 by row blocks (WDPM_DEVICES=a,b,c names them explicitly), one host thread per device,
 *     exchanging halo rows every WDPM_EXCHANGE_EVERY iterations (default 8) by RCCL send/recv
 *     (WDPM_HALO=peer: peer copies); results do not depend on N.
 */
#include <ctype.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <sys/wait.h>
#include <unistd.h>

#include "../../include/wdpm.h"
#include "arcascii.h"

#define ITER_PER_BLOCK 1000 /* IterationNum, WDPMCL.c:597 */

typedef struct {
  int module;               /* WDPM_ADD / WDPM_SUBTRACT / WDPM_DRAIN */
  char name[20];            /* module as typed */
  char dem[512], water[512], output[512], scratch[512];
  double addwater, subtractwater, rof, eltol, draintol, thres; /* as typed (mm, m3) */
  int cpu, gpu, iteration_limit;
} run_config;

static void put_lines(const char *const *lines) {
  for (; *lines; lines++) printf("%s\n", *lines);
}

/* ---- usage texts (WDPMCL.c:1658-1745) ------------------------------------------------------ */
static void usage_module(const char *module) {
  const int is_add = strcmp(module, "add") == 0, is_sub = strcmp(module, "subtract") == 0,
            is_drain = strcmp(module, "drain") == 0;
  printf("%s\n", "                                          ");
  printf("%s\n", "Program arguments in order of specification");
  if (is_add) printf("%s\n", "Add module specified");
  else if (is_sub) printf("%s\n", "Subtract module specified");
  else if (is_drain) printf("%s\n", "Drain module specified");
  printf("%s\n", "DEM file name (string) ");
  printf("%s\n", is_add ? "Water file name (string) - Optional, Use NULL to omit" : "Water file name (string)");
  printf("%s\n", "Output file name (string)");
  printf("%s\n", "Scratch file name (string) - Optional, use NULL to omit");
  if (is_add) {
    printf("%s\n", "Depth of water to add (mm) (real)");
    printf("%s\n", "Water runoff fraction (real)");
    printf("%s\n", "Elevation tolerance (mm) (real)");
  } else if (is_sub) {
    printf("%s\n", "Depth of water to remove (mm) (real)");
    printf("%s\n", "Elevation tolerance (mm) (real)");
  } else if (is_drain) {
    printf("%s\n", "Elevation tolerance (mm) (real)");
    printf("%s\n", "Drain tolerance (m3) (real)");
  }
  printf("%s\n", "Specify 0 for serial CPU and 1 for opencl ");
  printf("%s\n", "Specify 0 for OpenCL CPU and 1 for opencl GPU ");
  printf("%s\n", "Zero depth threshold (mm) (real)");
  printf("%s\n", "Maximum number of iterations (integer) - Optional, Use 0 to omit ");
  printf("%s\n", "                                          ");
}

static void usage_all(void) {
  static const char *const text[] = {
      "Module name: add",
      "DEM file name (string)",
      "Water file name (string) - Optional, use --NULL-- to omit",
      "Output file name (string)",
      "Scratch file name (string) - Optional, use --NULL-- to omit",
      "Depth of water to add (mm) (real)",
      "Water runoff fraction (real)",
      "Elevation tolerance (mm) (real)",
      "Specify 0 for serial CPU and 1 for opencl ",
      "Specify 0 for OpenCL CPU and 1 for opencl GPU ",
      "Zero depth threshold (mm) (real) ",
      "Maximum number of iterations (integer) - Optional, Use 0 to omit",
      "                                          ",
      "                                          ",
      "Module name: subtract",
      "Path and Name of Report file",
      "DEM file name (string)",
      "Water file name (string)",
      "Output file name (string)",
      "Scratch file name (string) - Optional, use --NULL-- to omit",
      "Depth of water to remove (mm) (real)",
      "Elevation tolerance (mm) (real)",
      "Specify 0 for serial CPU and 1 for opencl ",
      "Specify 0 for OpenCL CPU and 1 for opencl GPU ",
      "Zero depth threshold (mm) (real) ",
      "Maximum number of iterations (integer) - Optional, Use 0 to omit ",
      "                                          ",
      "                                          ",
      "Module name: drain",
      "Path and Name of Report file",
      "DEM file name (string)",
      "Water file name (string) ",
      "Output file name (string)",
      "Scratch file name (string) - Optional, use --NULL-- to omit",
      "Elevation tolerance (mm) (real)",
      "Drain tolerance (m3) (real)",
      "Specify 0 for serial CPU and 1 for opencl ",
      "Specify 0 for OpenCL CPU and 1 for opencl GPU ",
      "Zero depth threshold (mm) (real) ",
      "Maximum number of iterations (integer) - Optional, Use 0 to omit",
      "                                          ",
      NULL};
  put_lines(text);
}

/* ---- banner (WDPMCL.c:1616-1655) ---------------------------------------------------------- */
static void banner(int module) {
  static const char blank[] = "                                                                   ";
  static const char *const head[] = {
      blank, blank,
      "Wetland DEM Ponding Model version 2.0",
      "Copyright (c) 2010, 2012, 2014, 2020 Kevin Shook, Centre for Hydrology",
      "Developed by Oluwaseun Sharomi, Raymond Spiteri and Tonghe Liu",
      "Numerical Simulation Laboratory, University of Saskatchewan.\n",
      "--------------------------------------------------------------------",
      blank,
      "This program is free software: you can redistribute it and/or modify",
      "it under the terms of the GNU General Public License as published by",
      "the Free Software Foundation, either version 3 of the License, or",
      "(at your option) any later version.",
      blank,
      "This program is distributed in the hope that it will be useful,",
      "but WITHOUT ANY WARRANTY; without even the implied warranty of",
      "MERCHANTABILITY or FITNESS FOR A PARTICULAR PURPOSE.  See the",
      "GNU General Public License for more details.",
      blank,
      "You should have received a copy of the GNU General Public License",
      "along with this program.  If not, see <http://www.gnu.org/licenses/>.",
      blank, NULL};
  put_lines(head);
  if (module == WDPM_ADD) {
    printf("%s\n", "This program adds water to an ArcGIS ASCII file of water runoff");
    printf("%s\n", "and redistributes water over the DEM");
  } else if (module == WDPM_SUBTRACT) {
    printf("%s\n", "This program removes a depth water to an ArcGIS ASCII file of water depths");
    printf("%s\n", "and redistributes water over the DEM");
  } else if (module == WDPM_DRAIN) {
    printf("%s\n", "This program drains an ArcGIS ASCII file of water runoff");
    printf("%s\n", "from the lowest point in the DEM, which acts as a drain");
  }
  printf("%s\n", "From the algorithm of Shapiro, M., & Westervelt, J. (1992). ");
  printf("%s\n", "An Algebra for GIS and Image Processing (pp. 1-22).");
  printf("%s\n%s\n", blank, blank);
}

/* ---- parameter echo (WDPMCL.c:1748-1797) --------------------------------------------------- */
static void echo_parameters(const run_config *c) {
  printf("%30s\n", "WDPM Parameters");
  printf("%30s %s\n", "Function used:", c->name);
  printf("%30s %s\n", "DEM file:", c->dem);
  printf("%30s %s\n", "Water file:", c->water);
  printf("%30s %s\n", "Output file:", c->output);
  printf("%30s %s\n", "Scratch file:", c->scratch);
  if (c->module == WDPM_ADD) {
    printf("%30s %0.4f %s\n", "Water added:", c->addwater, "mm");
    printf("%30s %0.4f\n", "Runoff fraction:", c->rof);
    printf("%30s %0.4f %s\n", "Elevation tolerance:", c->eltol, "mm");
  } else if (c->module == WDPM_SUBTRACT) {
    printf("%30s %0.4f %s\n", "Water subtracted:", c->subtractwater, "mm");
    printf("%30s %0.4f %s\n", "Elevation tolerance:", c->eltol, "mm");
  } else {
    printf("%30s %0.4f %s\n", "Elevation tolerance:", c->eltol, "mm");
    printf("%30s %0.4f %s\n", "Drain tolerance:", c->draintol, "m3");
  }
  printf("%30s %0.4f %s\n", "Zero depth threshold:", c->thres, "mm");
  if (c->iteration_limit == 0) printf("%30s\n", "No iteration limitation is set");
  else printf("%30s %d\n", "Maximum number of iterations:", c->iteration_limit);
  printf("%s\n", "               ");
  if (c->cpu == 0) {
    printf("%41s\n", "Using Serial CPU for Computation");
  } else {
    printf("%41s\n", "Using Parallel OpenCL for Computation");
    if (c->gpu == 0) printf("%40s\n", "Using OpenCL CPU for Computation");
    if (c->gpu == 1) printf("%40s\n", "Using OpenCL GPU for Computation");
  }
}

static void iteration_headings(int module) {
  printf("%s\n", "               ");
  printf("%30s\n", "Doing calculations");
  if (module == WDPM_DRAIN) {
    printf("%15s %15s %15s %15s %15s\n", "iterations", "max diff", "vol change", "water left", "run time");
    printf("%13s %14s %15s %16s %17s\n", " ", "(m)", "(m3)", "(m3)", "(s)");
  } else {
    printf("%15s %15s %15s\n", "iterations", "max diff", "run time");
    printf("%13s %14s %15s\n", " ", "(m)", "(s)");
  }
}

/* ---- small helpers -------------------------------------------------------------------------- */
static int is_null_name(const char *s) { /* "NULL" in any case, WDPMCL.c:666-668 via upcase() */
  return strlen(s) == 4 && toupper((unsigned char)s[0]) == 'N' && toupper((unsigned char)s[1]) == 'U' &&
         toupper((unsigned char)s[2]) == 'L' && toupper((unsigned char)s[3]) == 'L';
}

static int file_exists(const char *path) {
  struct stat st;
  return stat(path, &st) == 0;
}

static double seconds_since(const struct timeval *t0) {
  struct timeval t;
  gettimeofday(&t, NULL);
  return (double)(t.tv_usec - t0->tv_usec) / 1000000 + (double)(t.tv_sec - t0->tv_sec);
}

static void die_usage(const char *module) {
  usage_module(module);
  exit(42);
}

static int module_from_name(const char *s) {
  if (strcmp(s, "add") == 0) return WDPM_ADD;
  if (strcmp(s, "subtract") == 0) return WDPM_SUBTRACT;
  if (strcmp(s, "drain") == 0) return WDPM_DRAIN;
  return -1;
}

/* positional values after the module name, in the order of WDPMCL.c:386-398 / :445-456 / :501-512 */
static void assign_fields(run_config *c, char **v) {
  snprintf(c->dem, sizeof c->dem, "%s", v[0]);
  snprintf(c->water, sizeof c->water, "%s", v[1]);
  snprintf(c->output, sizeof c->output, "%s", v[2]);
  snprintf(c->scratch, sizeof c->scratch, "%s", v[3]);
  int k = 4;
  if (c->module == WDPM_ADD) {
    c->addwater = atof(v[k++]);
    c->rof = atof(v[k++]);
    c->eltol = atof(v[k++]);
  } else if (c->module == WDPM_SUBTRACT) {
    c->subtractwater = atof(v[k++]);
    c->eltol = atof(v[k++]);
  } else {
    c->eltol = atof(v[k++]);
    c->draintol = atof(v[k++]);
  }
  c->cpu = (int)atof(v[k++]);
  c->gpu = (int)atof(v[k++]);
  c->thres = atof(v[k++]);
  c->iteration_limit = (int)atof(v[k++]);
}

/* whitespace-separated tokens of a parameter file (WDPMCL.c:334-343,366-385) */
static int read_param_file(const char *path, char tok[16][512]) {
  FILE *f = fopen(path, "r");
  if (!f) return -1;
  int n = 0;
  while (n < 16 && fscanf(f, "%511s", tok[n]) == 1) n++;
  fclose(f);
  return n;
}

static void parse_command_line(int argc, char **argv, run_config *c) {
  memset(c, 0, sizeof *c);
  if (argc == 1) { usage_all(); exit(42); }
  static char tok[16][512];
  char *vals[16];
  int nvals = 0, from_file = 0;
  if (argc == 2) {
    if (module_from_name(argv[1]) >= 0) die_usage(argv[1]);
    int n = read_param_file(argv[1], tok);
    if (n < 1) { usage_all(); exit(42); }
    snprintf(c->name, sizeof c->name, "%.19s", tok[0]);
    for (int i = 1; i < n; i++) vals[nvals++] = tok[i];
    from_file = 1;
  } else {
    snprintf(c->name, sizeof c->name, "%.19s", argv[1]);
    if (argc != 12 && argc != 13) die_usage(c->name);
    for (int i = 2; i < argc; i++) vals[nvals++] = argv[i];
  }
  c->module = module_from_name(c->name);
  banner(c->module);
  const int need = c->module == WDPM_ADD ? 11 : 10;
  if (c->module < 0 || (from_file ? nvals < need : nvals != need)) die_usage(c->name);
  assign_fields(c, vals);
}

/* ---- module set-up (WDPMCL.c:643-1034) ------------------------------------------------------ */
typedef struct {
  int R, C;
  double missing, cellsize, cellarea;
  double *dem, *water;          /* R x C, as in the files */
  int water_all_zero;           /* no water raster was read: the device starts from zeros, nothing is uploaded */
  int apply_module_water;       /* the module's water adjustment is still to be applied (not on a resumed run) */
  double initial_vol, basin_area, totaldrain0;
  int basincount, drainrow, draincol;
} raster_state;

static double volume_where(const raster_state *s, double dem_above) {
  double v = 0;
  const size_t n = (size_t)s->R * s->C;
  for (size_t i = 0; i < n; i++)
    if (s->dem[i] > dem_above) v += s->water[i];
  return v * s->cellarea;
}

/* a raster file that exists but cannot be read: the reference dereferences the NULL FILE* and crashes;
 * say what happened and stop (a SHORT file keeps the reference's behaviour: missing cells stay as they are) */
static void read_grid_or_die(const char *path, int R, int C, double *dst) {
  if (asc_read_grid(path, R, C, dst) != 0) {
    fprintf(stderr, "WDPMCL: cannot read raster file %s\n", path);
    exit(1);
  }
}

/* water-file branch shared by add and subtract: returns 1 when an existing file was read */
static int load_or_create_water(const run_config *c, raster_state *s) {
  if (!is_null_name(c->water)) {
    if (file_exists(c->water)) {
      printf("%30s\n", "Existing water file found");
      read_grid_or_die(c->water, s->R, s->C, s->water);
      return 1;
    }
    printf("%30s\n", "Water file missing, will be created");
  } else {
    printf("%30s\n", "Water file will be created");
  }
  s->water_all_zero = 1;       /* the calloc'ed raster: the device builds its zeros itself */
  return 0;
}

static void announce_no_scratch(int module) {
  printf("%s\n", "           ");
  printf("%30s\n", "No Scratch file found");
  printf("%s\n", "           ");
  printf("%30s\n", module == WDPM_SUBTRACT ? "New Scratch will be saved." : "New Scratch will be saved");
  printf("%s\n", "           ");
  printf("%30s\n", "Now proceeding with Waterfile checking");
  printf("%s\n", "           ");
}

/* WDPM_SCRATCH_BINARY=1: every checkpoint also leaves "<scratch>.f64", the same state in full
 * precision (arcascii.h), and a resume prefers it when it agrees with the ASCII scratch to the
 * 1e-6 m the text keeps (i.e. when it is the same checkpoint): the resumed run then continues
 * bit-for-bit where the interrupted one stopped, which the reference's text scratch cannot do. */
static int scratch_binary_enabled(void) {
  const char *e = getenv("WDPM_SCRATCH_BINARY");
  return e && atoi(e) != 0;
}

static char *sidecar_name(const char *scratch) {
  char *p = (char *)malloc(strlen(scratch) + 5);
  if (p) sprintf(p, "%s.f64", scratch);
  return p;
}

static void prefer_lossless_scratch(const run_config *c, raster_state *s) {
  if (!scratch_binary_enabled()) return;
  char *side = sidecar_name(c->scratch);
  const size_t n = (size_t)s->R * s->C;
  double *full = (double *)malloc(n * sizeof(double));
  if (side && full && asc_read_f64(side, s->R, s->C, full) == 0) {
    size_t bad = 0;
    for (size_t i = 0; i < n; i++) bad += !(fabs(full[i] - s->water[i]) <= 5.0000001e-7);
    if (bad == 0) {
      memcpy(s->water, full, n * sizeof(double));
      fprintf(stderr, "WDPMCL: resuming from the full-precision checkpoint %s\n", side);
    } else {
      fprintf(stderr, "WDPMCL: %s does not match the scratch file (%zu cells differ): ignored\n", side, bad);
    }
  }
  free(full);
  free(side);
}

/* the host half of the module set-up (WDPMCL.c:643-1034): which file the water raster comes from.  Applying
 * the module's water adjustment, padding, the basin count, the drain-cell search and the volumes are done next
 * to the rasters by the back-end (wdpm_group_upload_unpadded etc.). */
static void prepare_water(const run_config *c, raster_state *s) {
  const int have_scratch_name = !is_null_name(c->scratch);
  int resumed = 0;
  /* :656-664 / :813-821 sum the water raster before anything has been read into it: every term is
   * the +0.0 it was allocated with, so the sum is +0.0 without a pass over the raster */
  if (c->module == WDPM_ADD || c->module == WDPM_SUBTRACT) s->initial_vol = 0.0 * s->cellarea;
  if (have_scratch_name) {
    if (file_exists(c->scratch)) {                                             /* resume, :668-673 */
      printf("%s\n", "           ");
      printf("%30s\n", "Scratch file found");
      read_grid_or_die(c->scratch, s->R, s->C, s->water);
      prefer_lossless_scratch(c, s);
      resumed = 1;
    } else {
      announce_no_scratch(c->module);
    }
  }
  if (!resumed) {
    if (c->module == WDPM_DRAIN) {
      if (!file_exists(c->water)) {                                            /* :968-973, :983-988 */
        printf("%30s\n", "Error water file missing");
        exit(42);
      }
      printf("%30s\n", "Existing water file found");
      read_grid_or_die(c->water, s->R, s->C, s->water);
    } else {
      const int read_file = load_or_create_water(c, s);
      /* the reference recomputes the initial volume only on the scratch-name branch (:690-699, :846-855), from the
       * water file as read, with `dem > 0` for subtract: a sequential sum over the file raster, on the host */
      if (read_file && have_scratch_name)
        s->initial_vol = volume_where(s, c->module == WDPM_ADD ? s->missing : 0.0);
      s->apply_module_water = 1;
    }
  }
}

/* the drain cell on the host (WDPMCL.c:1005-1017 on the padded raster, whose border holds the NODATA value):
 * only for rasters whose NODATA value is not negative - there a NODATA cell can itself be the smallest
 * elevation > 0, which the device copy of the DEM (NODATA = +inf) cannot say */
static void find_drain_on_host(raster_state *s) {
  double mindrain = 100000000;
  for (int i = 0; i < s->R + 2; i++)
    for (int j = 0; j < s->C + 2; j++) {
      const int border = i == 0 || j == 0 || i == s->R + 1 || j == s->C + 1;
      const double d = border ? s->missing : s->dem[(size_t)(i - 1) * s->C + (j - 1)];
      if (d > 0 && d < mindrain) { mindrain = d; s->drainrow = i; s->draincol = j; }
    }
}

/* devices to spread the raster over: WDPM_DEVICES=a,b,c | WDPM_GPUS=N (0..N-1) | WDPM_DEVICE=n | 0 */
static int device_list(int32_t *dev, int max) {
  const char *lst = getenv("WDPM_DEVICES");
  int n = 0;
  if (lst && *lst) {
    char *copy = strdup(lst), *save = NULL;
    for (char *t = strtok_r(copy, ",", &save); t && n < max; t = strtok_r(NULL, ",", &save)) dev[n++] = atoi(t);
    free(copy);
  } else if (getenv("WDPM_GPUS") && atoi(getenv("WDPM_GPUS")) > 1) {
    const int want = atoi(getenv("WDPM_GPUS"));
    for (n = 0; n < want && n < max; n++) dev[n] = n;
  }
  if (n == 0) dev[n++] = getenv("WDPM_DEVICE") ? atoi(getenv("WDPM_DEVICE")) : 0;
  return n;
}

/* One context addresses at most 2e9 padded cells (wdpm_create: 32-bit cell indices inside a row block).  The reference's
 * serial path has no such limit, so a larger raster is cut into more row blocks on the SAME device(s): each device of the
 * list is named `per` times in a row and the usual row-block driver does the rest (halos between slabs of one device are
 * device-to-device copies).  WDPM_MAX_SLAB_CELLS overrides the threshold (tests). */
static int split_for_size(int32_t *dev, int n, const int max, const double padded_cells) {
  const double lim = getenv("WDPM_MAX_SLAB_CELLS") ? atof(getenv("WDPM_MAX_SLAB_CELLS")) : 1.9e9;
  if (!(lim > 0) || padded_cells <= lim * n) return n;
  int per = (int)((padded_cells + lim * n - 1) / (lim * n));
  if (n * per > max) per = max / n;
  if (per <= 1) return n;
  for (int i = n - 1; i >= 0; i--)
    for (int k = per - 1; k >= 0; k--) dev[i * per + k] = dev[i];
  return n * per;
}

/* ---- scratch (checkpoint) writer -------------------------------------------------------------
 * The reference rewrites the scratch raster after every block that does not end the run
 * (WDPMCL.c:1290-1372) and stalls the loop while fprintf runs.  Here the loop only pays for the
 * device-to-host copy: the text is produced by a writer thread while the next blocks compute.  The
 * writer always takes the NEWEST pending state (an older one that never reached the disk is simply
 * superseded) and main() waits for it before exiting, so the file left behind is the one the
 * reference leaves: the state after the last non-final block. */
typedef struct {
  pthread_t thread;
  pthread_mutex_t mu;
  pthread_cond_t cv;
  const char *path;
  char *sidecar;     /* "<path>.f64" when WDPM_SCRATCH_BINARY is set, else NULL */
  const asc_header *hdr;
  int R, C;
  double *pending;   /* newest state not yet picked up by the writer, or NULL */
  double *spare;     /* buffer the producer may fill next */
  int busy, quit, started;
} scratch_writer;

static void scratch_write_now(const scratch_writer *w, const double *water) {
  if (w->sidecar) asc_write_f64(w->sidecar, w->R, w->C, water);
  asc_write_grid(w->path, w->hdr, w->R, w->C, water);
}

static void *scratch_main(void *arg) {
  scratch_writer *w = (scratch_writer *)arg;
  pthread_mutex_lock(&w->mu);
  for (;;) {
    while (!w->pending && !w->quit) pthread_cond_wait(&w->cv, &w->mu);
    if (!w->pending) break;                       /* quit and nothing left to write */
    double *job = w->pending;
    w->pending = NULL;
    w->busy = 1;
    pthread_mutex_unlock(&w->mu);
    scratch_write_now(w, job);
    pthread_mutex_lock(&w->mu);
    w->busy = 0;
    if (!w->spare) w->spare = job; else free(job);
    pthread_cond_broadcast(&w->cv);
  }
  pthread_mutex_unlock(&w->mu);
  return NULL;
}

static void scratch_start(scratch_writer *w, const char *path, const asc_header *hdr, int R, int C) {
  memset(w, 0, sizeof *w);
  pthread_mutex_init(&w->mu, NULL);
  pthread_cond_init(&w->cv, NULL);
  w->path = path; w->hdr = hdr; w->R = R; w->C = C;
  w->sidecar = scratch_binary_enabled() ? sidecar_name(path) : NULL;
  w->started = pthread_create(&w->thread, NULL, scratch_main, w) == 0;
}

/* hand over a copy of the un-padded raster (takes the cells from `water`) */
static void scratch_submit(scratch_writer *w, const double *water) {
  const size_t bytes = (size_t)w->R * w->C * sizeof(double);
  if (!w->started) { scratch_write_now(w, water); return; }
  pthread_mutex_lock(&w->mu);
  double *buf = w->pending ? w->pending : w->spare;    /* supersede an unwritten older state */
  if (buf == w->spare) w->spare = NULL;
  w->pending = NULL;
  pthread_mutex_unlock(&w->mu);
  if (!buf) buf = (double *)malloc(bytes);
  if (!buf) { scratch_write_now(w, water); return; }
  memcpy(buf, water, bytes);
  pthread_mutex_lock(&w->mu);
  w->pending = buf;
  pthread_cond_broadcast(&w->cv);
  pthread_mutex_unlock(&w->mu);
}

static void scratch_finish(scratch_writer *w) {
  if (!w->started) { free(w->sidecar); return; }
  pthread_mutex_lock(&w->mu);
  w->quit = 1;
  pthread_cond_broadcast(&w->cv);
  pthread_mutex_unlock(&w->mu);
  pthread_join(w->thread, NULL);
  free(w->spare);
  free(w->pending);
  free(w->sidecar);
}

/* WDPM_TIMING=1: wall time of each phase of the run on stderr (never on stdout: the report stays
 * the reference's) */
static void phase(const char *name) {
  static struct timeval last;
  static int on = -1;
  if (on < 0) {
    on = getenv("WDPM_TIMING") && atoi(getenv("WDPM_TIMING")) != 0;
    gettimeofday(&last, NULL);
  }
  if (!on || !name) return;
  fprintf(stderr, "WDPMCL timing: %-28s %8.3f s\n", name, seconds_since(&last));
  gettimeofday(&last, NULL);
}

/* ---- gdaldem hand-off (reference: src/cmap_black.sh:1-10, run by the GUI's PNG button, src/WDPM.py:657-671) ----
 * The helper process is forked at program start, before anything has touched the GPU (a process that has
 * initialised the GPU must not exec another program), and sleeps on a pipe until the output raster is on disk. */
typedef struct { pid_t pid; int fd; char png[600]; } relief_helper;

static void relief_start(relief_helper *h, const char *output) {
  h->pid = -1;
  h->fd = -1;
  const char *map = getenv("WDPM_COLOR_RELIEF");
  if (!map || !*map) return;
  snprintf(h->png, sizeof h->png, "%s", output);
  char *dot = strrchr(h->png, '.');
  if (dot && !strchr(dot, '/')) *dot = 0;                          /* outfile="${infile%.*}" */
  strncat(h->png, ".png", sizeof h->png - strlen(h->png) - 1);
  int pfd[2];
  if (pipe(pfd) != 0) return;
  const pid_t pid = fork();
  if (pid < 0) { close(pfd[0]); close(pfd[1]); return; }
  if (pid == 0) {
    close(pfd[1]);
    char go = 0;
    if (read(pfd[0], &go, 1) != 1 || go != 'g') _exit(0);          /* the run ended without an output raster */
    execlp("gdaldem", "gdaldem", "color-relief", output, map, "-OF", "png", h->png, (char *)NULL);
    _exit(127);
  }
  close(pfd[0]);
  h->pid = pid;
  h->fd = pfd[1];
}

static void relief_finish(relief_helper *h) {
  if (h->pid < 0) return;
  const char go = 'g';
  int status = 0;
  if (write(h->fd, &go, 1) != 1) status = -1;
  close(h->fd);
  if (waitpid(h->pid, &status, 0) < 0) status = -1;
  if (WIFEXITED(status) && WEXITSTATUS(status) == 0) {
    char aux[640];
    snprintf(aux, sizeof aux, "%s.aux.xml", h->png);
    unlink(aux);                                                   /* "delete superfluous xml file" */
    fprintf(stderr, "WDPMCL: colour relief written to %s\n", h->png);
  } else if (WIFEXITED(status) && WEXITSTATUS(status) == 127) {
    fprintf(stderr, "WDPMCL: WDPM_COLOR_RELIEF is set but gdaldem could not be started\n");
  } else {
    fprintf(stderr, "WDPMCL: gdaldem color-relief failed\n");
  }
}

#define ABI_TRY(call)                                                      \
  do {                                                                     \
    if ((call) != 0) {                                                     \
      fprintf(stderr, "WDPMCL: %s: %s\n", #call, wdpm_last_error());       \
      exit(1);                                                             \
    }                                                                      \
  } while (0)

int main(int argc, char **argv) {
  setbuf(stdout, NULL);
  run_config cfg;
  parse_command_line(argc, argv, &cfg);
  relief_helper relief;
  relief_start(&relief, cfg.output);       /* before anything touches the GPU */
  echo_parameters(&cfg);

  phase(NULL);
  asc_header hdr;
  if (asc_read_header(cfg.dem, &hdr) != 0) {
    fprintf(stderr, "WDPMCL: cannot read DEM file %s\n", cfg.dem);
    return 1;
  }
  printf("%s\n", "                  ");                                         /* WDPMCL.c:1602-1613 */
  printf("%30s\n", "ArcGIS file header");
  printf("%30s %d\n", hdr.name[0], (int)hdr.value[0]);
  printf("%30s %d\n", hdr.name[1], (int)hdr.value[1]);
  printf("%30s %9.1f\n", hdr.name[2], hdr.value[2]);
  printf("%30s %9.1f\n", hdr.name[3], hdr.value[3]);
  printf("%30s %9.1f\n", hdr.name[4], hdr.value[4]);
  printf("%30s %9.1f\n", hdr.name[5], hdr.value[5]);

  raster_state st;
  memset(&st, 0, sizeof st);
  st.C = (int)hdr.value[0];
  st.R = (int)hdr.value[1];
  st.missing = hdr.value[5];
  st.cellsize = hdr.value[4];
  st.cellarea = st.cellsize * st.cellsize;
  if (st.R < 1 || st.C < 1) {
    fprintf(stderr, "WDPMCL: bad raster size in %s\n", cfg.dem);
    return 1;
  }
  printf("%30s\n", "Setting array sizes");
  const size_t ncell = (size_t)st.R * st.C;
  /* the host holds the two FILE rasters only; the padded rasters, the snapshot and the ping-pong partner live
   * on the device.  The water raster comes from the library (page-locked on the HIP back-end) when the run
   * writes checkpoints, i.e. downloads it after every block: page-locking costs ~0.3 s per GiB once and saves
   * ~40 ms per GiB on every transfer (measured at 8192^2).  WDPM_PINNED=0/1 overrides. */
  const int pinned = getenv("WDPM_PINNED") ? atoi(getenv("WDPM_PINNED")) != 0 : !is_null_name(cfg.scratch);
  st.dem = (double *)malloc(ncell * sizeof(double));
  if (pinned) {
    void *a = NULL;
    ABI_TRY(wdpm_host_alloc(ncell * sizeof(double), &a));
    st.water = (double *)a;
    if (st.water) memset(st.water, 0, ncell * sizeof(double));
  } else {
    st.water = (double *)calloc(ncell, sizeof(double));
  }
  if (!st.dem || !st.water) {
    fprintf(stderr, "WDPMCL: out of memory\n");
    return 1;
  }
  phase("allocate host rasters");
  read_grid_or_die(cfg.dem, st.R, st.C, st.dem);
  phase("read DEM");
  printf("%s\n", "           ");
  printf("%s\n", "           ");

  prepare_water(&cfg, &st);
  phase("set-up (water file)");

  /* unit conversions, WDPMCL.c:417-420 / :473-476 / :528-530 */
  const double eltol = cfg.eltol / 1000.0;
  const double thres = cfg.thres / 1000;
  const double draintol = cfg.draintol;

  wdpm_params p;
  memset(&p, 0, sizeof p);
  p.module = cfg.module;
  p.nrows = st.R;
  p.ncols = st.C;
  p.missingvalue = st.missing;
  /* drain: the outlet is found on the device once the DEM is there; a NODATA value >= 0 needs the host's search */
  const int host_drain_search = cfg.module == WDPM_DRAIN && !(st.missing < 0);
  if (host_drain_search) find_drain_on_host(&st);
  p.drainrow = cfg.module == WDPM_DRAIN ? (host_drain_search ? st.drainrow : -1) : 0;
  p.draincol = cfg.module == WDPM_DRAIN ? (host_drain_search ? st.draincol : -1) : 0;
  int32_t devices[64];
  const int ndev_asked = split_for_size(devices, device_list(devices, 64), 64, ((double)p.nrows + 2) * ((double)p.ncols + 2));
  const int every = getenv("WDPM_EXCHANGE_EVERY") ? atoi(getenv("WDPM_EXCHANGE_EVERY")) : 8;
  wdpm_setup su;
  memset(&su, 0, sizeof su);
  if (st.apply_module_water) {
    su.op = cfg.module == WDPM_ADD ? 1 : 2;
    su.add = cfg.addwater / 1000.0;                                            /* WDPMCL.c:419 */
    su.rof = cfg.rof;
    su.sub = cfg.subtractwater / 1000;                                         /* :475 */
  }
  wdpm_group *ctx = NULL;
  int ndev = 0;
  for (int attempt = 0; attempt < 2; attempt++) {
    ABI_TRY(wdpm_group_create(&ctx, &p, ndev_asked, devices, every));
    ndev = wdpm_group_size(ctx);
    /* padded rasters built on the device from the file rasters, module water applied on the way (:727-740, :796-807, :879-885) */
    ABI_TRY(wdpm_group_upload_unpadded(ctx, st.dem, st.water_all_zero ? NULL : st.water, &su));
    if (cfg.module != WDPM_DRAIN || host_drain_search || attempt == 1) break;
    double mindem = 0;
    int32_t dr = 0, dc = 0;
    ABI_TRY(wdpm_group_find_drain(ctx, &mindem, &dr, &dc));                    /* :1005-1017 */
    st.drainrow = dr;
    st.draincol = dc;
    const int rc = wdpm_group_set_drain(ctx, dr, dc);
    if (rc == 0) break;
    if (rc != 2) { fprintf(stderr, "WDPMCL: wdpm_group_set_drain: %s\n", wdpm_last_error()); return 1; }
    /* the outlet sits next to a slab boundary of the partition made without knowing it: partition again */
    wdpm_group_destroy(ctx);
    ctx = NULL;
    p.drainrow = dr;
    p.draincol = dc;
  }
  {
    static const char *const halo_name[] = {"", ", halos by RCCL send/recv", ", halos by peer copies", ", halos through the host"};
    const int hk = wdpm_group_halo(ctx);
    fprintf(stderr, "WDPMCL: redistribution loop, set-up and statistics on back-end %s, %d device%s (first: %d)%s\n",
            wdpm_backend_name(), ndev, ndev == 1 ? "" : "s, row-block decomposition, one host thread per device", devices[0],
            ndev > 1 && hk >= 0 && hk < 4 ? halo_name[hk] : "");
    /* ... and WHICH GPUs: one line per row block with the PCI bus id of the GPU it lives on (wdpm_device_info) - what the
     * reference's create_device() says about the one OpenCL device it picks (WDPMCL.c:80-121) */
    const int nblk = wdpm_group_size(ctx);
    fprintf(stderr, "WDPMCL: %d row block%s", nblk, nblk == 1 ? "" : "s");
    for (int i = 0; i < nblk; i++) {
      int32_t ord = -1;
      char bus[32] = "?";
      wdpm_rank *rk = wdpm_group_rank(ctx, i);
      if (rk && wdpm_device_info(wdpm_rank_ctx(rk), &ord, bus, (int32_t)sizeof bus) == 0)
        fprintf(stderr, "%s block %d on device %d [%s]", i ? "," : ":", i, (int)ord, bus);
    }
    fprintf(stderr, "\n");
    if (getenv("WDPM_REPORT_BACKEND") && atoi(getenv("WDPM_REPORT_BACKEND")) != 0)   /* opt-in: the report is no longer the reference's */
      printf("%41s %s, %d device%s%s\n", "Computation back-end:", wdpm_backend_name(), ndev, ndev == 1 ? "" : "s (row blocks)",
             ndev > 1 && hk >= 0 && hk < 4 ? halo_name[hk] : "");
  }
  {
    int64_t valid = 0;
    ABI_TRY(wdpm_group_count_stats(ctx, &valid, NULL, NULL));                  /* basincount, :643-650 */
    st.basincount = (int)valid;
  }
  if (cfg.module == WDPM_DRAIN) {
    double sum = 0, wd = 0, dmin = 0;
    ABI_TRY(wdpm_group_drain_stats(ctx, NULL, &sum));                          /* :1019-1028, the sequential sum */
    st.initial_vol = sum * st.cellarea;
    ABI_TRY(wdpm_group_get_cell(ctx, st.drainrow, st.draincol, &wd, &dmin));
    st.totaldrain0 = wd > 0 ? wd : 0;                                          /* :1029 */
    ABI_TRY(wdpm_group_set_totaldrain(ctx, st.totaldrain0));
    st.basin_area = st.basincount * st.cellarea;
    printf("%s\n", "               ");                                         /* :1820-1828 */
    printf("%30s\n", "Basin summary");
    printf("%20s %10.4f %s\n", "Basin area:", st.basin_area, "m2");
    printf("%20s %10.4f %s\n", "Initial volume:", st.initial_vol, "m3");
    printf("%20s %d\n", "Drain column:", st.draincol);
    printf("%20s %d\n", "Drain row:", st.drainrow);
    printf("%20s %10.4f %s\n", "Min DEM elevation:", dmin, "m");
  }
  iteration_headings(cfg.module);
  phase("contexts, upload, set-up on the device");

  /* block loop, WDPMCL.c:1049-1377 */
  struct timeval t0;
  gettimeofday(&t0, NULL);
  const int write_scratch = !is_null_name(cfg.scratch);
  scratch_writer scratch;
  if (write_scratch) scratch_start(&scratch, cfg.scratch, &hdr, st.R, st.C);
  int k = 0, done = 0;
  while (!done) {
    double max_diff = 0, diffdrain = 0, final_vol = 0;
    ABI_TRY(wdpm_group_run_block(ctx, ITER_PER_BLOCK, thres, &max_diff));
    k += ITER_PER_BLOCK;
    if (cfg.module == WDPM_DRAIN) {
      double final_sum = 0;
      ABI_TRY(wdpm_group_drain_stats(ctx, &diffdrain, &final_sum));
      diffdrain *= st.cellarea;                                                /* :1258 */
      final_vol = final_sum * st.cellarea;                                     /* :1267 */
      printf("%7s %d %7s %8.3f %5s %10.1f %5s %12.1f %5s %8.2f\n", "", k, "", max_diff, "", diffdrain, "",
             final_vol, "", seconds_since(&t0));
    } else {
      printf("%7s %d %7s %8.3f %5s %8.2f\n", "", k, "", max_diff, "", seconds_since(&t0));
    }
    done = max_diff <= eltol;                                                  /* :1324,1350 */
    if (cfg.module == WDPM_DRAIN && diffdrain < draintol) done = 1;            /* :1287,1303 */
    if (cfg.iteration_limit > 0 && k >= cfg.iteration_limit) done = 1;
    if (!done && write_scratch) {                                              /* checkpoint, :1290-1372 */
      /* un-padded (and, for add, NODATA-masked :1336-1344) on the device, straight into the file raster */
      ABI_TRY(wdpm_group_download_unpadded(ctx, cfg.module == WDPM_ADD, st.water));
      scratch_submit(&scratch, st.water);
    }
  }
  phase("block loop");
  if (write_scratch) scratch_finish(&scratch);
  phase("wait for checkpoint writer");

  /* final statistics, WDPMCL.c:1379-1467, next to the raster: the sequential volume sum (:1405-1412, exact),
   * the wet-cell count (:1397-1404) and the maximum (:1448-1457); the raster comes down un-padded and masked
   * (:1379-1392) for the output file only */
  double totaldrain = 0, watertotal = 0, dev_max = 0;
  int64_t wet = 0;
  ABI_TRY(wdpm_group_drain_stats(ctx, NULL, &watertotal));
  ABI_TRY(wdpm_group_count_stats(ctx, NULL, &wet, &dev_max));
  ABI_TRY(wdpm_group_download_unpadded(ctx, 1, st.water));
  if (cfg.module == WDPM_DRAIN) ABI_TRY(wdpm_group_get_totaldrain(ctx, &totaldrain));
  int64_t guard_bad = 0;
  if (getenv("WDPM_GUARD_KB")) {   /* debugging aid (include/wdpm.h: WDPM_OPT_GUARD_BAD): did any kernel write outside its buffer? */
    for (int i = 0; i < ndev; i++) {
      int64_t bad = 0;
      ABI_TRY(wdpm_get_option(wdpm_rank_ctx(wdpm_group_rank(ctx, i)), WDPM_OPT_GUARD_BAD, &bad));
      guard_bad += bad;
    }
    fprintf(stderr, "WDPMCL: guard bands of %d slab%s: %lld bytes overwritten\n", ndev, ndev == 1 ? "" : "s", (long long)guard_bad);
  }
  wdpm_group_destroy(ctx);
  if (guard_bad) return 3;
  phase("statistics + download + destroy");

  const int watercount = (int)wet;
  const double final_vol = watertotal * st.cellarea;
  const double meanwater = watertotal / ((float)watercount);
  const double waterfrac = (float)watercount / (float)st.basincount;
  double drainvol = 0, draindepth = 0;
  if (cfg.module == WDPM_DRAIN) {
    drainvol = totaldrain * st.cellarea;
    draindepth = (drainvol / ((float)st.basincount * st.cellarea)) * 1000;
  }
  /* the reference seeds the maximum with water[0][0] of the masked raster and lets `>` decide, so a NaN cell
   * never wins unless it is that seed: the device's maximum (from -inf) is folded into the seed the same way */
  double maxdepth = st.water[0];
  if (dev_max > maxdepth) maxdepth = dev_max;
  maxdepth = maxdepth * 1000;

  printf("%s\n", "                     ");                                      /* :1832-1857 */
  printf("%30s\n", "WDPM run summary");
  printf("%20s %10.2f %s\n", "Initial volume", st.initial_vol, "m3");
  printf("%20s %10.2f %s\n", "Final volume", final_vol, "m3");
  if (cfg.module == WDPM_DRAIN) {
    printf("%20s %10.2f %s\n", "Volume change", st.initial_vol - final_vol, "m3");
    printf("%20s %10.2f %s\n", "Volume drained", drainvol, "m3");
  } else {
    printf("%20s %10.2f %s\n", "Volume change", final_vol - st.initial_vol, "m3");
  }
  printf("%20s %10.4f %s\n", "Final water coverage", waterfrac, "");
  printf("%20s %10.2f %s\n", "Mean water depth", meanwater * 1000., "mm");
  if (cfg.module == WDPM_DRAIN) printf("%20s %10.2f %s\n", "Depth drained", draindepth, "mm ");
  printf("%20s %10.2f %s\n", "Max water depth", maxdepth, "mm ");

  asc_write_grid(cfg.output, &hdr, st.R, st.C, st.water);
  phase("write output raster");
  relief_finish(&relief);
  printf("%20s %10.2f %s\n", "Run Time", seconds_since(&t0), "s");
  free(st.dem);
  if (pinned) wdpm_host_free(st.water); else free(st.water);
  return 0;
}
