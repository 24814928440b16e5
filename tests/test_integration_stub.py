"""INTEGRATION.md §2 is code a maintainer is told to paste into the reference's WDPMCL.c.  This test takes that C block
VERBATIM out of the markdown, puts the reference's globals (src/WDPMCL.c:235-239: double** rasters, numrows, ...) around it
in a small harness, compiles it against include/wdpm.h, and runs it: on the CPU against the oracle's implementation of
the same ABI, on the GPU box against libwdpm_hip.so - results bit-identical to driving the ABI directly."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ORACLE_SO, ROOT
from helpers import n_bit_diff, pad, random_case

HARNESS = r'''
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
/* the reference's globals (src/WDPMCL.c:235-239) */
double **bigdem, **bigwater;
double missingvalue, totaldrain;
int numrows, numcols, drainrow, draincol;

%(setup)s

static double **rows_of(size_t nr, size_t nc) {
  double **p = malloc(nr * sizeof *p);
  for (size_t i = 0; i < nr; i++) p[i] = malloc(nc * sizeof **p);
  return p;
}

/* argv: module R C missing drainrow draincol blocks IterationNum thres cellarea in.bin out.bin */
int main(int argc, char **argv) {
  if (argc != 13) return 2;
  const int module = atoi(argv[1]);
  numrows = atoi(argv[2]); numcols = atoi(argv[3]); missingvalue = atof(argv[4]);
  drainrow = atoi(argv[5]); draincol = atoi(argv[6]);
  const int blocks = atoi(argv[7]), IterationNum = atoi(argv[8]);
  const double thres = atof(argv[9]), cellarea = atof(argv[10]);
  const char *activity = module == 2 ? "drain" : (module == 1 ? "subtract" : "add");
  const size_t nr = numrows + 2, nc = numcols + 2;
  bigdem = rows_of(nr, nc); bigwater = rows_of(nr, nc);
  FILE *f = fopen(argv[11], "rb");
  if (!f) return 3;
  for (size_t i = 0; i < nr; i++) if (fread(bigdem[i], sizeof(double), nc, f) != nc) return 3;
  for (size_t i = 0; i < nr; i++) if (fread(bigwater[i], sizeof(double), nc, f) != nc) return 3;
  if (fread(&totaldrain, sizeof(double), 1, f) != 1) return 3;
  fclose(f);
  int cpu = 2, k = 0;
  double max_diff = 0, diffdrain = 0, final_vol = 0;
  hip_setup(module);
  FILE *o = fopen(argv[12], "wb");
  for (int b = 0; b < blocks; b++) {
    if (cpu == 0) {
    }
%(loop)s
    fwrite(&max_diff, sizeof(double), 1, o);
  }
  for (size_t i = 0; i < nr; i++) memcpy(bigwater[i], flat_water + i * nc, nc * sizeof(double));   /* "then un-flatten" */
  for (size_t i = 0; i < nr; i++) fwrite(bigwater[i], sizeof(double), nc, o);
  const double tail[4] = {totaldrain, diffdrain, final_vol, (double)k};
  fwrite(tail, sizeof(double), 4, o);
  fclose(o);
  return 0;
}
'''


def stub_parts():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```c\n(.*?)```", md, re.S).group(1)
    cut = block.index("/* inside while(done == false)")
    return block[:cut], block[cut:]


def build(tmp_path, libdir, libname):
    setup, loop = stub_parts()
    src = tmp_path / "stub_harness.c"
    src.write_text(HARNESS % {"setup": setup, "loop": loop})
    exe = tmp_path / f"stub_{libname}"
    subprocess.check_call(["gcc", "-O1", "-Wall", "-Werror", "-Wno-unused-variable", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), "-o", str(exe),
                           str(src), "-L", libdir, f"-l{libname}", f"-Wl,-rpath,{libdir}", "-lm"])
    return exe


def run_stub(exe, tmp_path, module, bd, bw, miss, blocks, iters, thres, cellarea, drain=(0, 0), totaldrain=0.0):
    inp, out = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(np.ascontiguousarray(bd).tobytes())
        f.write(np.ascontiguousarray(bw).tobytes())
        f.write(np.float64(totaldrain).tobytes())
    R, C = bd.shape[0] - 2, bd.shape[1] - 2
    subprocess.check_call([str(exe), str(module), str(R), str(C), repr(miss), str(drain[0]), str(drain[1]), str(blocks), str(iters),
                           repr(thres), repr(cellarea), str(inp), str(out)])
    raw = np.fromfile(out, dtype=np.float64)
    md = raw[:blocks]
    water = raw[blocks:blocks + bd.size].reshape(bd.shape)
    return md, water, raw[blocks + bd.size:]


def direct(lib, module, bd, bw, miss, blocks, iters, thres, cellarea, drain=(0, 0), totaldrain=0.0):
    R, C = bd.shape[0] - 2, bd.shape[1] - 2
    kw = dict(drainrow=drain[0], draincol=drain[1]) if module == "drain" else {}
    with lib.context(module=module, nrows=R, ncols=C, missingvalue=miss, **kw) as ctx:
        ctx.upload(bd, bw)
        ctx.totaldrain = totaldrain
        md, dd, fs = [], 0.0, 0.0
        for _ in range(blocks):
            md.append(ctx.run_block(iters, thres))
            if module == "drain":
                dd, fs = ctx.drain_stats()
        return np.array(md), ctx.download_water(), (ctx.totaldrain, dd * cellarea, fs * cellarea)


def check(lib, exe, tmp_path):
    dem, water, miss = random_case(77, 61, 83)
    bd, bw = pad(dem, water, miss)
    for module, code in (("add", 0), ("subtract", 1)):
        md, w, tail = run_stub(exe, tmp_path, code, bd, bw, miss, 3, 40, 5e-6, 100.0)
        emd, ew, _ = direct(lib, module, bd, bw, miss, 3, 40, 5e-6, 100.0)
        assert np.array_equal(md.view(np.uint64), emd.view(np.uint64)) and n_bit_diff(w, ew) == 0 and tail[3] == 120
    k = int(np.argmin(np.where(dem > miss, dem, np.inf)))
    dr, dc = k // dem.shape[1] + 1, k % dem.shape[1] + 1
    td0 = max(float(bw[dr, dc]), 0.0)
    md, w, tail = run_stub(exe, tmp_path, 2, bd, bw, miss, 2, 30, 5e-6, 25.0, (dr, dc), td0)
    emd, ew, (etd, edd, efs) = direct(lib, "drain", bd, bw, miss, 2, 30, 5e-6, 25.0, (dr, dc), td0)
    assert np.array_equal(md.view(np.uint64), emd.view(np.uint64)) and n_bit_diff(w, ew) == 0
    assert (tail[0], tail[1], tail[2]) == (etd, edd, efs) and etd > td0


def test_integration_stub_compiles_and_runs_on_the_oracle_backend(oracle, tmp_path):
    exe = build(tmp_path, os.path.dirname(ORACLE_SO), "wdpm_oracle")
    check(oracle, exe, tmp_path)


@pytest.mark.gpu
def test_integration_stub_on_the_hip_library(hip, tmp_path):
    exe = build(tmp_path, os.path.join(ROOT, "wdpm_amd", "csrc"), "wdpm_hip")
    check(hip, exe, tmp_path)
