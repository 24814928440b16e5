#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the UNMODIFIED reference.

Runs only where /root/reference exists (the build container).  It drives
oracle/_ref/libwdpm_ref.so (reference src/WDPMCL.c compiled as-is by oracle/Makefile, serial
functions runoffs/runoffd/drain on the reference's own globals) and oracle/_ref/WDPMCL_ref (the
reference executable, serial path cpu=0), and writes DATA only:

  stencil_cases.npz   small synthetic rasters: inputs + full-precision water after each of the 9
                      colour passes of the first iteration and after 1/10/100(/1000) iterations
  basin5.asc.gz       the reference's sample DEM (dem/basin5.asc), a data file
  basin5_state.npz    full-precision basin5 states (sha256 + sampled rows) after 1000/3000
                      iterations of add 100 mm / add 300 mm, and drain after 1000 iterations
  basin5_cli.json     WDPMCL_ref report lines (iterations, max diff, volumes) and sha256 of the
                      output rasters for the validation trio (validation/validate_WDPM.sh) and
                      for BASELINE configs 1-2

  full_size.npz       BASELINE configs 3, 4, 5 at FULL size on the synthetic DEMs (wdpm_synth_dem, seed = size):
                      4096^2 add x20, 16384^2 add x2, 8192^2 add x3 then drain x5 - sha256 of the padded water
                      raster, an 8-byte hash of every row, sampled rows, max diff and totaldrain

                      ... and (round 5) the SETTLED states: 4096^2 add x1000 and x2000 (two blocks, the flush between them),
                      16384^2 add x100 (x300, x1000), 8192^2 add x1000 then ONE drain block seen after 100 and 1000

    python tests/golden/make_golden.py [full]      ("full": only full_size.npz's early-state entries)
    python tests/golden/make_golden.py settled cfg3|cfg4|cfg5|cfg4long   (one process each, minutes to hours of a core)
    python tests/golden/make_golden.py merge       (fold the settled parts into full_size.npz)
    python tests/golden/make_golden.py verify-oracle   (the CPU restatement against config 3's settled entries: 20 minutes)
    python tests/golden/make_golden.py synthcli    (synth_cli.json: the reference EXECUTABLE's add / drain / subtract chain on a synthetic 3072^2 DEM)
"""
import ctypes as C
import gzip
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libwdpm_ref.so")
REF_EXE = os.path.join(ROOT, "oracle", "_ref", "WDPMCL_ref")
BASIN5 = "/root/reference/dem/basin5.asc"

ADD, SUBTRACT, DRAIN = 0, 1, 2


def load_ref():
    ref = C.CDLL(REF_SO)
    ref.ref_setup.argtypes = [C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_int]
    ref.ref_pass.argtypes = [C.c_int, C.c_int, C.c_int]
    ref.ref_iterate.argtypes = [C.c_int, C.c_int]
    ref.ref_get_water.argtypes = [C.c_void_p]
    ref.ref_get_totaldrain.restype = C.c_double
    return ref


def pad(dem, water, missing):
    R, Cc = dem.shape
    bd = np.full((R + 2, Cc + 2), missing, dtype=np.float64)
    bw = np.zeros((R + 2, Cc + 2), dtype=np.float64)
    bd[1:-1, 1:-1] = dem
    bw[1:-1, 1:-1] = water
    return bd, bw


def find_drain(bd):
    """WDPMCL.c:1005-1017: first (row-major, strict <) minimum among bigdem > 0."""
    best, pos = 100000000.0, (0, 0)
    R, Cc = bd.shape
    for i in range(R):
        for j in range(Cc):
            v = bd[i, j]
            if v > 0 and v < best:
                best, pos = v, (i, j)
    return pos


def ref_water(ref, shape):
    out = np.empty(shape, dtype=np.float64)
    ref.ref_get_water(out.ctypes.data)
    return out


def synth_case(rng, R, Cc, missing_frac, dry_frac, depth):
    missing = -99999.0
    y, x = np.mgrid[0:R, 0:Cc]
    dem = 500.0 + 2.0 * np.sin(x / 3.1) * np.cos(y / 2.3) + rng.normal(0, 0.4, (R, Cc)) - 0.01 * (x + y)
    dem = np.round(dem, 4)
    dem[rng.random((R, Cc)) < missing_frac] = missing
    water = np.where(rng.random((R, Cc)) < dry_frac, 0.0, depth * rng.random((R, Cc)))
    water = np.where(dem > missing, water, 0.0)
    return dem, water, missing


def make_stencil_cases(ref):
    rng = np.random.default_rng(20261003)
    out = {}
    index = []
    shapes = [(1, 1, 0.0, 0.0), (2, 5, 0.0, 0.2), (3, 3, 0.0, 0.0), (7, 9, 0.1, 0.3), (20, 17, 0.05, 0.3),
              (33, 31, 0.08, 0.5), (64, 50, 0.05, 0.2), (48, 200, 0.03, 0.1), (95, 130, 0.5, 0.6)]
    for ci, (R, Cc, mf, df) in enumerate(shapes):
        dem, water, missing = synth_case(rng, R, Cc, mf, df, 0.3)
        bd, bw = pad(dem, water, missing)
        for module in (ADD, DRAIN):
            name = f"c{ci}_m{module}"
            dr, dc = (0, 0)
            td0 = 0.0
            if module == DRAIN:
                if not (bd > 0).any():
                    continue
                dr, dc = find_drain(bd)
                td0 = max(bw[dr, dc], 0.0)
            out[name + "_dem"] = bd
            out[name + "_w0"] = bw
            meta = dict(name=name, R=R, C=Cc, module=module, missing=missing, drainrow=int(dr), draincol=int(dc),
                        td0=td0, stages=[])
            # nine single passes of the first iteration
            ref.ref_setup(R, Cc, missing, bd.ctypes.data, bw.ctypes.data, td0, dr, dc)
            k = 0
            for oi in ((1, 2, 3) if R * Cc <= 64 * 50 else ()):  # single passes only for the small cases
                for oj in (1, 2, 3):
                    ref.ref_pass(module, oi, oj)
                    k += 1
                    out[f"{name}_p{k}"] = ref_water(ref, bd.shape)
                    meta.setdefault("td_pass", []).append(ref.ref_get_totaldrain())
            # 1, 10, 100 (1000 for small) iterations from the initial state
            iters = [1, 10, 100] + ([1000] if R * Cc <= 64 * 50 else [])
            ref.ref_setup(R, Cc, missing, bd.ctypes.data, bw.ctypes.data, td0, dr, dc)
            done = 0
            for n in iters:
                ref.ref_iterate(module, n - done)
                done = n
                out[f"{name}_i{n}"] = ref_water(ref, bd.shape)
                meta["stages"].append(dict(iters=n, totaldrain=ref.ref_get_totaldrain()))
            index.append(meta)
    out["index_json"] = np.frombuffer(json.dumps(index).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "stencil_cases.npz"), **out)
    print("stencil_cases.npz:", len(index), "cases")


def read_asc(path):
    with open(path) as f:
        hdr = [f.readline().split() for _ in range(6)]
        vals = np.array(f.read().split(), dtype=np.float64)
    ncols, nrows = int(float(hdr[0][1])), int(float(hdr[1][1]))
    return vals.reshape(nrows, ncols), {h[0]: float(h[1]) for h in hdr}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_basin5_state(ref):
    dem, hdr = read_asc(BASIN5)
    missing = hdr["NODATA_VALUE"]
    R, Cc = dem.shape
    out, index = {}, []
    for module, add_mm, iters in ((ADD, 100.0, (1000, 3000)), (ADD, 300.0, (1000, 3000))):
        water = np.where(dem > missing, add_mm / 1000.0, 0.0)  # NULL water file, rof 1.0 (WDPMCL.c:779-792)
        bd, bw = pad(dem, water, missing)
        thres = 0.005 / 1000
        ref.ref_setup(R, Cc, missing, bd.ctypes.data, bw.ctypes.data, 0.0, 0, 0)
        done = 0
        for n in iters:
            while done < n:
                # block structure of WDPMCL.c:1055-1125: flush then 1000 iterations
                w = ref_water(ref, bd.shape)
                w[w < thres] = 0
                ref.ref_setup(R, Cc, missing, bd.ctypes.data, w.ctypes.data, 0.0, 0, 0)
                ref.ref_iterate(module, 1000)
                done += 1000
            w = ref_water(ref, bd.shape)
            name = f"add{int(add_mm)}_k{n}"
            out[name + "_rows"] = w[::7].copy()
            index.append(dict(name=name, add_mm=add_mm, iters=n, thres=thres, sha256=sha(w),
                              sum=float(w[bd > missing].sum()), max=float(w.max())))
            print(name, index[-1]["sha256"][:16])
    out["index_json"] = np.frombuffer(json.dumps(index).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "basin5_state.npz"), **out)


def run_cli(args, cwd):
    p = subprocess.run([REF_EXE] + [str(a) for a in args], cwd=cwd, capture_output=True, text=True)
    return p.returncode, p.stdout


BLOCK_RE = re.compile(r"^\s+(\d+)\s+(-?\d+\.\d+)(?:\s+(-?\d+\.\d+)\s+(-?\d+\.\d+))?\s+(\d+\.\d+)\s*$")


def parse_report(text):
    blocks, summary = [], {}
    for ln in text.splitlines():
        m = BLOCK_RE.match(ln)
        if m and "." in ln:
            g = m.groups()
            blocks.append([int(g[0]), g[1]] + ([g[2], g[3]] if g[2] is not None else []))
        m2 = re.match(r"^\s*(Initial volume|Final volume|Volume change|Volume drained|Final water coverage|"
                      r"Mean water depth|Depth drained|Max water depth|Drain column:|Drain row:|"
                      r"Min DEM elevation:|Basin area:|Initial volume:)\s+(-?[\d.]+)", ln)
        if m2:
            summary.setdefault(m2.group(1), m2.group(2))
    return blocks, summary


def file_sha(path):
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def make_basin5_cli():
    res = {}
    with tempfile.TemporaryDirectory() as td:
        dem = os.path.join(td, "basin5.asc")
        shutil.copy(BASIN5, dem)

        def record(key, args, outfile):
            rc, text = run_cli(args, td)
            blocks, summary = parse_report(text)
            res[key] = dict(args=[str(a) for a in args], rc=rc, blocks=blocks, summary=summary,
                            out_sha256=file_sha(os.path.join(td, outfile)),
                            report_sha256_nontiming=hashlib.sha256(strip_timing(text).encode()).hexdigest())
            print(key, rc, len(blocks), "blocks", summary.get("Final volume"))

        # the validation trio, validation/validate_WDPM.sh:77,88,99 with run_type=0 (serial)
        record("val_add10", ["add", "basin5.asc", "NULL", "a10.asc", "NULL", 10, 1.0, 1.0, 0, 0, 0.005, 0], "a10.asc")
        record("val_drain", ["drain", "basin5.asc", "a10.asc", "a10d.asc", "NULL", 0.1, 1.0, 0, 0, 0.005, 0], "a10d.asc")
        record("val_sub10", ["subtract", "basin5.asc", "a10d.asc", "a10s.asc", "NULL", 10, 1.0, 0, 0, 0.005, 0], "a10s.asc")
        # BASELINE config 1 (limit 3000) and config 2 (add 300 mm, limit 1000)
        record("cfg1_add100_k3000", ["add", "basin5.asc", "NULL", "a100.asc", "NULL", 100, 1.0, 1.0, 0, 0, 0.005, 3000], "a100.asc")
        record("cfg2_add300_k1000", ["add", "basin5.asc", "NULL", "a300.asc", "NULL", 300, 1.0, 1.0, 0, 0, 0.005, 1000], "a300.asc")
        # scratch/resume + water-file path: add 20 mm on top of a10.asc with a scratch file, limit 2000
        record("add20_on_a10_scratch", ["add", "basin5.asc", "a10.asc", "a30.asc", "scr.asc", 20, 0.5, 1.0, 0, 0, 0.005, 2000], "a30.asc")
        res["add20_on_a10_scratch"]["scratch_sha256"] = file_sha(os.path.join(td, "scr.asc"))
        # resume: the scratch file left by the previous run exists, so it replaces the water file and the
        # 20 mm are NOT added again (WDPMCL.c:668-673); three more blocks, two more scratch rewrites
        record("resume_from_scratch", ["add", "basin5.asc", "a10.asc", "a30r.asc", "scr.asc", 20, 0.5, 0.5, 0, 0, 0.005, 3000],
               "a30r.asc")
        res["resume_from_scratch"]["scratch_sha256"] = file_sha(os.path.join(td, "scr.asc"))
        # BASELINE config 2 run to convergence at 10 mm (paper/paper.md:89: 179 000 iterations); ~7 min here.
        # Only the last block, a hash of all block lines and the raster hash are kept.
        rc, text = run_cli(["add", "basin5.asc", "NULL", "a300c.asc", "NULL", 300, 1.0, 10.0, 0, 0, 0.005, 0], td)
        blocks, summary = parse_report(text)
        res["cfg2_add300_converged"] = dict(
            args=["add", "basin5.asc", "NULL", "a300.asc", "NULL", "300", "1.0", "10.0", "0", "0", "0.005", "0"], rc=rc,
            n_blocks=len(blocks), last_block=blocks[-1],
            blocks_sha256=hashlib.sha256(json.dumps(blocks).encode()).hexdigest(), summary=summary,
            out_sha256=file_sha(os.path.join(td, "a300c.asc")),
            report_sha256_nontiming=hashlib.sha256(strip_timing(text).encode()).hexdigest())
        # keep the 2x3 pothole patch the validation awk scripts sum (lines 268-269, fields 59-61)
        for key, fn in (("val_add10", "a10.asc"), ("val_drain", "a10d.asc"), ("val_sub10", "a10s.asc")):
            w, _ = read_asc(os.path.join(td, fn))
            res[key]["patch_sum"] = float(w[268 - 7:270 - 7, 58:61].sum())
        # usage / error exits
        for key, args in (("usage_none", []), ("usage_add", ["add"]), ("usage_badargc", ["add", "x", "y"])):
            rc, text = run_cli(args, td)
            res[key] = dict(args=args, rc=rc, stdout=text)
    with open(os.path.join(HERE, "basin5_cli.json"), "w") as f:
        json.dump(res, f, indent=1)


def make_synth_cli(n=3072):
    """The validation chain (validation/validate_WDPM.sh: add, drain, subtract) of the REFERENCE EXECUTABLE on a synthetic n x n DEM -
    a size at which the product's marching kernel runs (basin5 is relay-kernel sized): report lines, summary and the sha256 of each
    output raster.  The DEM is wdpm_synth_dem(n, seed n) written with four decimals; limits of 1000 / 1000 / 1000 iterations keep the
    reference at about six minutes a step.  -> synth_cli.json"""
    sys.path.insert(0, ROOT)
    import wdpm_amd
    gen = wdpm_amd.load(os.path.join(ROOT, "oracle", "_build", "libwdpm_oracle.so"))
    res = {"n": n}
    with tempfile.TemporaryDirectory() as td:
        dem = gen.synth_dem(n, n)
        with open(os.path.join(td, "dem.asc"), "w") as f:
            f.write(f"ncols {n}\nnrows {n}\nxllcorner 0\nyllcorner 0\ncellsize 10\nNODATA_value -99999\n")
            np.savetxt(f, dem, fmt="%.4f")
        res["dem_sha256_of_values"] = hashlib.sha256(np.ascontiguousarray(dem).tobytes()).hexdigest()

        def record(key, args, outfile):
            rc, text = run_cli(args, td)
            blocks, summary = parse_report(text)
            res[key] = dict(args=[str(a) for a in args], rc=rc, blocks=blocks, summary=summary,
                            out_sha256=file_sha(os.path.join(td, outfile)),
                            report_sha256_nontiming=hashlib.sha256(strip_timing(text).encode()).hexdigest())
            print(key, rc, blocks, summary, flush=True)

        record("add100", ["add", "dem.asc", "NULL", "a.asc", "NULL", 100, 1.0, 1.0, 0, 0, 0.005, 1000], "a.asc")
        record("drain", ["drain", "dem.asc", "a.asc", "d.asc", "NULL", 0.1, 1.0, 0, 0, 0.005, 1000], "d.asc")
        record("sub10", ["subtract", "dem.asc", "d.asc", "s.asc", "NULL", 10, 1.0, 0, 0, 0.005, 1000], "s.asc")
    with open(os.path.join(HERE, "synth_cli.json"), "w") as f:
        json.dump(res, f, indent=1)


def verify_oracle_settled():
    """The CPU restatement (oracle/wdpm_oracle.c) against the SETTLED golden of config 3: two blocks of 1000 iterations at 4096^2, the
    threshold flush between them - twenty minutes of one core, so not part of the test suite (which pins the oracle at 4096^2 x 20 and on
    the small vectors); run when the goldens are made.  Prints True True per block (max diff, sha256 of the padded raster)."""
    import time
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import wdpm_amd
    orc = wdpm_amd.load(os.path.join(ROOT, "oracle", "_build", "libwdpm_oracle.so"))
    z = np.load(os.path.join(HERE, "full_size.npz"))
    idx = {m["name"]: m for m in json.loads(bytes(z["index_json"]).decode())}
    n, missing, thres = 4096, -99999.0, 0.005 / 1000
    bd, bw = pad(orc.synth_dem(n, n), np.full((n, n), 0.1), missing)
    t = time.time()
    with orc.context(module="add", nrows=n, ncols=n, missingvalue=missing) as c:
        c.upload(bd, bw)
        for name in ("cfg3_add_4096_i1000", "cfg3_add_4096_b2_i2000"):
            md = c.run_block(1000, thres)
            print(name, md == idx[name]["max_diff"], sha(c.download_water()) == idx[name]["sha256"], f"{time.time() - t:.0f} s", flush=True)


def strip_timing(text):
    """Report text with the run-time column / Run Time line removed (they are wall-clock)."""
    out = []
    for ln in text.splitlines():
        m = BLOCK_RE.match(ln)
        if m:
            ln = ln[:ln.rstrip().rfind(" ")].rstrip()
        if ln.strip().startswith("Run Time"):
            continue
        out.append(ln.rstrip())
    return "\n".join(out)


def row_hashes(w):
    """first 8 bytes of the sha256 of every row: says WHICH rows differ when the whole-raster hash does"""
    return np.array([np.frombuffer(hashlib.sha256(r.tobytes()).digest()[:8], dtype=np.uint64)[0] for r in w],
                    dtype=np.uint64)


def make_full_size(ref):
    """BASELINE configs 3-5 at full size, from the unmodified reference's runoffs()/runoffd()/drain().
    About a minute of CPU and 12 GB of memory at 16384^2."""
    import time
    sys.path.insert(0, ROOT)
    import wdpm_amd  # only its ctypes binding of OUR C generator (wdpm_synth_dem), loaded from the oracle library
    gen = wdpm_amd.load(os.path.join(ROOT, "oracle", "_build", "libwdpm_oracle.so"))
    missing, thres = -99999.0, 0.005 / 1000
    out, index = {}, []

    def record(name, n, module, w, w_start, bd, td, secs, sample_every, **extra):
        valid = bd > missing
        d = np.abs(w - w_start)
        md = float(d[0, 0])
        dv = d[valid]
        if dv.size and float(dv.max()) > md:
            md = float(dv.max())                                    # WDPMCL.c:1239-1254
        out[name + "_rowhash"] = row_hashes(w)
        out[name + "_rows"] = w[::sample_every].copy()
        index.append(dict(name=name, n=n, module=module, sha256=sha(w), max_diff=md, totaldrain=td,
                          sample_every=sample_every, ref_seconds=round(secs, 1), **extra))
        print(name, index[-1]["sha256"][:16], "max_diff", md, "totaldrain", td, f"{secs:.1f} s")

    for name, n, add_iters, drain_iters, every in (("cfg3_add_4096_i20", 4096, 20, 0, 512),
                                                   ("cfg4_add_16384_i2", 16384, 2, 0, 4096),
                                                   ("cfg5_drain_8192_a3_d5", 8192, 3, 5, 2048),
                                                   # nine iterations: with the default exchange interval (k = 4) the halos are
                                                   # refreshed twice and a ninth iteration consumes the second refresh
                                                   ("cfg4_add_16384_i9", 16384, 9, 0, 4096),
                                                   ("cfg5_drain_8192_a3_d9", 8192, 3, 9, 2048)):
        dem = gen.synth_dem(n, n)
        bd, bw = pad(dem, np.full((n, n), 0.1), missing)            # add 100 mm, rof 1.0 on an empty water raster
        del dem
        bw[bw < thres] = 0                                          # the block's threshold flush (a no-op here)
        t = time.perf_counter()
        ref.ref_setup(n, n, missing, bd.ctypes.data, bw.ctypes.data, 0.0, 0, 0)
        ref.ref_iterate(ADD, add_iters)
        w = ref_water(ref, bd.shape)
        if not drain_iters:
            record(name, n, ADD, w, bw, bd, 0.0, time.perf_counter() - t, every, add_iters=add_iters)
            continue
        k = int(np.argmin(np.where(bd > 0, bd, np.inf)))            # WDPMCL.c:1005-1017 (first row-major minimum)
        dr, dc = k // (n + 2), k % (n + 2)
        td0 = max(float(w[dr, dc]), 0.0)                            # :1029
        w[w < thres] = 0
        ref.ref_setup(n, n, missing, bd.ctypes.data, w.ctypes.data, td0, dr, dc)
        ref.ref_iterate(DRAIN, drain_iters)
        w2 = ref_water(ref, bd.shape)
        record(name, n, DRAIN, w2, w, bd, ref.ref_get_totaldrain(), time.perf_counter() - t, every,
               add_iters=add_iters, drain_iters=drain_iters, drainrow=dr, draincol=dc, td0=td0,
               volume_sum=float(np.add.accumulate(w2[bd > missing])[-1]))
        del w2
    out["index_json"] = np.frombuffer(json.dumps(index).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "full_size.npz"), **out)


# ---------------------------------------------------------------------------------------------------------------
# SETTLED states at full size (round 5).  full_size.npz's first entries are 2-20 iterations from a uniform sheet;
# these are the BASELINE configurations as SURVEY 8d words them: config 3 = one 1000-iteration block at 4096^2
# (and a second block behind its threshold flush), config 4 = 16384^2 over 100 (and, hours of one core, 1000)
# iterations, config 5 = drain at 8192^2 from "that raster's add-100-mm state after 1000 iterations".  Each case is
# a process of its own (`settled CASE`), writes a part file next to this script's output and `merge` folds the
# parts into full_size.npz.  Reference lines followed: WDPMCL.c:1055-1125 (block), :1239-1254 (max diff).
SETTLED_CASES = ("cfg3", "cfg4", "cfg5", "cfg4long", "cfg3x")


def _max_diff(w, w0, bd, missing):
    """WDPMCL.c:1239-1254, in slabs of rows so that a 16384^2 raster needs no third full-size temporary."""
    md = float(abs(w[0, 0] - w0[0, 0]))
    for r in range(0, w.shape[0], 1024):
        d = np.abs(w[r:r + 1024] - w0[r:r + 1024])[bd[r:r + 1024] > missing]
        if d.size and float(d.max()) > md:
            md = float(d.max())
    return md


def _seq_sum(w, bd, missing):
    """final_vol's row-major sequential sum (WDPMCL.c:1262-1267) without the cell area"""
    acc = 0.0
    for r in range(w.shape[0]):
        v = w[r][bd[r] > missing]
        if v.size:
            acc = float(np.add.accumulate(np.concatenate(([acc], v)))[-1])
    return acc


def make_settled(ref, case):
    import time
    sys.path.insert(0, ROOT)
    import wdpm_amd
    gen = wdpm_amd.load(os.path.join(ROOT, "oracle", "_build", "libwdpm_oracle.so"))
    missing, thres = -99999.0, 0.005 / 1000
    out, index = {}, []
    part = os.path.join(HERE, f"_settled_{case}.npz")

    def record(name, n, module, w, w_start, bd, td, secs, every, **extra):
        out[name + "_rowhash"] = row_hashes(w)
        out[name + "_rows"] = w[::every].copy()
        index.append(dict(name=name, n=n, module=module, sha256=sha(w), max_diff=_max_diff(w, w_start, bd, missing),
                          totaldrain=td, sample_every=every, ref_seconds=round(secs, 1),
                          wet_cells=int(np.count_nonzero(w > 0)), deepest=float(w.max()), **extra))
        print(name, index[-1], flush=True)
        out["index_json"] = np.frombuffer(json.dumps(index).encode(), dtype=np.uint8)
        np.savez_compressed(part, **out)                            # after every record: a killed run keeps its work

    def start(n):
        dem = gen.synth_dem(n, n)
        bd, bw = pad(dem, np.full((n, n), 0.1), missing)            # add 100 mm, rof 1.0 on an empty water raster
        bw[bw < thres] = 0
        return bd, bw

    t = time.perf_counter()
    if case == "cfg3":
        n = 4096
        bd, bw = start(n)
        ref.ref_setup(n, n, missing, bd.ctypes.data, bw.ctypes.data, 0.0, 0, 0)
        ref.ref_iterate(ADD, 1000)
        w1 = ref_water(ref, bd.shape)
        record("cfg3_add_4096_i1000", n, ADD, w1, bw, bd, 0.0, time.perf_counter() - t, 512, add_iters=1000, blocks=[1000])
        flushed = int(np.count_nonzero((w1 < thres) & (w1 != 0)))
        w1[w1 < thres] = 0                                          # second block: :1055-1065 now does something
        ref.ref_setup(n, n, missing, bd.ctypes.data, w1.ctypes.data, 0.0, 0, 0)
        ref.ref_iterate(ADD, 1000)
        w2 = ref_water(ref, bd.shape)
        record("cfg3_add_4096_b2_i2000", n, ADD, w2, w1, bd, 0.0, time.perf_counter() - t, 512, add_iters=2000,
               blocks=[1000, 1000], flushed_before_block2=flushed)
    elif case in ("cfg4", "cfg4long"):
        n = 16384
        bd, bw = start(n)
        ref.ref_setup(n, n, missing, bd.ctypes.data, bw.ctypes.data, 0.0, 0, 0)
        done = 0
        for upto in ((100,) if case == "cfg4" else (300, 1000)):
            ref.ref_iterate(ADD, upto - done)
            done = upto
            w = ref_water(ref, bd.shape)
            record(f"cfg4_add_16384_i{upto}", n, ADD, w, bw, bd, 0.0, time.perf_counter() - t, 4096, add_iters=upto,
                   blocks=[upto])
            del w
    elif case == "cfg5":
        n = 8192
        bd, bw = start(n)
        ref.ref_setup(n, n, missing, bd.ctypes.data, bw.ctypes.data, 0.0, 0, 0)
        ref.ref_iterate(ADD, 1000)
        w = ref_water(ref, bd.shape)
        record("cfg5_add_8192_i1000", n, ADD, w, bw, bd, 0.0, time.perf_counter() - t, 2048, add_iters=1000, blocks=[1000])
        k = int(np.argmin(np.where(bd > 0, bd, np.inf)))            # WDPMCL.c:1005-1017 (first row-major minimum)
        dr, dc = k // (n + 2), k % (n + 2)
        td0 = max(float(w[dr, dc]), 0.0)                            # :1029
        flushed = int(np.count_nonzero((w < thres) & (w != 0)))
        w[w < thres] = 0
        ref.ref_setup(n, n, missing, bd.ctypes.data, w.ctypes.data, td0, dr, dc)
        done = 0
        for upto in (100, 1000):                                    # ONE drain block, looked at after 100 and 1000
            ref.ref_iterate(DRAIN, upto - done)
            done = upto
            w2 = ref_water(ref, bd.shape)
            record(f"cfg5_drain_8192_a1000_d{upto}", n, DRAIN, w2, w, bd, ref.ref_get_totaldrain(),
                   time.perf_counter() - t, 2048, add_iters=1000, drain_iters=upto, drainrow=dr, draincol=dc, td0=td0,
                   flushed_before_drain=flushed, volume_sum=_seq_sum(w2, bd, missing))
            del w2
    elif case == "cfg3x":
        # The regimes the all-wet configurations above never reach at full size (after 1000 iterations of add 100 mm every cell is
        # still wet and the deepest pond is 1.1 m): DEEP water (the clamped neighbour step's guard, > 3 m in a wave's window) and a
        # mostly DRY raster (dry tiles, water arriving in tiles that were skipped).  4096^2, 12 m of water (flows of 1.5 m: a clamped step would be WRONG here) on one 256 x 256 block in
        # eight (a fixed pattern: block (bi, bj) is wet when (3 bi + 5 bj) % 8 == 0), two blocks of 200 iterations.
        n = 4096
        bd, _ = start(n)
        bi, bj = np.mgrid[0:n, 0:n] // 256
        bw = np.zeros_like(bd)
        bw[1:-1, 1:-1] = np.where((3 * bi + 5 * bj) % 8 == 0, 12.0, 0.0)
        del bi, bj
        ref.ref_setup(n, n, missing, bd.ctypes.data, bw.ctypes.data, 0.0, 0, 0)
        ref.ref_iterate(ADD, 200)
        w1 = ref_water(ref, bd.shape)
        record("cfg3x_ponds_4096_i200", n, ADD, w1, bw, bd, 0.0, time.perf_counter() - t, 512, add_iters=200, blocks=[200],
               water_in="12.0 m where (3 * (i // 256) + 5 * (j // 256)) % 8 == 0 (file coordinates), else 0")
        flushed = int(np.count_nonzero((w1 < thres) & (w1 != 0)))
        w1[w1 < thres] = 0
        ref.ref_setup(n, n, missing, bd.ctypes.data, w1.ctypes.data, 0.0, 0, 0)
        ref.ref_iterate(ADD, 200)
        w2 = ref_water(ref, bd.shape)
        record("cfg3x_ponds_4096_b2_i400", n, ADD, w2, w1, bd, 0.0, time.perf_counter() - t, 512, add_iters=400,
               blocks=[200, 200], flushed_before_block2=flushed)
    else:
        sys.exit(f"unknown case {case}; one of {SETTLED_CASES}")


def merge_settled():
    """fold the parts written by `settled CASE` into full_size.npz (entries of the same name are replaced)"""
    path = os.path.join(HERE, "full_size.npz")
    z = np.load(path)
    out = {k: z[k] for k in z.files if k != "index_json"}
    index = json.loads(bytes(z["index_json"]).decode())
    for case in SETTLED_CASES:
        part = os.path.join(HERE, f"_settled_{case}.npz")
        if not os.path.exists(part):
            continue
        p = np.load(part)
        new = json.loads(bytes(p["index_json"]).decode())
        names = {m["name"] for m in new}
        index = [m for m in index if m["name"] not in names] + new
        out.update({k: p[k] for k in p.files if k != "index_json"})
        print("merged", case, sorted(names))
    out["index_json"] = np.frombuffer(json.dumps(index).encode(), dtype=np.uint8)
    np.savez_compressed(path, **out)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "full":
        make_full_size(load_ref())
        return
    if len(sys.argv) > 2 and sys.argv[1] == "settled":
        make_settled(load_ref(), sys.argv[2])
        return
    if len(sys.argv) > 1 and sys.argv[1] == "verify-oracle":
        verify_oracle_settled()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "synthcli":
        make_synth_cli()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "merge":
        merge_settled()
        return
    if not os.path.exists(REF_SO):
        sys.exit("build oracle/_ref first: make -C oracle ref")
    ref = load_ref()
    make_stencil_cases(ref)
    with open(BASIN5, "rb") as f, gzip.GzipFile(os.path.join(HERE, "basin5.asc.gz"), "wb", mtime=0) as g:
        g.write(f.read())
    make_basin5_state(ref)
    make_basin5_cli()
    make_full_size(ref)


if __name__ == "__main__":
    main()
