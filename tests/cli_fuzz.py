"""TEST INFRASTRUCTURE: random small WDPMCL jobs with oddly formatted ArcASCII files (CRLF line ends, tabs, upper-case
header keys, %g / %e / signed numbers, trailing separators, blank last line, NODATA 0 / -1 / -9999, 1 x 1 ... 14 x 17 cells,
all three modules, water file or NULL, scratch or NULL or an existing scratch raster to resume from, iteration limits,
arguments on the command line or in a parameter file), run through two executables whose exit code,
report text (minus the wall-clock column) and output files are compared.  Used by tests/test_cli_differential.py:
the UNMODIFIED reference executable (oracle/_ref/WDPMCL_ref) against the product's command line."""

import os, subprocess, sys, random, hashlib, re, shutil
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_golden import strip_timing

def fmt_val(rng, v, style):
    if style == 0: return "%.4f" % v
    if style == 1: return "%g" % v
    if style == 2: return "%.6e" % v
    if style == 3: return ("+" if v >= 0 else "") + "%.3f" % v
    return repr(float(v))

def write_asc(path, a, nodata, rng, style):
    R, C = a.shape
    eol = "\r\n" if style["crlf"] else "\n"
    sep = style["sep"]
    names = ["ncols", "nrows", "xllcorner", "yllcorner", "cellsize", "NODATA_value"]
    if style["upper"]: names = [n.upper() for n in names]
    vals = [str(C), str(R), style["xll"], style["yll"], style["cell"], style["nd"]]
    with open(path, "w", newline="") as f:
        for n, v in zip(names, vals):
            f.write(n + style["hsep"] + v + eol)
        for r in range(R):
            f.write(sep.join(fmt_val(rng, a[r, c], style["num"]) for c in range(C)) + (sep if style["trail"] else "") + eol)
        if style["blank"]: f.write(eol)

def one(seed, work, ref_exe, exe, vary_env=False):
    rng = random.Random(seed); nrng = np.random.default_rng(seed)
    rng_amt = random.Random(seed * 7 + 1).choice([5, 50, 250.5])
    R, C = (rng.randint(1, 14), rng.randint(1, 17)) if rng.random() < 0.85 else (rng.randint(20, 70), rng.randint(20, 90))
    nodata = rng.choice([-9999.0, -99999.0, -1.0, 0.0])
    dem = np.round(nrng.uniform(1, 30, (R, C)) + 400 * rng.random(), rng.choice([0, 2, 4]))
    miss = nrng.random((R, C)) < rng.choice([0, 0.1, 0.5])
    dem[miss] = nodata
    water = np.round(np.where(nrng.random((R, C)) < 0.5, nrng.uniform(0, 0.5, (R, C)), 0.0), 6)
    water[miss] = nodata
    style = dict(crlf=rng.random() < 0.2, sep=rng.choice([" ", "  ", "\t", " \t"]), upper=rng.random() < 0.3,
                 hsep=rng.choice([" ", "     ", "\t"]), xll=rng.choice(["0", "1234.5", "-77.25", "5.0e2"]),
                 yll=rng.choice(["0", "99.125", "-3"]), cell=rng.choice(["10", "10.0", "2.5", "1", "30.000"]),
                 nd=("%g" % nodata) if rng.random() < 0.5 else ("%.1f" % nodata), num=rng.randint(0, 4),
                 trail=rng.random() < 0.5, blank=rng.random() < 0.3)
    outs = []
    module = rng.choice(["add", "subtract", "drain"])
    usewater = module == "drain" or rng.random() < 0.5
    scratch = rng.random() < 0.3
    limit = rng.choice([0, 1000, 3000])
    resume = scratch and rng.random() < 0.5
    as_param_file = rng.random() < 0.2
    for name, exe in (("ref", ref_exe), ("new", exe)):
        d = os.path.join(work, name); shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
        write_asc(os.path.join(d, "dem.asc"), dem, nodata, rng, style)
        if usewater: write_asc(os.path.join(d, "w.asc"), water, nodata, rng, dict(style, num=0))
        w = "w.asc" if usewater else "NULL"; s = "s.asc" if scratch else "NULL"
        if module == "add": args = ["add", "dem.asc", w, "out.asc", s, str(rng_amt), "0.8", "1.0", "0", "0", "0.005", str(limit)]
        elif module == "subtract": args = ["subtract", "dem.asc", w, "out.asc", s, str(rng_amt), "1.0", "0", "0", "0.005", str(limit)]
        else: args = ["drain", "dem.asc", w, "out.asc", s, "1.0", "1.0", "0", "0", "0.005", str(limit)]
        if resume:      # a scratch raster left by an earlier run: the reference resumes from it (WDPMCL.c:682-725)
            write_asc(os.path.join(d, "s.asc"), np.where(miss, nodata, np.round(water * 0.5, 6)), nodata, rng, dict(style, num=0))
        argv = args
        if as_param_file:    # one value per line in a file, the file as the only argument (WDPMCL.c:334-343)
            with open(os.path.join(d, "params.txt"), "w") as f:
                f.write("\n".join(args) + "\n")
            argv = ["params.txt"]
        env = dict(os.environ)
        for k in ("WDPM_DEVICES", "WDPM_GPUS", "WDPM_EXCHANGE_EVERY", "WDPM_SCRATCH_BINARY", "WDPM_IO_THREADS"):
            env.pop(k, None)
        if vary_env and name == "new":      # knobs of the product that must never show in the results
            erng = random.Random(seed * 13 + 5)
            if erng.random() < 0.6:
                env["WDPM_DEVICES"] = ",".join(["0"] * erng.randint(2, 5))
                env["WDPM_EXCHANGE_EVERY"] = str(erng.randint(1, 5))
            if erng.random() < 0.4:
                env["WDPM_SCRATCH_BINARY"] = "1"
            env["WDPM_IO_THREADS"] = str(erng.choice([1, 2, 5]))
            env["WDPM_HOST_PAR_MIN"] = "1"
        try:
            p = subprocess.run([exe] + argv, cwd=d, capture_output=True, text=True, timeout=60, errors="replace", env=env)
            rc, out = p.returncode, p.stdout
        except subprocess.TimeoutExpired:
            rc, out = "timeout", ""
        files = {}
        for fn in ("out.asc", "s.asc"):
            fp = os.path.join(d, fn)
            files[fn] = hashlib.sha256(open(fp, "rb").read()).hexdigest() if os.path.exists(fp) else None
        outs.append((rc, strip_timing(out), files, args))
    a, b = outs
    ok = a[0] == b[0] and a[1] == b[1] and a[2] == b[2]
    return ok, a, b, style, (R, C, nodata, module)



def first_difference(a, b):
    la, lb = a.splitlines(), b.splitlines()
    for i in range(max(len(la), len(lb))):
        x = la[i] if i < len(la) else "<none>"
        y = lb[i] if i < len(lb) else "<none>"
        if x != y:
            return f"reference: {x!r}  here: {y!r}"
    return ""
