#!/usr/bin/env python3
"""profiles/traffic.json from a tools/profile.sh session: make_traffic_json.py <gpurun_out/dir with pmc_summary.json> <profiles/rNN/name_pmc_summary.json>
Copies the summary to the second path and rewrites the entry of profiles/traffic.json that matches how the profiled run streamed the
DEM ("dem16" / "dem32" / "fp64_dem"): HBM bytes per launch of the dominant kernel (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, KiB), the
VALU issue share, the kernel's name and duration in the counter pass, and the library build the passes ran on (wdpm_build_info())."""
import json, os, re, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
src, dst = sys.argv[1], sys.argv[2]
summ = json.load(open(os.path.join(src, "pmc_summary.json")))
fused = {k: v for k, v in summ.items() if "fused_iteration_kernel" in k and "SQ_INSTS_VALU" in v}
name = max(fused, key=lambda k: fused[k]["SQ_INSTS_VALU"]["n"])          # the instance with the most launches: the plain one
e = fused[name]
m = re.search(r"fused_iteration_kernel<\s*(\d)\s*,\s*(\w+)\s*,\s*(\w+)", name)
d32 = m.group(3)
kind = {"2": "dem16", "1": "dem32", "true": "dem32", "0": "fp64_dem", "false": "fp64_dem"}[d32]
import wdpm_amd
build = wdpm_amd.load_hip().dll.wdpm_build_info().decode()
fetch, write = e["FETCH_SIZE"]["mean"], e["WRITE_SIZE"]["mean"]
valu = e["SQ_INSTS_VALU"]["mean"] * 4.0 / 1024.0 / (e["GRBM_GUI_ACTIVE"]["mean"] / 8.0)
shutil.copy(os.path.join(src, "pmc_summary.json"), os.path.join(root, dst))
path = os.path.join(root, "profiles", "traffic.json")
t = json.load(open(path))
t["note"] = ("HBM bytes per fused-kernel launch from separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `python bench.py --steps 10 --warmup 2` "
             "(gfx950: FETCH_SIZE x2, calibrated on kernels of known byte count in round 1; counters in KiB); valu_issue_frac = SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / "
             "(GRBM_GUI_ACTIVE / 8 XCDs) from the --pmc SQ_* pass (a share of the kernel's cycles in that pass), kernel_ms = the kernel's mean duration in that pass "
             "(profiled passes run at a lower clock than the un-profiled bench); build_info = wdpm_build_info() of the library the passes ran on: bench.py quotes an "
             "entry only for a library that says the same (tools/make_traffic_json.py writes this file)")
t[kind] = {"kernel": name, "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write, "fetch_correction": 2.0,
           "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0, "source": dst, "SQ_INSTS_VALU": e["SQ_INSTS_VALU"]["mean"],
           "GRBM_GUI_ACTIVE": e["GRBM_GUI_ACTIVE"]["mean"], "valu_issue_frac": round(valu, 3), "kernel_ms": round(e["kernel_ms"]["mean"], 4),
           "build_info": build}
json.dump(t, open(path, "w"), indent=1)
print(kind, name, "hbm GB/launch %.3f" % ((2 * fetch + write) * 1024 / 1e9), "valu_issue_frac %.3f" % valu, "kernel_ms %.4f" % e["kernel_ms"]["mean"], build)
