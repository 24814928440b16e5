"""bench.py prints ONE JSON line with the fields the driver reads (metric, value, unit, n_gpus, steps,
warmup, ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data, config.workload) plus the
`roofline` and `cpu_baseline` objects; checked on a small raster so the test takes seconds."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

gpu = pytest.mark.gpu


def run_bench(*args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


@gpu
def test_bench_line_has_the_contract_fields():
    d = run_bench("--size", "1024", "--steps", "7", "--warmup", "2", "--cpu-baseline-quick")
    assert d["metric"] == "cell-updates/sec on Add module, 1024x1024 DEM" and d["unit"] == "cell-updates/s"
    assert d["n_gpus"] == 1 and d["steps"] == 7 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "strong" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 1024 * 1024 * 7 / (d["ms_per_step"] * 7e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    assert list(r)[:2] == ["bound", "job_frac"] and 0 < r["job_frac"] <= r["frac_all_launches"] * 1.001
    # a short run also reports the same K iterations once more, after the timed block and beside it (the card's clocks have come up by then)
    nb = r["next_block"]
    assert 0 < nb["job_frac"] < 1 and nb["ms_per_step"] > 0 and nb["kernel_ms_per_iteration"] > 0 and "not `value`" in nb["what"]
    assert nb["untimed_iterations_before"] > 100        # 1024^2: microseconds per iteration, 50 ms of ramp
    # counter evidence (HBM bytes, VALU issue share) is only quoted for the configuration it was measured on
    assert r["traffic"] is None and r["hbm_real_frac"] is None and r["valu_issue_frac"] is None and r["counters_source"] is None
    # the dominant kernel is the plain instance (launches 2 .. K-1); the all-launch average (flush-on-load first launch,
    # max-diff last launch included) is reported beside it
    assert r["kernel_ms_per_iteration"] > 0 and r["kernel_ms_per_iteration_all_launches"] > 0 and "launches 2" in r["kernel"]
    assert abs(r["moved_frac"] * r["peak"] * r["kernel_ms_per_iteration"] * 1e6 - r["moved_bytes_per_launch"]) < 1e-6 * r["moved_bytes_per_launch"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] == 1 and c["value"] > 0 and c["unit"] == "cell-updates/s"
    assert "iterations" in c["sample"] and c["cpu_model"] and c["host_cores"] >= 1
    # the headline sample is on the bench's own raster (16384^2 x 2 for the default command, x 10 behind --cpu-baseline-full),
    # SURVEY §8d's 4096^2 and 1024^2 samples beside it; --cpu-baseline-quick (this test) runs the same three for a few iterations
    assert [x["size"] for x in c["samples"]] == [1024, 4096, 1024] and c["value"] == c["samples"][0]["value"] and "quick" in c
    assert "the metric's own raster" in c["sample"]


def test_cpu_baseline_samples_of_the_default_command():
    """which samples the un-flagged bench times (no GPU, nothing is run): the metric's own raster first"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_samples", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    seen = []
    mod.reference_baseline = lambda samples: (seen.append(list(samples)) or [{"size": n, "iterations": k, "seconds": 1.0, "value": 1.0} for n, k in samples])
    assert mod.cpu_baseline(16384)["samples"][0]["size"] == 16384 and seen[-1] == [(16384, 5), (4096, 24), (1024, 1000)]
    mod.cpu_baseline(16384, full=True)
    assert seen[-1][0] == (16384, 10)
    mod.cpu_baseline(1024)
    assert seen[-1][0] == (1024, 1000)


def test_counter_evidence_is_quoted_for_the_headline_configuration_and_build():
    """profiles/traffic.json: HBM bytes and VALU issue share per launch of the dominant kernel, from committed rocprofv3
    PMC passes of the default command - what `roofline.traffic / hbm_real_frac / valu_issue_frac` quote, and ONLY for the
    library build the passes ran on (wdpm_build_info(): a hash of the kernel sources).  No GPU needed."""
    import importlib.util
    import wdpm_amd
    spec = importlib.util.spec_from_file_location("bench_counters", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not os.path.exists(wdpm_amd.capi.HIP_LIB_PATH):      # a checkout that has built the oracle only
        pytest.skip("the HIP library is not built here: nothing to tie the counters to")
    lib = wdpm_amd.load_hip()
    build = lib.dll.wdpm_build_info().decode()
    assert build.startswith("kernels=") and " arch=" in build and " sched=" in build
    with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
        t = json.load(f)
    c = mod.measured_counters(lib, 16384, 1, "auto", "dem16")
    if t.get("dem16", {}).get("build_info") == build:
        assert 0.8 * 18 * 16384 ** 2 < c["traffic"] < 24 * 16384 ** 2 and 0.5 < c["valu_issue_frac"] < 1.0 and c["kernel_ms_at_collection"] > 0
        assert "fused_iteration_kernel" in c["kernel_name"] and c["build_info"] == build
        assert os.path.exists(os.path.join(ROOT, c["source"].split(" ")[0]))
    else:     # the kernels have changed since the passes: nothing may be quoted
        assert set(c) == {"mismatch"} and build in c["mismatch"]
    assert mod.measured_counters(lib, 4096, 1, "auto", "dem16") == {} and mod.measured_counters(lib, 16384, 2, "auto", "dem16") == {}

    class OtherBuild:          # a library built from other kernel sources never gets these counters
        class dll:
            @staticmethod
            def wdpm_build_info():
                return b"kernels=0123456789abcdef arch=gfx950 sched=max-ilp"
    assert set(mod.measured_counters(OtherBuild, 16384, 1, "auto", "dem32")) == {"mismatch"}


@gpu
def test_bench_drain_line():
    d = run_bench("--module", "drain", "--size", "1024", "--steps", "5", "--warmup", "1", "--drain-spinup", "20",
                  "--no-cpu-baseline")
    assert d["metric"].startswith("cell-updates/sec on Drain module") and d["value"] > 0 and "cpu_baseline" not in d


@gpu
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: the ranks are child processes of bench.py (one GPU here, so
    they share it: host-staged halos over gloo - the driver's multi-GPU boxes get RCCL halos), ONE line out"""
    d = run_bench("--gpus", "2", "--size", "1536", "--steps", "12", "--warmup", "3", "--exchange-every", "2")
    assert d["n_gpus"] == 2 and d["steps"] == 12 and d["value"] > 0
    c = d["config"]
    assert c["driver"] == "ranks" and c["halo"] == "host" and c["dist_backend"] == "gloo" and c["rccl_ranks"] is None
    assert c["exchange_every"] == 2 and c["exchange_every_requested"] == 2
    assert "cpu_baseline" not in d and "degraded" not in d
    one = run_bench("--size", "1536", "--steps", "12", "--warmup", "3", "--no-cpu-baseline")
    assert one["config"]["max_diff_m"] == c["max_diff_m"]          # same block, same bits


@gpu
def test_bench_group_driver():
    """--driver group: all ranks inside bench.py's process, one host thread per slab (what WDPMCL does)"""
    d = run_bench("--gpus", "3", "--driver", "group", "--size", "1536", "--steps", "12", "--warmup", "3",
                  "--exchange-every", "3")
    c = d["config"]
    assert d["n_gpus"] == 3 and c["driver"] == "group" and c["halo"] == "peer"
    assert c["enqueue_us_per_iteration_per_rank"] > 0


@gpu
def test_bench_ranks_fall_back_together_when_real_rccl_refuses():
    """Two ranks on the ONE GPU of the test box with RCCL halos forced: both processes take the id rank 0 made, both call
    ncclCommInitRank of the real RCCL - which refuses two ranks on one device - and then ALL ranks switch to host-staged halos
    together and finish.  (As far as one GPU can take the process-per-rank RCCL path.)"""
    env = dict(os.environ, WDPM_DIST_BACKEND="gloo", WDPM_HALO="rccl")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "1536", "--steps", "12", "--warmup", "3",
                        "--exchange-every", "2", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])
    assert "RCCL halos are not to be had" in p.stderr and "CommInitRank" in p.stderr
    assert d["n_gpus"] == 2 and d["config"]["halo"].startswith("host") and d["value"] > 0
    assert d["degraded"] is True and "CommInitRank" in d["degraded_reason"]      # not to be mistaken for a GPU-direct result
    one = run_bench("--size", "1536", "--steps", "12", "--warmup", "3", "--no-cpu-baseline")
    assert one["config"]["max_diff_m"] == d["config"]["max_diff_m"]


@gpu
def test_bench_launcher_starts_the_ranks_again_when_the_nccl_backend_fails():
    """the same two ranks with torch.distributed's nccl backend forced as well: that backend itself fails on a shared device,
    the rank processes die - and bench.py's own launcher starts them once more with host-staged halos over gloo"""
    env = dict(os.environ, WDPM_DIST_BACKEND="nccl", WDPM_HALO="rccl", WDPM_BENCH_RANKS_TIMEOUT="200")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "1536", "--steps", "12", "--warmup", "3",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "starting them once more with host-staged halos over gloo" in p.stderr and "budget" in p.stderr
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])
    assert d["n_gpus"] == 2 and d["config"]["halo"] == "host" and d["config"]["dist_backend"] == "gloo" and d["value"] > 0
    assert d["degraded"] is True and "exit status" in d["first_attempt"]


@gpu
def test_bench_reports_the_exchange_interval_the_library_used():
    """slabs too short for the interval asked for: the library shrinks it (every rank the same way) and the line says both"""
    d = run_bench("--gpus", "4", "--size", "256", "--steps", "6", "--warmup", "1", "--exchange-every", "12", "--no-cpu-baseline")
    c = d["config"]
    assert d["n_gpus"] == 4 and c["exchange_every_requested"] == 12 and 1 <= c["exchange_every"] < 12
