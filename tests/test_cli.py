"""The WDPMCL drop-in command line (wdpm_amd/csrc/wdpmcl_main.c) against what the unmodified
reference executable printed and wrote (tests/golden/basin5_cli.json): report text identical except
the wall-clock column, output and scratch rasters byte-identical, same exit codes.

CPU tests run the product's host code linked against the oracle back-end (oracle/_build/
WDPMCL_oracle, test-only); the gpu-marked tests run the shipped wdpm_amd/bin/WDPMCL on the HIP path."""
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

sys.path.insert(0, GOLDEN)
from make_golden import file_sha, parse_report, strip_timing  # noqa: E402

ORACLE_CLI = os.path.join(ROOT, "oracle", "_build", "WDPMCL_oracle")
HIP_CLI = os.path.join(ROOT, "wdpm_amd", "bin", "WDPMCL")


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(GOLDEN, "basin5_cli.json")) as f:
        return json.load(f)


@pytest.fixture()
def workdir(tmp_path):
    with gzip.open(os.path.join(GOLDEN, "basin5.asc.gz"), "rb") as f, open(tmp_path / "basin5.asc", "wb") as g:
        shutil.copyfileobj(f, g)
    return tmp_path


def run(exe, args, cwd):
    p = subprocess.run([exe] + [str(a) for a in args], cwd=cwd, capture_output=True, text=True, timeout=1500)
    return p.returncode, p.stdout, p.stderr


def check(exe, golden, key, cwd, outfile):
    g = golden[key]
    rc, out, err = run(exe, g["args"], cwd)
    assert rc == g["rc"], err
    blocks, summary = parse_report(out)
    assert blocks == g["blocks"]
    assert summary == g["summary"]
    assert hashlib.sha256(strip_timing(out).encode()).hexdigest() == g["report_sha256_nontiming"]
    assert file_sha(os.path.join(cwd, outfile)) == g["out_sha256"]
    return err


def chain(exe, golden, cwd):
    """validation/validate_WDPM.sh: add 10 mm -> drain -> subtract 10 mm, then water-file + scratch path"""
    check(exe, golden, "val_add10", cwd, "a10.asc")
    check(exe, golden, "val_drain", cwd, "a10d.asc")
    check(exe, golden, "val_sub10", cwd, "a10s.asc")
    check(exe, golden, "add20_on_a10_scratch", cwd, "a30.asc")
    assert file_sha(os.path.join(cwd, "scr.asc")) == golden["add20_on_a10_scratch"]["scratch_sha256"]
    # resume from that scratch file: several scratch rewrites by the writer thread, the one left behind
    # must be the state after the last non-final block, as with the reference
    check(exe, golden, "resume_from_scratch", cwd, "a30r.asc")
    assert file_sha(os.path.join(cwd, "scr.asc")) == golden["resume_from_scratch"]["scratch_sha256"]


def usage(exe, golden, cwd):
    for key in ("usage_none", "usage_add", "usage_badargc"):
        rc, out, _ = run(exe, golden[key]["args"], cwd)
        assert rc == 42 and out == golden[key]["stdout"]
    # drain without a water file: message + exit 42 (WDPMCL.c:983-988)
    rc, out, _ = run(exe, ["drain", "basin5.asc", "nope.asc", "o.asc", "NULL", 0.1, 1.0, 0, 0, 0.005, 0], cwd)
    assert rc == 42 and "Error water file missing" in out
    # add needs 13 arguments, subtract 12: the other count prints the usage after the banner
    rc, out, _ = run(exe, ["subtract", "basin5.asc", "NULL", "o.asc", "NULL", 10, 1.0, 1.0, 0, 0, 0.005, 0], cwd)
    assert rc == 42 and "Subtract module specified" in out and "Wetland DEM Ponding Model" in out


def param_file(exe, golden, cwd):
    g = golden["cfg2_add300_k1000"]
    with open(os.path.join(cwd, "params.txt"), "w") as f:
        f.write("\n".join(g["args"]) + "\n")
    rc, out, err = run(exe, ["params.txt"], cwd)
    assert rc == 0, err
    assert parse_report(out)[0] == g["blocks"]
    assert file_sha(os.path.join(cwd, "a300.asc")) == g["out_sha256"]


def test_cli_usage_and_exit_codes(workdir, golden):
    usage(ORACLE_CLI, golden, workdir)


def test_cli_validation_chain_on_oracle_backend(workdir, golden):
    chain(ORACLE_CLI, golden, workdir)


def test_cli_parameter_file(workdir, golden):
    param_file(ORACLE_CLI, golden, workdir)


@pytest.mark.gpu
def test_hip_cli_usage(workdir, golden):
    usage(HIP_CLI, golden, workdir)


@pytest.mark.gpu
def test_hip_cli_validation_chain(workdir, golden):
    """the reference's own end-to-end known answers (0.420810 m patch, 97577.54 / 86762.40 m3), via
    byte-identical rasters, on the HIP path"""
    chain(HIP_CLI, golden, workdir)


@pytest.mark.gpu
def test_hip_cli_baseline_configs(workdir, golden):
    err = check(HIP_CLI, golden, "cfg1_add100_k3000", workdir, "a100.asc")
    assert "hip-gfx950" in err
    param_file(HIP_CLI, golden, workdir)


@pytest.mark.gpu
def test_hip_cli_with_dem_codes_forced(workdir, golden):
    """basin5 is too small for the DEM codes to pay, so the library would not use them: WDPM_DEM32=2
    forces the code-streaming kernel and the reference's answers must not move"""
    env = dict(os.environ, WDPM_DEM32="2")
    for key, outfile in (("val_add10", "a10.asc"), ("cfg2_add300_k1000", "a300.asc")):
        g = golden[key]
        p = subprocess.run([HIP_CLI] + g["args"], cwd=workdir, capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, p.stderr
        blocks, summary = parse_report(p.stdout)
        assert blocks == g["blocks"] and summary == g["summary"]
        assert file_sha(os.path.join(workdir, outfile)) == g["out_sha256"]


def multi_device(exe, golden, cwd, devices, **extra_env):
    """WDPM_DEVICES spreads the raster over several contexts of one process (row blocks, halo copies):
    reports and rasters must not change"""
    env = dict(os.environ, WDPM_DEVICES=devices, WDPM_EXCHANGE_EVERY="3", **extra_env)
    for key, outfile in (("val_add10", "a10.asc"), ("val_drain", "a10d.asc"), ("val_sub10", "a10s.asc")):
        g = golden[key]
        p = subprocess.run([exe] + g["args"], cwd=cwd, capture_output=True, text=True, timeout=1500, env=env)
        assert p.returncode == 0, p.stderr
        assert f"{len(devices.split(','))} devices" in p.stderr
        blocks, summary = parse_report(p.stdout)
        assert blocks == g["blocks"] and summary == g["summary"]
        assert file_sha(os.path.join(cwd, outfile)) == g["out_sha256"]


def oversized_raster(exe, golden, cwd, quick=False):
    """A raster with more padded cells than one context takes (2e9; here the threshold is brought down to 60 000 by
    WDPM_MAX_SLAB_CELLS) is cut into row blocks on the same device without being asked: basin5's 228 932 padded cells ->
    4 slabs on device 0, 2 + 2 on `WDPM_DEVICES=0,0`; reports and rasters stay the reference's."""
    for devices, slabs in ((None, 4),) if quick else ((None, 4), ("0,0", 4)):
        env = dict(os.environ, WDPM_MAX_SLAB_CELLS="60000", WDPM_EXCHANGE_EVERY="3")
        env.pop("WDPM_DEVICES", None)
        env.pop("WDPM_GPUS", None)
        if devices:
            env["WDPM_DEVICES"] = devices
        for key, outfile in (("cfg2_add300_k1000", "a300.asc"),) if quick else (("val_add10", "a10.asc"), ("val_drain", "a10d.asc")):
            g = golden[key]
            p = subprocess.run([exe] + g["args"], cwd=cwd, capture_output=True, text=True, timeout=1500, env=env)
            assert p.returncode == 0, p.stderr
            assert f"{slabs} devices" in p.stderr, p.stderr
            blocks, summary = parse_report(p.stdout)
            assert blocks == g["blocks"] and summary == g["summary"]
            assert file_sha(os.path.join(cwd, outfile)) == g["out_sha256"]


def test_cli_cuts_an_oversized_raster_into_slabs_on_oracle_backend(workdir, golden):
    oversized_raster(ORACLE_CLI, golden, workdir, quick=True)      # one 1000-iteration job here; the GPU test runs two to convergence


@pytest.mark.gpu
def test_hip_cli_cuts_an_oversized_raster_into_slabs(workdir, golden):
    oversized_raster(HIP_CLI, golden, workdir)


def lossless_resume(exe, golden, cwd):
    """with WDPM_SCRATCH_BINARY=1 a checkpoint also leaves <scratch>.f64 and a resumed
    run continues bit-for-bit: 2000 iterations + checkpoint + 1000 resumed == 3000 uninterrupted (the
    reference's own golden output), which the 1e-6 m text scratch alone cannot reproduce."""
    g = golden["cfg1_add100_k3000"]
    env = dict(os.environ, WDPM_SCRATCH_BINARY="1")
    args = list(g["args"])
    args[4] = "ck.asc"
    p = subprocess.run([exe] + args, cwd=cwd, capture_output=True, text=True, timeout=1500, env=env)
    assert p.returncode == 0, p.stderr
    assert file_sha(os.path.join(cwd, "a100.asc")) == g["out_sha256"]          # checkpointing changes nothing
    assert os.path.exists(os.path.join(cwd, "ck.asc")) and os.path.exists(os.path.join(cwd, "ck.asc.f64"))
    assert not os.path.exists(os.path.join(cwd, "ck.asc.f64.tmp"))
    os.remove(os.path.join(cwd, "a100.asc"))
    args[-1] = "1000"                                                          # the checkpoint holds k = 2000
    p = subprocess.run([exe] + args, cwd=cwd, capture_output=True, text=True, timeout=1500, env=env)
    assert p.returncode == 0, p.stderr
    assert "full-precision checkpoint" in p.stderr
    blocks, _ = parse_report(p.stdout)
    assert [b[1:] for b in blocks] == [g["blocks"][2][1:]]                      # same max-diff as block 3 of the long run
    assert file_sha(os.path.join(cwd, "a100.asc")) == g["out_sha256"]
    # a sidecar that belongs to another checkpoint is ignored (it disagrees with the text scratch)
    import numpy as np
    raw = np.fromfile(os.path.join(cwd, "ck.asc.f64"), dtype=np.uint8)
    body = raw[32:].view(np.float64).copy()
    body[body > 0] += 1e-3
    with open(os.path.join(cwd, "ck.asc.f64"), "wb") as f:
        f.write(raw[:32].tobytes() + body.tobytes())
    args[4], args[3] = "ck.asc", "other.asc"
    shutil.copy(os.path.join(cwd, "ck.asc"), os.path.join(cwd, "keep.asc"))
    p = subprocess.run([exe] + args, cwd=cwd, capture_output=True, text=True, timeout=1500, env=env)
    assert p.returncode == 0 and "ignored" in p.stderr


def test_cli_lossless_checkpoint_resume_on_oracle_backend(workdir, golden):
    lossless_resume(ORACLE_CLI, golden, workdir)


@pytest.mark.gpu
def test_hip_cli_lossless_checkpoint_resume(workdir, golden):
    lossless_resume(HIP_CLI, golden, workdir)


def test_cli_set_up_and_statistics_across_slabs(workdir, golden):
    """set-up (module water, padding, basin count, drain-cell search, volumes) and the final statistics are done
    by the back-end next to the rasters, slab by slab when the raster is spread over several contexts, and the
    ArcASCII text is parsed / formatted on several host threads: the golden add run stays the reference's, and
    one block of every module gives the same report and the same bytes on three slabs + five I/O threads as on one"""
    env = dict(os.environ, WDPM_IO_THREADS="5", WDPM_DEVICES="0,0,0", WDPM_EXCHANGE_EVERY="2")
    g = golden["cfg2_add300_k1000"]
    p = subprocess.run([ORACLE_CLI] + g["args"], cwd=workdir, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr
    assert parse_report(p.stdout) == (g["blocks"], g["summary"])
    assert hashlib.sha256(strip_timing(p.stdout).encode()).hexdigest() == g["report_sha256_nontiming"]
    assert file_sha(os.path.join(workdir, "a300.asc")) == g["out_sha256"]
    runs = {"subtract": ["subtract", "basin5.asc", "a300.asc", "{}", "NULL", "40", "1.0", "0", "0", "0.005", "1000"],
            "drain": ["drain", "basin5.asc", "a300.asc", "{}", "NULL", "1.0", "1.0", "0", "0", "0.005", "1000"]}
    for name, args in runs.items():
        outs = {}
        for tag, e in (("t", env), ("s", dict(os.environ))):
            a = [x.format(f"{tag}.asc") for x in args]
            q = subprocess.run([ORACLE_CLI] + a, cwd=workdir, capture_output=True, text=True, timeout=600, env=e)
            assert q.returncode == 0, (name, q.stderr)
            outs[tag] = (strip_timing(q.stdout).replace(f"{tag}.asc", "X.asc"), file_sha(os.path.join(workdir, f"{tag}.asc")))
        assert outs["t"] == outs["s"], name


def test_cli_three_contexts_on_oracle_backend(workdir, golden):
    multi_device(ORACLE_CLI, golden, workdir, "0,0,0")


@pytest.mark.gpu
def test_hip_cli_four_contexts_on_one_gpu(workdir, golden):
    multi_device(HIP_CLI, golden, workdir, "0,0,0,0")


@pytest.mark.gpu
def test_hip_cli_three_contexts_with_dem_codes_forced(workdir, golden):
    """every slab encodes its own rows of the DEM (own offset k0); forced on, since basin5 slabs are small"""
    multi_device(HIP_CLI, golden, workdir, "0,0,0", WDPM_DEM32="2")


@pytest.mark.gpu
def test_hip_cli_add300_to_convergence(workdir, golden):
    """BASELINE config 2 all the way: basin5, add 300 mm, tolerance 10 mm, no iteration limit.  The
    unmodified reference needs 179 000 iterations (as paper/paper.md:89 reports) and 412 s on one core
    of the build container; the HIP path must print the same 179 progress lines and write the same
    bytes."""
    import time
    g = golden["cfg2_add300_converged"]
    t = time.time()
    rc, out, err = run(HIP_CLI, g["args"], workdir)
    dt = time.time() - t
    assert rc == 0, err
    blocks, summary = parse_report(out)
    assert len(blocks) == g["n_blocks"] == 179 and blocks[-1] == g["last_block"]
    assert hashlib.sha256(json.dumps(blocks).encode()).hexdigest() == g["blocks_sha256"]
    assert summary == g["summary"]
    assert hashlib.sha256(strip_timing(out).encode()).hexdigest() == g["report_sha256_nontiming"]
    assert file_sha(os.path.join(workdir, "a300.asc")) == g["out_sha256"]
    print(f"WDPMCL add 300 mm to convergence on the HIP path: {dt:.1f} s wall (reference serial: 412 s)")


def backend_line_and_gdaldem_handoff(exe, backend, workdir, golden):
    """opt-in conveniences for the GUI side.  WDPM_REPORT_BACKEND=1 adds ONE line to the report (and
    nothing else changes); WDPM_COLOR_RELIEF=<map> hands the output raster to `gdaldem color-relief` the way the
    reference's src/cmap_black.sh does (a stand-in `gdaldem` on PATH records how it was called)"""
    g = golden["cfg2_add300_k1000"]
    fake = os.path.join(workdir, "bin")
    os.makedirs(fake)
    with open(os.path.join(fake, "gdaldem"), "w") as f:
        f.write('#!/bin/sh\necho "$@" > "$(dirname "$0")/../gdaldem_args.txt"\ntouch "$6" "$6.aux.xml"\n')
    os.chmod(os.path.join(fake, "gdaldem"), 0o755)
    open(os.path.join(workdir, "colormap_black.txt"), "w").write("3,25,0,230\n0.001,25,0,230\n0,yellow\n-9999, black\n")
    env = dict(os.environ, WDPM_REPORT_BACKEND="1", WDPM_COLOR_RELIEF="colormap_black.txt",
               PATH=fake + os.pathsep + os.environ["PATH"])
    p = subprocess.run([exe] + g["args"], cwd=workdir, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr
    extra = [ln for ln in p.stdout.splitlines() if "Computation back-end:" in ln]
    assert len(extra) == 1 and f"{backend}, 1 device" in extra[0]
    rest = "\n".join(ln for ln in p.stdout.splitlines() if "Computation back-end:" not in ln) + "\n"
    assert hashlib.sha256(strip_timing(rest).encode()).hexdigest() == g["report_sha256_nontiming"]
    assert file_sha(os.path.join(workdir, "a300.asc")) == g["out_sha256"]
    assert open(os.path.join(workdir, "gdaldem_args.txt")).read().split() == \
        ["color-relief", "a300.asc", "colormap_black.txt", "-OF", "png", "a300.png"]
    assert os.path.exists(os.path.join(workdir, "a300.png")) and not os.path.exists(os.path.join(workdir, "a300.png.aux.xml"))
    assert "colour relief written to a300.png" in p.stderr


def test_cli_report_backend_line_and_gdaldem_handoff(workdir, golden):
    backend_line_and_gdaldem_handoff(ORACLE_CLI, "oracle-cpu", workdir, golden)


@pytest.mark.gpu
def test_hip_cli_report_backend_line_and_gdaldem_handoff(workdir, golden):
    """the same on the SHIPPED binary (wdpm_amd/bin/WDPMCL, HIP path): the helper that runs gdaldem is forked before
    anything touches the GPU - a process that has initialised it must not exec another program on this pool"""
    backend_line_and_gdaldem_handoff(HIP_CLI, "hip-gfx950", workdir, golden)


@pytest.mark.gpu
def test_hip_cli_chain_on_a_synthetic_dem_of_marching_kernel_size(tmp_path, hip):
    """round 5: the validation chain (add 100 mm -> drain -> subtract 10 mm, 1000 iterations each) of the shipped binary on a synthetic
    3072^2 DEM against what the REFERENCE EXECUTABLE printed and wrote for the same files (tests/golden/synth_cli.json, generated by
    tests/golden/make_golden.py synthcli: about 20 minutes of the reference).  basin5, the reference's own sample, is a relay-kernel
    raster here; this one takes the ArcASCII reader, the set-up on the device, the marching kernel (two waves per SIMD, DEM codes, XCD
    balance), the drain module's outlet and bookkeeping, the statistics and the writer through one command line at a size where
    that kernel runs by itself: report (minus wall clock) and rasters byte-identical."""
    path = os.path.join(GOLDEN, "synth_cli.json")
    if not os.path.exists(path):
        pytest.skip("tests/golden/synth_cli.json has not been generated (tests/golden/make_golden.py synthcli)")
    with open(path) as f:
        g = json.load(f)
    n = g["n"]
    dem = hip.synth_dem(n, n)
    assert hashlib.sha256(np.ascontiguousarray(dem).tobytes()).hexdigest() == g["dem_sha256_of_values"]
    with open(tmp_path / "dem.asc", "w") as f:
        f.write(f"ncols {n}\nnrows {n}\nxllcorner 0\nyllcorner 0\ncellsize 10\nNODATA_value -99999\n")
        np.savetxt(f, dem, fmt="%.4f")
    for key, outfile in (("add100", "a.asc"), ("drain", "d.asc"), ("sub10", "s.asc")):
        err = check(HIP_CLI, g, key, tmp_path, outfile)
        assert "hip-gfx950" in err
