"""CPU-side checks of the drop-in boundary: the product library loads and exports every symbol that
include/wdpm.h declares (no compute calls — there is no GPU here), and refuses to run without one."""
import os
import re
import subprocess

import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "wdpm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wdpm_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    import wdpm_amd.capi as capi
    assert declared_symbols() == sorted(capi.SYMBOLS)


def test_hip_library_exports_every_declared_symbol(hip):
    out = subprocess.check_output(["nm", "-D", "--defined-only", hip.path], text=True)
    exported = set(ln.split()[-1] for ln in out.splitlines() if " T " in ln)
    missing = [s for s in declared_symbols() if s not in exported]
    assert not missing, missing
    assert hip.backend == "hip-gfx950"


def test_oracle_exports_the_same_abi(oracle):
    out = subprocess.check_output(["nm", "-D", "--defined-only", oracle.path], text=True)
    exported = set(ln.split()[-1] for ln in out.splitlines() if " T " in ln)
    assert not [s for s in declared_symbols() if s not in exported]
    assert oracle.backend == "oracle-cpu"


def test_product_never_references_the_oracle():
    """The product tree must not load, link or mention the oracle (no CPU fallback)."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "wdpm_amd")):
        if "build" in base.split(os.sep):
            continue
        for fn in files:
            if fn.endswith((".py", ".hip", ".c", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(base, fn), errors="ignore").read()
                if re.search(r"libwdpm_oracle|oracle/_|wdpm_oracle\.c", txt):
                    bad.append(os.path.join(base, fn))
    assert not bad, bad


def test_hip_context_fails_loudly_without_a_gpu(hip):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import wdpm_amd
    with pytest.raises(wdpm_amd.WdpmError):
        hip.context(module="add", nrows=4, ncols=4, missingvalue=-1.0)


def test_synth_dem_is_deterministic_and_backend_independent(hip, oracle):
    a = hip.synth_dem(257, 4096)
    b = oracle.synth_dem(257, 4096)
    assert (a == b).all()
    assert a.shape == (257, 257)
    assert 450 < a.min() < a.max() < 560
    # quantised to 1e-4 m like basin5
    assert abs(a * 1e4 - (a * 1e4).round()).max() < 1e-6
