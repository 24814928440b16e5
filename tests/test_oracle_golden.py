"""Pins the oracle (oracle/wdpm_oracle.c) bit-for-bit against golden vectors produced by the
unmodified reference (tests/golden/make_golden.py) and the reference's validation constants."""
import json
import os

import numpy as np
import pytest

from helpers import bits_equal, n_bit_diff, pad, sha
from conftest import GOLDEN


def _ctx(lib, meta, z):
    name = meta["name"]
    ctx = lib.context(module=meta["module"], nrows=meta["R"], ncols=meta["C"], missingvalue=meta["missing"],
                      drainrow=meta["drainrow"], draincol=meta["draincol"])
    ctx.upload(z[name + "_dem"], z[name + "_w0"])
    ctx.totaldrain = meta["td0"]
    return ctx


def check_stencil_cases(lib, z, index):
    n_checked = 0
    for meta in index:
        name = meta["name"]
        # single colour passes of the first iteration
        if f"{name}_p1" in z:
            with _ctx(lib, meta, z) as ctx:
                k = 0
                for oi in (1, 2, 3):
                    for oj in (1, 2, 3):
                        ctx.single_pass(oi, oj)
                        k += 1
                        got = ctx.download_water()
                        assert bits_equal(got, z[f"{name}_p{k}"]), \
                            f"{name} pass {k}: {n_bit_diff(got, z[f'{name}_p{k}'])} cells differ"
                        if meta["module"] == 2:
                            assert ctx.totaldrain == meta["td_pass"][k - 1], f"{name} pass {k} totaldrain"
                        n_checked += 1
        # 1 / 10 / 100 / 1000 iterations
        with _ctx(lib, meta, z) as ctx:
            done = 0
            for st in meta["stages"]:
                ctx.iterate(st["iters"] - done)
                done = st["iters"]
                got = ctx.download_water()
                want = z[f"{name}_i{done}"]
                assert bits_equal(got, want), f"{name} after {done} iterations: {n_bit_diff(got, want)} cells differ"
                if meta["module"] == 2:
                    assert ctx.totaldrain == st["totaldrain"], f"{name} totaldrain after {done}"
                n_checked += 1
    return n_checked


def test_oracle_matches_reference_stencil_vectors(oracle, stencil_cases):
    z, index = stencil_cases
    assert check_stencil_cases(oracle, z, index) > 100


def basin5_blocks(lib, dem, missing, add_mm, n_blocks, thres=0.005 / 1000, **kw):
    water = np.where(dem > missing, add_mm / 1000.0, 0.0)
    bd, bw = pad(dem, water, missing)
    ctx = lib.context(module="add", nrows=dem.shape[0], ncols=dem.shape[1], missingvalue=missing, **kw)
    ctx.upload(bd, bw)
    out = []
    for _ in range(n_blocks):
        md = ctx.run_block(1000, thres)
        out.append((md, ctx.download_water()))
    ctx.close()
    return bd, out


def test_oracle_basin5_state_add100(oracle, basin5):
    dem, hdr = basin5
    z = np.load(os.path.join(GOLDEN, "basin5_state.npz"))
    index = {m["name"]: m for m in json.loads(bytes(z["index_json"]).decode())}
    bd, blocks = basin5_blocks(oracle, dem, hdr["NODATA_VALUE"], 100.0, 1)
    md, w = blocks[0]
    g = index["add100_k1000"]
    assert sha(w) == g["sha256"]
    assert bits_equal(w[::7], z["add100_k1000_rows"])
    assert f"{md:8.3f}".strip() == "1.779"  # WDPMCL_ref progress line, tests/golden/basin5_cli.json


@pytest.mark.parametrize("key", ["val_add10", "cfg1_add100_k3000", "cfg2_add300_k1000"])
def test_golden_cli_constants(key):
    """The reference's own pinned results (validation/validate_WDPM.sh:48-70) are what we recorded."""
    with open(os.path.join(GOLDEN, "basin5_cli.json")) as f:
        g = json.load(f)
    assert g["val_add10"]["summary"]["Final volume"] == "110035.85"
    assert abs(g["val_add10"]["patch_sum"] - 0.420810) < 5e-7
    assert g["val_drain"]["summary"]["Final volume"] == "97577.54"
    assert abs(g["val_drain"]["patch_sum"] - 0.420810) < 5e-7
    assert g["val_sub10"]["summary"]["Final volume"] == "86762.40"
    assert abs(g["val_sub10"]["patch_sum"] - 0.360810) < 5e-7
    assert g[key]["rc"] == 0
