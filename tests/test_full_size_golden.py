"""BASELINE configs 3, 4 and 5 at FULL size against the unmodified reference.

tests/golden/full_size.npz holds, for 4096^2 add x20, 16384^2 add x2 and 8192^2 add x3 + drain x5 on the
synthetic DEMs, what the reference's own runoffs()/runoffd()/drain() produce (tests/golden/make_golden.py,
run where /root/reference exists): sha256 of the padded water raster, an 8-byte hash of every row (says
which rows differ when the whole hash does), sampled rows, max diff, totaldrain and the sequential
volume sum.  Bit-exact is the bar: every comparison below is on hashes of the fp64 bits."""
import hashlib
import json
import os

import numpy as np
import pytest

import wdpm_amd
from conftest import GOLDEN
from helpers import pad, sha

MISS, THRES = -99999.0, 0.005 / 1000


@pytest.fixture(scope="module")
def golden():
    z = np.load(os.path.join(GOLDEN, "full_size.npz"))
    return z, {m["name"]: m for m in json.loads(bytes(z["index_json"]).decode())}


def row_hashes(w):
    return np.array([np.frombuffer(hashlib.sha256(r.tobytes()).digest()[:8], dtype=np.uint64)[0] for r in w],
                    dtype=np.uint64)


def assert_matches(z, meta, w):
    name = meta["name"]
    if sha(w) != meta["sha256"]:
        bad = np.nonzero(row_hashes(w) != z[name + "_rowhash"])[0]
        raise AssertionError(f"{name}: {bad.size} rows differ from the reference, first {bad[:8].tolist()}")
    assert np.array_equal(w[::meta["sample_every"]].view(np.uint64), z[name + "_rows"].view(np.uint64))


def inputs(lib, n):
    dem = lib.synth_dem(n, n)
    return pad(dem, np.full((n, n), 0.1), MISS)


def run_add(lib, meta, **ctx_kw):
    n = meta["n"]
    bd, bw = inputs(lib, n)
    with lib.context(module="add", nrows=n, ncols=n, missingvalue=MISS, **ctx_kw) as c:
        c.upload(bd, bw)
        return c, c.run_block(meta["add_iters"], THRES), c.download_water()


def test_oracle_is_pinned_at_4096(oracle, golden):
    """the CPU restatement against the reference at config 3's full size (the small golden vectors cover at
    most two 171-column strips; this covers 24)"""
    z, idx = golden
    meta = idx["cfg3_add_4096_i20"]
    _, md, w = run_add(oracle, meta)
    assert md == meta["max_diff"]
    assert_matches(z, meta, w)


@pytest.mark.gpu
@pytest.mark.parametrize("name,variant", [("cfg3_add_4096_i20", "fused"), ("cfg3_add_4096_i20", "fused-fp64dem"),
                                          ("cfg3_add_4096_i20", "pass"), ("cfg3_add_4096_i20", "fused-chunk48"),
                                          ("cfg4_add_16384_i2", "fused"), ("cfg4_add_16384_i2", "fused-fp64dem")])
def test_add_at_full_size_equals_the_reference(hip, golden, name, variant):
    z, idx = golden
    meta = idx[name]
    n = meta["n"]
    bd, bw = inputs(hip, n)
    kw = dict(kernel=wdpm_amd.KERNEL_PASS) if variant == "pass" else dict(kernel=wdpm_amd.KERNEL_FUSED)
    if variant == "fused-chunk48":
        kw["chunk_rows"] = 48
    with hip.context(module="add", nrows=n, ncols=n, missingvalue=MISS, **kw) as c:
        c.upload(bd, bw)
        if variant == "fused-fp64dem":
            c.set_option(wdpm_amd.OPT_DEM32, 0)
        elif variant.startswith("fused"):
            assert c.get_option(wdpm_amd.OPT_DEM32) == 1       # the synthetic DEMs are decimal: the codes are in use
        md = c.run_block(meta["add_iters"], THRES)
        w = c.download_water()
    assert md == meta["max_diff"]
    assert_matches(z, meta, w)


@pytest.mark.gpu
@pytest.mark.parametrize("devices,k", [([0, 0, 0, 0], 1), ([0] * 8, 2)])
def test_row_blocks_at_16384_equal_the_reference(hip, golden, devices, k):
    """config 4's decomposition (here: the slabs of one GPU, one host thread each) against the REFERENCE's
    bits, not only against one slab"""
    from wdpm_amd.rowblock import Group
    z, idx = golden
    meta = idx["cfg4_add_16384_i2"]
    n = meta["n"]
    bd, bw = inputs(hip, n)
    with Group(hip, "add", n, n, MISS, devices, exchange_every=k) as g:
        assert g.size == len(devices)
        g.upload(bd, bw)
        md = g.run_block(meta["add_iters"], THRES)
        w = g.download_water()
    assert md == meta["max_diff"]
    assert_matches(z, meta, w)


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
def test_drain_at_8192_equals_the_reference(hip, golden, devices):
    """config 5 at full size: add x3, flush, drain x5 - water, max diff, totaldrain and the sequential
    volume sum; 8192^2 is where the drain kernel's two-waves-per-SIMD launch shape switches on"""
    from wdpm_amd.rowblock import Group
    z, idx = golden
    meta = idx["cfg5_drain_8192_a3_d5"]
    n = meta["n"]
    bd, bw = inputs(hip, n)
    with hip.context(module="add", nrows=n, ncols=n, missingvalue=MISS) as c:
        c.upload(bd, bw)
        c.iterate(meta["add_iters"])
        w3 = c.download_water()
    dr, dc = meta["drainrow"], meta["draincol"]
    k = int(np.argmin(np.where(bd > 0, bd, np.inf)))
    assert (k // (n + 2), k % (n + 2)) == (dr, dc)
    assert max(float(w3[dr, dc]), 0.0) == meta["td0"]
    with Group(hip, "drain", n, n, MISS, devices, exchange_every=2, drainrow=dr, draincol=dc) as g:
        g.upload(bd, w3)
        g.set_totaldrain(meta["td0"])
        md = g.run_block(meta["drain_iters"], THRES)
        diffdrain, vol = g.drain_stats()
        td = g.totaldrain()
        w = g.download_water()
    assert md == meta["max_diff"] and td == meta["totaldrain"]
    assert diffdrain == abs(meta["totaldrain"] - meta["td0"])
    assert vol == meta["volume_sum"]
    assert_matches(z, meta, w)
