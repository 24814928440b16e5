"""GPU parity tests: the HIP path, through the C ABI, against (i) the committed golden vectors of
the unmodified reference and (ii) the CPU oracle on the same seeded inputs.  Bit-exact: the
north_star tolerance (1e-4 mm) is slack, these tests demand 0 differing bits."""
import json
import os

import numpy as np
import pytest

import wdpm_amd
from conftest import GOLDEN
from helpers import bits_equal, find_drain, n_bit_diff, pad, random_case, sha
from test_oracle_golden import basin5_blocks, check_stencil_cases

pytestmark = pytest.mark.gpu

TOL_MM = 1e-4  # north_star: max abs depth difference in mm; we assert 0 and report against this


class KernelLib:
    """the hip Lib with a fixed kernel / chunk choice, so shared checkers can take it as `lib`"""

    def __init__(self, lib, **kw):
        self.lib, self.kw = lib, kw

    def context(self, **kw):
        kw = dict(self.kw, **kw)
        return self.lib.context(**kw)


@pytest.mark.parametrize("kernel,chunk", [(wdpm_amd.KERNEL_PASS, 0), (wdpm_amd.KERNEL_FUSED, 0),
                                          (wdpm_amd.KERNEL_FUSED, 3), (wdpm_amd.KERNEL_FUSED, 12)])
def test_golden_stencil_vectors(hip, stencil_cases, kernel, chunk):
    z, index = stencil_cases
    assert check_stencil_cases(KernelLib(hip, kernel=kernel, chunk_rows=chunk), z, index) > 100


def _compare_with_oracle(hip, oracle, module, R, C, seed, iters, kernel, chunk=0, thres=None, dem32=None,
                         dem_digits=4, **case_kw):
    dem, water, miss = random_case(seed, R, C, **case_kw)
    if dem_digits is None:        # elevations that are not decimal fractions: the fp64 DEM must stay in charge
        dem = np.where(dem > miss, dem * (1.0 + 2.0 ** -30), dem)
    bd, bw = pad(dem, water, miss)
    kw = dict(module=module, nrows=R, ncols=C, missingvalue=miss)
    td0 = 0.0
    if module == "drain":
        dr, dc = find_drain(bd)
        td0 = max(bw[dr, dc], 0.0)
        kw.update(drainrow=dr, draincol=dc)
    with hip.context(kernel=kernel, chunk_rows=chunk, **kw) as g, oracle.context(**kw) as o:
        for c in (g, o):
            c.upload(bd, bw)
            c.totaldrain = td0
        if dem32 is not None:
            # random_case rounds elevations to 4 decimals: the device finds them encodable by itself
            encodable = dem_digits is not None and bool((dem > miss).any())
            assert g.get_option(wdpm_amd.OPT_DEM32) == int(encodable)
            g.set_option(wdpm_amd.OPT_DEM32, 2 if dem32 else 0)      # 2: also on launches this small
            assert g.get_option(wdpm_amd.OPT_DEM32) == int(bool(dem32) and encodable)
        for n in iters:
            if thres is not None:
                mg, mo = g.run_block(n, thres), o.run_block(n, thres)
                assert mg == mo
            else:
                g.iterate(n)
                o.iterate(n)
            wg, wo = g.download_water(), o.download_water()
            nd = n_bit_diff(wg, wo)
            maxmm = float(np.abs(wg - wo).max() * 1000)
            assert nd == 0, f"{nd} cells differ, max |d| = {maxmm} mm (tolerance {TOL_MM} mm)"
            assert g.totaldrain == o.totaldrain
            if module == "drain":
                assert g.drain_stats() == o.drain_stats()


@pytest.mark.parametrize("kernel", [wdpm_amd.KERNEL_PASS, wdpm_amd.KERNEL_FUSED])
@pytest.mark.parametrize("module", ["add", "drain"])
@pytest.mark.parametrize("R,C,chunk", [(38, 398, 12), (100, 700, 0), (301, 170, 48), (64, 1100, 30), (5, 175, 3)])
def test_random_rasters_match_oracle(hip, oracle, module, kernel, R, C, chunk):
    _compare_with_oracle(hip, oracle, module, R, C, seed=R * 1000 + C, iters=(1, 2, 25), kernel=kernel, chunk=chunk)


@pytest.mark.parametrize("dem32", [0, 1])
@pytest.mark.parametrize("module", ["add", "subtract", "drain"])      # (drain streams the codes from round 4 on)
@pytest.mark.parametrize("R,C,chunk", [(38, 398, 12), (100, 700, 0), (301, 170, 48), (64, 1100, 30), (5, 175, 3), (1, 1, 0)])
def test_dem_codes_on_and_off(hip, oracle, module, R, C, chunk, dem32):
    """the one-iteration kernel with the DEM streamed as verified 32-bit codes (WDPM_OPT_DEM32) and with
    the fp64 DEM: same bits as the oracle either way, edge strips and ragged sizes included"""
    _compare_with_oracle(hip, oracle, module, R, C, seed=R * 1000 + C + 1, iters=(1, 2, 25),
                         kernel=wdpm_amd.KERNEL_FUSED, chunk=chunk, dem32=dem32)


@pytest.mark.parametrize("module", ["add", "subtract", "drain"])
@pytest.mark.parametrize("R,C,chunk", [(38, 398, 12), (100, 700, 0), (301, 170, 48), (64, 1100, 30), (5, 175, 3), (1, 1, 0)])
def test_dem_codes_as_16_bit_offsets_on_and_off(hip, oracle, module, R, C, chunk):
    """round 4: the verified 32-bit codes once more as 16-bit offsets from one base per 48 columns of a row (18.1 B of HBM traffic
    per cell-update).  Gentle terrain (0.1 m of noise, a slope) is encodable; the kernel must give the same bits with the offsets,
    with the 32-bit codes and with the fp64 DEM - edge strips, NODATA cells (0xFFFF) and whole NODATA groups included"""
    rng = np.random.default_rng(R * 1000 + C)
    miss = -99999.0
    y, x = np.mgrid[0:R, 0:C]
    dem = np.round(500.0 + 0.3 * np.sin(x / 5.1) * np.cos(y / 3.3) + rng.normal(0, 0.05, (R, C)) - 0.002 * (x + y), 4)
    dem[rng.random((R, C)) < 0.05] = miss
    if C > 200:
        dem[:, 96:144] = miss                       # a whole group of 48 columns without a valid cell (and the ones beside it cut)
    water = np.where(dem > miss, np.where(rng.random((R, C)) < 0.3, 0.0, 0.3 * rng.random((R, C))), 0.0)
    bd, bw = pad(dem, water, miss)
    kw = dict(module=module, nrows=R, ncols=C, missingvalue=miss)
    td0 = 0.0
    if module == "drain":
        if not (dem > miss).any():
            pytest.skip("no valid cell: no outlet")
        dr, dc = find_drain(bd)
        td0 = max(bw[dr, dc], 0.0)
        kw.update(drainrow=dr, draincol=dc)
    out = {}
    for mode in ("fp64", "codes32", "codes16"):
        with hip.context(kernel=wdpm_amd.KERNEL_FUSED, chunk_rows=chunk, **kw) as g:
            g.upload(bd, bw)
            g.totaldrain = td0
            can16 = bool((dem > miss).any()) and os.environ.get("WDPM_DEM16", "1") != "0"      # (the forced-variant suites switch it off)
            small = 1 if os.environ.get("WDPM_DEM32") == "2" else 2                     # 2: available, but a raster this small keeps the 32-bit codes
            assert g.get_option(wdpm_amd.capi.OPT_DEM16) == (small if can16 else 0)       # (1 under the forced-variant suites' WDPM_DEM32=2)
            g.set_option(wdpm_amd.OPT_DEM32, 0 if mode == "fp64" else 2)                # 2: the codes on launches of any size
            g.set_option(wdpm_amd.capi.OPT_DEM16, 1 if mode == "codes16" else 0)
            assert g.get_option(wdpm_amd.capi.OPT_DEM16) == int(mode == "codes16" and can16)
            g.iterate(7)
            md = g.run_block(5, 1e-5)
            out[mode] = (g.download_water(), md, g.totaldrain)
    with oracle.context(**kw) as o:
        o.upload(bd, bw)
        o.totaldrain = td0
        o.iterate(7)
        want = (o.run_block(5, 1e-5), o.download_water(), o.totaldrain)
    for mode, (w, md, td) in out.items():
        assert n_bit_diff(w, want[1]) == 0 and md == want[0] and td == want[2], mode


def test_16_bit_offsets_are_refused_where_the_relief_is_too_steep(hip, oracle):
    """a group of 48 columns spanning 65 534 quanta is the last one that fits; 65 535 (0xFFFF is NODATA) and more keep the 32-bit codes"""
    R, C, miss = 40, 300, -99999.0
    for span, ok in ((65534, 1), (65535, 0), (200001, 0)):      # (quanta of 1e-4 m: the other cells keep e = 4 in charge)
        dem = np.full((R, C), 500.0003)
        dem[7, 100] = np.round(500.0003 + span * 1e-4, 4)       # one cell that far above its group
        dem[3, 50] = miss
        bd, bw = pad(dem, np.where(dem > miss, 0.05, 0.0), miss)
        kw = dict(module="add", nrows=R, ncols=C, missingvalue=miss)
        with hip.context(kernel=wdpm_amd.KERNEL_FUSED, chunk_rows=12, **kw) as g, oracle.context(**kw) as o:
            g.upload(bd, bw)
            o.upload(bd, bw)
            assert g.get_option(wdpm_amd.OPT_DEM32) == 1 and g.get_option(wdpm_amd.capi.OPT_DEM16) == ((1 if os.environ.get("WDPM_DEM32") == "2" else 2) * ok if os.environ.get("WDPM_DEM16", "1") != "0" else 0), span
            g.set_option(wdpm_amd.OPT_DEM32, 2)
            g.iterate(6)
            o.iterate(6)
            assert n_bit_diff(g.download_water(), o.download_water()) == 0, span


def test_a_rough_dem_uploaded_over_a_smooth_one_inherits_no_16_bit_offsets(hip, oracle):
    """the 16-bit level is decided afresh by every upload.  A smooth DEM (offsets accepted), then - into the same context - a
    DEM that passes the 32-bit check only: asking for the offsets must be refused, and the options must switch offsets, group bases
    and their pitch together (DEM32 off and on again brings the offsets back only where they are valid)"""
    if os.environ.get("WDPM_DEM16", "1") == "0":
        pytest.skip("the forced-variant suite without offsets")
    R, C, miss = 40, 300, -99999.0
    kw = dict(module="add", nrows=R, ncols=C, missingvalue=miss)
    smooth = np.full((R, C), 500.0003)
    rough = smooth.copy()
    rough[7, 100] = np.round(500.0003 + 200001 * 1e-4, 4)
    rough[3, 50] = miss
    avail = 1 if os.environ.get("WDPM_DEM32") == "2" else 2
    with hip.context(kernel=wdpm_amd.KERNEL_FUSED, chunk_rows=12, **kw) as g:
        g.upload(*pad(smooth, np.full((R, C), 0.05), miss))
        assert g.get_option(wdpm_amd.capi.OPT_DEM16) == avail
        g.set_option(wdpm_amd.OPT_DEM32, 0)
        assert g.get_option(wdpm_amd.capi.OPT_DEM16) == 0
        g.set_option(wdpm_amd.OPT_DEM32, 2)
        assert g.get_option(wdpm_amd.capi.OPT_DEM16) == 1          # back, with their bases
        g.iterate(3)
        bd, bw = pad(rough, np.where(rough > miss, 0.05, 0.0), miss)
        g.upload(bd, bw)
        assert g.get_option(wdpm_amd.OPT_DEM32) == 1 and g.get_option(wdpm_amd.capi.OPT_DEM16) == 0
        g.set_option(wdpm_amd.OPT_DEM32, 2)
        g.set_option(wdpm_amd.capi.OPT_DEM16, 1)                    # must be refused: the offsets in memory are the smooth DEM's
        assert g.get_option(wdpm_amd.capi.OPT_DEM16) == 0
        g.iterate(6)
        with oracle.context(**kw) as o:
            o.upload(bd, bw)
            o.iterate(6)
            assert n_bit_diff(g.download_water(), o.download_water()) == 0


@pytest.mark.parametrize("R,C,chunk", [(60, 400, 12), (33, 190, 0), (150, 700, 30)])
def test_drain_on_codes_with_nodata_around_the_outlet(hip, oracle, R, C, chunk):
    """the drain kernels stream the DEM codes too, and a NODATA code decodes to NaN, not +inf (wdpm_stencil.h::dem32_decode_nan
    has the argument why both behave alike).  Here NODATA sits where each branch of that argument is exercised: cells that are
    centres and neighbours all over the raster, a ring of NODATA on three sides of the outlet and one directly above it - with
    the codes forced on launches of this size (32-bit and 16-bit), against the oracle on the fp64 DEM"""
    rng = np.random.default_rng(R * 7 + C)
    miss = -99999.0
    y, x = np.mgrid[0:R, 0:C]
    dem = np.round(500.0 + 0.2 * np.sin(x / 4.7) * np.cos(y / 3.1) + rng.normal(0, 0.03, (R, C)) + 0.003 * ((x - C // 2) ** 2 + (y - R // 2) ** 2) ** 0.5, 4)
    dem[rng.random((R, C)) < 0.06] = miss
    r0, c0 = R // 2, C // 2
    dem[r0, c0] = 495.0                                     # the outlet: the lowest valid cell by far (5 m: its group of 48 columns still fits 16-bit offsets)
    dem[r0 - 1, c0] = miss                                  # NODATA straight above it,
    dem[r0 + 1, c0 - 1:c0 + 2] = miss                       # below it
    dem[r0 - 1:r0 + 2, c0 + 1] = miss                       # and to its right: it can be reached from the left and the upper left only
    dem[r0, c0 - 1], dem[r0 - 1, c0 - 1] = 499.5, 499.6
    water = np.where(dem > miss, 0.05 + 0.2 * rng.random((R, C)), 0.0)
    bd, bw = pad(dem, water, miss)
    dr, dc = find_drain(bd)
    assert (dr, dc) == (r0 + 1, c0 + 1)
    kw = dict(module="drain", nrows=R, ncols=C, missingvalue=miss, drainrow=dr, draincol=dc)
    td0 = max(bw[dr, dc], 0.0)
    with oracle.context(**kw) as o:
        o.upload(bd, bw)
        o.totaldrain = td0
        o.iterate(9)
        want = (o.run_block(16, 1e-5), o.download_water(), o.totaldrain)
    assert want[2] > td0                                    # water did reach the outlet
    for d16 in (0, 1):
        with hip.context(kernel=wdpm_amd.KERNEL_FUSED, chunk_rows=chunk, **kw) as g:
            g.upload(bd, bw)
            g.totaldrain = td0
            g.set_option(wdpm_amd.OPT_DEM32, 2)
            g.set_option(wdpm_amd.capi.OPT_DEM16, d16)
            assert g.get_option(wdpm_amd.OPT_DEM32) == 1
            if d16 and os.environ.get("WDPM_DEM16", "1") != "0":
                assert g.get_option(wdpm_amd.capi.OPT_DEM16) == 1
            g.iterate(9)
            md = g.run_block(16, 1e-5)
            assert n_bit_diff(g.download_water(), want[1]) == 0 and md == want[0] and g.totaldrain == want[2], d16


@pytest.mark.parametrize("module", ["add", "subtract", "drain"])
@pytest.mark.parametrize("R,C", [(120, 300), (150, 471), (61, 1000)])
def test_steady_iterations_of_small_rasters_replayed_as_hip_graphs(hip, oracle, module, R, C):
    """round 5: small rasters are launch-bound, so wdpm_iterate replays the iterations between a block's first and last launch as HIP
    graphs of 32 launches captured from its own loop (include/wdpm.h: WDPM_OPT_GRAPH_LAUNCHES).  Same bits as the oracle through
    counts on both sides of every threshold (33 iterations: none; 34: one graph; 35, 100, 131), through whole blocks (flush on the
    first launch, max diff on the last), after an option change and a second upload into the same context (cached graphs carry what they
    were given by value: they must go), and with the launches timed (no graphs then)"""
    dem, water, miss = random_case(R * 31 + C, R, C)
    bd, bw = pad(dem, water, miss)
    kw = dict(module=module, nrows=R, ncols=C, missingvalue=miss)
    td0 = 0.0
    if module == "drain":
        dr, dc = find_drain(bd)
        td0 = max(bw[dr, dc], 0.0)
        kw.update(drainrow=dr, draincol=dc)
    graphs_on = os.environ.get("WDPM_GRAPH", "1") != "0"
    with hip.context(**kw) as g, oracle.context(**kw) as o:
        small = os.environ.get("WDPM_RELAY") != "0" or os.environ.get("WDPM_TRI") != "0"     # (a forced-variant suite may send every size to the marching kernel)
        for c in (g, o):
            c.upload(bd, bw)
            c.totaldrain = td0
        seen = 0
        for n in (33, 34, 35, 100, 131):
            g.iterate(n)
            o.iterate(n)
            assert n_bit_diff(g.download_water(), o.download_water()) == 0 and g.totaldrain == o.totaldrain, n
            now = g.get_option(wdpm_amd.capi.OPT_GRAPH_LAUNCHES)
            if graphs_on and small and os.environ.get("WDPM_RELAY") is None and os.environ.get("WDPM_TRI") is None:
                assert now - seen == (n - 2) // 32, (n, now, seen)          # the first and the last iteration of a call are never in a graph
            seen = now
        for n, thres in ((200, 1e-5), (67, 2e-4)):
            assert g.run_block(n, thres) == o.run_block(n, thres)
            assert n_bit_diff(g.download_water(), o.download_water()) == 0 and g.totaldrain == o.totaldrain
        g.set_option(wdpm_amd.OPT_DEM32, 2)                     # other codes in the launches from here on
        g.iterate(70)
        o.iterate(70)
        assert n_bit_diff(g.download_water(), o.download_water()) == 0
        dem2 = np.where(dem > miss, np.round(dem * 0.5 + 100.0, 3), dem)          # another DEM into the same context, same buffers
        bd2, bw2 = pad(dem2, water * 0.5, miss)
        if module == "drain":
            dr, dc = find_drain(bd2)
            for c in (g, o):
                c.lib.check(c.lib.dll.wdpm_set_drain(c._h, dr, dc))
        for c in (g, o):
            c.upload(bd2, bw2)
            c.totaldrain = 0.0
            c.iterate(90)
        assert n_bit_diff(g.download_water(), o.download_water()) == 0 and g.totaldrain == o.totaldrain
        before = g.get_option(wdpm_amd.capi.OPT_GRAPH_LAUNCHES)
        g.timing_reset()                                         # timed launches are queued one by one
        g.iterate(80)
        o.iterate(80)
        assert g.get_option(wdpm_amd.capi.OPT_GRAPH_LAUNCHES) == before and g.timing()[0] == 80
        assert n_bit_diff(g.download_water(), o.download_water()) == 0


def test_dem_codes_are_refused_for_non_decimal_elevations(hip, oracle):
    _compare_with_oracle(hip, oracle, "add", 60, 400, seed=77, iters=(5,), kernel=wdpm_amd.KERNEL_FUSED, dem32=1,
                         dem_digits=None)


@pytest.mark.parametrize("name", ["six_digits_high", "needs_offset", "relief_too_large", "negative", "neg_zero",
                                  "mixed_digits", "integers", "huge"])
def test_dem_code_edge_cases(hip, oracle, name):
    """which DEMs the device accepts as k / 10^e with 32-bit k - and that results never depend on it"""
    rng = np.random.default_rng(5)
    R, C, miss = 40, 230, -99999.0
    base = rng.normal(0, 3, (R, C))
    dem, want = {
        "six_digits_high": (np.round(1500.0 + base, 6), 1),                      # k ~ 1.5e9: fits without the offset too
        "needs_offset": (np.round(3000.0 + base, 6), 1),            # k ~ 3e9 > 2^31: only k - k0 fits
        "relief_too_large": (np.where(rng.random((R, C)) < 0.5, 1e-6, 5000.000001), 0),
        "negative": (np.round(-12.0 + base, 3), 1),
        "neg_zero": (np.where(rng.random((R, C)) < 0.01, -0.0, np.round(3.0 + base, 2)), 0),
        "mixed_digits": (np.where(rng.random((R, C)) < 0.5, np.round(400 + base, 1), np.round(400 + base, 5)), 1),
        "integers": (np.round(700.0 + base, 0), 1),
        "huge": (np.where(rng.random((R, C)) < 0.02, 1e300, np.round(10 + base, 2)), 0),
    }[name]
    dem = dem.copy()
    dem[rng.random((R, C)) < 0.04] = miss
    water = np.where(dem > miss, 0.2 * rng.random((R, C)), 0.0)
    bd, bw = pad(dem, water, miss)
    kw = dict(module="add", nrows=R, ncols=C, missingvalue=miss)
    with hip.context(**kw) as g, oracle.context(**kw) as o:
        g.upload(bd, bw)
        o.upload(bd, bw)
        assert g.get_option(wdpm_amd.OPT_DEM32) == want
        g.set_option(wdpm_amd.OPT_DEM32, 2)                           # honoured only if the DEM passed the check
        assert g.get_option(wdpm_amd.OPT_DEM32) == want
        g.iterate(12)
        o.iterate(12)
        assert n_bit_diff(g.download_water(), o.download_water()) == 0


@pytest.mark.parametrize("module", ["add", "subtract", "drain"])
def test_block_loop_matches_oracle(hip, oracle, module):
    """flush + snapshot + iterations + max_diff (WDPMCL.c:1055-1073,1239-1254), thres > 0 and = 0"""
    _compare_with_oracle(hip, oracle, module, 120, 333, seed=7, iters=(40, 40), kernel=wdpm_amd.KERNEL_FUSED,
                         thres=0.005 / 1000)
    _compare_with_oracle(hip, oracle, module, 61, 200, seed=8, iters=(30,), kernel=wdpm_amd.KERNEL_FUSED, thres=0.0)


def test_all_missing_and_all_dry(hip, oracle):
    _compare_with_oracle(hip, oracle, "add", 30, 200, seed=3, iters=(3,), kernel=wdpm_amd.KERNEL_FUSED,
                         missing_frac=1.0)
    _compare_with_oracle(hip, oracle, "add", 30, 200, seed=4, iters=(3,), kernel=wdpm_amd.KERNEL_FUSED,
                         dry_frac=1.0)
    _compare_with_oracle(hip, oracle, "add", 30, 200, seed=5, iters=(3,), kernel=wdpm_amd.KERNEL_FUSED,
                         missing_frac=0.0, dry_frac=0.0)


def test_basin5_add100_and_add300_state_hashes(hip, basin5):
    """BASELINE configs 1-2 on the HIP path: full-precision state after 1000 / 3000 iterations equals
    the unmodified reference's (sha256 of the fp64 raster), and so does the printed max diff."""
    dem, hdr = basin5
    z = np.load(os.path.join(GOLDEN, "basin5_state.npz"))
    index = {m["name"]: m for m in json.loads(bytes(z["index_json"]).decode())}
    with open(os.path.join(GOLDEN, "basin5_cli.json")) as f:
        cli = json.load(f)
    for add_mm, key in ((100.0, "cfg1_add100_k3000"), (300.0, None)):
        bd, blocks = basin5_blocks(hip, dem, hdr["NODATA_VALUE"], add_mm, 3)
        for k, (md, w) in zip((1000, 2000, 3000), blocks):
            name = f"add{int(add_mm)}_k{k}"
            if name in index:
                assert sha(w) == index[name]["sha256"], name
                assert bits_equal(w[::7], z[name + "_rows"])
            if key:
                want = cli[key]["blocks"][k // 1000 - 1]
                assert want[0] == k and f"{md:8.3f}".strip() == want[1]


def test_synthetic_1024_matches_oracle(hip, oracle):
    """config-3 generator at 1024^2, add 100 mm, 60 iterations: HIP fused == oracle, bit for bit"""
    n = 1024
    dem = hip.synth_dem(n, n)
    miss = -99999.0
    bd, bw = pad(dem, np.full((n, n), 0.1), miss)
    kw = dict(module="add", nrows=n, ncols=n, missingvalue=miss)
    with hip.context(**kw) as g, oracle.context(**kw) as o:
        g.upload(bd, bw)
        o.upload(bd, bw)
        mg, mo = g.run_block(60, 0.005 / 1000), o.run_block(60, 0.005 / 1000)
        wg, wo = g.download_water(), o.download_water()
        assert n_bit_diff(wg, wo) == 0, f"max |d| = {np.abs(wg - wo).max() * 1000} mm"
        assert mg == mo


@pytest.mark.parametrize("n,iters", [(4096, 20), (16384, 2)])
def test_full_size_properties(hip, n, iters):
    """BASELINE sizes, where the oracle is too slow: size-independent properties.
    (1) the two independent HIP implementations (9 launches/iteration in place vs fused marching
        window) agree bit for bit; (2) water volume is conserved to rounding; (3) depths stay >= 0."""
    dem = hip.synth_dem(n, n)
    miss = -99999.0
    bd, bw = pad(dem, np.full((n, n), 0.1), miss)
    del dem
    kw = dict(module="add", nrows=n, ncols=n, missingvalue=miss)
    with hip.context(kernel=wdpm_amd.KERNEL_FUSED, **kw) as f:
        f.upload(bd, bw)
        f.iterate(iters)
        wf = f.download_water()
    with hip.context(kernel=wdpm_amd.KERNEL_PASS, **kw) as p:
        p.upload(bd, bw)
        p.iterate(iters)
        wp = p.download_water()
    assert n_bit_diff(wf, wp) == 0
    assert wf.min() >= 0.0
    total0, total1 = 0.1 * n * n, float(wf.sum())
    assert abs(total1 - total0) <= 1e-9 * total0
    assert wf[0].max() == 0 and wf[-1].max() == 0 and wf[:, 0].max() == 0 and wf[:, -1].max() == 0


def test_rowblock_solver_single_rank_on_gpu(hip, oracle):
    from wdpm_amd.rowblock import RowBlockSolver
    dem, water, miss = random_case(22, 90, 200)
    bd, bw = pad(dem, water, miss)
    s = RowBlockSolver(hip, "add", 90, 200, miss)
    s.upload_global(bd, bw)
    with oracle.context(module="add", nrows=90, ncols=200, missingvalue=miss) as o:
        o.upload(bd, bw)
        assert s.run_block(30, 1e-6) == o.run_block(30, 1e-6)
        assert bits_equal(s.owned_water(), o.download_water())
    s.close()


def module_kw(module, bd, nrows, ncols, miss):
    kw = dict(module=module, nrows=nrows, ncols=ncols, missingvalue=miss)
    if module == "drain":
        kw["drainrow"], kw["draincol"] = find_drain(bd)
    return kw


@pytest.mark.parametrize("module", ["add", "drain"])
def test_negative_zero_depths_keep_their_sign(hip, oracle, module):
    """a -0.0 depth in the input makes the library pick the sign-preserving stencil variant by
    itself; results equal the oracle bit for bit, sign of zero included"""
    dem, water, miss = random_case(31, 60, 220, dry_frac=0.5)
    water[(water == 0) & (np.arange(water.size).reshape(water.shape) % 3 == 0)] = -0.0
    bd, bw = pad(dem, water, miss)
    assert np.signbit(bw[bw == 0]).any()
    kw = module_kw(module, bd, 60, 220, miss)
    with hip.context(**kw) as g, oracle.context(**kw) as o:
        g.upload(bd, bw)
        o.upload(bd, bw)
        assert g.get_option(wdpm_amd.OPT_SIGNED_ZERO_SAFE) == 1
        g.iterate(7)
        o.iterate(7)
        assert n_bit_diff(g.download_water(), o.download_water()) == 0
        # a later upload without -0.0 goes back to the fast variant, still exact
        g.upload(bd, np.abs(bw))
        o.upload(bd, np.abs(bw))
        assert g.get_option(wdpm_amd.OPT_SIGNED_ZERO_SAFE) == 0
        g.iterate(7)
        o.iterate(7)
        assert n_bit_diff(g.download_water(), o.download_water()) == 0
        # and forcing the safe variant on clean data gives the same bits again
        g.upload(bd, np.abs(bw))
        g.set_option(wdpm_amd.OPT_SIGNED_ZERO_SAFE, 1)
        g.iterate(7)
        assert n_bit_diff(g.download_water(), o.download_water()) == 0


@pytest.mark.parametrize("module", ["add", "drain"])
def test_negative_and_nan_inputs_are_handled_like_the_reference(hip, oracle, module):
    """odd water files: negative depths and NaN cells never give water and are carried through"""
    dem, water, miss = random_case(32, 40, 200)
    water[3, 5] = -0.25
    water[10, 100] = np.nan
    water[20:22, 50:60] = -1e-9
    water[30, 20:40] = -3.0          # deep negative cells next to wet ones: the clamp to [0, w_c] matters
    bd, bw = pad(dem, water, miss)
    kw = module_kw(module, bd, 40, 200, miss)
    with hip.context(**kw) as g, oracle.context(**kw) as o:
        g.upload(bd, bw)
        o.upload(bd, bw)
        g.iterate(5)
        o.iterate(5)
        assert bits_equal(g.download_water(), o.download_water())


def test_synthetic_1024_drain_matches_oracle(hip, oracle):
    """BASELINE config 5 at the reduced size SURVEY §8d names (1024^2): drain from the add-100-mm state,
    totaldrain / vol change / water left and the raster equal the CPU oracle bit for bit"""
    n, miss = 1024, -99999.0
    dem = hip.synth_dem(n, n)
    bd, bw = pad(dem, np.full((n, n), 0.1), miss)
    with hip.context(module="add", nrows=n, ncols=n, missingvalue=miss) as a:
        a.upload(bd, bw)
        a.run_block(200, 0.005 / 1000)
        w0 = a.download_water()
    dr, dc = find_drain(bd)
    kw = dict(module="drain", nrows=n, ncols=n, missingvalue=miss, drainrow=dr, draincol=dc)
    with hip.context(**kw) as g, oracle.context(**kw) as o:
        for c in (g, o):
            c.upload(bd, w0)
            c.totaldrain = max(w0[dr, dc], 0.0)
        mg, mo = g.run_block(40, 0.005 / 1000), o.run_block(40, 0.005 / 1000)
        assert mg == mo
        assert n_bit_diff(g.download_water(), o.download_water()) == 0
        assert g.totaldrain == o.totaldrain and g.totaldrain > 0
        assert g.drain_stats() == o.drain_stats()


def test_subnormal_depths_are_not_flushed(hip, oracle):
    """thres = 0 lets depths decay into the fp64 subnormal range (the reference's slow case on x86);
    the GPU must keep them, bit for bit — no flush-to-zero anywhere in the kernels"""
    dem, water, miss = random_case(41, 50, 210, dry_frac=0.2)
    water = water * 1e-308                                  # 0 .. 3e-309: subnormal and near-subnormal
    water[::7, ::5] = 5e-324
    bd, bw = pad(dem, water, miss)
    kw = dict(module="add", nrows=50, ncols=210, missingvalue=miss)
    with hip.context(**kw) as g, oracle.context(**kw) as o:
        g.upload(bd, bw)
        o.upload(bd, bw)
        assert g.run_block(30, 0.0) == o.run_block(30, 0.0)
        wg, wo = g.download_water(), o.download_water()
        assert n_bit_diff(wg, wo) == 0
        assert ((wg > 0) & (wg < 2.2e-308)).any()           # subnormals survived


@pytest.mark.parametrize("module", ["add", "drain"])
@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_adversarial_operands(hip, oracle, module, seed):
    """Elevations and depths drawn from pools built to hit the corners of the neighbour step on the GPU's own instructions
    (v_ldexp / v_max / v_min without canonicalisation, the -inf gate, the ldexp(ht, 2200) sign trick of the drain step):
    exact ties of water surfaces, depths from one subnormal ulp to 1e300 next to elevations from 1e-4 to 1e15, neighbours
    one ulp apart, water that exactly fills the step to its neighbour.  Several iterations, bit for bit against the oracle."""
    rng = np.random.default_rng(100 + seed)
    R, C, miss = 120, 400, -99999.0
    dpool = np.array([1e-4, 0.5, 1.0, 499.9999, 500.0, np.nextafter(500.0, 501), 500.0001, 500.1, 512.0, 1e6, 1e15])
    if seed == 4:       # decimal elevations only: the DEM goes through the 32-bit codes (forced below)
        dpool = np.array([1e-4, 0.5, 1.0, 499.9999, 500.0, 500.0001, 500.1, 512.0, 99999.9999])
    wpool = np.array([0.0, 0.0, 0.0, 5e-324, 1e-310, 2.3e-308, 1e-300, 1e-16, 1.1368683772161603e-13, 1e-4, 1e-4, 0.1, 0.1,
                      np.nextafter(0.1, 1), 0.7999999999999, 0.8, 1.0, 8.0, 1e6, 1e300])
    dem = dpool[rng.integers(0, len(dpool), (R, C))]
    water = wpool[rng.integers(0, len(wpool), (R, C))]
    # runs of equal elevations / equal surfaces along rows so that ties and one-ulp differences meet as neighbours
    same = rng.random((R, C)) < 0.4
    for arr in (dem, water):
        arr[:, 1:] = np.where(same[:, 1:], arr[:, :-1], arr[:, 1:])
    fill = rng.random((R, C)) < 0.1                           # depth that brings the surface exactly to the left neighbour's
    water[:, 1:] = np.where(fill[:, 1:] & (dem[:, :-1] > dem[:, 1:]), dem[:, :-1] - dem[:, 1:], water[:, 1:])
    dem[rng.random((R, C)) < 0.05] = miss
    water[dem <= miss] = 0.0
    bd, bw = pad(dem, water, miss)
    kw = dict(module=module, nrows=R, ncols=C, missingvalue=miss)
    if module == "drain":
        dr, dc = find_drain(bd)
        kw.update(drainrow=dr, draincol=dc)
    # the triangle kernel (a raster this small), the marching kernel (a chunk height is asked for), the per-pass kernel
    for kernel, chunk in ((wdpm_amd.KERNEL_FUSED, 0), (wdpm_amd.KERNEL_FUSED, 12), (wdpm_amd.KERNEL_PASS, 0)):
        with hip.context(kernel=kernel, chunk_rows=chunk, **kw) as g, oracle.context(**kw) as o:
            for c in (g, o):
                c.upload(bd, bw)
            if seed == 4 and module == "add":
                g.set_option(wdpm_amd.OPT_DEM32, 2)
                assert g.get_option(wdpm_amd.OPT_DEM32) == 1
            for n in (1, 2, 9):
                g.iterate(n)
                o.iterate(n)
                wg, wo = g.download_water(), o.download_water()
                assert n_bit_diff(wg, wo) == 0, (kernel, chunk, n)
                assert g.totaldrain == o.totaldrain
            assert g.run_block(4, 1e-5) == o.run_block(4, 1e-5)
            assert n_bit_diff(g.download_water(), o.download_water()) == 0


def test_negative_elevations_and_other_nodata(hip, oracle):
    """terrain below datum and a different NODATA code"""
    rng = np.random.default_rng(42)
    R, C, miss = 44, 260, -9999.0
    dem = np.round(-120.0 + 3 * rng.standard_normal((R, C)).cumsum(axis=1) * 0.1, 3)
    dem[rng.random((R, C)) < 0.07] = miss
    water = np.where(dem > miss, rng.random((R, C)) * 0.2, 0.0)
    bd, bw = pad(dem, water, miss)
    for module in ("add", "drain"):
        kw = dict(module=module, nrows=R, ncols=C, missingvalue=miss)
        if module == "drain":
            # no positive elevation: the reference's outlet search (bigdem > 0) finds nothing and keeps 0,0
            kw.update(drainrow=0, draincol=0)
        with hip.context(**kw) as g, oracle.context(**kw) as o:
            g.upload(bd, bw)
            o.upload(bd, bw)
            g.iterate(12)
            o.iterate(12)
            assert n_bit_diff(g.download_water(), o.download_water()) == 0
            assert g.totaldrain == o.totaldrain


@pytest.mark.parametrize("R,C", [(1, 1), (1, 700), (2, 3), (600, 1), (3, 171), (4, 172), (23, 178), (24, 179)])
def test_degenerate_shapes(hip, oracle, R, C):
    """rasters thinner than a block, exactly one strip wide, one column past a strip, ..."""
    _compare_with_oracle(hip, oracle, "add", R, C, seed=R * 7 + C, iters=(1, 5), kernel=wdpm_amd.KERNEL_FUSED)
    _compare_with_oracle(hip, oracle, "drain", R, C, seed=R * 7 + C + 1, iters=(4,), kernel=wdpm_amd.KERNEL_FUSED)


@pytest.mark.parametrize("R,C", [(1250, 1300), (1700, 950), (700, 2900)])
def test_triangle_kernel_in_several_rounds(hip, oracle, R, C):
    """rasters whose 3-row chunks no longer fit on the chip at once but which the launcher still gives to the triangle kernel
    (up to 2.7 rounds of three-row waves; add / subtract then take six rows per wave, K = 2): 3 300 - 6 600 work items here,
    against the oracle, flush and max change included"""
    # the first block runs on the marching kernel (it keeps the dry-tile flags and so finds out that the raster is mostly wet),
    # the blocks after it on the triangle kernel; with tile tracking off the triangle kernel runs from the first launch
    _compare_with_oracle(hip, oracle, "add", R, C, seed=R + C, iters=(2, 2, 1), kernel=wdpm_amd.KERNEL_FUSED, thres=5e-6)
    _compare_with_oracle(hip, oracle, "drain", R, C, seed=R + C + 1, iters=(2, 3), kernel=wdpm_amd.KERNEL_FUSED, thres=5e-6)
    os.environ["WDPM_TILES"] = "0"
    try:
        _compare_with_oracle(hip, oracle, "add", R, C, seed=R + C + 2, iters=(3,), kernel=wdpm_amd.KERNEL_FUSED)
    finally:
        del os.environ["WDPM_TILES"]


def test_every_width_around_strip_boundaries(hip, oracle):
    """every raster width from 165 to 200, 335 to 350 and 505 to 520 columns (a strip stores 171 columns and loads 192: widths
    that end a strip exactly, one short, one over, inside the halo ...), marching kernel (12-row chunks) and triangle kernel"""
    for C in list(range(165, 201)) + list(range(335, 351)) + list(range(505, 521)):
        for R, chunk in ((9, 0), (26, 12)):
            _compare_with_oracle(hip, oracle, "add", R, C, seed=C * 3 + R, iters=(2,), kernel=wdpm_amd.KERNEL_FUSED, chunk=chunk)
        _compare_with_oracle(hip, oracle, "drain", 14, C, seed=C * 5, iters=(2,), kernel=wdpm_amd.KERNEL_FUSED, chunk=12 if C % 2 else 0)


def test_every_height_around_chunk_boundaries(hip, oracle):
    """every raster height from 1 to 45 rows with the triangle kernel, 6-row and 12-row marching chunks (a chunk's last step,
    the warm-up rows of the next, the raster's last rows in every position of the 3-row cadence), add and drain"""
    for R in range(1, 46):
        for chunk in (0, 6, 12):
            _compare_with_oracle(hip, oracle, "add", R, 200, seed=R * 7 + chunk, iters=(3,), kernel=wdpm_amd.KERNEL_FUSED, chunk=chunk)
        _compare_with_oracle(hip, oracle, "drain", R, 75, seed=R * 11, iters=(2,), kernel=wdpm_amd.KERNEL_FUSED, chunk=(0, 6, 12)[R % 3])


@pytest.mark.parametrize("R,C", [(2, 50000), (50000, 2), (7, 30000), (30000, 7), (1, 100000), (100000, 1)])
def test_extreme_aspect_ratios(hip, oracle, R, C):
    """hundreds of strips one chunk high, and one strip thousands of chunks high: the launch geometry at its ends
    (both the marching and, where it is chosen, the triangle kernel), against the oracle"""
    _compare_with_oracle(hip, oracle, "add", R, C, seed=R + 3 * C, iters=(1, 3), kernel=wdpm_amd.KERNEL_FUSED)
    _compare_with_oracle(hip, oracle, "drain", R, C, seed=R + 3 * C + 1, iters=(3,), kernel=wdpm_amd.KERNEL_FUSED)
    _compare_with_oracle(hip, oracle, "add", R, C, seed=R + 3 * C + 2, iters=(2, 2), kernel=wdpm_amd.KERNEL_FUSED, thres=5e-6)


def test_one_block_of_1000_iterations_at_8192(hip):
    """a whole reference block (1000 iterations) at 8192^2: the kernel streaming the DEM as verified 32-bit
    codes and the one reading the fp64 DEM end on identical bits and identical max diff; volume conserved;
    nothing negative"""
    n, miss = 8192, -99999.0
    dem = hip.synth_dem(n, n)
    bd, bw = pad(dem, np.full((n, n), 0.1), miss)
    del dem
    kw = dict(module="add", nrows=n, ncols=n, missingvalue=miss)
    res = {}
    for dem32 in (0, 1):
        with hip.context(kernel=wdpm_amd.KERNEL_FUSED, **kw) as c:
            c.upload(bd, bw)
            c.set_option(wdpm_amd.OPT_DEM32, dem32)
            md = c.run_block(1000, 0.005 / 1000)
            res[dem32] = (md, c.download_water())
    (m1, w1), (m2, w2) = res[0], res[1]
    assert m1 == m2 and n_bit_diff(w1, w2) == 0
    assert w1.min() >= 0.0
    assert abs(float(w1.sum()) - 0.1 * n * n) <= 1e-9 * 0.1 * n * n


@pytest.mark.parametrize("dem32", [0, 2])
@pytest.mark.parametrize("R,C,top,bottom", [(300, 500, 33, 44), (200, 700, 0, 40), (150, 180, 25, 0), (90, 400, 30, 30),
                                            (60, 200, 30, 30), (2079, 900, 33, 33)])
def test_overlapped_iterate_equals_plain_iterate(hip, R, C, top, bottom, dem32):
    """wdpm_iterate_overlapped: boundary rows first (two short launches), the interior on a side stream —
    same bits as the single launch, whatever the windows; too-small interiors fall back to one launch"""
    dem, water, miss = random_case(R + C, R, C)
    bd, bw = pad(dem, water, miss)
    kw = dict(module="add", nrows=R, ncols=C, missingvalue=miss)
    with hip.context(**kw) as a, hip.context(**kw) as b:
        a.upload(bd, bw)
        b.upload(bd, bw)
        b.set_option(wdpm_amd.OPT_DEM32, dem32)     # the windowed launches with the fp64 DEM / with the codes
        for n in (1, 3, 4):
            a.iterate(n)
            b.iterate_overlapped(n, top, bottom)
            # rows the neighbours need are complete on the context's stream alone; everything after a join
            assert bits_equal(a.download_rows(0, max(top, 1)), b.download_rows(0, max(top, 1)))
            assert n_bit_diff(a.download_water(), b.download_water()) == 0
        assert a.run_block(5, 1e-6) == b.run_block(5, 1e-6)


def test_overlapped_iterate_on_a_full_size_slab(hip):
    """the 8-GPU slab of BASELINE config 4 (2079 x 16384): interior launch with DEM codes on two waves per
    SIMD, boundary windows on the fp64 DEM - same bits as the single launch"""
    R, C = 2079, 16384
    dem = hip.synth_dem(C, C)[:R]
    miss = -99999.0
    bd, bw = pad(dem, np.full((R, C), 0.1), miss)
    kw = dict(module="add", nrows=R, ncols=C, missingvalue=miss)
    with hip.context(**kw) as a, hip.context(**kw) as b:
        a.upload(bd, bw)
        b.upload(bd, bw)
        assert b.get_option(wdpm_amd.OPT_DEM32) == 1
        for n in (4, 4):
            a.iterate(n)
            b.iterate_overlapped(n, 33, 44)
            assert n_bit_diff(a.download_water(), b.download_water()) == 0


def _volume_cases():
    from test_seqsum_model import cases
    return list(cases())


@pytest.mark.parametrize("name", [c[0] for c in _volume_cases()])
def test_volume_sum_is_the_sequential_sum(hip, name):
    """wdpm_volume_partial (the drain module's `final_vol`, WDPMCL.c:1259-1266) evaluates the reference's
    left-to-right fp64 sum in parallel on the device (tests/seqsum_model.py): bit-identical to the sequential
    sum on ties, binade crossings, subnormals, negative / NaN / inf depths, chained start values"""
    from seqsum_model import sequential_sum
    x, start = next((c[1], c[2]) for c in _volume_cases() if c[0] == name)
    C = 997
    R = max(1, -(-len(x) // C))
    rng = np.random.default_rng(11)
    water = np.zeros(R * C)
    water[:len(x)] = x
    water = water.reshape(R, C)
    dem = np.round(500 + rng.random((R, C)), 3)
    miss = -99999.0
    dem[rng.random((R, C)) < 0.03] = miss
    water_at_nodata = np.where(dem > miss, water, -99999.0)      # what a drain run reads from an add run's output
    bd, bw = pad(dem, water_at_nodata, miss)
    valid = (bd > miss).ravel()
    want = sequential_sum(bw.ravel()[valid], start)
    with hip.context(module="add", nrows=R, ncols=C, missingvalue=miss) as g:
        g.upload(bd, bw)
        got = g.volume_partial(0, R + 2, start)
        assert np.float64(got).view(np.uint64) == np.float64(want).view(np.uint64) or (np.isnan(got) and np.isnan(want)), \
            (name, got, want)
        # and row ranges chain: top half, then bottom half continued from it
        half = (R + 2) // 2
        chained = g.volume_partial(half, R + 2, g.volume_partial(0, half, start))
        assert np.float64(chained).view(np.uint64) == np.float64(want).view(np.uint64) or np.isnan(want)


def _play(c, script, R, module):
    out = []
    for op in script:
        if op[0] == "begin":
            c.begin_block(op[1])
        elif op[0] == "expect":
            c.expect_max_diff(op[1], op[2])
        elif op[0] == "maxdiff":
            out.append(c.max_diff(*op[1:]))
        elif op[0] == "download":
            out.append(c.download_water())
        elif op[0] == "upload_water":
            c.upload_water(op[1])
        elif op[0] == "upload_rows":
            c.upload_rows(op[1], op[2])
        elif op[0] == "iterate":
            c.iterate(op[1])
        elif op[0] == "iterate_overlapped":
            c.iterate_overlapped(op[1], op[2], op[3])
        elif op[0] == "pass":
            c.single_pass(op[1], op[2])
        elif op[0] == "outlet":
            c.drain_outlet()
        elif op[0] == "volume":
            out.append(c.volume_partial(0, R + 2, 0.0))
    if module == "drain":
        out.append(c.totaldrain)
    return out


@pytest.mark.parametrize("module", ["add", "drain"])
@pytest.mark.parametrize("kernel", [wdpm_amd.KERNEL_FUSED, wdpm_amd.KERNEL_PASS])
def test_lazy_flush_and_snapshot_rotation(hip, oracle, module, kernel):
    """wdpm_begin_block on the HIP back-end copies nothing and flushes nothing by itself: the current raster
    BECOMES the snapshot and the threshold flush rides on the next iteration's loads (or is applied when
    somebody reads or writes the raster first).  Every order of calls must still look exactly like the
    reference's eager flush + copy (WDPMCL.c:1055-1073), which is what the oracle does."""
    R, C = 70, 260
    dem, water, miss = random_case(91, R, C)
    water = np.where(water < 0.05, water * 1e-3, water)          # plenty of cells below the thresholds used
    bd, bw = pad(dem, water, miss)
    rng = np.random.default_rng(5)
    w2 = np.where(bd > miss, 0.2 * rng.random(bd.shape), 0.0)
    rows = 0.3 * rng.random((4, C + 2))
    kw = dict(module=module, nrows=R, ncols=C, missingvalue=miss)
    if module == "drain":
        dr, dc = find_drain(bd)
        kw.update(drainrow=dr, draincol=dc)
    t1, t2 = 1e-3, 4e-3

    scripts = [
        [("begin", t1), ("maxdiff",)],
        [("begin", t1), ("download",)],
        [("begin", t1), ("upload_water", w2), ("iterate", 2), ("maxdiff",), ("download",)],
        [("begin", t1), ("pass", 1, 1), ("maxdiff",), ("download",)],
        [("begin", t1), ("iterate", 1), ("maxdiff",), ("begin", t2), ("iterate", 2), ("maxdiff",), ("download",)],
        [("begin", t1), ("upload_rows", 9, rows), ("iterate", 1), ("maxdiff",), ("download",)],
        [("begin", t1), ("begin", t2), ("iterate", 3), ("maxdiff",), ("download",)],
        [("begin", t2), ("begin", t1), ("maxdiff",), ("iterate", 1), ("maxdiff",)],
        [("begin", t1), ("volume",), ("iterate", 2), ("volume",), ("maxdiff",)],
        [("iterate", 2), ("begin", t1), ("iterate", 4), ("begin", t2), ("maxdiff",), ("iterate", 1), ("maxdiff",), ("download",)],
        [("begin", t1), ("iterate_overlapped", 2, 12, 15), ("maxdiff",), ("download",)],
        [("begin", t1), ("outlet",), ("maxdiff",), ("download",)],
        # the max-change reduction folded into the block's last iteration launch (wdpm_expect_max_diff)
        [("begin", t1), ("expect", 0, R + 2), ("it", 5), ("maxdiff",), ("download",)],
        [("begin", t1), ("expect", 0, R + 2), ("it", 1), ("maxdiff",), ("maxdiff",)],
        [("begin", t1), ("expect", 6, 40), ("it", 4), ("maxdiff", 6, 40), ("maxdiff",), ("maxdiff", 6, 41)],
        [("begin", t1), ("expect", 6, 40), ("iterate_overlapped", 3, 12, 15), ("maxdiff", 6, 40), ("download",)],
        [("begin", t1), ("expect", 0, R + 2), ("it", 3), ("upload_rows", 9, rows), ("maxdiff",)],
        [("begin", t1), ("expect", 0, R + 2), ("it", 3), ("it", 2), ("maxdiff",)],
        [("it", 3), ("expect", 0, R + 2), ("it", 2), ("maxdiff",), ("begin", t2), ("expect", 0, R + 2), ("it", 2), ("maxdiff",)],
    ]

    play = lambda c, script: _play(c, script, R, module)   # noqa: E731

    for script in scripts:
        with hip.context(kernel=kernel, **kw) as g, oracle.context(**kw) as o:
            for c in (g, o):
                c.upload(bd, bw)
                c.totaldrain = 0.25
            got, want = play(g, script), play(o, script)
        assert len(got) == len(want)
        for a, b in zip(got, want):
            if isinstance(a, np.ndarray):
                assert n_bit_diff(a, b) == 0, [op[0] for op in script]
            else:
                assert a == b, ([op[0] for op in script], a, b)


@pytest.mark.parametrize("module", ["add", "drain"])
def test_random_call_sequences(hip, oracle, module):
    """The library's lazy state (pending flush, owed drain(), raster rotation, tile flags, the folded max-change) driven by
    RANDOM sequences of ABI calls on rasters of random shape, chunk height and kernel: after every observing call the HIP
    context must show exactly what the oracle's eager implementation shows."""
    import random
    lo, hi = (int(v) for v in os.environ.get("WDPM_FUZZ_SEEDS", "0:60").split(":"))     # a longer hunt: WDPM_FUZZ_SEEDS=60:2000
    for seed in range(lo, hi):
        rng = random.Random(seed * 3 + (module == "drain"))
        R, C = rng.randint(4, 90), rng.randint(3, 420)
        dem, water, miss = random_case(500 + seed, R, C, missing_frac=rng.choice([0.0, 0.05, 0.4]), dry_frac=rng.choice([0.1, 0.6, 0.95]))
        water = np.where(water < 0.05, water * 1e-3, water)
        bd, bw = pad(dem, water, miss)
        nrng = np.random.default_rng(seed)
        kw = dict(module=module, nrows=R, ncols=C, missingvalue=miss)
        if module == "drain":
            dr, dc = find_drain(bd)
            kw.update(drainrow=dr, draincol=dc)
        script = []
        for _ in range(rng.randint(6, 14)):
            op = rng.choice(["begin", "begin", "it", "it", "it", "iterate_overlapped", "maxdiff", "maxdiff", "expect", "download",
                             "upload_rows", "upload_water", "pass", "outlet", "volume"])
            if op == "begin":
                script.append(("begin", rng.choice([0.0, 1e-3, 4e-3])))
            elif op == "it":
                script.append(("iterate", rng.randint(1, 5)))
            elif op == "iterate_overlapped":
                if R + 2 >= 40:
                    top, bot = rng.randint(0, 15), rng.randint(0, 15)
                    script.append(("iterate_overlapped", rng.randint(1, 3), top, bot))
            elif op == "maxdiff":
                lo = rng.randint(0, R); hi = rng.randint(lo + 1, R + 2)
                script.append(("maxdiff",) if rng.random() < 0.6 else ("maxdiff", lo, hi))
            elif op == "expect":
                lo = rng.randint(0, R); hi = rng.randint(lo + 1, R + 2)
                script.append(("expect", 0, R + 2) if rng.random() < 0.6 else ("expect", lo, hi))
            elif op == "upload_rows":
                n = rng.randint(1, min(5, R + 2)); r0 = rng.randint(0, R + 2 - n)
                rows = np.where(bd[r0:r0 + n] > miss, 0.3 * nrng.random((n, C + 2)), 0.0)
                script.append(("upload_rows", r0, rows))
            elif op == "upload_water":
                script.append(("upload_water", np.where(bd > miss, 0.2 * nrng.random(bd.shape), 0.0)))
            elif op == "pass":
                script.append(("pass", rng.randint(1, 3), rng.randint(1, 3)))
            elif op == "outlet":
                if module == "drain":
                    script.append(("outlet",))
            else:
                script.append((op,))
        # round 5: one script in three also holds a LONG run of iterations somewhere (wdpm_iterate replays those of small rasters as HIP graphs
        # of 32 launches; a generator of its own, so that the scripts of earlier rounds stay what they were)
        rng2 = random.Random(seed * 7 + 1)
        if rng2.random() < 0.34:
            script.insert(rng2.randint(0, len(script)), ("iterate", rng2.randint(34, 75)))
        script.append(("maxdiff",))
        script.append(("download",))
        kernel = rng.choice([wdpm_amd.KERNEL_FUSED, wdpm_amd.KERNEL_FUSED, wdpm_amd.KERNEL_PASS])
        chunk = rng.choice([0, 0, 6, 12, 30])
        with hip.context(kernel=kernel, chunk_rows=chunk, **kw) as g, oracle.context(**kw) as o:
            for c in (g, o):
                c.upload(bd, bw)
                c.totaldrain = 0.25
            if rng.random() < 0.3:
                g.set_option(wdpm_amd.capi.OPT_TILES, 0)
            if module == "add" and rng.random() < 0.4:
                g.set_option(wdpm_amd.OPT_DEM32, 2)
            got, want = _play(g, script, R, module), _play(o, script, R, module)
        names = [op[0] for op in script]
        assert len(got) == len(want)
        for a, b in zip(got, want):
            if isinstance(a, np.ndarray):
                assert n_bit_diff(a, b) == 0, (seed, R, C, kernel, chunk, names)
            else:
                assert a == b, (seed, R, C, kernel, chunk, names, a, b)


@pytest.mark.parametrize("chunk", [0, 6])
def test_drain_outlet_at_every_window_position(hip, oracle, chunk):
    """the outlet forced onto every row offset of a 9-row window and onto columns at every lane / strip
    alignment (first and last lane of a strip, both sides of a strip boundary, the halo columns): water and
    totaldrain after a few iterations, triangle kernel (chunk 0 on a raster this small) and marching kernel"""
    R, C = 32, 400
    dem, water, miss = random_case(77, R, C, missing_frac=0.02, dry_frac=0.1)
    bd, bw = pad(dem, water, miss)
    cols = [1, 2, 3, 4, 168, 170, 171, 172, 173, 178, 179, 180, 189, 190, 191, 192, 193, 340, 341, 342, 343, 399, 400]
    for dr in list(range(1, 14)) + [R - 1, R]:
        for dc in cols[(dr * 5) % 3::3]:
            kw = dict(module="drain", nrows=R, ncols=C, missingvalue=miss, drainrow=dr, draincol=dc)
            with hip.context(kernel=wdpm_amd.KERNEL_FUSED, chunk_rows=chunk, **kw) as g, oracle.context(**kw) as o:
                for c in (g, o):
                    c.upload(bd, bw)
                    c.totaldrain = 0.125
                    c.iterate(3)
                assert g.totaldrain == o.totaldrain, (dr, dc)
                assert n_bit_diff(g.download_water(), o.download_water()) == 0, (dr, dc)
                for c in (g, o):
                    c.begin_block(1e-3)
                    c.iterate(2)
                assert g.drain_stats() == o.drain_stats() and g.max_diff() == o.max_diff(), (dr, dc)


def test_guard_bands_notice_a_stray_write(hip):
    """the net the whole GPU suite runs in (tests/conftest.py: WDPM_GUARD_KB, checked whenever a context closes) does catch:
    three bytes written just in front of the current water raster and two just behind its dump area are counted"""
    import ctypes as C
    assert int(os.environ.get("WDPM_GUARD_KB", "0")) > 0
    R, Cc = 20, 30
    with open("/proc/self/maps") as f:      # the HIP runtime this process already runs on (never a second one)
        path = next(line.split()[-1] for line in f if "libamdhip64.so" in line)
    rt = C.CDLL(path)
    rt.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    c = hip.context(module="add", nrows=R, ncols=Cc, missingvalue=-99999.0)
    try:
        c.upload(np.full((R + 2, Cc + 2), 500.0), np.full((R + 2, Cc + 2), 0.1))
        c.iterate(2)
        assert c.get_option(wdpm_amd.capi.OPT_GUARD_BAD) == 0
        p = c.water_ptr()
        assert rt.hipMemset(p - 3, 0, 3) == 0
        assert rt.hipMemset(p + ((R + 2) * (Cc + 2) + 192) * 8, 0, 2) == 0      # 192 doubles: the dump area (wdpm_create)
        assert rt.hipDeviceSynchronize() == 0
        assert c.get_option(wdpm_amd.capi.OPT_GUARD_BAD) == 5
    finally:
        hip.dll.wdpm_destroy(c._h)          # not through close(): that one would (rightly) complain
        c._h = None


@pytest.mark.parametrize("module", ["add", "drain"])
def test_water_kinds_and_the_gate_free_variants(hip, oracle, module):
    """WDPM_OPT_WATER_KINDS / WDPM_OPT_PLAIN_WATER (round 3): the kernels without the centre gate run exactly while the library
    knows that every cell which may not give water holds +0.0 - what its scan of every upload finds, what a threshold flush
    removes, what a partial upload or a raw pointer brings back - and the bits are the oracle's in every one of these states
    (WDPMCL.c:1099 is the test they leave out)"""
    from wdpm_amd.capi import OPT_PLAIN_WATER, OPT_WATER_KINDS
    R, C = 130, 420
    dem, water, miss = random_case(4242, R, C)
    nodata = dem <= miss
    assert nodata.any() and (water[nodata] == 0).all() and (water >= 0).all()
    cases = {
        "clean": (water, 0),
        "minus_99999_on_nodata": (np.where(nodata, -99999.0, water), 2),       # what the reference's own output rasters hold there
        "negative_on_a_valid_cell": (np.where((np.arange(R * C).reshape(R, C) % 97) == 0, -0.25, water), 2),
        "water_on_nodata": (np.where(nodata, 0.125, water), 4),
        "nan": (np.where((np.arange(R * C).reshape(R, C) % 1013) == 5, np.nan, water), 4),
        "negative_zero": (np.where(water == 0, -0.0, water), 1),
    }
    for name, (w, kinds) in cases.items():
        bd, bw = pad(dem, w, miss)
        kw = dict(module=module, nrows=R, ncols=C, missingvalue=miss)
        if module == "drain":
            dr, dc = find_drain(bd)
            kw.update(drainrow=dr, draincol=dc)
        with hip.context(kernel=wdpm_amd.KERNEL_FUSED, **kw) as g, oracle.context(**kw) as o:
            for c in (g, o):
                c.upload(bd, bw)
                c.totaldrain = 0.0
            assert g.get_option(OPT_WATER_KINDS) == kinds, name
            assert g.get_option(OPT_PLAIN_WATER) == int(kinds == 0), name
            for block in range(3):
                assert g.run_block(7, 1e-5) == o.run_block(7, 1e-5), (name, block)     # NaN == NaN is False: max diff never is NaN
                assert n_bit_diff(g.download_water(), o.download_water()) == 0, (name, block)
                assert g.totaldrain == o.totaldrain
                # the flush of the first launch removes what is negative; the rest stays for good
                assert g.get_option(OPT_WATER_KINDS) == (kinds & ~2), (name, block)
            # a partial upload brings negative depths back (until the next flush) ...
            rows = bw[10:13].copy()
            rows[1, 5:9] = -1.0
            g.upload_rows(10, rows)
            o.upload_rows(10, rows)
            assert g.get_option(OPT_WATER_KINDS) == (kinds & ~2) | 2 and g.get_option(OPT_PLAIN_WATER) == 0
            g.iterate(3)                                                        # no flush: still there (gated variants)
            o.iterate(3)
            assert g.get_option(OPT_WATER_KINDS) & 2
            assert g.run_block(5, 1e-5) == o.run_block(5, 1e-5)
            assert n_bit_diff(g.download_water(), o.download_water()) == 0, name
            assert g.get_option(OPT_WATER_KINDS) == (kinds & ~2)
            # ... and whoever takes the raw device pointer may write anything
            g.water_ptr()
            assert g.get_option(OPT_WATER_KINDS) & 4 and g.get_option(OPT_PLAIN_WATER) == 0
            g.iterate(4)
            o.iterate(4)
            assert n_bit_diff(g.download_water(), o.download_water()) == 0, name
