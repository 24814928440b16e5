"""Rasters of more than 4 GiB each (byte offsets past 2^32, 5.8 * 10^8 cells): three copies of one 8001 x 24000 basin stacked with
NODATA rows between them.  The copies cannot exchange water, start on the same row colour (8001 = 0 mod 3), and so must
come out identical bit for bit - to each other and to the basin run on its own in a raster of 1.5 GiB.  A size-independent
property, no oracle run of half a billion cells needed; every kernel of the block loop addresses past 4 GiB here."""
import numpy as np
import pytest

import wdpm_amd
from helpers import n_bit_diff

pytestmark = pytest.mark.gpu


def enough_hbm(gib):
    """skip (not fail) on a card that is not an MI355X-sized one, or is shared: these tests take up to 80 GB of HBM"""
    import torch
    free, _ = torch.cuda.mem_get_info(0)
    if free < gib * 2 ** 30:
        pytest.skip("needs %d GiB of free device memory, %.0f GiB are free" % (gib, free / 2 ** 30))

@pytest.mark.parametrize("rows,cols,copies", [(8000, 24000, 3), (8798, 44000, 5)])
def test_stacked_basins_past_4gib(hip, rows, cols, copies):
    """(8000, 24000, 3): 4.6 GB per raster.  (8798, 44000, 5): 43994 x 44000, 1.936e9 padded cells - just under the 2e9 cells one
    context takes (wdpm_create), 15.5 GB per raster, 70 GB of HBM"""
    psutil = pytest.importorskip("psutil")
    need = (48 if copies == 3 else 128) * 2 ** 30
    if psutil.virtual_memory().available < need:
        pytest.skip("needs %d GiB of free host memory" % (need >> 30))
    enough_hbm(24 if copies == 3 else 80)
    miss, thres = -99999.0, 5e-6
    rng = np.random.default_rng(cols)
    y = np.arange(rows)[:, None]
    x = np.arange(cols)[None, :]
    dem = np.round(500.0 + 3.0 * np.sin(x / 37.0) * np.cos(y / 53.0) - 1e-4 * x + 0.05 * rng.random((rows, cols)), 4)
    dem[rng.random((rows, cols)) < 0.01] = miss
    water = np.where(dem > miss, 0.1, 0.0)
    water[rng.random((rows, cols)) < 0.3] = 0.0

    def padded(ncopies):
        R = ncopies * (rows + 1) - 1
        bd = np.full((R + 2, cols + 2), miss)
        bw = np.zeros((R + 2, cols + 2))
        for k in range(ncopies):
            r0 = 1 + k * (rows + 1)
            bd[r0:r0 + rows, 1:-1] = dem
            bw[r0:r0 + rows, 1:-1] = water
        return R, bd, bw

    def run(ncopies):
        R, bd, bw = padded(ncopies)
        with hip.context(module="add", nrows=R, ncols=cols, missingvalue=miss, kernel=wdpm_amd.KERNEL_FUSED) as c:
            c.upload(bd, bw)
            del bd, bw
            md = [c.run_block(3, thres), c.run_block(2, thres)]
            dem32 = c.get_option(wdpm_amd.OPT_DEM32)
            return md, c.download_water(), dem32

    md1, w1, _ = run(1)
    md3, w3, dem32 = run(copies)
    assert w3.nbytes > 2 ** 32 and dem32 == 1
    assert md1 == md3                                     # the maximum over identical basins
    for k in range(copies):
        r0 = 1 + k * (rows + 1)
        assert n_bit_diff(w3[r0:r0 + rows], w1[1:1 + rows]) == 0, k
    for k in range(1, copies):
        assert not w3[k * (rows + 1)].any()                                # the separator rows stay dry


def test_more_cells_than_one_context_takes(hip):
    """46 007 x 46 000 = 2.12e9 padded cells: refused by wdpm_create, run as two row blocks on the one device (what WDPMCL does
    by itself for such a raster) through the calls WDPMCL makes - set-up from the un-padded file rasters, counts, outlet search,
    block loop, volume sum, masked un-padded download.  Six stacked copies of a basin that must all equal the basin run alone."""
    from wdpm_amd.rowblock import Group
    psutil = pytest.importorskip("psutil")
    if psutil.virtual_memory().available < 200 * 2 ** 30:
        pytest.skip("needs 200 GiB of free host memory")
    enough_hbm(96)
    rows, cols, copies, miss, thres = 7667, 46000, 6, -99999.0, 5e-6
    rng = np.random.default_rng(46000)
    y = np.arange(rows)[:, None]
    x = np.arange(cols)[None, :]
    dem = np.round(500.0 + 2.0 * np.sin(x / 41.0) * np.cos(y / 29.0) - 1e-4 * x + 0.05 * rng.random((rows, cols)), 4)
    dem[rng.random((rows, cols)) < 0.01] = miss
    wet = (rng.random((rows, cols)) < 0.6) & (dem > miss)
    R = copies * (rows + 1) - 1
    with pytest.raises(wdpm_amd.capi.WdpmError, match="too large"):
        hip.context(module="add", nrows=R, ncols=cols, missingvalue=miss)
    fdem = np.full((R, cols), miss)                       # the file rasters: un-padded
    fwat = np.zeros((R, cols))
    for k in range(copies):
        r0 = k * (rows + 1)
        fdem[r0:r0 + rows] = dem
        fwat[r0:r0 + rows] = np.where(wet, 0.05, 0.0)
    assert (R + 2) * (cols + 2) > 2e9
    with Group(hip, "add", R, cols, miss, [0, 0], exchange_every=2) as grp:
        grp.upload_unpadded(fdem, fwat, op=1, add=0.05, rof=0.0)          # WDPMCL.c:727-740: wet cells + 0.05, dry cells stay dry
        del fwat
        nvalid, nwet, wmax = grp.count_stats()
        assert (nvalid, nwet, wmax) == (copies * int((dem > miss).sum()), copies * int(wet.sum()), 0.1)
        k = int(np.argmin(np.where(dem > 0, dem, np.inf)))                # :1005-1017 the first row-major minimum
        assert grp.find_drain() == (float(dem.flat[k]), k // cols + 1, k % cols + 1)
        md = [grp.run_block(3, thres), grp.run_block(2, thres)]
        _, vol = grp.drain_stats()
        w = grp.download_unpadded(True)
    with hip.context(module="add", nrows=rows, ncols=cols, missingvalue=miss, kernel=wdpm_amd.KERNEL_FUSED) as c:
        c.upload(np.pad(dem, 1, constant_values=miss), np.pad(np.where(wet, 0.1, 0.0), 1))
        md1 = [c.run_block(3, thres), c.run_block(2, thres)]
        w1 = np.where(dem > miss, c.download_water()[1:-1, 1:-1], miss)
    assert md == md1
    for k in range(copies):
        r0 = k * (rows + 1)
        assert n_bit_diff(w[r0:r0 + rows], w1) == 0, k
    assert vol == float(np.add.accumulate(w[fdem > miss])[-1])            # the sequential row-major sum, :1262-1268
