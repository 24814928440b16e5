"""Worker for tests/test_mock_rccl.py: thread-per-rank groups whose halos go through the library's RCCL path
(wdpm_comm_exchange), with tests/mock_rccl/libmock_rccl.so standing in for librccl (set by the parent through
WDPM_RCCL_LIB before this process loads the library)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import wdpm_amd
from helpers import bits_equal, find_drain, pad, random_case
from wdpm_amd.rowblock import Group

hip = wdpm_amd.load_hip()
assert hip.dll.wdpm_comm_available() == 1 and b"2.99.99" in hip.dll.wdpm_comm_version(), hip.dll.wdpm_comm_version()
n_ok = 0
for module, R, C, n, k, blocks in (("add", 300, 420, 3, 3, [13, 8]), ("add", 260, 200, 4, 1, [9, 4]), ("drain", 240, 380, 2, 2, [9, 6]),
                                   ("drain", 330, 250, 3, 1, [7, 5]), ("add", 1300, 1100, 5, 4, [21, 10])):
    dem, water, miss = random_case(50 + n, R, C)
    if R > 1000:                                  # big enough for the marching kernel, dry tiles and the folded max diff
        water[:, :] = 0.0
        water[100:160, 300:420] = 0.3
        water[900:960, 100:200] = 0.2
        water[dem <= miss] = 0.0
    bd, bw = pad(dem, water, miss)
    kw = {}
    if module == "drain":
        dr, dc = find_drain(bd)
        kw = dict(drainrow=dr, draincol=dc)
    res = {}
    for name, devices in (("one", [0]), ("many", [0] * n)):
        with Group(hip, module, R, C, miss, devices, exchange_every=k, **kw) as g:
            if name == "many":
                assert g.size == n and wdpm_amd.HALO_NAMES[g.halo_kind] == "rccl", (g.size, g.halo_kind)
            g.upload(bd, bw)
            if kw:
                g.set_totaldrain(max(bw[dr, dc], 0.0))
            out = []
            for b in blocks:
                out.append(g.run_block(b, 1e-4))
                if kw:
                    out += list(g.drain_stats()) + [g.totaldrain()]
            res[name] = (g.download_water(), out)
    assert bits_equal(res["one"][0], res["many"][0]), (module, n, k)
    assert res["one"][1] == res["many"][1], (module, n, k, res["one"][1], res["many"][1])
    n_ok += 1
print("MOCK_RCCL_GROUPS_OK", n_ok)

# the random row-block jobs of tests/test_rowblock.py (2-7 slabs, any module, outlet anywhere, every slab-edge geometry of the
# outlet) once more with the halos going through wdpm_comm_exchange instead of peer copies, against the oracle
from test_rowblock import _check_outlet_on_slab_edges, _random_group_jobs   # noqa: E402
oracle = wdpm_amd.load(os.path.join(ROOT, "oracle", "_build", "libwdpm_oracle.so"))
lo, hi = (int(v) for v in os.environ.get("WDPM_FUZZ_SEEDS", "3000:3060").split(":"))
_random_group_jobs(hip, oracle, range(lo, hi))
_check_outlet_on_slab_edges(hip, oracle)
print("MOCK_RCCL_RANDOM_JOBS_OK", hi - lo)
