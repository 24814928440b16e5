import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "libwdpm_oracle.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (oracle/wdpm_oracle.c), built on demand.  The checker, never the product."""
    import wdpm_amd
    if not os.path.exists(ORACLE_SO):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    return wdpm_amd.load(ORACLE_SO)


# Guard bands around the library's big device buffers (include/wdpm.h: WDPM_OPT_GUARD_BAD; the GPU pool has no address
# sanitizer): on for every test process and for the WDPMCL children (which exit with status 3 if a guard byte changed).
os.environ.setdefault("WDPM_GUARD_KB", "64")


@pytest.fixture(scope="session")
def hip():
    """The product library; fails loudly (no fallback) if it has not been built.  Every context and every group of this library
    is asked, when it is closed, whether any kernel wrote outside its buffers."""
    import ctypes as C

    import wdpm_amd
    from wdpm_amd import capi, rowblock
    lib = wdpm_amd.load_hip()

    def guard_damage(ctx_handle):
        v = C.c_int64()
        lib.check(lib.dll.wdpm_get_option(ctx_handle, capi.OPT_GUARD_BAD, C.byref(v)))
        return v.value

    if not getattr(capi.Context, "_guard_checked", False):
        ctx_close, grp_close = capi.Context.close, rowblock.Group.close

        def checked_ctx_close(self):
            bad = guard_damage(self._h) if self._h and self.lib is lib else 0
            ctx_close(self)
            assert bad == 0, f"{bad} guard bytes around a context's device buffers were overwritten"

        def checked_grp_close(self):
            bad = 0
            if self._h and self.lib is lib:
                bad = sum(guard_damage(self.rank_ctx(i)) for i in range(self.size))
            grp_close(self)
            assert bad == 0, f"{bad} guard bytes around a group's device buffers were overwritten"

        capi.Context.close, rowblock.Group.close = checked_ctx_close, checked_grp_close
        capi.Context._guard_checked = True
    return lib


@pytest.fixture(scope="session")
def stencil_cases():
    z = np.load(os.path.join(GOLDEN, "stencil_cases.npz"))
    index = json.loads(bytes(z["index_json"]).decode())
    return z, index


@pytest.fixture(scope="session")
def basin5():
    import gzip
    with gzip.open(os.path.join(GOLDEN, "basin5.asc.gz"), "rt") as f:
        hdr = [f.readline().split() for _ in range(6)]
        vals = np.array(f.read().split(), dtype=np.float64)
    ncols, nrows = int(float(hdr[0][1])), int(float(hdr[1][1]))
    return vals.reshape(nrows, ncols), {h[0]: float(h[1]) for h in hdr}
