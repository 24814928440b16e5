"""ctypes binding of the C ABI declared in include/wdpm.h.

The same binding serves any shared library exporting that ABI.  The product library is
``wdpm_amd/csrc/libwdpm_hip.so`` (HIP kernels for gfx950); :func:`load_hip` loads it and fails
loudly when it is missing — there is no CPU fallback in this package.  Tests load the CPU
oracle through :func:`load` with an explicit path; the package itself never does.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

ADD, SUBTRACT, DRAIN = 0, 1, 2
MODULES = {"add": ADD, "subtract": SUBTRACT, "drain": DRAIN}
KERNEL_AUTO, KERNEL_PASS, KERNEL_FUSED = 0, 1, 2
OPT_SIGNED_ZERO_SAFE = 1
OPT_DEM32 = 2
OPT_TILES, OPT_TILES_SEEN, OPT_TILES_WORKED, OPT_SPARSE, OPT_GUARD_BAD, OPT_WATER_KINDS, OPT_PLAIN_WATER, OPT_DEM16 = 3, 4, 5, 6, 7, 8, 9, 10
OPT_GRAPH_LAUNCHES = 11
HALO_AUTO, HALO_RCCL, HALO_PEER, HALO_HOST = 0, 1, 2, 3
HALO_NAMES = {0: "none", 1: "rccl", 2: "peer", 3: "host"}
COMM_ID_BYTES = 128

_HERE = os.path.dirname(os.path.abspath(__file__))
# WDPM_HIP_LIB lets a tuning run point at an alternative build of the same HIP library
HIP_LIB_PATH = os.environ.get("WDPM_HIP_LIB") or os.path.join(_HERE, "csrc", "libwdpm_hip.so")


class Params(C.Structure):
    """struct wdpm_params (include/wdpm.h)."""
    _fields_ = [
        ("module", C.c_int32), ("nrows", C.c_int32), ("ncols", C.c_int32),
        ("drainrow", C.c_int32), ("draincol", C.c_int32),
        ("slab_row0", C.c_int32), ("slab_rows", C.c_int32),
        ("device", C.c_int32), ("kernel", C.c_int32), ("chunk_rows", C.c_int32),
        ("missingvalue", C.c_double),
    ]


class SlabStruct(C.Structure):
    """struct wdpm_slab"""
    _fields_ = [("own_lo", C.c_int32), ("own_hi", C.c_int32), ("row0", C.c_int32), ("rows", C.c_int32),
                ("up", C.c_int32), ("down", C.c_int32)]


class SetupStruct(C.Structure):
    """struct wdpm_setup"""
    _fields_ = [("op", C.c_int32), ("add", C.c_double), ("rof", C.c_double), ("sub", C.c_double)]


class HaloOp(C.Structure):
    """struct wdpm_halo_op"""
    _fields_ = [("peer", C.c_int32), ("row", C.c_int32), ("nrows", C.c_int32)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                          C.POINTER(C.c_void_p), C.POINTER(C.c_int64))
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_double))


class HostTransportStruct(C.Structure):
    """struct wdpm_host_transport"""
    _fields_ = [("user", C.c_void_p), ("exchange", EXCHANGE_FN), ("allgather", ALLGATHER_FN)]


# name -> (restype, argtypes); every symbol include/wdpm.h declares
_dp = C.POINTER(C.c_double)
_vp = C.c_void_p
_ip = C.POINTER(C.c_int32)
SYMBOLS = {
    "wdpm_create": (C.c_int, [C.POINTER(_vp), C.POINTER(Params)]),
    "wdpm_destroy": (None, [_vp]),
    "wdpm_last_error": (C.c_char_p, []),
    "wdpm_backend_name": (C.c_char_p, []),
    "wdpm_abi_version": (C.c_int, []),
    "wdpm_upload": (C.c_int, [_vp, _vp, _vp]),
    "wdpm_upload_water": (C.c_int, [_vp, _vp]),
    "wdpm_download_water": (C.c_int, [_vp, _vp]),
    "wdpm_download_rows": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp]),
    "wdpm_upload_rows": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp]),
    "wdpm_set_totaldrain": (C.c_int, [_vp, C.c_double]),
    "wdpm_get_totaldrain": (C.c_int, [_vp, _dp]),
    "wdpm_begin_block": (C.c_int, [_vp, C.c_double]),
    "wdpm_iterate": (C.c_int, [_vp, C.c_int32]),
    "wdpm_iterate_overlapped": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32]),
    "wdpm_pass": (C.c_int, [_vp, C.c_int32, C.c_int32]),
    "wdpm_drain_outlet": (C.c_int, [_vp]),
    "wdpm_max_diff": (C.c_int, [_vp, C.c_int32, C.c_int32, _dp]),
    "wdpm_expect_max_diff": (C.c_int, [_vp, C.c_int32, C.c_int32]),
    "wdpm_drain_stats": (C.c_int, [_vp, _dp, _dp]),
    "wdpm_volume_partial": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_double, _dp]),
    "wdpm_run_block": (C.c_int, [_vp, C.c_int32, C.c_double, _dp]),
    "wdpm_water_ptr": (C.c_int, [_vp, C.POINTER(_vp)]),
    "wdpm_dem_ptr": (C.c_int, [_vp, C.POINTER(_vp)]),
    "wdpm_set_stream": (C.c_int, [_vp, _vp]),
    "wdpm_synchronize": (C.c_int, [_vp]),
    "wdpm_timing_reset": (C.c_int, [_vp]),
    "wdpm_timing_get": (C.c_int, [_vp, C.POINTER(C.c_int64), _dp]),
    "wdpm_timing_get_steady": (C.c_int, [_vp, C.POINTER(C.c_int64), _dp]),
    "wdpm_timing_get_exchange": (C.c_int, [_vp, C.POINTER(C.c_int64), _dp]),
    "wdpm_build_info": (C.c_char_p, []),
    "wdpm_balance_info": (C.c_int, [_vp, C.POINTER(C.c_int32), _dp]),
    "wdpm_device_info": (C.c_int, [_vp, C.POINTER(C.c_int32), C.c_char_p, C.c_int32]),
    "wdpm_copy_rows": (C.c_int, [_vp, C.c_int32, _vp, C.c_int32, C.c_int32]),
    "wdpm_set_last_error": (None, [C.c_char_p]),
    "wdpm_enable_peer_access": (C.c_int, [_vp, _vp]),
    "wdpm_comm_available": (C.c_int, []),
    "wdpm_comm_version": (C.c_char_p, []),
    "wdpm_comm_unique_id": (C.c_int, [_vp]),
    "wdpm_comm_init_rank": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp]),
    "wdpm_comm_init_all": (C.c_int, [C.POINTER(_vp), C.c_int32]),
    "wdpm_comm_size": (C.c_int, [_vp, _ip, _ip]),
    "wdpm_comm_exchange": (C.c_int, [_vp, C.c_int32, C.POINTER(HaloOp), C.c_int32, C.POINTER(HaloOp)]),
    "wdpm_comm_allgather": (C.c_int, [_vp, _dp, C.c_int32, _dp]),
    "wdpm_comm_abort": (C.c_int, [_vp]),
    "wdpm_partition": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(SlabStruct)]),
    "wdpm_rank_create": (C.c_int, [C.POINTER(_vp), C.POINTER(Params), C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp, _vp]),
    "wdpm_rank_destroy": (None, [_vp]),
    "wdpm_rank_ctx": (_vp, [_vp]),
    "wdpm_rank_slab": (C.c_int, [_vp, C.c_int32, C.POINTER(SlabStruct)]),
    "wdpm_rank_info": (C.c_int, [_vp, _ip, _ip, _ip]),
    "wdpm_rank_upload": (C.c_int, [_vp, _vp, _vp]),
    "wdpm_rank_upload_global": (C.c_int, [_vp, _vp, _vp]),
    "wdpm_rank_set_totaldrain": (C.c_int, [_vp, C.c_double]),
    "wdpm_rank_get_totaldrain": (C.c_int, [_vp, _dp]),
    "wdpm_rank_begin_block": (C.c_int, [_vp, C.c_double]),
    "wdpm_rank_iterate": (C.c_int, [_vp, C.c_int32]),
    "wdpm_rank_exchange": (C.c_int, [_vp]),
    "wdpm_rank_max_diff": (C.c_int, [_vp, _dp]),
    "wdpm_rank_run_block": (C.c_int, [_vp, C.c_int32, C.c_double, _dp]),
    "wdpm_rank_drain_stats": (C.c_int, [_vp, _dp, _dp]),
    "wdpm_rank_download_owned": (C.c_int, [_vp, _vp]),
    "wdpm_upload_unpadded": (C.c_int, [_vp, _vp, _vp, C.POINTER(SetupStruct)]),
    "wdpm_count_stats": (C.c_int, [_vp, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _dp]),
    "wdpm_find_drain": (C.c_int, [_vp, C.c_int32, C.c_int32, _dp, _ip, _ip]),
    "wdpm_set_drain": (C.c_int, [_vp, C.c_int32, C.c_int32]),
    "wdpm_get_cell": (C.c_int, [_vp, C.c_int32, C.c_int32, _dp, _dp]),
    "wdpm_download_unpadded": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, _vp]),
    "wdpm_group_upload_unpadded": (C.c_int, [_vp, _vp, _vp, C.POINTER(SetupStruct)]),
    "wdpm_group_count_stats": (C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _dp]),
    "wdpm_group_find_drain": (C.c_int, [_vp, _dp, _ip, _ip]),
    "wdpm_group_set_drain": (C.c_int, [_vp, C.c_int32, C.c_int32]),
    "wdpm_group_get_cell": (C.c_int, [_vp, C.c_int32, C.c_int32, _dp, _dp]),
    "wdpm_group_download_unpadded": (C.c_int, [_vp, C.c_int32, _vp]),
    "wdpm_group_halo": (C.c_int, [_vp]),
    "wdpm_group_rank": (_vp, [_vp, C.c_int32]),
    "wdpm_group_enqueue_stats": (C.c_int, [_vp, _dp, _dp, C.POINTER(C.c_int64)]),
    "wdpm_group_create": (C.c_int, [C.POINTER(_vp), C.POINTER(Params), C.c_int32, C.POINTER(C.c_int32), C.c_int32]),
    "wdpm_group_destroy": (None, [_vp]),
    "wdpm_group_size": (C.c_int, [_vp]),
    "wdpm_group_upload": (C.c_int, [_vp, _vp, _vp]),
    "wdpm_group_download_water": (C.c_int, [_vp, _vp]),
    "wdpm_group_set_totaldrain": (C.c_int, [_vp, C.c_double]),
    "wdpm_group_get_totaldrain": (C.c_int, [_vp, _dp]),
    "wdpm_group_run_block": (C.c_int, [_vp, C.c_int32, C.c_double, _dp]),
    "wdpm_group_drain_stats": (C.c_int, [_vp, _dp, _dp]),
    "wdpm_get_option": (C.c_int, [_vp, C.c_int32, C.POINTER(C.c_int64)]),
    "wdpm_set_option": (C.c_int, [_vp, C.c_int32, C.c_int64]),
    "wdpm_synth_dem": (C.c_int, [C.c_int32, C.c_uint64, _vp]),
    "wdpm_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(_vp)]),
    "wdpm_host_free": (None, [_vp]),
}


class WdpmError(RuntimeError):
    pass


_runtime_preloaded = False


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64.so (SONAME
    libamdhip64.so.7) and load it by file name; libwdpm_hip.so needs libamdhip64.so.7 by SONAME.
    If our library came first the dynamic linker would map /opt/rocm's copy and torch would later
    map its own beside it — two runtimes, and stream handles / device pointers shared between
    torch (RCCL halo exchange) and this library would be meaningless.  Mapping torch's copy first
    (without importing torch) makes both resolve to the same runtime.  No torch, no preload: the
    system runtime is used, as the C command line does."""
    global _runtime_preloaded
    if _runtime_preloaded or os.environ.get("WDPM_NO_TORCH_RUNTIME"):
        return
    _runtime_preloaded = True
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


_SINCE_ROUND_5 = {"wdpm_device_info"}


class Lib:
    """A loaded shared library exporting the wdpm C ABI."""

    def __init__(self, path: str):
        if not os.path.exists(path):
            raise WdpmError(f"wdpm library not found: {path}")
        self.path = path
        _preload_torch_hip_runtime()
        self.dll = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            if name in _SINCE_ROUND_5 and os.path.basename(path).startswith("alt_") and not hasattr(self.dll, name):
                continue                  # an A/B build of an earlier revision (tools/build_alt.sh): timing runs do not call these
            fn = getattr(self.dll, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if self.dll.wdpm_abi_version() != 1:
            raise WdpmError("wdpm ABI version mismatch")

    @property
    def backend(self) -> str:
        return self.dll.wdpm_backend_name().decode()

    def check(self, rc: int):
        if rc != 0:
            raise WdpmError(self.dll.wdpm_last_error().decode() or f"wdpm error {rc}")

    def synth_dem(self, n: int, seed: int) -> np.ndarray:
        out = np.empty((n, n), dtype=np.float64)
        self.check(self.dll.wdpm_synth_dem(n, seed, out.ctypes.data))
        return out

    def context(self, **kw) -> "Context":
        return Context(self, **kw)


@dataclass
class _Slab:
    row0: int
    rows: int


class Context:
    """One raster (or one row slab of a raster) resident on one device — wraps wdpm_ctx."""

    def __init__(self, lib: Lib, module, nrows: int, ncols: int, missingvalue: float,
                 drainrow: int = 0, draincol: int = 0, slab_row0: int = 0, slab_rows: int = 0,
                 device: int = 0, kernel: int = KERNEL_AUTO, chunk_rows: int = 0):
        self.lib = lib
        if isinstance(module, str):
            module = MODULES[module]
        self.module = module
        self.nrows, self.ncols = nrows, ncols
        self.ncp = ncols + 2
        self.slab = _Slab(slab_row0, slab_rows if slab_rows > 0 else nrows + 2)
        p = Params(module=module, nrows=nrows, ncols=ncols, drainrow=drainrow, draincol=draincol,
                   slab_row0=slab_row0, slab_rows=slab_rows, device=device, kernel=kernel,
                   chunk_rows=chunk_rows, missingvalue=missingvalue)
        h = C.c_void_p()
        lib.check(lib.dll.wdpm_create(C.byref(h), C.byref(p)))
        self._h = h

    # -- lifetime
    def close(self):
        if self._h:
            self.lib.dll.wdpm_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def shape(self):
        return (self.slab.rows, self.ncp)

    @staticmethod
    def _arr(a, shape):
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a.shape != tuple(shape):
            raise ValueError(f"expected array of shape {tuple(shape)}, got {a.shape}")
        return a

    # -- data movement
    def upload(self, bigdem, bigwater):
        d, w = self._arr(bigdem, self.shape), self._arr(bigwater, self.shape)
        self.lib.check(self.lib.dll.wdpm_upload(self._h, d.ctypes.data, w.ctypes.data))

    def upload_water(self, bigwater):
        w = self._arr(bigwater, self.shape)
        self.lib.check(self.lib.dll.wdpm_upload_water(self._h, w.ctypes.data))

    def download_water(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=np.float64)
        self.lib.check(self.lib.dll.wdpm_download_water(self._h, out.ctypes.data))
        return out

    def download_rows(self, row: int, nrows: int) -> np.ndarray:
        out = np.empty((nrows, self.ncp), dtype=np.float64)
        self.lib.check(self.lib.dll.wdpm_download_rows(self._h, row, nrows, out.ctypes.data))
        return out

    def upload_rows(self, row: int, src):
        src = np.ascontiguousarray(src, dtype=np.float64)
        assert src.ndim == 2 and src.shape[1] == self.ncp
        self.lib.check(self.lib.dll.wdpm_upload_rows(self._h, row, src.shape[0], src.ctypes.data))

    @property
    def totaldrain(self) -> float:
        v = C.c_double()
        self.lib.check(self.lib.dll.wdpm_get_totaldrain(self._h, C.byref(v)))
        return v.value

    @totaldrain.setter
    def totaldrain(self, v: float):
        self.lib.check(self.lib.dll.wdpm_set_totaldrain(self._h, float(v)))

    # -- block loop pieces
    def begin_block(self, thres: float):
        self.lib.check(self.lib.dll.wdpm_begin_block(self._h, thres))

    def iterate(self, n: int):
        self.lib.check(self.lib.dll.wdpm_iterate(self._h, n))

    def iterate_overlapped(self, n: int, top_rows: int, bottom_rows: int):
        self.lib.check(self.lib.dll.wdpm_iterate_overlapped(self._h, n, top_rows, bottom_rows))

    def single_pass(self, oi: int, oj: int):
        self.lib.check(self.lib.dll.wdpm_pass(self._h, oi, oj))

    def drain_outlet(self):
        self.lib.check(self.lib.dll.wdpm_drain_outlet(self._h))

    def expect_max_diff(self, row_lo: int = 0, row_hi: int | None = None):
        self.lib.check(self.lib.dll.wdpm_expect_max_diff(self._h, row_lo, self.slab.rows if row_hi is None else row_hi))

    def max_diff(self, row_lo: int = 0, row_hi: int | None = None) -> float:
        v = C.c_double()
        hi = self.slab.rows if row_hi is None else row_hi
        self.lib.check(self.lib.dll.wdpm_max_diff(self._h, row_lo, hi, C.byref(v)))
        return v.value

    def drain_stats_diff(self) -> float:
        """|totaldrain - olddrain| only (no raster download)"""
        a = C.c_double()
        self.lib.check(self.lib.dll.wdpm_drain_stats(self._h, C.byref(a), None))
        return a.value

    def drain_stats(self):
        a, b = C.c_double(), C.c_double()
        self.lib.check(self.lib.dll.wdpm_drain_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def volume_partial(self, row_lo: int, row_hi: int, start: float = 0.0) -> float:
        v = C.c_double()
        self.lib.check(self.lib.dll.wdpm_volume_partial(self._h, row_lo, row_hi, start, C.byref(v)))
        return v.value

    def run_block(self, n_iter: int, thres: float) -> float:
        v = C.c_double()
        self.lib.check(self.lib.dll.wdpm_run_block(self._h, n_iter, thres, C.byref(v)))
        return v.value

    # -- plumbing
    def water_ptr(self) -> int:
        p = C.c_void_p()
        self.lib.check(self.lib.dll.wdpm_water_ptr(self._h, C.byref(p)))
        return p.value

    def dem_ptr(self) -> int:
        p = C.c_void_p()
        self.lib.check(self.lib.dll.wdpm_dem_ptr(self._h, C.byref(p)))
        return p.value

    def set_stream(self, stream_handle: int):
        self.lib.check(self.lib.dll.wdpm_set_stream(self._h, C.c_void_p(stream_handle)))

    def get_option(self, key: int) -> int:
        v = C.c_int64()
        self.lib.check(self.lib.dll.wdpm_get_option(self._h, key, C.byref(v)))
        return v.value

    def set_option(self, key: int, value: int):
        self.lib.check(self.lib.dll.wdpm_set_option(self._h, key, value))

    def synchronize(self):
        self.lib.check(self.lib.dll.wdpm_synchronize(self._h))

    def timing_reset(self):
        self.lib.check(self.lib.dll.wdpm_timing_reset(self._h))

    def timing(self):
        n, ms = C.c_int64(), C.c_double()
        self.lib.check(self.lib.dll.wdpm_timing_get(self._h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def timing_steady(self):
        """(launches, ms) of the launches between the first and the last of each iterate call"""
        n, ms = C.c_int64(), C.c_double()
        self.lib.check(self.lib.dll.wdpm_timing_get_steady(self._h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def balance_info(self):
        """(rebalances so far, [eight per-XCD chunk-height weights, the factor on a strip's last chunk])"""
        n, w = C.c_int32(), (C.c_double * 9)()
        self.lib.check(self.lib.dll.wdpm_balance_info(self._h, C.byref(n), w))
        return n.value, [float(v) for v in w]

    def device_info(self):
        """(HIP ordinal as this process sees it, PCI bus id of the GPU the context lives on) - (-1, "host") on the CPU restatement"""
        o, b = C.c_int32(), C.create_string_buffer(32)
        self.lib.check(self.lib.dll.wdpm_device_info(self._h, C.byref(o), b, 32))
        return o.value, b.value.decode()

    def timing_exchange(self):
        """(halo refreshes into this context, ms on its stream from queueing a transfer to its rows' arrival) since the last reset"""
        n, ms = C.c_int64(), C.c_double()
        self.lib.check(self.lib.dll.wdpm_timing_get_exchange(self._h, C.byref(n), C.byref(ms)))
        return n.value, ms.value


def load(path: str) -> Lib:
    return Lib(path)


_hip: Lib | None = None


def load_hip() -> Lib:
    """The product library.  Raises (never falls back) when it has not been built."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise WdpmError(
                f"HIP library {HIP_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or `make -C wdpm_amd/csrc`). There is no CPU fallback.")
        _hip = Lib(HIP_LIB_PATH)
    return _hip
