"""Row-block domain decomposition of the WDPM block loop across ranks (one rank = one GPU).

Python mirror of the C driver ``wdpm_amd/csrc/wdpm_rowblock.c`` (``wdpm_rank_*`` / ``wdpm_group_*`` of
include/wdpm.h), which holds ALL of the logic — the partition with its halo rule, the exchange
schedule, the overlapped last iteration, the block loop of src/WDPMCL.c:1049-1377 across ranks, the
drain module's chained volume sum.  The same C code drives the ranks of one process in ``WDPMCL``
(``WDPM_GPUS=N``, one host thread per GPU) and the one-process-per-GPU ranks ``bench.py`` starts.
What is left here is what only the caller can provide:

* how the ranks find each other — the 128-byte RCCL id made by rank 0 (``wdpm_comm_unique_id``) is
  handed to every rank through ``torch.distributed`` (:func:`rccl_id`); from then on halo rows move
  GPU to GPU by ``ncclSend``/``ncclRecv`` issued by the library on its own stream;
* a host-staged transport (:class:`HostTransport`: the library stages rows in host memory and calls
  back; the bytes travel by gloo) for CPU tests against the oracle back-end and for several ranks
  sharing the one GPU of a test box, which RCCL refuses.

Exactness (derivation in the C file; ``tests/test_rowblock.py`` re-derives the numbers with a
cell-level dependency simulation): k iterations without communication need 3k-1 halo rows above and
6k-2 below a slab whose boundaries are = 2 (mod 3); owned rows then stay bit-identical to the
single-device result.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from .capi import (HALO_AUTO, HALO_HOST, HALO_RCCL, MODULES, Context, HostTransportStruct, Lib, Params, SlabStruct,
                   EXCHANGE_FN, ALLGATHER_FN, COMM_ID_BYTES)


@dataclass(frozen=True)
class Slab:
    rank: int
    nranks: int
    own_lo: int      # first owned padded row (global)
    own_hi: int      # last owned padded row (global, inclusive)
    row0: int        # first padded row held (global), multiple of 3
    rows: int        # padded rows held
    up: int          # halo rows above own_lo actually held
    down: int        # halo rows below own_hi actually held

    @property
    def lo(self):    # slab-local index of own_lo
        return self.own_lo - self.row0

    @property
    def hi(self):    # slab-local index one past own_hi
        return self.own_hi - self.row0 + 1


def halo_depth(k: int):
    """(rows above, rows below) needed for k communication-free iterations (worst case over the
    data; tests/test_rowblock.py re-derives both by a dependency simulation of the pass order)."""
    return 3 * k - 1, 6 * k - 2


def partition(lib: Lib, nrows: int, nranks: int, k: int, module="add", drainrow: int = -1) -> list[Slab]:
    """wdpm_partition: the (nrows+2)-row padded raster as nranks row slabs with halos for k iterations."""
    out = (SlabStruct * nranks)()
    m = MODULES[module] if isinstance(module, str) else module
    if lib.dll.wdpm_partition(nrows, nranks, k, m, drainrow, out) != 0:
        raise ValueError(f"a slab of {nrows + 2} rows / {nranks} ranks is smaller than the halo depth of {k} "
                         f"iterations; use fewer ranks or a smaller exchange interval")
    return [Slab(g, nranks, s.own_lo, s.own_hi, s.row0, s.rows, s.up, s.down) for g, s in enumerate(out)]


def rccl_id(lib: Lib, dist, rank: int) -> bytes:
    """the id every rank needs for ncclCommInitRank: made on rank 0, broadcast by torch.distributed"""
    box = [None]
    if rank == 0:
        buf = C.create_string_buffer(COMM_ID_BYTES)
        lib.check(lib.dll.wdpm_comm_unique_id(buf))
        box[0] = buf.raw
    dist.broadcast_object_list(box, src=0)
    return box[0]


class HostTransport:
    """The caller's half of WDPM_HALO_HOST: the library hands over host buffers, torch.distributed
    (gloo) moves them.  Used by the CPU tests and by multi-rank rehearsals on one GPU."""

    def __init__(self, dist, group=None):
        self.dist, self.group = dist, group
        self.error = None
        self._exchange = EXCHANGE_FN(self._exchange_cb)
        self._allgather = ALLGATHER_FN(self._allgather_cb)
        self.struct = HostTransportStruct(None, self._exchange, self._allgather)

    def _exchange_cb(self, user, n_ops, is_send, peer, buf, count):
        try:
            import torch
            ops = []
            for i in range(n_ops):
                n = int(count[i])
                if n == 0:
                    continue
                a = np.ctypeslib.as_array(C.cast(buf[i], C.POINTER(C.c_double)), shape=(n,))
                t = torch.from_numpy(a)
                ops.append(self.dist.P2POp(self.dist.isend if is_send[i] else self.dist.irecv, t, int(peer[i]), self.group))
            for w in self.dist.batch_isend_irecv(ops) if ops else []:
                w.wait()
            return 0
        except Exception as e:  # noqa: BLE001 - must not propagate through the C frame
            self.error = e
            return 1

    def _allgather_cb(self, user, mine, n, out):
        try:
            import torch
            world = self.dist.get_world_size(self.group)
            m = torch.from_numpy(np.ctypeslib.as_array(mine, shape=(n,)).copy())
            parts = [torch.empty(n, dtype=torch.float64) for _ in range(world)]
            self.dist.all_gather(parts, m, group=self.group)
            np.ctypeslib.as_array(out, shape=(world * n,))[:] = torch.cat(parts).numpy()
            return 0
        except Exception as e:  # noqa: BLE001
            self.error = e
            return 1


class _RankContext(Context):
    """the slab context a wdpm_rank owns (not destroyed from here)"""

    def __init__(self, lib, handle, module, nrows, ncols, slab):
        self.lib, self._h, self.module = lib, handle, module
        self.nrows, self.ncols, self.ncp = nrows, ncols, ncols + 2
        self.slab = slab

    def close(self):
        self._h = None


class RowBlockSolver:
    """The WDPM block loop on one rank's slab (wdpm_rank_*).  With nranks == 1 it is the plain
    single-GPU loop.  halo: "rccl" (needs dist, backend nccl or gloo, to hand the id round), "host"
    (needs a HostTransport) or None = rccl when no transport is given."""

    def __init__(self, lib: Lib, module, nrows: int, ncols: int, missingvalue: float, rank: int = 0,
                 nranks: int = 1, exchange_every: int = 8, transport: HostTransport | None = None, dist=None,
                 drainrow: int = 0, draincol: int = 0, device: int = 0, kernel: int = 0, chunk_rows: int = 0,
                 halo: str | None = None):
        self.lib, self.rank, self.nranks, self.dist = lib, rank, nranks, dist
        m = MODULES[module] if isinstance(module, str) else module
        p = Params(module=m, nrows=nrows, ncols=ncols, drainrow=drainrow, draincol=draincol, slab_row0=0,
                   slab_rows=0, device=device, kernel=kernel, chunk_rows=chunk_rows, missingvalue=missingvalue)
        self.transport = transport
        kind, idbuf, host = HALO_AUTO, None, None
        if nranks > 1:
            if halo is None:
                halo = "host" if transport is not None else "rccl"
            if halo == "rccl":
                if dist is None:
                    raise ValueError("RCCL halos need torch.distributed to hand the communicator id to every rank")
                kind, idbuf = HALO_RCCL, rccl_id(lib, dist, rank)
            elif halo == "host":
                if transport is None:
                    raise ValueError("host-staged halos need a HostTransport")
                kind, host = HALO_HOST, C.byref(transport.struct)
            else:
                raise ValueError(f"unknown halo transport {halo!r}")
        h = C.c_void_p()
        self._check(lib.dll.wdpm_rank_create(C.byref(h), C.byref(p), rank, nranks, exchange_every, kind, idbuf, host))
        self._h = h
        self.slabs = [self._slab(g) for g in range(nranks)]
        self.slab = self.slabs[rank]
        halo_kind, k, owner = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(lib.dll.wdpm_rank_info(h, C.byref(halo_kind), C.byref(k), C.byref(owner)))
        self.halo_kind, self.k, self.drain_owner = halo_kind.value, k.value, owner.value
        self.ctx = _RankContext(lib, C.c_void_p(lib.dll.wdpm_rank_ctx(h)), m, nrows, ncols,
                                self.slab if nranks > 1 else Slab(0, 1, 0, nrows + 1, 0, nrows + 2, 0, 0))
        self.module = m

    def _check(self, rc):
        if rc != 0:
            if self.transport is not None and self.transport.error is not None:
                err, self.transport.error = self.transport.error, None
                raise err
            self.lib.check(rc)

    def _slab(self, g):
        s = SlabStruct()
        self.lib.check(self.lib.dll.wdpm_rank_slab(self._h, g, C.byref(s)))
        return Slab(g, self.nranks, s.own_lo, s.own_hi, s.row0, s.rows, s.up, s.down)

    def close(self):
        if self._h:
            self.ctx.close()
            self.lib.dll.wdpm_rank_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def abort_comm(self):
        """wdpm_comm_abort: end this rank's communicator now (queued transfers end, the peers' fail instead of waiting)"""
        if self._h:
            self.lib.dll.wdpm_comm_abort(self.ctx._h)

    def rccl_ranks(self) -> int | None:
        """communicator size as RCCL itself reports it (ncclCommCount), None without RCCL halos"""
        if self.halo_kind != HALO_RCCL:
            return None
        n = C.c_int32()
        self.lib.check(self.lib.dll.wdpm_comm_size(self.ctx._h, C.byref(n), None))
        return n.value

    # -- data
    def upload(self, slab_dem: np.ndarray, slab_water: np.ndarray):
        d, w = Context._arr(slab_dem, self.ctx.shape), Context._arr(slab_water, self.ctx.shape)
        self._check(self.lib.dll.wdpm_rank_upload(self._h, d.ctypes.data, w.ctypes.data))

    def upload_global(self, bigdem: np.ndarray, bigwater: np.ndarray):
        shape = (self.ctx.nrows + 2, self.ctx.ncp)
        d, w = Context._arr(bigdem, shape), Context._arr(bigwater, shape)
        self._check(self.lib.dll.wdpm_rank_upload_global(self._h, d.ctypes.data, w.ctypes.data))

    def owned_water(self) -> np.ndarray:
        s = self.slab
        out = np.empty((s.own_hi - s.own_lo + 1 if self.nranks > 1 else self.ctx.nrows + 2, self.ctx.ncp))
        self.lib.check(self.lib.dll.wdpm_rank_download_owned(self._h, out.ctypes.data))
        return out

    # -- the block loop pieces (WDPMCL.c:1055-1125, 1239-1254)
    def exchange(self):
        self._check(self.lib.dll.wdpm_rank_exchange(self._h))

    def begin_block(self, thres: float):
        self._check(self.lib.dll.wdpm_rank_begin_block(self._h, thres))

    def iterate(self, n_iter: int):
        self._check(self.lib.dll.wdpm_rank_iterate(self._h, n_iter))

    def max_diff(self) -> float:
        v = C.c_double()
        self._check(self.lib.dll.wdpm_rank_max_diff(self._h, C.byref(v)))
        return v.value

    def run_block(self, n_iter: int, thres: float) -> float:
        v = C.c_double()
        self._check(self.lib.dll.wdpm_rank_run_block(self._h, n_iter, thres, C.byref(v)))
        return v.value

    # -- drain bookkeeping (WDPMCL.c:1257-1268) across ranks
    def set_totaldrain(self, v: float):
        self._check(self.lib.dll.wdpm_rank_set_totaldrain(self._h, float(v)))

    def totaldrain(self) -> float:
        v = C.c_double()
        self._check(self.lib.dll.wdpm_rank_get_totaldrain(self._h, C.byref(v)))
        return v.value

    def drain_stats(self):
        """(|totaldrain - olddrain|, sum of water over valid cells in the reference's row-major order)"""
        a, b = C.c_double(), C.c_double()
        self._check(self.lib.dll.wdpm_rank_drain_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value


def spread_over_devices(ndev: int, nslabs: int | None = None, limit: int = 8) -> list[int]:
    """The device list of a row-block job on a box with `ndev` GPUs: one slab per GPU, at most `limit` of them
    (the reference picks one OpenCL device, WDPMCL.c:80-121,598-638; WDPMCL here takes WDPM_GPUS / WDPM_DEVICES) - or, if
    `nslabs` is given, that many slabs dealt in contiguous runs, so that neighbouring slabs share a device wherever
    slabs outnumber devices: 8 slabs on 2 GPUs -> [0, 0, 0, 0, 1, 1, 1, 1], 3 on 2 -> [0, 0, 1].  The multi-GPU tests
    (tests/test_multi_gpu.py) size themselves with this from torch.cuda.device_count()."""
    if ndev < 1:
        raise ValueError("no device")
    use = min(ndev, limit)
    if nslabs is None:
        return list(range(use))
    if nslabs < 1:
        raise ValueError("no slab")
    use = min(use, nslabs)
    return [i * use // nslabs for i in range(nslabs)]


class Group:
    """wdpm_group_*: the ranks of ONE process, one host thread per device — what the WDPMCL drop-in
    uses with WDPM_GPUS=N.  devices may repeat (several slabs on one GPU: peer-copy halos)."""

    def __init__(self, lib: Lib, module, nrows: int, ncols: int, missingvalue: float, devices, exchange_every: int = 8,
                 drainrow: int = 0, draincol: int = 0, kernel: int = 0, chunk_rows: int = 0):
        self.lib = lib
        m = MODULES[module] if isinstance(module, str) else module
        p = Params(module=m, nrows=nrows, ncols=ncols, drainrow=drainrow, draincol=draincol, slab_row0=0,
                   slab_rows=0, device=0, kernel=kernel, chunk_rows=chunk_rows, missingvalue=missingvalue)
        dev = (C.c_int32 * len(devices))(*devices)
        h = C.c_void_p()
        lib.check(lib.dll.wdpm_group_create(C.byref(h), C.byref(p), len(devices), dev, exchange_every))
        self._h, self.shape = h, (nrows + 2, ncols + 2)
        self.size = lib.dll.wdpm_group_size(h)
        self.halo_kind = lib.dll.wdpm_group_halo(h)

    def close(self):
        if self._h:
            self.lib.dll.wdpm_group_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def rank_ctx(self, i: int) -> C.c_void_p:
        return C.c_void_p(self.lib.dll.wdpm_rank_ctx(C.c_void_p(self.lib.dll.wdpm_group_rank(self._h, i))))

    def upload(self, bigdem, bigwater):
        d, w = Context._arr(bigdem, self.shape), Context._arr(bigwater, self.shape)
        self.lib.check(self.lib.dll.wdpm_group_upload(self._h, d.ctypes.data, w.ctypes.data))

    def download_water(self) -> np.ndarray:
        out = np.empty(self.shape)
        self.lib.check(self.lib.dll.wdpm_group_download_water(self._h, out.ctypes.data))
        return out

    def set_totaldrain(self, v: float):
        self.lib.check(self.lib.dll.wdpm_group_set_totaldrain(self._h, float(v)))

    def totaldrain(self) -> float:
        v = C.c_double()
        self.lib.check(self.lib.dll.wdpm_group_get_totaldrain(self._h, C.byref(v)))
        return v.value

    def run_block(self, n_iter: int, thres: float) -> float:
        v = C.c_double()
        self.lib.check(self.lib.dll.wdpm_group_run_block(self._h, n_iter, thres, C.byref(v)))
        return v.value

    def drain_stats(self):
        a, b = C.c_double(), C.c_double()
        self.lib.check(self.lib.dll.wdpm_group_drain_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    # -- set-up and final statistics next to the rasters
    def upload_unpadded(self, dem, water=None, op: int = 0, add: float = 0.0, rof: float = 0.0, sub: float = 0.0):
        from .capi import SetupStruct
        shape = (self.shape[0] - 2, self.shape[1] - 2)
        d = Context._arr(dem, shape)
        w = None if water is None else Context._arr(water, shape)
        su = SetupStruct(op, add, rof, sub)
        self.lib.check(self.lib.dll.wdpm_group_upload_unpadded(self._h, d.ctypes.data, None if w is None else w.ctypes.data,
                                                               C.byref(su)))

    def count_stats(self):
        """(valid cells, valid cells with water > 0.001, max over cells of (valid ? water : missing))"""
        a, b, m = C.c_int64(), C.c_int64(), C.c_double()
        self.lib.check(self.lib.dll.wdpm_group_count_stats(self._h, C.byref(a), C.byref(b), C.byref(m)))
        return a.value, b.value, m.value

    def find_drain(self):
        m, r, c = C.c_double(), C.c_int32(), C.c_int32()
        self.lib.check(self.lib.dll.wdpm_group_find_drain(self._h, C.byref(m), C.byref(r), C.byref(c)))
        return m.value, r.value, c.value

    def set_drain(self, drainrow: int, draincol: int) -> int:
        """0 = set; 2 = the outlet is too close to a slab boundary of this partition (create the group again)"""
        rc = self.lib.dll.wdpm_group_set_drain(self._h, drainrow, draincol)
        if rc not in (0, 2):
            self.lib.check(rc)
        return rc

    def get_cell(self, row: int, col: int):
        w, d = C.c_double(), C.c_double()
        self.lib.check(self.lib.dll.wdpm_group_get_cell(self._h, row, col, C.byref(w), C.byref(d)))
        return w.value, d.value

    def download_unpadded(self, mask_missing: bool = True) -> np.ndarray:
        out = np.empty((self.shape[0] - 2, self.shape[1] - 2))
        self.lib.check(self.lib.dll.wdpm_group_download_unpadded(self._h, int(mask_missing), out.ctypes.data))
        return out

    def enqueue_stats(self):
        """(seconds queueing launches, seconds in halo refreshes) summed over ranks, iterations of one rank"""
        s, e, n = C.c_double(), C.c_double(), C.c_int64()
        self.lib.check(self.lib.dll.wdpm_group_enqueue_stats(self._h, C.byref(s), C.byref(e), C.byref(n)))
        return s.value, e.value, n.value
