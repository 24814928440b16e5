"""Row-block domain decomposition of the WDPM block loop across ranks (one rank = one GPU).

Host-side driver above the C ABI (include/wdpm.h), mirroring the reference's block loop
(src/WDPMCL.c:1049-1377) for a raster split into contiguous row slabs.  New work relative to the
reference, which is single-device.

Exactness.  Rows outside a slab act as NODATA, so the rows next to a slab edge go wrong and the
error creeps inward at the rate at which a cell can depend on other cells.  Inside a 3x3 block the
centre depends on all nine cells and every neighbour on the centre and on the neighbours visited
before it; chaining this over the three column alignments of a row alignment makes every row of
a block depend on all three.  The block grid moves down one row per row alignment, so per
iteration an error at the lower slab edge climbs 4 rows in the first iteration and 6 in each
further one, and an error at the upper edge descends 2 rows, then 3 per iteration: k iterations
need 3k-1 halo rows above and 6k-2 below when the slab boundary L satisfies L % 3 == 2
(halo_depth(); tests/test_rowblock.py re-derives both numbers with a cell-level dependency
("taint") simulation of the pass order, and shows on rasters that the rule is sufficient).  A rank
that owns rows [L, H] and holds that halo runs k iterations with no communication and its owned
rows stay bit-identical to the single-device result; the halos are then refreshed from the
neighbours' owned rows (one send/recv pair per neighbour every k iterations, over RCCL/xGMI when
the tensors live on GPUs).  The slab's first row L-(3k-1) is a multiple of 3, which keeps the
colour alignment of every slab equal to the whole raster's.
tests/test_rowblock.py checks all of this bit-for-bit on CPU ranks (gloo, world size 2 and 3).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .capi import Context, Lib


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


def partition(nrows: int, nranks: int, k: int) -> list[Slab]:
    """Split the (nrows+2)-row padded raster into nranks row slabs with halos for k iterations."""
    P = nrows + 2
    up, down = halo_depth(k)
    bounds = [0]
    for g in range(1, nranks):
        b = (P * g) // nranks
        b -= (b - 2) % 3            # boundaries must be = 2 (mod 3)
        bounds.append(b)
    bounds.append(P)
    slabs = []
    for g in range(nranks):
        lo, hi = bounds[g], bounds[g + 1] - 1
        if nranks > 1 and hi - lo + 1 < max(up, down):
            raise ValueError(f"slab of rank {g} ({hi - lo + 1} rows) is smaller than the halo depth; "
                             f"use fewer ranks or a smaller exchange interval")
        r0 = max(lo - up, 0) if g > 0 else 0
        r1 = min(hi + down, P - 1) if g < nranks - 1 else P - 1
        assert r0 % 3 == 0
        slabs.append(Slab(g, nranks, lo, hi, r0, r1 - r0 + 1, lo - r0, r1 - hi))
    return slabs


class HostTransport:
    """Halo rows travel through host numpy buffers and torch.distributed point-to-point ops (gloo).
    Works with any backend library; used by the CPU tests and as a fallback."""

    def __init__(self, dist, group=None):
        self.dist, self.group = dist, group

    def exchange(self, ctx: Context, sends, recvs):
        import torch
        ops, bufs = [], []
        for peer, row, n in sends:
            t = torch.from_numpy(ctx.download_rows(row, n))
            ops.append(self.dist.P2POp(self.dist.isend, t, peer, self.group))
            bufs.append(t)
        landing = []
        for peer, row, n in recvs:
            t = torch.empty((n, ctx.ncp), dtype=torch.float64)
            ops.append(self.dist.P2POp(self.dist.irecv, t, peer, self.group))
            landing.append((row, t))
        for w in self.dist.batch_isend_irecv(ops):
            w.wait()
        for row, t in landing:
            ctx.upload_rows(row, t.numpy())


class _DeviceMemory:
    """zero-copy view of library-owned HBM for torch (via __cuda_array_interface__)"""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def device_view(torch, ptr: int, rows: int, ncp: int, device):
    """torch tensor (rows x ncp, fp64) aliasing library-owned device memory at `ptr`"""
    return torch.as_tensor(_DeviceMemory(ptr, rows * ncp), device=device).view(rows, ncp)


class DeviceTransport:
    """Halo rows go GPU-to-GPU: torch tensors alias the library's water raster in HBM and are handed
    to torch.distributed (backend nccl = RCCL, over xGMI inside a node) as one batched
    send/recv group per exchange.  The context runs on torch's current stream so kernels and
    transfers are ordered on the device without host synchronisation."""

    def __init__(self, dist, device):
        import torch
        self.dist, self.torch, self.device = dist, torch, device
        self._views = {}

    def _view(self, ctx: Context):
        ptr = ctx.water_ptr()
        v = self._views.get(ptr)
        if v is None:
            v = device_view(self.torch, ptr, ctx.slab.rows, ctx.ncp, self.device)
            self._views[ptr] = v
        return v

    def exchange(self, ctx: Context, sends, recvs):
        w = self._view(ctx)
        ops = [self.dist.P2POp(self.dist.isend, w[row:row + n], peer) for peer, row, n in sends]
        ops += [self.dist.P2POp(self.dist.irecv, w[row:row + n], peer) for peer, row, n in recvs]
        for work in self.dist.batch_isend_irecv(ops):
            work.wait()


class RowBlockSolver:
    """The WDPM block loop on one rank's slab.  With nranks == 1 it is the plain single-GPU loop."""

    def __init__(self, lib: Lib, module, nrows: int, ncols: int, missingvalue: float, rank: int = 0,
                 nranks: int = 1, exchange_every: int = 4, transport=None, dist=None, drainrow: int = 0,
                 draincol: int = 0, fallback_transport=None, overlap: bool = True, **ctx_kw):
        self.rank, self.nranks, self.k = rank, nranks, max(1, exchange_every)
        self.slabs = partition(nrows, nranks, self.k)
        self.slab = self.slabs[rank]
        self.transport, self.dist, self.fallback_transport = transport, dist, fallback_transport
        if nranks > 1 and (transport is None or dist is None):
            raise ValueError("multi-rank solver needs a transport and torch.distributed")
        s = self.slab
        self.ctx = lib.context(module=module, nrows=nrows, ncols=ncols, missingvalue=missingvalue,
                               drainrow=drainrow, draincol=draincol, slab_row0=s.row0,
                               slab_rows=s.rows if nranks > 1 else 0, **ctx_kw)
        self.module = self.ctx.module
        self.overlap = overlap and nranks > 1
        self._since_exchange = 0
        # drain module: the rank whose OWNED rows hold the outlet has the raster's totaldrain
        self.drain_owner = next((sl.rank for sl in self.slabs if sl.own_lo <= drainrow <= sl.own_hi), 0)

    def close(self):
        self.ctx.close()

    # -- data
    def upload_global(self, bigdem: np.ndarray, bigwater: np.ndarray):
        s = self.slab
        self.ctx.upload(bigdem[s.row0:s.row0 + s.rows], bigwater[s.row0:s.row0 + s.rows])
        self._since_exchange = 0
        self.agree_on_options()

    def agree_on_options(self):
        """A -0.0 depth anywhere in the raster makes every rank use the sign-preserving stencil
        variant (halo rows may be written straight into device memory by the transport, past the
        library's own upload scan)."""
        if self.nranks > 1:
            import torch
            from .capi import OPT_SIGNED_ZERO_SAFE
            t = torch.tensor([self.ctx.get_option(OPT_SIGNED_ZERO_SAFE)], dtype=torch.int64)
            if self.dist.get_backend() == "nccl":
                t = t.cuda()
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            self.ctx.set_option(OPT_SIGNED_ZERO_SAFE, int(t.item()))

    def owned_water(self) -> np.ndarray:
        s = self.slab
        return self.ctx.download_rows(s.lo, s.hi - s.lo)

    # -- halo refresh
    def _plan(self):
        s, sends, recvs = self.slab, [], []
        if s.rank > 0:
            above = self.slabs[s.rank - 1]
            sends.append((s.rank - 1, s.lo, above.down))          # my first rows are its lower halo
            recvs.append((s.rank - 1, 0, s.up))
        if s.rank < s.nranks - 1:
            below = self.slabs[s.rank + 1]
            sends.append((s.rank + 1, s.hi - below.up, below.up))  # my last rows are its upper halo
            recvs.append((s.rank + 1, s.hi, s.down))
        return sends, recvs

    def exchange(self):
        if self.nranks > 1:
            sends, recvs = self._plan()
            try:
                self.transport.exchange(self.ctx, sends, recvs)
            except Exception as e:  # noqa: BLE001 - e.g. a GPU-direct transport the platform refuses
                if self.fallback_transport is None:
                    raise
                import sys
                print(f"[wdpm rank {self.rank}] halo transport {type(self.transport).__name__} failed ({e!r}); "
                      f"switching to {type(self.fallback_transport).__name__}", file=sys.stderr, flush=True)
                self.transport, self.fallback_transport = self.fallback_transport, None
                self.transport.exchange(self.ctx, sends, recvs)
        self._since_exchange = 0

    # -- the block loop pieces (WDPMCL.c:1055-1125, 1239-1254)
    def begin_block(self, thres: float):
        self.ctx.begin_block(thres)

    def iterate(self, n_iter: int):
        done = 0
        while done < n_iter:
            room = self.k - self._since_exchange
            if room <= 0:
                self.exchange()
                room = self.k
            step = min(room, n_iter - done)
            if self.overlap and step == room:
                # this step ends a group and an exchange follows: produce the rows the neighbours
                # need first, so that the send/recv overlaps the rest of the last iteration
                s = self.slab
                top = s.lo + self.slabs[s.rank - 1].down if s.rank > 0 else 0
                bottom = s.rows - (s.hi - self.slabs[s.rank + 1].up) if s.rank < s.nranks - 1 else 0
                self.ctx.iterate_overlapped(step, top, bottom)
            else:
                self.ctx.iterate(step)
            done += step
            self._since_exchange += step

    def max_diff(self) -> float:
        if self._since_exchange and self.nranks > 1:
            self.exchange()
        s = self.slab
        m = self.ctx.max_diff(s.lo, s.hi)
        if self.nranks > 1:
            import torch
            t = torch.tensor([m], dtype=torch.float64)
            if self.dist.get_backend() == "nccl":
                t = t.cuda()
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            m = float(t.item())
        return m

    def run_block(self, n_iter: int, thres: float) -> float:
        self.begin_block(thres)
        self.iterate(n_iter)
        return self.max_diff()

    # -- drain bookkeeping (WDPMCL.c:1257-1268) across ranks
    def _bcast(self, values, src):
        import torch
        t = torch.tensor(values, dtype=torch.float64)
        if self.dist.get_backend() == "nccl":
            t = t.cuda()
        self.dist.broadcast(t, src=src)
        return [float(v) for v in t.cpu()]

    def set_totaldrain(self, v: float):
        self.ctx.totaldrain = v

    def totaldrain(self) -> float:
        td = self.ctx.totaldrain
        return self._bcast([td], self.drain_owner)[0] if self.nranks > 1 else td

    def drain_stats(self):
        """(|totaldrain - olddrain|, sum of water over valid cells in the reference's row-major
        order).  The sum is chained rank to rank so that its rounding equals the single-raster sum."""
        if self.nranks == 1:
            return self.ctx.drain_stats()
        import torch
        s, nccl = self.slab, self.dist.get_backend() == "nccl"
        run = torch.zeros(1, dtype=torch.float64, device="cuda" if nccl else "cpu")
        if s.rank > 0:
            self.dist.recv(run, src=s.rank - 1)
        part = self.ctx.volume_partial(s.lo, s.hi, float(run.item()))
        if s.rank < s.nranks - 1:
            self.dist.send(torch.tensor([part], dtype=torch.float64, device=run.device), dst=s.rank + 1)
        final_sum = self._bcast([part], s.nranks - 1)[0]
        diffdrain = self._bcast([self.ctx.drain_stats_diff()], self.drain_owner)[0]
        return diffdrain, final_sum
