"""The library's own RCCL halo path (wdpm_amd/csrc/wdpm_rccl.hip: wdpm_comm_*) against real RCCL on the one
GPU a test box has.  RCCL refuses two ranks on one device, so the communicator has ONE rank and the halo
rows travel as a send to self: the same entry points, the same grouped ncclSend/ncclRecv on the context's
stream between the library's kernels, real RCCL kernels moving the rows.  More ranks need more GPUs: the
driver's SCALE run; the multi-rank logic itself is covered on CPU ranks and over host-staged halos."""
import ctypes as C

import numpy as np
import pytest

from helpers import bits_equal
from wdpm_amd.capi import COMM_ID_BYTES, HaloOp

pytestmark = pytest.mark.gpu

R, Cc, MISS = 64, 300, -99999.0


def inputs():
    rng = np.random.default_rng(0)
    bd = np.full((R + 2, Cc + 2), MISS)
    bd[1:-1, 1:-1] = np.round(500 + rng.random((R, Cc)), 3)
    return bd, rng.random((R + 2, Cc + 2))


def expected(hip, bd, bw):
    """3 iterations, rows 10..14 copied onto rows 40..44, 2 more iterations - through the host"""
    with hip.context(module="add", nrows=R, ncols=Cc, missingvalue=MISS, device=0) as ref:
        ref.upload(bd, bw)
        ref.iterate(3)
        w = ref.download_water()
        w[40:45] = w[10:15]
        ref.upload_water(w)
        ref.iterate(2)
        return ref.download_water()


@pytest.mark.parametrize("init", ["rank", "all"])
def test_rows_move_through_rccl_between_kernels(hip, init):
    assert hip.dll.wdpm_comm_available() == 1, "RCCL could not be bound on a GPU box"
    bd, bw = inputs()
    want = expected(hip, bd, bw)
    with hip.context(module="add", nrows=R, ncols=Cc, missingvalue=MISS, device=0) as c:
        if init == "rank":
            ident = C.create_string_buffer(COMM_ID_BYTES)
            hip.check(hip.dll.wdpm_comm_unique_id(ident))
            hip.check(hip.dll.wdpm_comm_init_rank(c._h, 1, 0, ident))
        else:
            arr = (C.c_void_p * 1)(c._h)
            hip.check(hip.dll.wdpm_comm_init_all(arr, 1))
        n, r = C.c_int32(-1), C.c_int32(-1)
        hip.check(hip.dll.wdpm_comm_size(c._h, C.byref(n), C.byref(r)))
        assert (n.value, r.value) == (1, 0)
        c.upload(bd, bw)
        c.iterate(3)                                      # kernels queued on the stream right before the transfer ...
        send, recv = (HaloOp * 1)(HaloOp(0, 10, 5)), (HaloOp * 1)(HaloOp(0, 40, 5))
        hip.check(hip.dll.wdpm_comm_exchange(c._h, 1, send, 1, recv))   # ... rows by RCCL, no host synchronisation ...
        c.iterate(2)                                      # ... and kernels right behind it
        assert bits_equal(c.download_water(), want)
        mine = (C.c_double * 3)(1.5, -2.0, 7.0)
        out = (C.c_double * 3)()
        hip.check(hip.dll.wdpm_comm_allgather(c._h, mine, 3, out))
        assert list(out) == [1.5, -2.0, 7.0]


def test_rccl_shares_the_process_with_torch(hip):
    """inside a PyTorch process the library must bind the RCCL and HIP runtime PyTorch mapped (one of each per
    process): a torch collective and the library's own communicator side by side"""
    import os
    import socket
    import torch
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        t = torch.tensor([3.0], device="cuda")
        dist.all_reduce(t)
        assert t.item() == 3.0
        from wdpm_amd.rowblock import RowBlockSolver
        bd, bw = inputs()
        s1 = RowBlockSolver(hip, "add", R, Cc, MISS, rank=0, nranks=1, dist=dist)
        s1.upload(bd, bw)
        a = s1.run_block(5, 0.0)
        with hip.context(module="add", nrows=R, ncols=Cc, missingvalue=MISS, device=0) as ref:
            ref.upload(bd, bw)
            assert ref.run_block(5, 0.0) == a
        s1.close()
    finally:
        dist.destroy_process_group()
