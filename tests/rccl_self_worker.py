"""Worker for tests/test_rccl_transport.py: a one-rank RCCL process group on the GPU.  RCCL refuses two ranks on
one device, so the GPU-direct halo transport (wdpm_amd.rowblock.DeviceTransport: torch tensors aliasing the
library's water raster, one batched isend/irecv group) is exercised as a send to self: same code path, same
stream hand-over between the library's kernels and RCCL, real RCCL kernels moving the rows."""
import os, sys
import numpy as np
sys.path.insert(0, ".")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29533")
import torch, torch.distributed as dist
import wdpm_amd
from wdpm_amd.rowblock import DeviceTransport
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
lib = wdpm_amd.load_hip()
R, C = 64, 300
rng = np.random.default_rng(0)
bd = np.full((R + 2, C + 2), -99999.0); bd[1:-1, 1:-1] = np.round(500 + rng.random((R, C)), 3)
bw = rng.random((R + 2, C + 2))
with lib.context(module="add", nrows=R, ncols=C, missingvalue=-99999.0, device=0) as c:
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    c.upload(bd, bw)
    class S: rows = R + 2
    c.slab = S()
    t = torch.tensor([3.0], device="cuda"); dist.all_reduce(t); print("all_reduce ok", t.item())
    tr = DeviceTransport(dist, torch.device("cuda", 0))
    c.iterate(3)                                    # kernels queued on the stream right before the transfer ...
    before = c.download_water()
    c.upload_water(before)
    c.iterate(2)
    want = None
    with lib.context(module="add", nrows=R, ncols=C, missingvalue=-99999.0, device=0) as ref:
        ref.upload(bd, before)
        ref.iterate(2)
        want = ref.download_water()
        want[40:45] = want[10:15]
        ref.upload_water(want)
        ref.iterate(2)
        want = ref.download_water()
    tr.exchange(c, sends=[(0, 10, 5)], recvs=[(0, 40, 5)])      # ... rows 10..14 -> rows 40..44 by RCCL, no host sync
    c.iterate(2)                                                # ... and kernels right behind it
    w = c.download_water()
    ok = np.array_equal(w.view(np.uint64), want.view(np.uint64))
    print("RCCL_SELF_TRANSPORT", "OK" if ok else "MISMATCH")
dist.destroy_process_group()
