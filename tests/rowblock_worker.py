"""Worker for tests/test_rowblock.py: one rank of a row-block-decomposed WDPM run on CPU ranks
(gloo) with the oracle as the compute back-end behind the same C ABI the GPUs use."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(rank, world, port, libpath, case, outdir):
    import torch.distributed as dist
    import wdpm_amd
    from wdpm_amd.rowblock import HostTransport, RowBlockSolver
    from helpers import find_drain, pad, random_case

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    lib = wdpm_amd.load(libpath)
    dem, water, miss = random_case(case["seed"], case["R"], case["C"])
    if case.get("ponds"):                         # mostly dry: dry tiles, water arriving through refreshed halos
        water[:, :] = 0.0
        for r0, r1, c0, c1, d in case["ponds"]:
            water[r0:r1, c0:c1] = d
        water[dem <= miss] = 0.0
    bd, bw = pad(dem, water, miss)
    kw = dict(case.get("ctx_kw", {}))
    if case.get("devices"):                       # tests/test_multi_gpu.py: every rank process on a GPU of its own
        kw["device"] = case["devices"][rank]
    drain = case["module"] == "drain"
    if drain:
        dr, dc = find_drain(bd)
        kw.update(drainrow=dr, draincol=dc)
    transport = HostTransport(dist)
    if case.get("failing_transport"):
        class Refusing(HostTransport):       # the caller's transport raises: the error must surface, not hang
            def _exchange_cb(self, *a):
                self.error = RuntimeError("simulated: transport refused")
                return 1
        transport = Refusing(dist)
    if case.get("halo") == "rccl":
        # halos through the library's own RCCL path, one process per rank: wdpm_comm_unique_id on rank 0 -> broadcast (gloo) ->
        # ncclCommInitRank -> grouped ncclSend/ncclRecv on the context's stream, ncclAllGather for the block scalars
        s = RowBlockSolver(lib, case["module"], case["R"], case["C"], miss, rank=rank, nranks=world,
                           exchange_every=case["k"], halo="rccl", dist=dist, **kw)
        assert s.rccl_ranks() == world
    else:
        s = RowBlockSolver(lib, case["module"], case["R"], case["C"], miss, rank=rank, nranks=world,
                           exchange_every=case["k"], transport=transport, dist=dist, **kw)
    s.upload_global(bd, bw)
    if drain:
        s.set_totaldrain(max(bw[dr, dc], 0.0))
    mds, stats = [], []
    if case.get("failing_transport"):
        try:
            s.run_block(case["blocks"][0], case["thres"])
        except RuntimeError as e:
            assert "simulated" in str(e)
            np.savez(os.path.join(outdir, f"rank{rank}.npz"), refused=1)
            s.close()
            dist.destroy_process_group()
            return
        raise SystemExit("the refused transport went unnoticed")
    for n in case["blocks"]:
        mds.append(s.run_block(n, case["thres"]))
        if drain:
            stats.append(list(s.drain_stats()) + [s.totaldrain()])
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), own=s.owned_water(), lo=s.slab.own_lo, hi=s.slab.own_hi,
             mds=np.array(mds), stats=np.array(stats))
    s.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import json
    rank, world, port, libpath, case, outdir = sys.argv[1:7]
    run(int(rank), int(world), int(port), libpath, json.loads(case), outdir)
