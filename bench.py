#!/usr/bin/env python3
"""bench.py — cell-updates/s of the WDPM Add module on a synthetic 16384 x 16384 DEM (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 16384]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one iteration (all 9 colour passes) over the whole raster.  The timed region is one
block of the reference's loop: threshold flush + snapshot, K iterations, max-|dw| reduction
(src/WDPMCL.c:1055-1125,1239-1254), with the rasters already resident in HBM.  N > 1: row-block
decomposition, one rank per GPU, halo refresh by RCCL send/recv (wdpm_amd/rowblock.py); total work
is fixed, so scaling is strong.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_CELL_UPDATE = 24.0   # fp64 dem read + water read + water write
HBM_PEAK_GBS = 8000.0               # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
THRES = 0.005 / 1000.0              # zero-depth threshold 0.005 mm
ADD_M = 0.1                         # add 100 mm, runoff fraction 1.0
MISSING = -99999.0


def build_slab_inputs(lib, n, slab):
    """padded dem / water rows [row0, row0+rows) of the config-4 workload"""
    dem = lib.synth_dem(n, n)                      # integer-seeded generator, seed = size
    ncp = n + 2
    bd = np.full((slab.rows, ncp), MISSING)
    lo, hi = max(slab.row0, 1), min(slab.row0 + slab.rows, n + 1)   # padded interior rows held
    bd[lo - slab.row0:hi - slab.row0, 1:-1] = dem[lo - 1:hi - 1]
    del dem
    bw = np.where(bd > MISSING, ADD_M, 0.0)
    return bd, bw


def measured_traffic(n, world, kernel, dem32):
    """HBM bytes per fused-kernel launch from the committed rocprofv3 PMC passes (separate
    --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this same command, gfx950 x2 FETCH correction,
    calibrated on kernels of known byte count: profiles/).  Only for the configuration they
    were taken on; otherwise null.  With the DEM streamed as 32-bit codes the kernel moves 20 B per
    cell-update, i.e. LESS than the 24 algorithmic bytes `achieved` is priced at."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if t["size"] == n and t["n_gpus"] == world and kernel in ("auto", "fused"):
            return t["dem32" if dem32 else "fp64_dem"]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def reference_baseline(n, iters):
    """The UNMODIFIED reference's serial functions (runoffs() on its own double** globals, driven in
    the loop order of WDPMCL.c:1097-1103) from oracle/_ref/libwdpm_ref.so, which oracle/Makefile
    builds from the sources where they lie under /root/reference and which travels to the GPU box as
    a binary.  None if that library is not there or does not load."""
    import ctypes as C
    import wdpm_amd
    so = os.path.join(ROOT, "oracle", "_ref", "libwdpm_ref.so")
    try:
        ref = C.CDLL(so)
        ref.ref_setup.argtypes = [C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_int]
        ref.ref_iterate.argtypes = [C.c_int, C.c_int]
    except (OSError, AttributeError):
        return None
    dem = wdpm_amd.load_hip().synth_dem(n, n)
    bd = np.full((n + 2, n + 2), MISSING)
    bd[1:-1, 1:-1] = dem
    bw = np.where(bd > MISSING, ADD_M, 0.0)
    ref.ref_setup(n, n, MISSING, bd.ctypes.data, bw.ctypes.data, 0.0, 0, 0)
    ref.ref_iterate(0, 1)
    t = time.perf_counter()
    ref.ref_iterate(0, iters)
    dt = time.perf_counter() - t
    return {"value": n * n * iters / dt, "unit": "cell-updates/s", "cores": 1, "kind": "reference",
            "sample": f"runoffs() of the unmodified src/WDPMCL.c (oracle/_ref/libwdpm_ref.so), synthetic {n}x{n} "
                      f"all-wet add 100 mm, {iters} iterations, {dt:.1f} s"}


def cpu_baseline(n=4096, iters=32):
    """The reference's own serial code if its prebuilt library is present (kind "reference"), else the
    CPU oracle (bit-equal port of it, kind "port"), timed on one host core on a bounded sample of the
    same workload.  A reported baseline, never the product path."""
    import subprocess
    import wdpm_amd
    ref = reference_baseline(n, 24)
    if ref is not None:
        return ref
    so = os.path.join(ROOT, "oracle", "_build", "libwdpm_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], stdout=subprocess.DEVNULL)
    orc = wdpm_amd.load(so)
    dem = orc.synth_dem(n, n)
    bd = np.full((n + 2, n + 2), MISSING)
    bd[1:-1, 1:-1] = dem
    bw = np.where(bd > MISSING, ADD_M, 0.0)
    with orc.context(module="add", nrows=n, ncols=n, missingvalue=MISSING) as c:
        c.upload(bd, bw)
        c.iterate(1)
        t = time.perf_counter()
        c.run_block(iters, THRES)
        dt = time.perf_counter() - t
    return {"value": n * n * iters / dt, "unit": "cell-updates/s", "cores": 1, "kind": "port",
            "sample": f"oracle/wdpm_oracle.c, synthetic {n}x{n} all-wet add 100 mm, {iters} iterations, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)   # one block of the reference loop
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", type=int, default=16384)
    ap.add_argument("--exchange-every", type=int, default=int(os.environ.get("WDPM_EXCHANGE_EVERY", "4")))
    ap.add_argument("--kernel", choices=["auto", "pass", "fused", "fused2", "fused2w"], default="auto")
    ap.add_argument("--module", choices=["add", "drain"], default="add",
                    help="drain = BASELINE config 5: water-in is the add-100-mm state after --drain-spinup iterations")
    ap.add_argument("--drain-spinup", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import wdpm_amd
    from wdpm_amd.rowblock import DeviceTransport, RowBlockSolver

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # rehearsal on a box with fewer GPUs than ranks: WDPM_DIST_BACKEND=gloo WDPM_HALO=host lets several
    # ranks share one GPU (RCCL refuses that); the driver's real runs use nccl, one GPU per rank
    backend = os.environ.get("WDPM_DIST_BACKEND", "nccl")
    ngpu = torch.cuda.device_count()
    local_rank = local_rank % max(ngpu, 1)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    lib = wdpm_amd.load_hip()
    n = args.size
    kernel = {"auto": wdpm_amd.KERNEL_AUTO, "pass": wdpm_amd.KERNEL_PASS, "fused": wdpm_amd.KERNEL_FUSED,
              "fused2": wdpm_amd.KERNEL_FUSED2, "fused2w": wdpm_amd.KERNEL_FUSED2W}[args.kernel]
    transport = fallback = None
    if world > 1:
        from wdpm_amd.rowblock import HostTransport
        if backend == "nccl":
            # GPU-direct halos over RCCL; a host-staged gloo path stands by should the platform refuse them
            host = HostTransport(dist, dist.new_group(backend="gloo"))
            transport, fallback = DeviceTransport(dist, torch.device("cuda", local_rank)), host
            if os.environ.get("WDPM_HALO", "device") == "host":
                transport, fallback = host, None
        else:
            transport = HostTransport(dist, None)
    drain_kw = {}
    if args.module == "drain":
        # the outlet is the first row-major minimum of the DEM (WDPMCL.c:1005-1017), padded coordinates
        full = lib.synth_dem(n, n)
        k = int(np.argmin(full))
        drain_kw = dict(drainrow=k // n + 1, draincol=k % n + 1)
        del full
    solver = RowBlockSolver(lib, "add", n, n, MISSING, rank=rank, nranks=world, exchange_every=args.exchange_every,
                            transport=transport, dist=dist, device=local_rank, kernel=kernel,
                            fallback_transport=fallback)
    solver.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    bd, bw = build_slab_inputs(lib, n, solver.slab)
    solver.ctx.upload(bd, bw)
    solver.agree_on_options()
    if args.module == "drain":
        # spin the water up with the add module, then hand the state to a drain solver
        solver.run_block(args.drain_spinup, THRES)
        solver.exchange()
        bw = solver.ctx.download_water()
        solver.close()
        solver = RowBlockSolver(lib, "drain", n, n, MISSING, rank=rank, nranks=world,
                                exchange_every=args.exchange_every, transport=solver.transport, dist=dist,
                                device=local_rank, kernel=kernel, **drain_kw)
        solver.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        solver.ctx.upload(bd, bw)
        solver.agree_on_options()
        s = solver.slab
        dr = drain_kw["drainrow"]
        w_out = float(bw[dr - s.row0, drain_kw["draincol"]]) if s.row0 <= dr < s.row0 + s.rows else 0.0
        if world > 1:
            t = torch.tensor([w_out], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            w_out = float(t.item())
        solver.set_totaldrain(max(w_out, 0.0))
    del bd, bw

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    solver.run_block(args.warmup, THRES)           # untimed warm-up steps
    if world > 1:
        solver.exchange()                          # the transport's first use (communicator set-up) is never timed
    sync()
    solver.ctx.timing_reset()
    t0 = time.perf_counter()
    max_diff = solver.run_block(args.steps, THRES) # exactly K timed steps
    stats_s = None
    if args.module == "drain":
        # the drain module's per-block bookkeeping (WDPMCL.c:1257-1268) belongs to the block loop: |d totaldrain|
        # and the sequential row-major volume sum (host, streamed down in chunks; rank-chained at N > 1)
        torch.cuda.synchronize()
        ts = time.perf_counter()
        solver.drain_stats()
        stats_s = time.perf_counter() - ts
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    launches, kernel_ms = solver.ctx.timing()
    dem32 = bool(solver.ctx.get_option(wdpm_amd.OPT_DEM32)) and args.module == "add"

    if rank == 0:
        cells = float(n) * n
        value = cells * args.steps / dt
        own_cells = float(solver.slab.own_hi - solver.slab.own_lo + 1) * n if world > 1 else cells
        iter_ms = kernel_ms / max(args.steps, 1)     # device time of one iteration's stencil launch(es)
        achieved = ALGO_BYTES_PER_CELL_UPDATE * own_cells / (iter_ms * 1e-3) / 1e9 if iter_ms > 0 else 0.0
        out = {
            "metric": "cell-updates/sec on Add module, 16k x 16k DEM" if args.module == "add" else
                      "cell-updates/sec on Drain module (BASELINE config 5)",
            "value": value, "unit": "cell-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"synthetic {n}x{n} diamond-square DEM (seed {n}), Add 100 mm, rof 1.0, "
                                    f"thres 0.005 mm, one block of {args.steps} iterations") if args.module == "add" else
                                   (f"synthetic {n}x{n} DEM (seed {n}), Drain from the add-100-mm state after "
                                    f"{args.drain_spinup} iterations, one block of {args.steps} iterations"),
                       "kernel": args.kernel, "decomposition": f"row-block x{world}" if world > 1 else "single GPU",
                       "exchange_every": args.exchange_every if world > 1 else None,
                       "max_diff_m": max_diff,
                       **({"drain_bookkeeping_ms_per_block": stats_s * 1e3} if stats_s is not None else {})},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(n, world, args.kernel, dem32),
                         "dem": "32-bit codes, verified lossless on upload (20 B of HBM traffic per cell-update)" if dem32
                                else "fp64 (24 B of HBM traffic per cell-update)",
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_CELL_UPDATE * own_cells,
                         "kernel_ms_per_iteration": iter_ms, "launches": launches,
                         "job_frac": value * ALGO_BYTES_PER_CELL_UPDATE / 1e9 / (HBM_PEAK_GBS * world)},
        }
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    solver.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
