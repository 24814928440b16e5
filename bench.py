#!/usr/bin/env python3
"""bench.py — cell-updates/s of the WDPM Add module on a synthetic 16384 x 16384 DEM (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 16384] [--driver ranks|group]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one iteration (all 9 colour passes) over the whole raster.  The timed region is one
block of the reference's loop: threshold flush + snapshot, K iterations, max-|dw| reduction
(src/WDPMCL.c:1055-1125,1239-1254), with the rasters already resident in HBM.  N > 1: row-block
decomposition, one rank per GPU, the C driver wdpm_amd/csrc/wdpm_rowblock.c on every rank, halo rows
by RCCL send/recv issued by the library on its own stream; total work is fixed, so scaling is strong.
Rank 0 prints ONE JSON line.

`--gpus N` without a launcher (WORLD_SIZE unset) starts the N ranks itself, as child processes
(`python -m torch.distributed.run ...` of this same file, before anything here has touched the GPU),
relays rank 0's line and exits with the launcher's status.  `--driver group` instead runs all N ranks
inside THIS process, one host thread per GPU (wdpm_group_*: what the WDPMCL drop-in does with
WDPM_GPUS=N).  On a box with fewer GPUs than ranks (rehearsal) the ranks share GPUs and the halos are
staged through the host (RCCL wants one GPU per rank); the line says so.

Nothing in an N-rank run can end without a line.  torch.distributed (gloo by default: the control plane -
barriers, the communicator id, a handful of scalars - needs no GPU fabric) only brackets the timed region;
the halos travel by the library's own RCCL communicator, whose set-up and first transfer run with deadlines
(WDPM_RCCL_TIMEOUT_S / WDPM_SYNC_TIMEOUT_S, include/wdpm.h).  If any rank's RCCL refuses, fails or runs past
a deadline, ALL ranks agree on it (one all-reduce) and go on with host-staged halos over gloo in the same
processes; the line then says `"degraded": true` and why.  The synthetic DEM is generated ONCE per node
(local rank 0 -> /dev/shm, the others map their rows).
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


class SharedDem:
    """The n x n synthetic DEM of the workload (integer-seeded C generator, seed = size), made ONCE per node: local rank 0
    writes it to /dev/shm, everybody else maps it read-only after a barrier (round 2: every rank generated the whole raster
    to keep its eighth - 2.1 GB and 4-27 s of one core per rank at 16384^2).  Falls back to generating it in this process
    when there is nothing to share with (one rank) or /dev/shm cannot be used."""

    def __init__(self, lib, n, dist, local_rank):
        self.n, self.path, self.owner, self.dist = n, None, False, dist
        if dist is None:
            self.dem = lib.synth_dem(n, n)
            return
        # (the launcher of this file names its runs, so that it can sweep what a killed run left behind: WDPM_BENCH_SHM_TAG)
        run = os.environ.get("WDPM_BENCH_SHM_TAG", "") + os.environ.get("TORCHELASTIC_RUN_ID", "") + "_" + os.environ.get("MASTER_PORT", "0")
        path = f"/dev/shm/wdpm_bench_{os.getuid()}_{''.join(c for c in run if c.isalnum() or c == '_')}_{n}.npy"
        ok = 1.0
        if local_rank == 0:
            try:
                m = np.lib.format.open_memmap(path, mode="w+", dtype=np.float64, shape=(n, n))
                lib.check(lib.dll.wdpm_synth_dem(n, n, m.ctypes.data))
                m.flush()
                del m
                self.owner = True
            except Exception as e:  # noqa: BLE001 - no /dev/shm, no room: every rank makes its own
                print(f"bench.py: cannot share the DEM through {path} ({e}); every rank generates it", file=sys.stderr)
                ok = 0.0
        import torch
        t = torch.tensor([ok], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)        # also the barrier behind which the file is complete
        if t.item() > 0:
            self.path = path
            self.dem = np.load(path, mmap_mode="r")
            # every rank has its mapping (it survives the unlink): the name can go at once, so that a run that is killed later
            # - the launcher's deadline, a rank that raises - leaves nothing behind in RAM-backed /dev/shm
            dist.barrier()
            if self.owner:
                try:
                    os.unlink(path)
                except OSError:
                    pass
                self.path = None
        else:
            self.dem = lib.synth_dem(n, n)

    def slab_inputs(self, slab):
        """padded dem / water rows [row0, row0+rows) of the config-4 workload"""
        n, ncp = self.n, self.n + 2
        bd = np.full((slab.rows, ncp), MISSING)
        lo, hi = max(slab.row0, 1), min(slab.row0 + slab.rows, n + 1)   # padded interior rows held
        bd[lo - slab.row0:hi - slab.row0, 1:-1] = self.dem[lo - 1:hi - 1]
        bw = np.where(bd > MISSING, ADD_M, 0.0)
        return bd, bw

    def whole_inputs(self):
        n = self.n
        bd = np.full((n + 2, n + 2), MISSING)
        bd[1:-1, 1:-1] = self.dem
        return bd, np.where(bd > MISSING, ADD_M, 0.0)

    def argmin(self):
        return int(np.argmin(self.dem))

    def close(self):
        """collective: nobody needs the file any more"""
        self.dem = None
        if self.dist is not None:
            self.dist.barrier()
        if self.owner and self.path:
            try:
                os.unlink(self.path)
            except OSError:
                pass


def measured_counters(lib, n, world, kernel, dem_kind):
    """Counter evidence for the dominant kernel from the committed rocprofv3 PMC passes (separate
    --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE runs of this same command,
    gfx950 x2 FETCH correction, calibrated on kernels of known byte count: profiles/).  Quoted ONLY for the
    configuration AND the library build they were taken on: profiles/traffic.json names the build
    (wdpm_build_info(): a hash of the kernel sources) and the kernel's name; a library built from other kernel
    sources gets no counters, only the note that the file is of another build.  With the DEM streamed as 32-bit
    codes the kernel moves 20 B per cell-update, i.e. LESS than the 24 algorithmic bytes `achieved` is priced at.
    -> dict(traffic=HBM bytes per launch, valu_issue_frac=share of the kernel's cycles in which a SIMD issues a VALU
    instruction (SQ_INSTS_VALU x 4 cycles / SIMDs / (GRBM_GUI_ACTIVE / XCDs)), kernel_ms_at_collection, source=...)"""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    build = lib.dll.wdpm_build_info().decode()
    try:
        with open(path) as f:
            t = json.load(f)
        if t["size"] == n and t["n_gpus"] == world and kernel in ("auto", "fused"):
            e = t.get(dem_kind)        # "dem16" | "dem32" | "fp64_dem": how the kernel of this run streams the DEM
            if e is None or e.get("build_info") != build:
                e = e or {}
                return {"mismatch": f"profiles/traffic.json is of build [{e.get('build_info')}], this library is [{build}]: no counters quoted"}
            return {"traffic": e["hbm_bytes_per_launch"], "valu_issue_frac": e.get("valu_issue_frac"),
                    "kernel_ms_at_collection": e.get("kernel_ms"), "kernel_name": e.get("kernel"), "source": e.get("source"),
                    "build_info": build}
    except (OSError, KeyError, ValueError):
        pass
    return {}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def reference_baseline(samples):
    """The UNMODIFIED reference's serial functions (runoffs() on its own double** globals, driven in
    the loop order of WDPMCL.c:1097-1103) from oracle/_ref/libwdpm_ref.so, which oracle/Makefile
    builds from the sources where they lie under /root/reference and which travels to the GPU box as
    a binary.  samples = [(n, iterations), ...] -> list of results, or None if that library is not
    there or does not load."""
    import ctypes as C
    import wdpm_amd
    so = os.path.join(ROOT, "oracle", "_ref", "libwdpm_ref.so")
    try:
        ref = C.CDLL(so)
        ref.ref_setup.argtypes = [C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_int]
        ref.ref_iterate.argtypes = [C.c_int, C.c_int]
    except (OSError, AttributeError):
        return None
    out = []
    for n, iters in samples:
        dem = wdpm_amd.load_hip().synth_dem(n, n)
        bd = np.full((n + 2, n + 2), MISSING)
        bd[1:-1, 1:-1] = dem
        del dem
        bw = np.where(bd > MISSING, ADD_M, 0.0)
        ref.ref_setup(n, n, MISSING, bd.ctypes.data, bw.ctypes.data, 0.0, 0, 0)
        del bd, bw
        ref.ref_iterate(0, 1)
        t = time.perf_counter()
        ref.ref_iterate(0, iters)
        dt = time.perf_counter() - t
        out.append({"size": n, "iterations": iters, "seconds": round(dt, 2), "value": n * n * iters / dt})
    return out


def oracle_baseline(samples):
    """the CPU oracle (bit-equal port of the reference's serial path) on the same samples"""
    import subprocess
    import wdpm_amd
    so = os.path.join(ROOT, "oracle", "_build", "libwdpm_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], stdout=subprocess.DEVNULL)
    orc = wdpm_amd.load(so)
    out = []
    for n, iters in samples:
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
        out.append({"size": n, "iterations": iters, "seconds": round(dt, 2), "value": n * n * iters / dt})
    return out


def cpu_baseline(n=16384, full=False, quick=False):
    """The reference's own serial code if its prebuilt library is present (kind "reference"), else the
    CPU oracle (bit-equal port of it, kind "port"), timed on ONE host core (the reference's serial path is
    single-threaded) on bounded samples of the same workload.  The headline `value` is on the metric's own
    configuration: the synthetic 16384^2 raster, 5 iterations of the serial loop (WDPMCL.c:1094-1106; about 30 s at
    4e7 cell-updates/s, the top of the bench contract's 10 - 30 s - SURVEY.md §8d names 10 iterations, which
    `--cpu-baseline-full` runs: about a minute);
    beside it §8d's 1024^2 x 1000 iterations (cache-resident; later iterations of a settling raster take the slower
    branches) and 4096^2 x 24.  A reported baseline, never the product path."""
    # the headline sample: the bench's own raster size, as many iterations as take about 30 s (never fewer than two): FIVE at 16384^2
    # (round 4 timed two; SURVEY 8d names ten, which `--cpu-baseline-full` runs in about a minute)
    head_iters = 10 if full and n >= 16384 else max(2, min(1000, int(round(1.35e9 / (float(n) * n)))))
    samples = [(n, head_iters), (4096, 24), (1024, 1000)]
    if quick:        # tests of the line's shape only (--cpu-baseline-quick): seconds instead of a minute, not a baseline anybody should quote
        samples = [(n, 3), (4096, 1), (1024, 10)]
    kind, res = "reference", reference_baseline(samples)
    if res is None:
        kind, res = "port", oracle_baseline(samples)
    head = res[0]
    what = ("runoffs() of the unmodified src/WDPMCL.c (oracle/_ref/libwdpm_ref.so)" if kind == "reference" else
            "oracle/wdpm_oracle.c")
    return {"value": head["value"], "unit": "cell-updates/s", "cores": 1, "kind": kind, "cpu_model": cpu_model(),
            "host_cores": os.cpu_count(),
            **({"quick": "shape test only: a few iterations per sample"} if quick else {}),
            "sample": f"{what}, synthetic {head['size']}x{head['size']} all-wet add 100 mm (the metric's own raster), "
                      f"{head['iterations']} iterations after one untimed, {head['seconds']:.1f} s, one thread (the reference's serial path has no other)",
            "spread_note": "the rate falls with the raster's size, not with the number of iterations timed: at 16384^2 the serial loop streams "
                           "4.3 GB per iteration from DRAM through one core (rows of 131 KB behind row pointers), at 4096^2 and 1024^2 most "
                           "of it stays in the host's caches; the 1000-iteration block of a settling raster is no slower than its first iterations",
            "samples": res}


def self_launch(args):
    """`python bench.py --gpus N` with no launcher: start the N ranks as CHILD processes (never exec: this
    process has not touched the GPU and will not), relay rank 0's JSON line, return the launcher's status.
    The ranks settle a refused or stuck RCCL among themselves (host-staged halos in the same processes, see main()); what
    is left for this level is a run that crashes or does not finish at all: it is started ONCE more with host-staged
    halos from the outset, within what is left of the time budget (WDPM_BENCH_BUDGET_S, default 560 s: under the driver's
    600 s), and the line of that second run carries `"degraded": true` and the first attempt's status, so that it cannot
    be mistaken for a GPU-direct result."""
    import signal
    import socket
    import subprocess
    import glob
    t_start = time.monotonic()
    # Time limits: the default command gets the tight pair the driver's 600 s call for - 200 s for the first attempt,
    # 560 s in all.  A longer job (more steps, a spin-up, a bigger raster) is a healthy run, not a hung one: its limits grow with
    # the work it was asked for (about 1.5e11 cell-updates/s per GPU, DEM generation and set-up on top), unless the environment
    # says otherwise (WDPM_BENCH_RANKS_TIMEOUT / WDPM_BENCH_BUDGET_S).
    size, steps, warm = float(getattr(args, "size", 16384)), getattr(args, "steps", 1000), getattr(args, "warmup", 20)
    spin = getattr(args, "drain_spinup", 0) if getattr(args, "module", "add") == "drain" else 0
    expected = 60.0 + size ** 2 / 4.0e7 + size ** 2 * (steps + warm + spin) / (1.0e11 * max(args.gpus, 1))
    first_default = max(200.0, 2.5 * expected)
    budget = float(os.environ.get("WDPM_BENCH_BUDGET_S", str(max(560.0, 2.0 * first_default + 60.0) if first_default > 200.0 else 560.0)))
    shm_tag = f"L{os.getpid()}x{int(time.time()) % 100000}"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), WDPM_BENCH_SHM_TAG=shm_tag)
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}"]
    first_limit = min(float(os.environ.get("WDPM_BENCH_RANKS_TIMEOUT", str(first_default))), budget)

    def sweep_shm():
        """the shared DEM of THIS launcher's ranks, if they were killed before they could unlink it (2.1 GB of RAM each)"""
        for f in glob.glob(f"/dev/shm/wdpm_bench_{os.getuid()}_{shm_tag}*"):
            try:
                os.unlink(f)
            except OSError:
                pass

    def elapsed():
        return time.monotonic() - t_start

    def launch(rdzv, env, limit):
        """-> (return code or None on timeout, stdout, stderr); the children get their own process group so that a
        hung run can be ended as a whole (by that exact group id, nothing else)"""
        p = subprocess.Popen(base + rdzv + [os.path.abspath(__file__), *sys.argv[1:]], stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, text=True, env=env, start_new_session=True)
        try:
            out, err = p.communicate(timeout=limit)
            return p.returncode, out, err
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            out, err = p.communicate()
            sweep_shm()
            return None, out, err

    def run(env, limit):
        t_end = time.monotonic() + limit
        for attempt in range(3):
            if attempt == 0:       # the launcher picks and holds its own rendezvous port
                rdzv = ["--standalone", "--local-addr", "127.0.0.1"]
            else:                  # a port that was free a moment ago
                with socket.socket() as s:
                    s.bind(("127.0.0.1", 0))
                    rdzv = ["--master-addr", "127.0.0.1", "--master-port", str(s.getsockname()[1])]
            rc, out, err = launch(rdzv, env, max(t_end - time.monotonic(), 1.0))
            sys.stderr.write(err)
            if rc == 0 or rc is None or "EADDRINUSE" not in err:
                break              # only a lost race for the rendezvous port is worth another try
        return rc, out

    rc, out = run(env, first_limit)
    lines = [ln for ln in out.splitlines() if ln.startswith('{"metric"')]
    gpu_direct = env.get("WDPM_HALO", "rccl") == "rccl"
    first = None
    if (rc != 0 or not lines) and gpu_direct:
        first = f"no result within {first_limit:.0f} s" if rc is None else f"exit status {rc}"
        left = budget - elapsed() - 5.0
        print(f"bench.py: the ranks ended with {first} after {elapsed():.0f} s of a {budget:.0f} s budget; "
              f"starting them once more with host-staged halos over gloo ({left:.0f} s left)", file=sys.stderr, flush=True)
        if left > 20.0:
            rc, out = run(dict(env, WDPM_HALO="host", WDPM_DIST_BACKEND="gloo"), left)
            lines = [ln for ln in out.splitlines() if ln.startswith('{"metric"')]
    for ln in out.splitlines():
        if not ln.startswith('{"metric"'):
            print(ln, file=sys.stderr)
    if lines:
        line = lines[-1]
        if first is not None:      # the second attempt's line: say what it is
            try:
                d = json.loads(line)
                d["degraded"] = True
                d["first_attempt"] = f"{first} on RCCL halos; this line is a second run with host-staged halos"
                line = json.dumps(d)
            except ValueError:
                pass
        print(line, flush=True)
    sweep_shm()
    print(f"bench.py: {elapsed():.0f} s of the {budget:.0f} s budget used", file=sys.stderr, flush=True)
    if rc is None:
        return 124
    return rc if rc != 0 or lines else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)   # one block of the reference loop
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", type=int, default=16384)
    ap.add_argument("--exchange-every", type=int, default=int(os.environ.get("WDPM_EXCHANGE_EVERY", "8")))
    ap.add_argument("--kernel", choices=["auto", "pass", "fused"], default="auto")
    ap.add_argument("--driver", choices=["ranks", "group"], default="ranks",
                    help="ranks = one process per GPU (torch.distributed.run); group = all ranks in this process, one "
                         "host thread per GPU (what WDPMCL does with WDPM_GPUS=N)")
    ap.add_argument("--module", choices=["add", "drain"], default="add",
                    help="drain = BASELINE config 5: water-in is the add-100-mm state after --drain-spinup iterations")
    ap.add_argument("--drain-spinup", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="also time SURVEY §8d's 16384^2 x 10 iterations on the host (about a minute) and make it the headline")
    ap.add_argument("--cpu-baseline-quick", action="store_true",
                    help="tests only: a few iterations per CPU sample (the line's shape, not a baseline to quote)")
    args = ap.parse_args()

    launched = "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched and args.driver == "ranks":
        sys.exit(self_launch(args))

    # multi-process GPU work on this driver needs dmabuf IPC (RCCL, tensors shared across processes); exported on the pool's
    # boxes already - set here too, before the runtime is loaded, for a shell that lacks it
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # the library's deadlines for communicator set-up / the first transfer and for stream waits: tighter than its defaults,
    # so that a stuck fabric costs the bench a minute, not its whole time budget
    os.environ.setdefault("WDPM_RCCL_TIMEOUT_S", "60")
    os.environ.setdefault("WDPM_SYNC_TIMEOUT_S", "120")
    t_proc = time.monotonic()
    import torch
    import wdpm_amd
    from wdpm_amd.rowblock import Group, HostTransport, RowBlockSolver

    world = int(os.environ.get("WORLD_SIZE", "1")) if args.driver == "ranks" else 1
    rank = int(os.environ.get("RANK", "0")) if args.driver == "ranks" else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if args.driver == "ranks" else 0
    if args.driver == "ranks":
        args.gpus = world
    ngpu = torch.cuda.device_count()
    if ngpu < 1 or not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # fewer GPUs than ranks (rehearsal on a one-GPU box): ranks share GPUs, which RCCL refuses - host halos
    shared = ngpu < args.gpus
    # The control plane (barriers around the timed region, the communicator id, a few scalars) runs on gloo: it needs no
    # GPU fabric, so a node whose RCCL is in trouble still gets as far as finding that out - and past it.  The halos go
    # through the library's own RCCL communicator.  WDPM_DIST_BACKEND=nccl puts the control plane on RCCL too.
    backend = os.environ.get("WDPM_DIST_BACKEND", "gloo")
    halo = os.environ.get("WDPM_HALO", "host" if shared else "rccl")
    device = local_rank % ngpu
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    lib = wdpm_amd.load_hip()
    n = args.size
    kernel = {"auto": wdpm_amd.KERNEL_AUTO, "pass": wdpm_amd.KERNEL_PASS, "fused": wdpm_amd.KERNEL_FUSED}[args.kernel]
    shared_dem = SharedDem(lib, n, dist, local_rank)
    drain_kw = {}
    if args.module == "drain":
        # the outlet is the first row-major minimum of the DEM (WDPMCL.c:1005-1017), padded coordinates
        k = shared_dem.argmin()
        drain_kw = dict(drainrow=k // n + 1, draincol=k % n + 1)

    def dist_reduce(v, op):
        if world == 1:
            return v
        t = torch.tensor([v], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=op)
        return float(t.item())

    def dist_max(v):
        return dist_reduce(v, dist.ReduceOp.MAX) if world > 1 else v

    stats_s = None
    rccl_ranks = None
    degraded = None               # why the run is not the GPU-direct one that was asked for
    again = None                  # (seconds, steady timing) of the block after the timed one (the `ranks` driver, short add runs)
    k_used = args.exchange_every
    if args.driver == "group":
        # ---- all ranks in this process: wdpm_group_* (thread per GPU, RCCL via ncclCommInitAll or peer copies)
        devs = [int(d) for d in os.environ["WDPM_DEVICES"].split(",")] if os.environ.get("WDPM_DEVICES") else \
               [g % ngpu for g in range(args.gpus)]
        bd, bw = shared_dem.whole_inputs()
        grp = Group(lib, "add", n, n, MISSING, devs, exchange_every=args.exchange_every, kernel=kernel)
        grp.upload(bd, bw)
        if args.module == "drain":
            grp.run_block(args.drain_spinup, THRES)
            bw = grp.download_water()
            grp.close()
            grp = Group(lib, "drain", n, n, MISSING, devs, exchange_every=args.exchange_every, kernel=kernel, **drain_kw)
            grp.upload(bd, bw)
            grp.set_totaldrain(max(float(bw[drain_kw["drainrow"], drain_kw["draincol"]]), 0.0))
        del bd, bw
        shared_dem.close()
        ranks_used, halo = grp.size, wdpm_amd.HALO_NAMES[grp.halo_kind]
        ctx0 = grp.rank_ctx(0)
        import ctypes as C
        kk = C.c_int32()
        lib.check(lib.dll.wdpm_rank_info(lib.dll.wdpm_group_rank(grp._h, 0), None, C.byref(kk), None))
        k_used = kk.value
        grp.run_block(args.warmup, THRES)           # untimed warm-up steps (ends with a synchronous reduction)
        e0, x0, _ = grp.enqueue_stats()
        for i in range(grp.size):
            lib.check(lib.dll.wdpm_timing_reset(grp.rank_ctx(i)))
        t0 = time.perf_counter()
        max_diff = grp.run_block(args.steps, THRES)  # exactly K timed steps
        if args.module == "drain":
            ts = time.perf_counter()
            grp.drain_stats()
            stats_s = time.perf_counter() - ts
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        e1, x1, _ = grp.enqueue_stats()
        enqueue_us = (e1 - e0) / max(ranks_used, 1) / max(args.steps, 1) * 1e6
        refresh_us = (x1 - x0) / max(ranks_used, 1) / max(args.steps, 1) * 1e6
        la, ms = C.c_int64(), C.c_double()
        lib.check(lib.dll.wdpm_timing_get(ctx0, C.byref(la), C.byref(ms)))
        launches, kernel_ms = la.value, ms.value
        lib.check(lib.dll.wdpm_timing_get_steady(ctx0, C.byref(la), C.byref(ms)))
        steady_launches, steady_ms = la.value, ms.value
        per_rank = []
        for i in range(grp.size):
            h, row = grp.rank_ctx(i), {}
            for key, fn in (("all", lib.dll.wdpm_timing_get), ("steady", lib.dll.wdpm_timing_get_steady), ("xch", lib.dll.wdpm_timing_get_exchange)):
                lib.check(fn(h, C.byref(la), C.byref(ms)))
                row[key] = (la.value, ms.value)
            nb, wb = C.c_int32(), (C.c_double * 9)()
            lib.check(lib.dll.wdpm_balance_info(h, C.byref(nb), wb))
            row["balance"] = (nb.value, [float(v) for v in wb])
            od, bus = C.c_int32(), C.create_string_buffer(32)
            lib.check(lib.dll.wdpm_device_info(h, C.byref(od), bus, 32))
            row["device"], row["pid"] = (od.value, bus.value.decode()), os.getpid()
            per_rank.append(row)
        v = C.c_int64()
        lib.check(lib.dll.wdpm_get_option(ctx0, wdpm_amd.OPT_DEM32, C.byref(v)))
        dem32 = bool(v.value)
        lib.check(lib.dll.wdpm_get_option(ctx0, wdpm_amd.capi.OPT_DEM16, C.byref(v)))
        dem16 = dem32 and v.value == 1
        sl = wdpm_amd.capi.SlabStruct()
        lib.check(lib.dll.wdpm_rank_slab(lib.dll.wdpm_group_rank(grp._h, 0), 0, C.byref(sl)))
        own_rows0 = sl.own_hi - sl.own_lo + 1 if ranks_used > 1 else n
        decomposition = f"row-block x{ranks_used}, one process, one host thread per GPU" if ranks_used > 1 else "single GPU"
        closer = grp.close
    else:
        # ---- one process per GPU: wdpm_rank_* with RCCL halos (or host-staged ones when ranks share a GPU)
        # the host-staged transport's bytes travel by gloo: its group is made by every rank, up front
        host_group = dist.new_group(backend="gloo") if world > 1 and backend == "nccl" else None
        mk = dict(rank=rank, nranks=world, exchange_every=args.exchange_every, dist=dist, device=device, kernel=kernel)

        def make_solver(module, **extra):
            if world == 1:
                return RowBlockSolver(lib, module, n, n, MISSING, **mk, **extra)
            if halo == "rccl":
                return RowBlockSolver(lib, module, n, n, MISSING, halo="rccl", **mk, **extra)
            return RowBlockSolver(lib, module, n, n, MISSING, halo="host", transport=HostTransport(dist, host_group),
                                  **mk, **extra)

        solver, refused = None, ""
        try:
            solver = make_solver("add")                # RCCL: communicator set-up, with the library's deadline
            bd, bw = shared_dem.slab_inputs(solver.slab)
            solver.upload(bd, bw)
            if world > 1:
                solver.exchange()                      # the first transfer: peer mappings, links - with a deadline too
                solver.ctx.synchronize()
                # ... and once through everything a block does across ranks (iteration groups, the overlapped last
                # iteration with the transfer beside it, the refresh, the all-gather of max diff), then back to the start
                solver.run_block(2 * solver.k + 1, THRES)
                solver.upload(bd, bw)
        except Exception as e:  # noqa: BLE001 - a platform that refuses GPU-direct halos, or one that never answers
            refused = f"{type(e).__name__}: {e}"
        if world > 1 and halo == "rccl":
            # decided by ALL ranks together (a rank on its own must never change transport): any refusal anywhere
            # puts every rank on host-staged halos, in these same processes
            if dist_reduce(1.0 if refused else 0.0, dist.ReduceOp.MAX) > 0:
                why = [None] * world
                dist.all_gather_object(why, refused)
                why = next((w for w in why if w), "")
                if rank == 0:
                    print(f"bench.py: RCCL halos are not to be had after {time.monotonic() - t_proc:.0f} s ({why}); every rank "
                          f"switches to host-staged halos", file=sys.stderr, flush=True)
                if solver is not None:
                    solver.abort_comm()                # the peers' queued receives end; nothing of it is waited for
                    solver.close()
                halo, refused, degraded = "host (RCCL refused)", "", f"RCCL halos refused or timed out: {why}"
                solver = RowBlockSolver(lib, "add", n, n, MISSING, halo="host", transport=HostTransport(dist, host_group), **mk)
                bd, bw = shared_dem.slab_inputs(solver.slab)
                solver.upload(bd, bw)
        if refused:
            raise SystemExit(f"bench.py: {refused}")
        if args.module == "drain":
            # spin the water up with the add module, then hand the state to a drain solver
            solver.run_block(args.drain_spinup, THRES)
            solver.exchange()
            bw = solver.ctx.download_water()
            solver.close()
            solver = make_solver("drain", **drain_kw)
            s = solver.slab
            if s.rows != bd.shape[0]:
                # the drain partition keeps the outlet three rows clear of every boundary and moved one: the add
                # run's slabs no longer fit (the synthetic DEMs drain at a corner, so this does not happen there)
                sys.exit("bench.py: the drain partition differs from the add run's; use --driver group")
            solver.upload(bd, bw)
            dr = drain_kw["drainrow"]
            w_out = float(bw[dr - s.row0, drain_kw["draincol"]]) if s.row0 <= dr < s.row0 + s.rows else 0.0
            solver.set_totaldrain(max(dist_max(w_out), 0.0))
        del bd, bw
        shared_dem.close()
        ranks_used = world
        rccl_ranks = solver.rccl_ranks()
        k_used = solver.k
        if world > 1 and k_used != args.exchange_every and rank == 0:
            print(f"bench.py: exchange interval {args.exchange_every} does not fit these slabs; the library uses {k_used}",
                  file=sys.stderr, flush=True)
        solver.run_block(args.warmup, THRES)           # untimed warm-up steps
        if world > 1:
            solver.exchange()                          # the transport's first use is never timed
        # the opening bracket: everybody's queue is empty, everybody is here
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        solver.ctx.timing_reset()
        t0 = time.perf_counter()
        max_diff = solver.run_block(args.steps, THRES) # exactly K timed steps (ends with the ranks' all-gather of max diff)
        if args.module == "drain":
            # the drain module's per-block bookkeeping (WDPMCL.c:1257-1268) belongs to the block loop: |d totaldrain|
            # and the sequential row-major volume sum (evaluated exactly on the device; rank-chained at N > 1)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            solver.drain_stats()
            stats_s = time.perf_counter() - ts
        # the closing bracket.  Each rank stops its own clock when ITS device is idle, then all meet; the job's time is
        # the MAX over ranks (the block's last step is a collective, so no rank can be early by more than that step)
        torch.cuda.synchronize()
        dt_mine = time.perf_counter() - t0
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        dt = dist_max(dt_mine)
        launches, kernel_ms = solver.ctx.timing()
        steady_launches, steady_ms = solver.ctx.timing_steady()
        mine = {"all": (launches, kernel_ms), "steady": (steady_launches, steady_ms), "xch": solver.ctx.timing_exchange(),
                "seconds": dt_mine, "balance": solver.ctx.balance_info(), "device": solver.ctx.device_info(), "pid": os.getpid()}
        per_rank = [mine]
        if world > 1:
            per_rank = [None] * world
            dist.all_gather_object(per_rank, mine)
        # The same K steps ONCE MORE, outside the contract's timed region and reported beside it, never as `value`: an MI355X whose
        # compute units have idled for 5 ms runs this kernel 10 - 35 % slower and takes ~30 ms of THIS kernel to come back
        # (profiles/r04/first_iterations2.txt: no sleep, copy or other kernel shortens that), so `--warmup 5 --steps 20` times
        # the ramp itself - and at N = 8, where its 25 iterations last 4 ms, the slowest part of it.  After the timed block:
        # untimed iterations until 50 ms have gone by since the warm-up began (the same number on every rank), then K timed
        # ones - the same box at the clocks the reference's 1000-iteration blocks run at.  Short add runs only (a long one is
        # past the ramp by itself).
        if args.module == "add" and 1 <= args.steps <= 200 and not degraded:
            per_step = dt / max(args.steps, 1)
            more = int(min(4000, max(0.0, 0.050 - dt - args.warmup * per_step) / per_step)) if per_step > 0 else 0
            if more > 0:
                solver.run_block(more, THRES)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
                torch.cuda.synchronize()
            solver.ctx.timing_reset()
            t1 = time.perf_counter()
            solver.run_block(args.steps, THRES)
            torch.cuda.synchronize()
            dt2 = dist_max(time.perf_counter() - t1)
            again = (dt2, solver.ctx.timing_steady(), more)
        dem32 = bool(solver.ctx.get_option(wdpm_amd.OPT_DEM32))      # (drain too, from round 4: launches of two waves per SIMD)
        dem16 = dem32 and solver.ctx.get_option(wdpm_amd.capi.OPT_DEM16) == 1     # (2: available, but this slab is too small for them to pay)
        own_rows0 = solver.slab.own_hi - solver.slab.own_lo + 1 if world > 1 else n
        decomposition = f"row-block x{world}, one process per GPU" if world > 1 else "single GPU"
        enqueue_us = refresh_us = None
        closer = solver.close

    if rank == 0:
        cells = float(n) * n
        value = cells * args.steps / dt
        own_cells = float(own_rows0) * n                 # rank 0's share: what its kernel launches process
        # Device time of one iteration's stencil launch, rank 0 (HIP events on the kernel's stream).  The DOMINANT kernel is
        # the plain iteration kernel: the first launch of a block is its flush-on-load variant and the last its max-diff
        # variant (other template instances, listed separately by rocprofv3), so the roofline figure is taken over the
        # launches between them; the average over ALL launches of the block is reported next to it.
        all_ms = kernel_ms / max(args.steps, 1)
        iter_ms = steady_ms / steady_launches if steady_launches > 0 else all_ms
        achieved = ALGO_BYTES_PER_CELL_UPDATE * own_cells / (iter_ms * 1e-3) / 1e9 if iter_ms > 0 else 0.0
        # what the kernel really streams: the DEM as fp64, as 4-byte codes, or as 2-byte offsets + one 4-byte base per 48 columns
        moved = ((16.0 + 2.0 + 4.0 / 48.0) if dem16 else 20.0 if dem32 else 24.0) * own_cells
        pmc = measured_counters(lib, n, ranks_used, args.kernel, "dem16" if dem16 else "dem32" if dem32 else "fp64_dem") if args.module == "add" else {}
        traffic = pmc.get("traffic")
        hbm_real = traffic / (iter_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic and iter_ms > 0 else None
        # VALU issue share as the counters gave it: a share of the kernel's cycles (instructions per launch are a property
        # of the binary, and the share does not depend on the clock of the profiled pass)
        valu = pmc.get("valu_issue_frac")
        # `bound` names the roofline `achieved` / `peak` are priced against (the contract's "hbm" | "mfma": HBM - there is no contraction in
        # this path); what the kernel actually runs into first, by the counters, is said beside it
        bound = "hbm"
        limiter = None
        if valu and valu > (hbm_real if hbm_real is not None else moved / (iter_ms * 1e-3) / 1e9 / HBM_PEAK_GBS):
            limiter = "fp64 VALU issue and the board's power cap, ahead of HBM (valu_issue_frac against hbm_real_frac; no MFMA in this path)"
        out = {
            "metric": (f"cell-updates/sec on Add module, {n}x{n} DEM" if args.module == "add" else
                       f"cell-updates/sec on Drain module, {n}x{n} DEM (BASELINE config 5)"),
            "value": value, "unit": "cell-updates/s", "n_gpus": ranks_used, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"synthetic {n}x{n} diamond-square DEM (seed {n}), Add 100 mm, rof 1.0, "
                                    f"thres 0.005 mm, one block of {args.steps} iterations") if args.module == "add" else
                                   (f"synthetic {n}x{n} DEM (seed {n}), Drain from the add-100-mm state after "
                                    f"{args.drain_spinup} iterations, one block of {args.steps} iterations"),
                       "kernel": args.kernel, "driver": args.driver, "decomposition": decomposition,
                       "exchange_every": k_used if ranks_used > 1 else None,
                       "exchange_every_requested": args.exchange_every if ranks_used > 1 else None,
                       "halo": halo if ranks_used > 1 else None, "rccl_ranks": rccl_ranks,
                       "rccl": lib.dll.wdpm_comm_version().decode() if ranks_used > 1 and halo == "rccl" else None,
                       "dist_backend": backend if world > 1 else None,
                       "max_diff_m": max_diff,
                       **({"enqueue_us_per_iteration_per_rank": enqueue_us,
                           "halo_refresh_host_us_per_iteration_per_rank": refresh_us} if enqueue_us is not None else {}),
                       **({"drain_bookkeeping_ms_per_block": stats_s * 1e3} if stats_s is not None else {})},
            "roofline": {"bound": bound,
                         # the whole job against N x 8 TB/s at the 24 algorithmic bytes: the figure that includes everything
                         "job_frac": value * ALGO_BYTES_PER_CELL_UPDATE / 1e9 / (HBM_PEAK_GBS * ranks_used),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         # counter bytes / kernel time / 8 TB/s: what the memory system really carries
                         "hbm_real_frac": hbm_real,
                         # share of the kernel's cycles in which a SIMD issues a VALU instruction: the OTHER ceiling
                         "valu_issue_frac": valu, "limited_by": limiter,
                         "counters_source": (pmc.get("mismatch") or ("profiles/traffic.json <- " + str(pmc.get("source")))) if pmc else None,
                         # the counters' own run: the kernel they were read on, its duration there (profiled passes run at a
                         # lower clock) and the library build - the same as this one, or they would not be quoted
                         "counters_kernel": pmc.get("kernel_name"), "kernel_ms_at_collection": pmc.get("kernel_ms_at_collection"),
                         "counters_build": pmc.get("build_info"), "build": lib.dll.wdpm_build_info().decode(),
                         "dem": ("16-bit offsets from one 32-bit base per 48 columns, an exact identity with the 32-bit codes verified "
                                 "lossless on upload (18.1 B of HBM traffic per cell-update)") if dem16 else
                                "32-bit codes, verified lossless on upload (20 B of HBM traffic per cell-update)" if dem32
                                else "fp64 (24 B of HBM traffic per cell-update)",
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_CELL_UPDATE * own_cells,
                         "moved_bytes_per_launch": moved,
                         "moved_frac": moved / (iter_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if iter_ms > 0 else 0.0,
                         "kernel_ms_per_iteration": iter_ms, "launches": launches,
                         "kernel": "the iteration kernel's plain instance: launches 2 .. K-1 of the block" if steady_launches > 0
                                   else "all launches of the block",
                         "kernel_ms_per_iteration_all_launches": all_ms,
                         "frac_all_launches": ALGO_BYTES_PER_CELL_UPDATE * own_cells / (all_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if all_ms > 0 else 0.0},
        }
        if again is not None:
            dt2, (l2, ms2), more = again
            out["roofline"]["next_block"] = {
                "what": f"the same {args.steps} iterations once more, after the timed block and {more} further untimed ones (not `value`: the "
                        "contract's block is the one above); the card leaves its idle clocks over ~30 ms of this kernel and of nothing "
                        "else, profiles/r04/first_iterations2.txt",
                "untimed_iterations_before": more,
                "ms_per_step": dt2 * 1e3 / args.steps,
                "job_frac": cells * args.steps / dt2 * ALGO_BYTES_PER_CELL_UPDATE / 1e9 / (HBM_PEAK_GBS * ranks_used),
                "kernel_ms_per_iteration": ms2 / l2 if l2 else None,
                "frac": ALGO_BYTES_PER_CELL_UPDATE * own_cells / (ms2 / l2 * 1e-3) / 1e9 / HBM_PEAK_GBS if l2 and ms2 > 0 else None}
        # Which physical GPUs ran this (an N-rank line must prove it ran on N distinct GPUs, and a rehearsal must say so).
        # Per rank: the HIP ordinal its process sees and the PCI bus id of that GPU, as the library reports them for the rank's
        # context (wdpm_device_info).  `rehearsal` whenever the ranks share GPUs, or the RCCL that answered is not the ROCm one
        # (tests/mock_rccl via WDPM_RCCL_LIB: its path is in `rccl`), or the halos do not travel GPU to GPU at all.
        devices = [{"rank": i, "hip_ordinal": r["device"][0], "pci_bus_id": r["device"][1], "pid": r.get("pid")} for i, r in enumerate(per_rank)]
        distinct = len({d["pci_bus_id"] for d in devices})
        out["config"]["devices"] = devices
        out["config"]["distinct_gpus"] = distinct
        rccl_from = out["config"]["rccl"] or ""
        standin = ranks_used > 1 and halo == "rccl" and "/opt/rocm" not in rccl_from and "librccl" not in os.path.basename(rccl_from.rstrip(")").split("(")[-1])
        why = []
        if distinct < ranks_used:
            why.append(f"{ranks_used} ranks on {distinct} physical GPU{'s' if distinct != 1 else ''}")
        if standin:
            why.append(f"the RCCL entry points were bound from a stand-in ({rccl_from})")
        if ranks_used > 1 and not str(halo).startswith(("rccl", "peer")):
            why.append(f"halos are staged through the host ({halo})")
        if why:
            out["rehearsal"] = True
            out["rehearsal_reason"] = "; ".join(why) + ": not a multi-GPU measurement"
        # chunk heights by XCD as rank 0 ended up with them: rebalances so far, the eight weights (which XCDs of this
        # box are slow, and by how much) and the factor on a strip's last chunk; all 1 / 0.95 where the balance did not engage
        out["config"]["xcd_balance"] = {"updates": per_rank[0]["balance"][0], "weights": [round(v, 4) for v in per_rank[0]["balance"][1]]}
        if ranks_used > 1:
            # what an N-GPU line needs to explain its own scaling: per rank, the stencil launches' device time per
            # iteration (HIP events on the rank's stream; `steady` = the plain instance between a block's first and last launch),
            # the halo refreshes INTO the rank - how many, and the us on its stream from queueing a transfer to its rows' arrival,
            # waiting for the neighbour included - and the rank's own wall clock for the timed block
            def per_iter(row, key):
                n_l, ms_l = row[key]
                return ms_l / n_l if n_l else None
            k_all = [r["all"][1] / max(args.steps, 1) for r in per_rank]
            out["per_rank"] = {
                "kernel_ms_per_iteration": k_all,
                "kernel_ms_per_iteration_steady": [per_iter(r, "steady") for r in per_rank],
                "refreshes": [r["xch"][0] for r in per_rank],
                "refresh_us": [r["xch"][1] * 1e3 / r["xch"][0] if r["xch"][0] else None for r in per_rank],
                "refresh_ms_total": [r["xch"][1] for r in per_rank],
                "block_seconds": [r.get("seconds") for r in per_rank],
                "xcd_balance_updates": [r["balance"][0] for r in per_rank],
                "kernel_ms_min_max": [min(k_all), max(k_all)],
                "note": "refresh_us is measured on the receiving rank's stream: it contains the wait for the sender's kernels; a refresh that "
                        "hides behind the overlapped interior launch still shows its full length here",
            }
        if degraded:
            out["degraded"] = True
            out["degraded_reason"] = degraded
        if not args.no_cpu_baseline and ranks_used == 1:   # rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(n, full=args.cpu_baseline_full, quick=args.cpu_baseline_quick)
        print(json.dumps(out), flush=True)
    if degraded:
        # a helper thread may still sit inside the RCCL call that never came back: leave without running anybody's destructors
        if world > 1:
            dist.barrier()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
    closer()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
