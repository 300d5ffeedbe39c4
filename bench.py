#!/usr/bin/env python3
"""Headline benchmark: Chamfer point-pairs/sec (B x N x M) on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one forward chamfer_distance over one batch of synthetic clouds already resident in
HBM: BASELINE.json configs[1] (B=32, N=M=4096, Float32) per GPU.  With N GPUs the global batch is
32*N (configs[4]: B=256 sharded 32/GPU on 8), each rank runs the kernel on its shard and the two
Float64 partial sums are all-reduced over RCCL (weak scaling).  value = global pairs / max-rank time.

Protocol: [K cold steps, timed separately -> cold_ms_per_step] -> burn-in (>= --burn-ms of the same launches, so that the
counted steps run at the clocks a training loop sees, not at the clocks of an idle device) -> W warm-up
steps -> barrier + device sync -> EXACTLY K timed steps -> device sync + barrier; max over ranks.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline             -- bound "mfma": `achieved` = the ALGORITHMIC f16 flops of one launch (SURVEY.md 8(d) MFMA clause: every
                          ordered pair at the K = 16 of the fp16-split filter, 2 * 2BNM * 16 = 34.36 GFLOP) / the dominant
                          kernel's average duration in THIS run (HIP events on its stream), `peak` 2500 TF dense f16,
                          `frac` = achieved / peak.  Since round 6 the kernel SKIPS lane tiles whose bounding box cannot
                          hold a nearest neighbour (spatial pruning): what it executes on the matrix cores is
                          `executed_f16_flops_per_launch` (SQ_INSTS_MFMA x 32768, PMC) and `mfma_pipe_frac` (busy
                          cycles) -- both named extras, lower than `frac`.  Other extras, never `frac`:
                          `algorithmic_fp32_over_valu_peak` (16 flop per pair / t / 157.3 TF; > 1 because the kernel does
                          not execute those flops), `achieved_hbm` / `hbm_frac` (algorithmic bytes / t against 8 TB/s:
                          BASELINE.json asks; not the bound), `traffic` (HBM bytes per launch, PMC), `valu_per_mfma` (PMC)
  issue_occupancy      -- a named extra, NOT a roofline fraction: SIMD issue cycles ((VALU - MFMA) x 4 + MFMA busy) per
                          kernel cycle; it goes UP when a kernel wastes instructions
  protocol             -- benchmarks/metrics.jl:24-38 style numbers: min / median per call, forward and forward+backward
  configs              -- the other BASELINE configs (C1, C3 with the CDFs rebuilt per call + cached, C4, C4', C5's
                          N=1024 shard, C2's shape on surface samples of real meshes), each with its own SURVEY 8(d)
                          roofline; reference_harness: benchmarks/metrics.jl's full sweep n = 2^6 .. 2^14 (chamfer,
                          edge_loss, laplacian_loss; forward and forward+backward)
  cpu_baseline         -- the oracle's KD-tree twin of the reference CPU path, timed on this host (N=1 only)
Everything beyond the timed K steps runs after them (N=1, rank 0) and does not enter `value`.

N > 1 (one process per GPU): NO torch in this file.  torchrun only exports RANK / WORLD_SIZE / MASTER_*; the library
bootstraps its own RCCL communicator (fx3d_comm_bootstrap over tcp://MASTER_ADDR:MASTER_PORT+1), the data-plane
collective is its all-reduce(sum) of 2 Float64 per evaluation, and the control plane (barrier, max over ranks of the
elapsed time) is its all-reduce(max).  The run fails loudly unless the communicator reports exactly N ranks.
`--comm torch` keeps the torch.distributed variant of round 1.  The collective is one all-reduce of 2 Float64 PER
EVALUATION (north_star: "RCCL all-reduce of the scalar loss"), either on a second stream behind an event so that it
overlaps the next evaluation's kernel (`overlap`) or on the compute stream (`serial`); the default `auto` times 40 steps
of each before the protocol starts and runs the faster one on this node (the overlapped form needs more host work per
step and loses on a host with few free cores; world size 1: serial 50.6 vs overlap 57.6 us).  The other placements and
`deferred` (one collective per 32 evaluations) are timed right after the K steps and printed as `modes`.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU, NPTS, MPTS, DIM = 32, 4096, 4096, 3
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 vector peak == fp32-input MFMA peak (dense)
F16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16 / bf16 MFMA peak (the 2:1-sparsity headline is never used)
PEAK_CLOCK_GHZ = 2.4
N_SIMD = 1024              # 256 CUs x 4


def kernel_source_hash():
    """sha256 over the sources the nn1 kernel is built from: ties profiles/pmc_latest.json to a binary."""
    h = hashlib.sha256()
    for f in ("chamfer.hip", "fx3d_common.h", "Makefile"):
        with open(os.path.join(ROOT, "flux3d.jl_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def ubench_loop_ceiling():
    """The issue bound of the filter loop, DERIVED from the committed micro-benchmark table (profiles/r06_ubench_overlap.txt, produced
    by tools/ubench_overlap.hip on the MI355X; VERDICT r5 #1: not a constant in this file): cycles per tile and SIMD of the MFMA alone
    and of the loop's own instruction mix at four waves per SIMD, and their ratio = the share of the loop's cycles the matrix pipe
    can be busy."""
    path = os.path.join(ROOT, "profiles", "r06_ubench_overlap.txt")
    out = {"source": "profiles/r06_ubench_overlap.txt (tools/ubench_overlap.hip, W = 4 waves per SIMD, slowest-wave column)"}
    try:
        import re
        mf = lp = None
        for ln in open(path):
            m = re.search(r"W=4 .*\(avg wave\)\s+([0-9.]+) \(slowest wave\)", ln)
            if not m:
                continue
            if ln.startswith("mfma only"):
                mf = float(m.group(1))
            elif ln.startswith("nn1 loop (fold after issue)") and "prio=1" in ln:
                lp = float(m.group(1))
        out.update({"mfma_only_cycles_per_tile": mf, "loop_cycles_per_tile": lp, "pipe_share_of_the_loop": (mf / lp) if mf and lp else None})
    except Exception as e:  # noqa: BLE001
        out["note"] = f"{type(e).__name__}: {e}"
    return out


def live_pmc_traffic(timeout_s=120):
    """HBM bytes of ONE nn1 launch at the bench workload, measured now: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE -- counters in
    their own runs, no trace domains next to them) of tools/pmc_driver.py nn1 in subprocesses, read back from the result databases, with
    the guide's gfx950 correction (FETCH_SIZE counts 128-byte requests as 64: the read side doubled; KiB units).  Any failure -- no
    rocprofv3, a refused counter, a timeout -- leaves the committed profiles/pmc_latest.json figure in place and says why."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"bytes_per_launch": None, "note": "rocprofv3 not found"}
    out = {"bytes_per_launch": None, "tool": "rocprofv3 --pmc <counter> -- python tools/pmc_driver.py nn1 (one pass per counter)"}
    vals = {}
    tmp = tempfile.mkdtemp(prefix="fx3d_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            r = subprocess.run([exe, "--pmc", ctr, "-d", d, "-o", "r", "--", sys.executable, os.path.join(ROOT, "tools", "pmc_driver.py"), "nn1",
                                "--reps", "6"], cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                out["note"] = f"{ctr} pass failed (rc {r.returncode}): {(r.stderr or r.stdout)[-300:]}"
                return out
            rows = sqlite3.connect(dbs[0]).execute(
                "select avg(value), count(*) from counters_collection where counter_name = ? and kernel_name like '%nn1_f16_kernel%'", (ctr,)).fetchall()
            if not rows or rows[0][0] is None:
                out["note"] = f"{ctr}: no nn1_f16_kernel dispatch in the counter database"
                return out
            vals[ctr] = {"avg_kib": rows[0][0], "dispatches": rows[0][1]}
        out.update({"bytes_per_launch": int(round(2 * vals["FETCH_SIZE"]["avg_kib"] * 1024 + vals["WRITE_SIZE"]["avg_kib"] * 1024)),
                    "raw": vals, "correction": "gfx950: FETCH_SIZE x 2 (128-byte requests counted as 64 bytes), WRITE_SIZE as is; KiB"})
    except Exception as e:  # noqa: BLE001
        out["note"] = f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: N ranks of this script, one per device, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* exported as torchrun would (the library's TCP rendezvous does the rest).  Fewer than N visible
    devices ends the run with a non-zero status BEFORE anything is printed: a line with n_gpus != --gpus must never exist.
    Returns the exit status (0 only when every rank ended with 0)."""
    import socket
    import subprocess
    import flux3d_jl_amd as fx
    try:
        have = fx.device_count()
    except Exception as e:  # noqa: BLE001 -- no HIP runtime / no device at all
        print(f"[bench] --gpus {n}: cannot count devices ({e})", file=sys.stderr)
        return 3
    if have < n:
        print(f"[bench] --gpus {n} but only {have} device(s) visible: refusing to run (no line is printed)", file=sys.stderr)
        return 3
    port = os.environ.get("MASTER_PORT")
    if port is None:  # two free consecutive ports: MASTER_PORT and the library's rendezvous at MASTER_PORT + 1
        for _ in range(64):
            with socket.socket() as s0:
                s0.bind(("127.0.0.1", 0))
                p = s0.getsockname()[1]
            try:
                with socket.socket() as s1:
                    s1.bind(("127.0.0.1", p + 1))
                port = str(p)
                break
            except OSError:
                continue
        else:
            port = "29533"
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=port,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    deadline = time.time() + float(os.environ.get("FX3D_BENCH_LAUNCH_TIMEOUT", "1800"))
    while any(pr.poll() is None for pr in procs):
        failed = any(pr.poll() not in (None, 0) for pr in procs)
        if failed or time.time() > deadline:
            for pr in procs:        # a rank died (its peers wait in a collective) or the job hangs: end the exact
                if pr.poll() is None:   # children we started, never a pattern
                    pr.kill()
            break
        time.sleep(0.05)
    rcs = [pr.wait() for pr in procs]
    if any(rcs):
        print(f"[bench] rank exit codes {rcs}", file=sys.stderr)
        return 1
    return 0


def single_gpu_rate(fx, x, y, steps=200, burn_ms=80.0):
    """Pairs per second of the plain single-GPU step on the current device (burn-in, 20 warm-up, `steps` timed launches): the
    denominator of scaling_efficiency, measured in the same run, on the same box, before a communicator exists."""
    import numpy as np
    s = fx.Stream.create()
    loss_dev = fx.DeviceArray.empty((1,), np.float32)
    with fx.stream(s):
        for _ in range(int(burn_ms / 0.045) + 21):
            fx.chamfer_distance(x, y, loss_out=loss_dev, sync=False)
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fx.chamfer_distance(x, y, loss_out=loss_dev, sync=False)
        s.synchronize()
        t1 = time.perf_counter()
    B, N, M = int(x.shape[2]), int(x.shape[1]), int(y.shape[1])
    return {"value": B * N * M * steps / (t1 - t0), "ms_per_step": (t1 - t0) * 1e3 / steps, "steps": steps}


def run_inproc(args, why, value_1=None):
    """`--launcher inproc`: ONE process, N devices (fx3d_comm_init_all = ncclCommInitAll + a worker thread, a stream and the sharded
    evaluation's scratch per device; fx3d_chamfer_fwd_multi = kernel -> all-reduce(sum) of 2 Float64 over RCCL -> finalisation with the
    global batch size on every device).  The same protocol as the process-per-GPU path: burn-in, W warm-up, EXACTLY K steps between two
    device-wide synchronisations of ALL devices, one JSON line.  Returns the exit status."""
    import numpy as np
    import ctypes as C
    import flux3d_jl_amd as fx
    from flux3d_jl_amd import _lib
    from flux3d_jl_amd.distributed import MultiDevice
    n = args.gpus
    try:
        have = fx.device_count()
    except Exception as e:  # noqa: BLE001
        print(f"[bench] --launcher inproc: cannot count devices ({e})", file=sys.stderr)
        return 3
    if have < n:
        print(f"[bench] --gpus {n} but only {have} device(s) visible: refusing to run (no line is printed)", file=sys.stderr)
        return 3
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ids = [fx.device_identity(d) for d in range(n)]
    if len(set(ids)) != n:
        print(f"[bench] {n} devices but {len(set(ids))} distinct identities: {ids}", file=sys.stderr)
        return 1
    fx.set_device(0)
    if value_1 is None:
        x0 = fx.gpu(fx.synth.uniform_cloud(fx.synth.SEED_A, DIM, NPTS, B_PER_GPU))
        y0 = fx.gpu(fx.synth.uniform_cloud(fx.synth.SEED_B, DIM, MPTS, B_PER_GPU))
        value_1 = single_gpu_rate(fx, x0, y0)
        del x0, y0
    md = MultiDevice(devices=list(range(n)))
    info = md.info()
    if info["ndev"] != n:
        print(f"[bench] the in-process communicators span {info['ndev']} devices, --gpus {n}", file=sys.stderr)
        return 1
    Bg = B_PER_GPU * n
    xs = [md.shard(fx.synth.uniform_cloud(fx.synth.SEED_A, DIM, NPTS, B_PER_GPU, batch_offset=d * B_PER_GPU), d) for d in range(n)]
    ys = [md.shard(fx.synth.uniform_cloud(fx.synth.SEED_B, DIM, MPTS, B_PER_GPU, batch_offset=d * B_PER_GPU), d) for d in range(n)]
    loss = float(md.chamfer_distance(xs, ys, Bg))   # (blocks: the first evaluation, code objects loaded on every device)
    for _ in range(int(args.burn_ms / 0.045) + 1 + args.warmup):
        md.enqueue(xs, ys, Bg)
    md.synchronize()
    _lib.call("fx3d_profile_enable", 5)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        md.enqueue(xs, ys, Bg)
    md.synchronize()
    t1 = time.perf_counter()
    _lib.call("fx3d_profile_enable", 0)
    avg, mn, mx, cnt = C.c_double(0), C.c_double(0), C.c_double(0), C.c_int64(0)
    _lib.call("fx3d_profile_kernel_stats", b"nn1", C.byref(avg), C.byref(mn), C.byref(mx), C.byref(cnt))
    elapsed = t1 - t0
    value = Bg * NPTS * MPTS * args.steps / elapsed
    kern_s = avg.value * 1e-3 if cnt.value else None
    hw_flops = 2.0 * 16 * 1024 * (2.0 * B_PER_GPU * NPTS * MPTS / 1024.0)
    out = {
        "metric": "chamfer_point_pairs_per_sec", "value": value, "unit": "pairs/s", "n_gpus": n, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"chamfer_distance fwd B={Bg} ({B_PER_GPU}/GPU) N=M={NPTS} D=3 Float32 U[0,1)^3 (BASELINE configs[1]; configs[4] shape at 8 GPUs)",
                   "global_batch": Bg, "points": NPTS,
                   "parallelism": f"batch-sharded x{n}, one all-reduce(sum) of 2 Float64 per evaluation over RCCL (fx3d_chamfer_fwd_multi)"},
        "loss": loss,
        "launcher": "inproc (one process, a worker thread per device, ncclCommInitAll)" + (f" -- FALLBACK: {why}" if why else ""),
        "value_1gpu_same_run": value_1, "scaling_efficiency": value / (n * value_1["value"]),
        "roofline": {"bound": "mfma", "achieved": (hw_flops / kern_s / 1e12) if kern_s else None, "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": (hw_flops / kern_s / 1e12 / F16_MFMA_PEAK_TFLOPS) if kern_s else None, "traffic": None,
                     "kernel_avg_ms": avg.value if cnt.value else None, "launches_timed": cnt.value,
                     "note": "per device: the algorithmic f16 flops of one rank's launch / the kernel's average duration over all devices' "
                             "profiled launches (the single-GPU line carries the counters)"},
        "comm": {"nranks": n, "rccl_version": info["rccl_version"], "backend": "fx3d_comm_init_all (ncclCommInitAll) + fx3d_chamfer_fwd_multi",
                 "devices": [{"index": d, "pci_bus_id": ids[d][0], "device_uuid": ids[d][1]} for d in range(n)], "distinct_devices": len(set(ids))},
    }
    try:
        C.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    print(json.dumps(out), flush=True)
    md.close()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the protocol / configs measurements after the timed steps")
    ap.add_argument("--burn-ms", type=float, default=80.0,
                    help="milliseconds of the same launches before the counted warm-up (steady clocks); 0 = none")
    ap.add_argument("--comm", choices=["torch", "native"], default="native",
                    help="multi-GPU: the library's own RCCL communicator for data AND control plane (default, no torch), or "
                         "torch.distributed for both")
    ap.add_argument("--mode", choices=["auto", "overlap", "serial", "deferred"], default="auto",
                    help="multi-GPU: where the per-evaluation all-reduce of the two Float64 sums runs -- `overlap`: on a second "
                         "stream behind an event (hides the collective's latency behind the next evaluation's kernel, costs "
                         "two cross-stream events per step and more host work), `serial`: on the compute stream; `auto` "
                         "(default) times 40 untimed steps of each before the protocol starts and takes the faster one on "
                         "this node; `deferred`: one collective per --allreduce-every evaluations (opt-in: an eval loop's shape)")
    ap.add_argument("--allreduce-every", type=int, default=32, help="evaluations per collective in mode `deferred`")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the multi-GPU code path (RCCL communicator, collectives) even at world size 1")
    ap.add_argument("--launcher", choices=["procs", "inproc"], default="procs",
                    help="multi-GPU: `procs` = one process per GPU (torchrun, or this script spawning N ranks), the library's "
                         "RCCL communicator bootstrapped over TCP; `inproc` = ONE process driving the N devices through "
                         "fx3d_comm_init_all / fx3d_chamfer_fwd_multi (ncclCommInitAll, a worker thread per device) -- also the "
                         "FALLBACK rank 0 takes by itself when the process-per-GPU bootstrap fails (the line then says so)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit(f"[bench] --gpus {args.gpus}: need >= 1")
    if args.launcher == "inproc" and (args.gpus > 1 or args.force_dist):
        if int(os.environ.get("RANK", "0")) != 0:   # under torchrun: rank 0 drives every device, the other ranks have nothing to do
            return
        raise SystemExit(run_inproc(args, None))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started as a plain `python bench.py --gpus N` (no torchrun): this process becomes the launcher of N ranks
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:   # never print a line whose n_gpus differs from --gpus
        raise SystemExit(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}")

    import numpy as np
    dist = None
    torch = None
    use_dist = world > 1 or args.force_dist
    use_torch = use_dist and args.comm == "torch"
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
    if use_torch:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # nccl == RCCL on ROCm

    import flux3d_jl_amd as fx
    from flux3d_jl_amd import _lib
    from flux3d_jl_amd.distributed import (DeferredShardedChamfer, NativeComm, NativeShardedChamfer, ShardedChamfer,
                                           default_rendezvous)
    import ctypes as C

    fx.set_device(local_rank)
    Bg = B_PER_GPU * world
    # this rank's contiguous slab of the global synthetic batch (documented SplitMix64 stream)
    x = fx.gpu(fx.synth.uniform_cloud(fx.synth.SEED_A, DIM, NPTS, B_PER_GPU, batch_offset=rank * B_PER_GPU))
    y = fx.gpu(fx.synth.uniform_cloud(fx.synth.SEED_B, DIM, MPTS, B_PER_GPU, batch_offset=rank * B_PER_GPU))

    # the single-GPU rate of THIS run, on rank 0, before any communicator exists: scaling_efficiency = value_N / (N value_1)
    value_1 = None
    if use_dist and rank == 0:
        value_1 = single_gpu_rate(fx, x, y)
    native_comm = None
    comm_info = None
    if use_dist and not use_torch:
        # the library's own RCCL communicator, bootstrapped without torch: data plane AND control plane.  No fallback: a
        # rank that cannot join, or a communicator of the wrong size, ends the run (the driver must not record a
        # single-GPU number as an N-GPU one).
        rdv = default_rendezvous()
        try:
            native_comm = NativeComm(rank, world, rendezvous=rdv)
        except Exception as e:  # noqa: BLE001 -- the bootstrap is bounded (120 s) and collective: it fails on every rank or on none
            if world == 1:
                raise
            if rank != 0:
                print(f"[bench] rank {rank}: process-per-GPU bootstrap failed ({e}); rank 0 takes the in-process launcher", file=sys.stderr)
                return
            del x, y
            raise SystemExit(run_inproc(args, f"process-per-GPU bootstrap failed on rank 0: {e}", value_1=value_1))
        comm_info = dict(native_comm.info(), backend="fx3d_comm (RCCL behind the C ABI): data + control plane, no torch",
                         bootstrap=rdv.split(":", 1)[0])
        if comm_info["nranks"] != args.gpus or comm_info["nranks"] != world:
            raise SystemExit(f"[bench] communicator has {comm_info['nranks']} ranks, --gpus {args.gpus}, WORLD_SIZE {world}")
    if use_torch:
        v = C.c_int32(0)
        _lib.load().fx3d_comm_info(None, None, None, C.byref(v))
        comm_info = {"nranks": dist.get_world_size(), "rank": dist.get_rank(), "rccl_version": v.value,
                     "backend": "torch.distributed nccl (RCCL)", "bootstrap": "torch"}
        if comm_info["nranks"] != args.gpus:
            raise SystemExit(f"[bench] process group has {comm_info['nranks']} ranks, --gpus {args.gpus}")

    def make_runner(mode):
        """(step, sync_all, read_loss, stream, description) for one way of running a step."""
        if not use_dist or mode == "single_plain":
            s = fx.Stream.create()
            loss_dev = fx.DeviceArray.empty((1,), np.float32)

            def step():  # chamfer_distance(A, B) forward: ONE launch (the last block finalises the loss)
                with fx.stream(s):
                    fx.chamfer_distance(x, y, loss_out=loss_dev, sync=False)

            def read_loss():
                with fx.stream(s):
                    return float(loss_dev.item())
            return step, s.synchronize, read_loss, s, "single GPU, no collective"
        if native_comm is None:  # torch data plane
            if mode == "deferred":
                s = fx.Stream(torch.cuda.current_stream().cuda_stream)
                sh = DeferredShardedChamfer(comm=None, group=max(1, args.allreduce_every))

                def step():
                    with fx.stream(s):
                        sh(x, y, Bg)

                def sync_all():
                    with fx.stream(s):
                        sh.flush()
                    torch.cuda.synchronize()

                def read_loss():
                    with fx.stream(s):
                        return float(sh.losses.to_host()[max(sh.last_count - 1, 0)])
                return step, sync_all, read_loss, s, f"torch all_reduce of {2 * sh.group} f64 per {sh.group} steps"
            sh = ShardedChamfer(overlap=(mode == "overlap"))
            s = fx.Stream(torch.cuda.current_stream().cuda_stream)

            def sync_all():
                sh.synchronize()
                torch.cuda.synchronize()
            return (lambda: sh(x, y, Bg, sync=False)), sync_all, (lambda: (sh.synchronize(), float(sh.loss.item()))[1]), s, \
                "torch all_reduce of 2 f64 per step" + (" on a side stream" if mode == "overlap" else "")
        s = fx.Stream.create()
        if mode == "deferred":
            sh = DeferredShardedChamfer(comm=native_comm, group=max(1, args.allreduce_every))

            def step():
                with fx.stream(s):
                    sh(x, y, Bg)

            def sync_all():
                with fx.stream(s):
                    sh.flush()
                s.synchronize()

            def read_loss():
                with fx.stream(s):
                    return float(sh.losses.to_host()[max(sh.last_count - 1, 0)])
            return step, sync_all, read_loss, s, f"fx3d_comm all-reduce of {2 * sh.group} f64 per {sh.group} steps"
        sh = NativeShardedChamfer(native_comm, overlap=(mode == "overlap"))

        def step():
            with fx.stream(s):
                sh(x, y, Bg, sync=False)

        def sync_all():
            sh.synchronize()
            s.synchronize()

        def read_loss():
            with fx.stream(s):
                return float(sh.result())
        return step, sync_all, read_loss, s, "fx3d_comm all-reduce of 2 f64 per step" + \
            (" on a second stream, overlapping the next step's kernel" if mode == "overlap" else " on the compute stream")

    def barrier():
        fx.synchronize()
        if native_comm is not None:
            native_comm.barrier()           # 8-byte all-reduce(max) on the default stream + read-back
        elif dist is not None:
            dist.barrier()
        fx.synchronize()

    def max_over_ranks(v):
        if native_comm is not None:
            return native_comm.max_over_ranks(v)
        if dist is not None:
            t = torch.tensor([v], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return v

    def run_timed(runner, steps, warmup, profile_every=0):
        step, sync_all, read_loss, s, desc = runner
        for _ in range(warmup):
            step()
        sync_all()
        barrier()
        _lib.call("fx3d_profile_enable", profile_every)
        e0, e1 = fx.Event(), fx.Event()
        sync_all()
        barrier()
        t0 = time.perf_counter()
        e0.record(s)
        for _ in range(steps):
            step()
        e1.record(s)
        sync_all()
        barrier()
        t1 = time.perf_counter()
        _lib.call("fx3d_profile_enable", 0)
        elapsed = max_over_ranks(t1 - t0)
        return {"elapsed": elapsed, "t_local": t1 - t0, "loss": read_loss(), "event_ms": e0.elapsed_ms(e1), "desc": desc,
                "runner": (step, sync_all, s)}

    def timed(mode, steps, warmup, profile_every=0, burn_ms=0.0, cold=False):
        runner = make_runner(mode)
        out = {}
        if cold:  # the first K steps this process ever runs (after ONE launch that loads the code object): idle-device clocks
            runner[0]()
            runner[1]()
            out["cold_ms_per_step"] = run_timed(runner, steps, 0)["elapsed"] * 1e3 / steps
        if burn_ms > 0:
            n = int(burn_ms / 0.045) + 1  # a step is >= 45 us of GPU work: >= burn_ms in total
            for _ in range(n):
                runner[0]()
            runner[1]()
            out["burn_steps"] = n
        out.update(run_timed(runner, steps, warmup, profile_every))
        return out

    # HIP events around every 5th nn1 launch, on the launch's own stream (bracketing every launch
    # costs ~5 us per step in event records; measured)
    main_mode = args.mode if use_dist else "single"
    calibration = None
    if main_mode == "auto":  # both per-evaluation placements, 40 steps each after 10 of warm-up, max over ranks: outside the protocol
        calibration = {m: run_timed(make_runner(m), 40, 10)["elapsed"] * 1e3 / 40 for m in ("overlap", "serial")}
        main_mode = min(calibration, key=calibration.get)   # (max-over-ranks times: every rank picks the same one)
    res = timed(main_mode, args.steps, args.warmup, 0 if os.environ.get("FX3D_BENCH_NOPROFILE") else 5,
                burn_ms=args.burn_ms, cold=True)
    avg, mn, mx, cnt = C.c_double(0), C.c_double(0), C.c_double(0), C.c_int64(0)
    _lib.call("fx3d_profile_kernel_stats", b"nn1", C.byref(avg), C.byref(mn), C.byref(mx), C.byref(cnt))
    _lib.call("fx3d_profile_enable", 0)
    step, sync_all, bench_stream = res["runner"]
    if cnt.value < 50 and not os.environ.get("FX3D_BENCH_NOPROFILE"):
        # short runs (--steps 20): the kernel average still comes from >= 50 launches, timed outside the K steps
        _lib.call("fx3d_profile_enable", 1)
        for _ in range(60):
            step()
        sync_all()
        _lib.call("fx3d_profile_kernel_stats", b"nn1", C.byref(avg), C.byref(mn), C.byref(mx), C.byref(cnt))
        _lib.call("fx3d_profile_enable", 0)

    modes = None
    if use_dist:  # the other ways of placing the collective, same steps, right after (every rank takes part)
        modes = {main_mode: {"ms_per_step": res["elapsed"] * 1e3 / args.steps, "collective": res["desc"]}}
        if world == 1:  # --force-dist on one device: the plain single-launch loss of the same data is the reference every mode must equal
            plain = timed("single_plain", 2, 1)
            modes[main_mode]["loss_equal"] = bool(np.float32(plain["loss"]) == np.float32(res["loss"]))
            modes["plain_loss"] = float(plain["loss"])
        if calibration is not None:
            modes["auto_calibration_ms_per_step"] = calibration
        for m in ("overlap", "serial", "deferred"):
            if m != main_mode:
                r2 = timed(m, args.steps, min(args.warmup, 20))
                modes[m] = {"ms_per_step": r2["elapsed"] * 1e3 / args.steps, "collective": r2["desc"],
                            "loss_equal": bool(np.float32(r2["loss"]) == np.float32(res["loss"]))}

    # ---- who ran where: every rank's device identity, its own kernel average and its own time per step,
    #      gathered over the control plane; two ranks on one physical device end the run (an N-rank line must be N devices)
    if use_dist:
        pci, uuid_hex = fx.device_identity(local_rank)
        dom, bus, devfn = pci.split(":")
        dv, fnn = devfn.split(".")
        u32 = [int(uuid_hex[8 * i:8 * i + 8], 16) for i in range(4)]
        local_ms = (res["t_local"] * 1e3 / args.steps) if "t_local" in res else float("nan")
        rec = [float(int(dom, 16)), float(int(bus, 16)), float(int(dv, 16)), float(int(fnn, 16))] + [float(v) for v in u32] + \
              [avg.value, local_ms, float(comm_info["nranks"]), float(local_rank), float(res["event_ms"] / args.steps)]
        if native_comm is not None:
            table = native_comm.allgather_f64(rec)
        else:
            t = torch.tensor(rec, dtype=torch.float64, device="cuda")
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t)
            table = np.stack([q.cpu().numpy() for q in parts])
        ranks = []
        for r in range(world):
            row = table[r]
            ranks.append({"rank": r, "pci_bus_id": "%04x:%02x:%02x.%x" % tuple(int(v) for v in row[:4]),
                          "device_uuid": "".join("%08x" % int(v) for v in row[4:8]), "kernel_avg_ms": row[8], "ms_per_step_local": row[9],
                          "comm_nranks": int(row[10]), "local_device_index": int(row[11]), "stream_event_ms_per_step": row[12]})
        ids = [(q["pci_bus_id"], q["device_uuid"]) for q in ranks]
        if len(set(ids)) != world:
            raise SystemExit(f"[bench] {world} ranks on {len(set(ids))} distinct devices: {ids}")
        if any(q["comm_nranks"] != world for q in ranks):
            raise SystemExit(f"[bench] a rank's communicator does not span {world} ranks: {[q['comm_nranks'] for q in ranks]}")
        comm_info["ranks"] = ranks
        comm_info["distinct_devices"] = len(set(ids))

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    elapsed, loss = res["elapsed"], res["loss"]
    pairs_per_step = Bg * NPTS * MPTS
    value = pairs_per_step * args.steps / elapsed
    ms_per_step = elapsed * 1e3 / args.steps
    kern_s = avg.value * 1e-3
    # algorithmic work of ONE nn1 launch on one GPU:
    #   flops: 8 per ordered pair evaluation (3 sub, 3 mul, 2 add), both directions = 16*B*N*M
    #   bytes: read both clouds once (4*D*B*(N+M)) + the per-block partial sums written
    flops = 16.0 * B_PER_GPU * NPTS * MPTS
    abytes = 4.0 * DIM * B_PER_GPU * (NPTS + MPTS) + 8.0 * 2 * B_PER_GPU * 8
    n_mfma = 2.0 * B_PER_GPU * NPTS * MPTS / 1024.0            # ALL 32 x 32 pair tiles of both directions: one v_mfma_f32_32x32x16_f16 each
    hw_flops = 2.0 * 16 * 1024 * n_mfma                          # algorithmic f16 flops of the launch (MFMA clause: K = 16)
    hw_tf = hw_flops / kern_s / 1e12 if kern_s else None
    ub = ubench_loop_ceiling()
    roof = {
        "bound": "mfma", "achieved": hw_tf, "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": (hw_tf / F16_MFMA_PEAK_TFLOPS) if kern_s else None, "traffic": None,
        "mfma_pipe_frac": None, "executed_f16_flops_per_launch": None,
        "algorithmic_fp32_over_valu_peak": (flops / kern_s / 1e12 / FP32_PEAK_TFLOPS) if kern_s else None,
        "algorithmic_fp32_tflops": flops / kern_s / 1e12 if kern_s else None, "fp32_valu_peak": FP32_PEAK_TFLOPS,
        "achieved_hbm": abytes / kern_s / 1e9 if kern_s else None, "hbm_peak": HBM_PEAK_GBS,
        "hbm_frac": (abytes / kern_s / 1e9 / HBM_PEAK_GBS) if kern_s else None,
        "algorithmic_f16_flops_per_launch": hw_flops,
        "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": abytes,
        "kernel": "nn1_f16_kernel<false, false, true> (spatial pruning)", "kernel_avg_ms": avg.value, "kernel_min_ms": mn.value, "launches_timed": cnt.value,
        "main_loop_issue_bound": ub,
        "note": "bound = the f16 matrix pipe.  achieved = the ALGORITHMIC f16 flops of the launch (SURVEY.md 8(d), MFMA clause: "
                "2*(2*B*N*M)*K with K = 16, i.e. one v_mfma_f32_32x32x16_f16 per 32 x 32 pairs and direction) / the kernel's average "
                "duration in this run (HIP events on its stream); peak = 2500 TF dense f16; frac = achieved / peak.  The kernel does "
                "NOT execute all of them since round 6: candidates lie in the LDS image along a Hilbert curve through an 8^3 grid, queries "
                "are taken in the same order, and a wave runs the filter only over the lane tiles whose bounding box can hold a nearest "
                "neighbour of one of its 32 queries (22 % of them on uniform clouds; results bit-identical) -- executed_f16_flops_per_launch "
                "and mfma_pipe_frac (busy cycles per kernel cycle at 2.4 GHz) are what the matrix cores really do (PMC, "
                "profiles/pmc_latest.json).  Why skipping and not a better schedule: main_loop_issue_bound -- tools/ubench_overlap.hip "
                "(ISA-checked, profiles/r06_ubench_overlap.txt) shows the filter loop at the SIMD's VALU-issue bound for its "
                "instruction mix (4.25 cycles per VALU + ~13 per MFMA issue; 10.7 VALU per MFMA in the table's loop => 58.8 cycles per tile, "
                "the pipe busy 0.55 of the loop; the shipped loop tracks minima per 32-candidate block: 13.2 VALU per MFMA) under every "
                "schedule tried.  The whole kernel is VALU-issue bound (3 490 VALU per wave, SQ_INSTS_VALU): DESIGN.md 3.1.  traffic: 20.5 MB "
                "of it are the blocks' scratch rows of the pruning (written once, read back from L2), not re-reads of the clouds.  "
                "algorithmic_fp32_over_valu_peak is a NAMED EXTRA, not the fraction of the bound: the reference's exact Float32 "
                "form (16 flop per pair, both directions) / time against the fp32 vector peak; it exceeds 1 because the kernel "
                "does not execute those flops.  achieved_hbm / hbm_frac: algorithmic bytes / time against 8 TB/s (BASELINE.json "
                "asks; an exact all-pairs method cannot approach it: 70 % would be 0.56 us)"}
    for k in ("frac", "hbm_frac"):
        assert roof[k] is None or roof[k] <= 1.0, (k, roof[k])   # no field called *frac may exceed 1
    issue = None
    pmc_note = "no profiles/pmc_latest.json"
    try:  # PMC-derived numbers per launch, collected by separate rocprofv3 --pmc passes (tools/profile_round.sh)
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as fh:
            pmc = json.load(fh)
        stale = pmc.get("kernel_source_sha16") != kernel_source_hash()
        valu, mfma = float(pmc["SQ_INSTS_VALU"]), float(pmc["SQ_INSTS_MFMA"])
        roof.update({"traffic": pmc.get("nn1_hbm_bytes_per_launch"), "valu_per_mfma": (valu - mfma) / mfma,
                     "counters_from": pmc.get("source"), "counters_stale": stale,
                     "executed_f16_flops_per_launch": mfma * 32768.0, "executed_share_of_algorithmic": mfma * 32768.0 / hw_flops,
                     "mfma_pipe_frac": (float(pmc["SQ_VALU_MFMA_BUSY_CYCLES"]) / N_SIMD / kern_s / 1e9 / PEAK_CLOCK_GHZ) if kern_s else None})
        assert roof["mfma_pipe_frac"] is None or roof["mfma_pipe_frac"] <= 1.0
        cyc = ((valu - mfma) * 4.0 + float(pmc["SQ_VALU_MFMA_BUSY_CYCLES"])) / N_SIMD
        issue = {"value": (cyc / kern_s / 1e9 / PEAK_CLOCK_GHZ) if kern_s else None, "unit": "SIMD issue cycles per kernel cycle at 2.4 GHz",
                 "counters_stale": stale,
                 "note": "NOT a roofline fraction (VERDICT r2): ((SQ_INSTS_VALU - SQ_INSTS_MFMA) x 4 + MFMA busy cycles) / 1024 SIMDs "
                         "/ kernel time.  An occupancy of the issue ports that goes UP when the kernel issues more instructions for "
                         "the same work; kept because it shows how little idle time is left (the kernel is issue bound: every MFMA takes "
                         "~13 cycles out of its SIMD's VALU issue stream, tools/ubench_overlap.hip)"}
        pmc_note = None
    except Exception as e:  # noqa: BLE001
        pmc_note = f"profiles/pmc_latest.json unusable: {e}"
    if pmc_note:
        roof["counters_note"] = pmc_note
    if world == 1 and not use_dist and not args.no_extras and not os.environ.get("FX3D_BENCH_NO_LIVE_PMC"):
        live = live_pmc_traffic()   # (round 5) the HBM traffic of one launch measured IN THIS RUN, not read from a committed file
        if live.get("bytes_per_launch"):
            roof["traffic_committed_profile"] = roof.get("traffic")
            roof["traffic"] = live["bytes_per_launch"]
        roof["traffic_live"] = live
    pipe = None
    out = {
        "metric": "chamfer_point_pairs_per_sec", "value": value, "unit": "pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"chamfer_distance fwd B={Bg} ({B_PER_GPU}/GPU) N=M={NPTS} D=3 Float32 U[0,1)^3 (BASELINE configs[1]; configs[4] shape at 8 GPUs)",
                   "global_batch": Bg, "points": NPTS,
                   "parallelism": f"batch-sharded x{world}, " + (res["desc"] if use_dist else "no collective")},
        "loss": loss,
        "roofline": roof,
        "issue_occupancy": issue,
        "cold_ms_per_step": res.get("cold_ms_per_step"), "burn_in_steps": res.get("burn_steps"),
        "stream_event_ms_per_step": res["event_ms"] / args.steps,
    }
    out["launcher"] = "procs (one process per GPU)" if use_dist else "single process, one GPU"
    if value_1 is not None:
        out["value_1gpu_same_run"] = value_1
        out["scaling_efficiency"] = value / (world * value_1["value"])
    if comm_info is not None:
        out["comm"] = comm_info
    if modes is not None:
        out["modes"] = modes
    if world == 1 and not use_dist and not args.no_extras:
        out["protocol"] = protocol_numbers(fx, x, y)
        out["configs"] = config_one_liners(fx)
        out["reference_harness"] = reference_harness(fx)
        out["hbm_bound_kernels"] = hbm_bound_kernels()
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(fx)
    try:  # C stdio of the loaded libraries first (RCCL prints its NCCL_DEBUG=VERSION banner there): the JSON goes last
        C.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def _per_call_ms(fx, fn, n=100, warm=5):
    """min / median of n single calls, each between two HIP events on one stream (benchmarks/metrics.jl:24-31: warm-up,
    many samples, minimum -- plus the median SURVEY 8d asks for).  Calls are enqueued back to back."""
    import numpy as np
    s = fx.Stream.create()
    with fx.stream(s):
        for _ in range(warm):
            fn()
        s.synchronize()
        # burn-in: >= 15 ms of the same calls.  A measurement that follows host work (uploads, topology builds) starts on a
        # device that has dropped its clocks; the first of two otherwise identical measurements read 6-7 us high for calls
        # of 7-15 us (round 4: the harness's forward column, and with it `back` = total - forward, were off by that much)
        t_burn = time.perf_counter()
        while time.perf_counter() - t_burn < 0.015:
            for _ in range(8):
                fn()
        s.synchronize()
        ev = [fx.Event() for _ in range(n + 1)]
        ev[0].record(s)
        for i in range(n):
            fn()
            ev[i + 1].record(s)
        s.synchronize()
        ts = np.array([ev[i].elapsed_ms(ev[i + 1]) for i in range(n)])
    return {"min_ms": float(ts.min()), "median_ms": float(np.median(ts)), "samples": n}


def protocol_numbers(fx, x, y):
    """SURVEY 8(d) / benchmarks/metrics.jl:24-38: forward, and forward + backward (loss with indices, then the adjoint
    w.r.t. both clouds), min and median over 100 event-bracketed calls at the headline config."""
    import numpy as np
    loss_dev = fx.DeviceArray.empty((1,), np.float32)
    fwd = _per_call_ms(fx, lambda: fx.chamfer_distance(x, y, loss_out=loss_dev, sync=False))

    def fb():
        fx.chamfer_value_and_grad(x, y, loss_out=loss_dev, sync=False)   # one ABI call: forward with indices + adjoint
    fwdbwd = _per_call_ms(fx, fb)
    pairs = B_PER_GPU * NPTS * MPTS
    return {"forward": dict(fwd, pairs_per_s_at_min=pairs / (fwd["min_ms"] * 1e-3)),
            "forward_backward": dict(fwdbwd, pairs_per_s_at_min=pairs / (fwdbwd["min_ms"] * 1e-3)),
            "note": "single calls between HIP events on one stream (event records included); `value` above is K back-to-back steps"}


def _roof(ms_min, flops=None, nbytes=None):
    """SURVEY.md 8(d) per config: algorithmic flops (and / or bytes) per call / the call's minimum time, against the fp32
    vector peak (a named ratio that may exceed 1 for the filtered kernels, never called `frac`) and / or the 8 TB/s HBM peak."""
    t = ms_min * 1e-3
    out = {}
    if flops is not None:
        out.update({"algorithmic_flops": flops, "achieved_tflops": flops / t / 1e12, "algorithmic_fp32_over_valu_peak": flops / t / 1e12 / FP32_PEAK_TFLOPS})
    if nbytes is not None:
        out.update({"algorithmic_bytes": nbytes, "achieved_gbs": nbytes / t / 1e9, "frac_hbm_peak": nbytes / t / 1e9 / HBM_PEAK_GBS})
    return out


def surface_clouds(fx, n=4096, B=32, seed=0x5EED0C2, normalise=True):
    """C2's shape (B = 32, N = M = 4096) on SURFACE samples of real meshes instead of U[0,1)^3: cloud b of A is sampled from
    mesh b mod 10 of {teapot, sphere, the 8 ModelNet OFF files of tests/golden/modelnet}, cloud b of B from mesh (b + 3) mod 10
    -- points on 2-D manifolds, the geometry chamfer_distance sees in fit_mesh / ModelNet evaluation.  ``normalise``: every
    mesh scaled to the unit sphere first (an evaluation pipeline's preprocessing); False keeps the files' own units, which
    pairs clouds whose extents differ by up to 250 x (teapot: 3, one ModelNet table: 825)."""
    import shutil
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import modelnet_chamfer_eval as ev
    tmp = tempfile.mkdtemp(prefix="fx3d_bench_mn_")
    try:
        for z in ("ModelNet10.zip", "ModelNet40.zip"):
            shutil.copy(os.path.join(ROOT, "tests", "golden", "modelnet", z), tmp)
        meshes = [(m[1], m[2]) for m in ev.listing(tmp)]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    meshes = [fx.load_obj(os.path.join(ROOT, "tests", "golden", f)) for f in ("teapot.obj", "sphere.obj")] + meshes
    if normalise:
        meshes = [(ev.unit_sphere(v), f) for v, f in meshes]
    ia = [b % len(meshes) for b in range(B)]
    ib = [(b + 3) % len(meshes) for b in range(B)]
    ta = fx.gpu(fx.TriMesh([meshes[i][0] for i in ia], [meshes[i][1] for i in ia]))
    tb = fx.gpu(fx.TriMesh([meshes[i][0] for i in ib], [meshes[i][1] for i in ib]))
    return fx.sample_points(ta, n, seed=seed), fx.sample_points(tb, n, seed=seed + 1)


def config_one_liners(fx):
    """The other BASELINE configs, one entry each: min / median over 100 single calls (ms) + the config's own SURVEY 8(d)
    roofline (algorithmic flops / bytes of the call / its minimum time)."""
    import numpy as np
    out = {}
    loss_dev = fx.DeviceArray.empty((1,), np.float32)

    def chamfer_entry(a, b, n, m, B):
        r = _per_call_ms(fx, lambda: fx.chamfer_distance(a, b, loss_out=loss_dev, sync=False))
        r["pairs_per_s_at_min"] = B * n * m / (r["min_ms"] * 1e-3)
        r["roofline"] = _roof(r["min_ms"], flops=16.0 * B * n * m, nbytes=4.0 * 3 * B * (n + m))
        return r

    a = fx.gpu(fx.synth.uniform_cloud(fx.synth.SEED_A, 3, 1024, 2))
    b = fx.gpu(fx.synth.uniform_cloud(fx.synth.SEED_B, 3, 1024, 2))
    out["C1 chamfer fwd B=2 N=M=1024"] = chamfer_entry(a, b, 1024, 1024, 2)
    # the headline's shape on numpy's default_rng(7) uniforms: this dataset has a query whose wave takes the exact fall-back, and
    # in a one-round launch that wave is the tail
    rng = np.random.default_rng(7)
    af, bf = (fx.gpu(np.asfortranarray(rng.random((3, 4096, 32)).astype(np.float32))) for _ in range(2))
    out["C2 shape, numpy default_rng(7) uniforms: one slow query (B=32 N=M=4096)"] = chamfer_entry(af, bf, 4096, 4096, 32)
    del af, bf
    sa, sb = surface_clouds(fx)
    out["C2 shape on surface-sampled clouds (teapot, sphere, 8 ModelNet OFF meshes, each normalised to the unit sphere; B=32 N=M=4096)"] = \
        chamfer_entry(sa, sb, 4096, 4096, 32)
    sa, sb = surface_clouds(fx, normalise=False)
    out["C2 shape on surface-sampled clouds, the files' own units (extent ratios up to 250 within a pair; B=32 N=M=4096)"] = \
        chamfer_entry(sa, sb, 4096, 4096, 32)
    del sa, sb
    a5 = fx.gpu(fx.synth.uniform_cloud(fx.synth.SEED_A, 3, 1024, 32))  # C5 "also report 1024" (one rank's 32-cloud shard)
    b5 = fx.gpu(fx.synth.uniform_cloud(fx.synth.SEED_B, 3, 1024, 32))
    out["C5 shard at N=M=1024: chamfer fwd B=32 per GPU"] = chamfer_entry(a5, b5, 1024, 1024, 32)
    # C3: chamfer_distance(mesh, mesh, 5000) = 2 x (areas -> probabilities -> CDF -> 5000 draws) + chamfer.  The reference redoes the
    # CDFs on every call (src/transforms/mesh_func.jl:27-47): that is the figure of record; `cdf_cached` = the same call when the mesh
    # objects still hold their CDFs (vertices unchanged since the last call, e.g. the target of a fitting loop)
    t = os.path.join(ROOT, "tests", "golden", "teapot.obj")
    m8, m8b = fx.gpu(fx.load_trimesh(*[t] * 8)), fx.gpu(fx.load_trimesh(*[t] * 8))
    V, F, n3 = m8.V, m8.F, 5000
    c3_bytes = 2 * (12.0 * V * 8 + 12.0 * F * 8 + 8.0 * F * 8 + 12.0 * n3 * 8)  # verts + faces in, Float64 CDF, samples out
    r = _per_call_ms(fx, lambda: fx.chamfer_distance(m8, m8b, n3, seed=5, loss_out=loss_dev, sync=False, reuse_cdf=False))
    r["roofline"] = _roof(r["min_ms"], flops=16.0 * 8 * n3 * n3, nbytes=c3_bytes + 4.0 * 3 * 8 * 2 * n3)
    r["cdf_cached"] = _per_call_ms(fx, lambda: fx.chamfer_distance(m8, m8b, n3, seed=5, loss_out=loss_dev, sync=False))
    out["C3 chamfer_distance(mesh, mesh, 5000) B=8 teapots, CDFs rebuilt per call (2 x (CDF + draw) + chamfer)"] = r
    # C3 as BASELINE.json names it -- the fit_mesh.jl LOOP ITERATION (examples/fit_mesh.jl:98-110; timed the `gradient(...)` way by
    # benchmarks/metrics.jl:27-32): sample 5000 points of the offset source and of the target, chamfer, 0.1 laplacian + edge loss,
    # the gradient w.r.t. the vertex offsets through every term, and the Momentum step.  (a) the tutorial's own pair -- one source
    # sphere (2562 vertices) fitted to the teapot -- eager and as the captured hipGraph (one launch per iteration);
    # (b) B = 8 teapot-class meshes (BASELINE configs[2]), eager and replayed.
    g = os.path.join(ROOT, "tests", "golden")

    def fit_entry(src, tgt, label):
        xo = fx.DeviceArray.zeros((3, int(src.dev("verts_packed").shape[1])), np.float32)
        opt = fx.Momentum(1.0, 0.9)
        it = [0]

        def eager():
            _, grad = fx.loss_dolphin(xo, src, tgt, 5000, seed=100 + 2 * it[0], with_grad=True, sync=False)
            opt.update(xo, grad)
            it[0] += 1
        e = _per_call_ms(fx, eager)

        def replayed(ordered, fold=True):
            xg = fx.DeviceArray.zeros((3, int(src.dev("verts_packed").shape[1])), np.float32)
            step = fx.FitStepGraph(xg, src, tgt, fx.Momentum(1.0, 0.9), num_samples=5000, ordered=ordered, fold=fold)
            for _ in range(20):
                step.step()
            step.synchronize()
            ev = [fx.Event() for _ in range(101)]
            ev[0].record(step.stream)
            for i in range(100):
                step.step()
                ev[i + 1].record(step.stream)
            step.synchronize()
            ts = np.array([ev[i].elapsed_ms(ev[i + 1]) for i in range(100)])
            return {"min_ms": float(ts.min()), "median_ms": float(np.median(ts)), "samples": 100}, float(step.loss.item())
        det, loss_after = replayed(True)
        unfolded, _ = replayed(True, fold=False)
        fast, _ = replayed(False)
        B, V, F = src.N, src.V, src.F
        nb = 2 * (12.0 * V * B + 12.0 * F * B + 8.0 * F * B + 12.0 * 5000 * B) + 4.0 * 3 * B * 2 * 5000 * 2 + 3 * 12.0 * V * B
        return {"what": label, "eager": e, "graph_replay": det,
                "graph_replay_unfolded": dict(unfolded, note="FitStepGraph(fold=False): the two regularisers as launches of their own (seven launches per "
                                                             "iteration); the default (graph_replay) carries them as spare blocks of the draw launch and of "
                                                             "the launch of the chamfer adjoint's rows (fx3d_mesh_reg: five launches, the same bits)"),
                "graph_replay_scatter": dict(fast, note="FitStepGraph(ordered=False): the sampling adjoint scatters with float atomics (sums in arrival order), "
                                                       "six launches; the default replay (graph_replay) gathers in a fixed order -- the gradient is the oracle's "
                                                       "bit for bit and the same on every run (sample_gather.h)"),
                "loss_after": loss_after,
                "roofline": _roof(det["min_ms"], flops=16.0 * B * 5000 * 5000, nbytes=nb)}
    tv, tf = fx.load_obj(os.path.join(g, "teapot.obj"))   # the tutorial's preprocessing of its target (examples/fit_mesh.jl:46-54):
    tv = tv - tv.mean(1, keepdims=True)                    # zero mean, scaled into the source sphere's bounding box
    tv = np.asfortranarray((tv / np.abs(tv).max()).astype(np.float32))
    out["C3 fit_mesh.jl loop iteration (loss + gradient + Momentum), sphere -> teapot, 5000 samples (examples/fit_mesh.jl:98-110)"] = \
        fit_entry(fx.gpu(fx.load_trimesh(os.path.join(g, "sphere.obj"))), fx.gpu(fx.TriMesh([tv], [tf])),
                  "one source mesh (2562 V / 5120 F) against one target (1202 V / 2256 F, centred and scaled as the tutorial does): the tutorial's loop")
    out["C3 fit_mesh loop iteration, B = 8 teapot-class meshes (BASELINE configs[2]), 5000 samples"] = \
        fit_entry(fx.gpu(fx.load_trimesh(*[t] * 8)), fx.gpu(fx.load_trimesh(*[t] * 8)), "eight source meshes against eight targets (1202 V / 2256 F each)")
    # benchmarks/triangle_mesh.jl:30-34: the two workloads that file times on the teapot
    tp = fx.gpu(fx.load_trimesh(t))
    r = _per_call_ms(fx, lambda: fx.sample_points(tp, 10000, seed=7))
    r["roofline"] = _roof(r["min_ms"], nbytes=12.0 * tp.V + 12.0 * tp.F + 8.0 * tp.F + 12.0 * 10000)
    out["triangle_mesh.jl: sample_points(teapot, 10000) (benchmarks/triangle_mesh.jl:33)"] = r
    tp2 = fx.gpu(fx.load_trimesh(t))
    r = _per_call_ms(fx, lambda: fx.chamfer_distance(tp, tp2, 10000, seed=5, loss_out=loss_dev, sync=False, reuse_cdf=False))
    r["roofline"] = _roof(r["min_ms"], flops=16.0 * 10000 * 10000, nbytes=2 * (12.0 * tp.V + 20.0 * tp.F + 12.0 * 10000) + 4.0 * 3 * 2 * 10000)
    out["triangle_mesh.jl: chamfer_distance(teapot, teapot, 10000) (benchmarks/triangle_mesh.jl:30)"] = r
    c4 = fx.gpu(fx.synth.uniform_cloud(0x5EED0004, 3, 1024, 32))
    r = _per_call_ms(fx, lambda: fx.knn(c4, 20, drop_first=True))
    r["roofline"] = _roof(r["min_ms"], flops=8.0 * 32 * 1024 * 1024, nbytes=4.0 * 3 * 1024 * 32 + 2 * 4.0 * 20 * 1024 * 32)
    out["C4 kNN k=20 self graph B=32 N=1024 D=3"] = r
    f64 = fx.gpu(np.asfortranarray(np.random.default_rng(1).standard_normal((64, 1024, 32)).astype(np.float32)))
    r = _per_call_ms(fx, lambda: fx.knn(f64, 20, drop_first=True))
    r["roofline"] = _roof(r["min_ms"], flops=3.0 * 64 * 32 * 1024 * 1024, nbytes=4.0 * 64 * 1024 * 32 + 2 * 4.0 * 20 * 1024 * 32)
    out["C4' kNN k=20 self graph B=32 N=1024 D=64 (second EdgeConv)"] = r
    # off the BASELINE shapes (round 3): k + drop in 33..64 stays on the matrix cores; one large cloud runs as candidate slices
    out["kNN k=40 self graph at C4's / C4''s shape (D=3 / D=64)"] = {
        "D3": _per_call_ms(fx, lambda: fx.knn(c4, 40, drop_first=True)), "D64": _per_call_ms(fx, lambda: fx.knn(f64, 40, drop_first=True))}
    one = fx.gpu(np.asfortranarray(np.random.default_rng(2).standard_normal((64, 8192, 1)).astype(np.float32)))
    r = _per_call_ms(fx, lambda: fx.knn(one, 20, drop_first=True))
    r["roofline"] = _roof(r["min_ms"], flops=3.0 * 64 * 8192 * 8192, nbytes=4.0 * 64 * 8192 + 2 * 4.0 * 20 * 8192)
    out["kNN k=20 self graph, ONE cloud of 8192 points, D=64 (candidate slices of fx3d_knn_ws)"] = r
    return out


def hbm_bound_kernels():
    """SURVEY.md 8(d)'s HBM-bound members of the path at a bandwidth-bound size (a 3.9 M-face sheet; the C4' graph; the chamfer
    adjoint at B = 256): one roofline object per kernel -- algorithmic bytes / the kernel's own duration (the library's HIP
    events around the launch) against 8 TB/s nominal and the 6.3 TB/s a plain copy reaches (tools/hbm_roofline.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import hbm_roofline
    rows = hbm_roofline.measure(cells=1400, reps=10)
    return {r["kernel"]: {"bound": "hbm", "achieved": r["achieved_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": r["frac_hbm_peak"],
                          "frac_of_achievable_6300": r["frac_achievable"], "algorithmic_bytes": r["algorithmic_bytes"],
                          "kernel_avg_ms": r["kernel_avg_ms"], "size": r["size"]} for r in rows}


def reference_harness(fx):
    """benchmarks/metrics.jl:17-63 in full: n = 2^6 .. 2^14; chamfer_distance on generate_pcloud(n) (p_i = (i,i,i)/n, A == B,
    B = 1), edge_loss and laplacian_loss on generate_trimesh(n) (the same points as vertices, faces (i,i,i)); forward, and
    forward + backward (the harness times `gradient`: total - forward = its "back" column).  min over 100 single calls, ms."""
    import numpy as np
    rows = {}
    loss_dev = fx.DeviceArray.empty((1,), np.float32)
    for n in (64, 256, 1024, 4096, 16384):
        p = fx.gpu(fx.synth.reference_bench_cloud(n))
        row = {}
        f = _per_call_ms(fx, lambda: fx.chamfer_distance(p, p, loss_out=loss_dev, sync=False))

        def cfb():  # `gradient(...)`: value and gradient in ONE ABI call (fx3d_chamfer_fwd_bwd; two calls left the device idle
            fx.chamfer_value_and_grad(p, p, loss_out=loss_dev, sync=False)   # between the launches: "back" 25-31 us in round 3)
        t = _per_call_ms(fx, cfb)
        row["chamfer_distance"] = {"forward_ms": f["min_ms"], "total_ms": t["min_ms"], "back_ms": t["min_ms"] - f["min_ms"],
                                   "roofline": _roof(f["min_ms"], flops=16.0 * n * n, nbytes=4.0 * 3 * 2 * n)}
        v = fx.synth.reference_bench_cloud(n)
        faces = np.asfortranarray(np.tile(np.arange(1, n + 1, dtype=np.int32), (3, 1)))
        m = fx.gpu(fx.TriMesh([v], [faces]))
        E = int(fx.get_edges_packed(m).shape[0])
        nnz = 2 * E + n
        f = _per_call_ms(fx, lambda: fx.edge_loss(m, sync=False))

        def efb():
            fx.edge_loss(m, sync=False)
            fx.edge_loss_grad(m)
        t = _per_call_ms(fx, efb)
        row["edge_loss"] = {"forward_ms": f["min_ms"], "total_ms": t["min_ms"], "back_ms": t["min_ms"] - f["min_ms"], "edges": E,
                            "roofline": _roof(f["min_ms"], nbytes=12.0 * n + 8.0 * E)}
        f = _per_call_ms(fx, lambda: fx.laplacian_loss(m, sync=False))

        def lfb():
            fx.laplacian_loss(m, sync=False)
            fx.laplacian_loss_grad(m)
        t = _per_call_ms(fx, lfb)
        row["laplacian_loss"] = {"forward_ms": f["min_ms"], "total_ms": t["min_ms"], "back_ms": t["min_ms"] - f["min_ms"], "nnz": nnz,
                                 "roofline": _roof(f["min_ms"], nbytes=12.0 * n + 8.0 * nnz + 4.0 * n)}
        rows[str(n)] = row
    return {"protocol": "benchmarks/metrics.jl:24-38: minimum over samples; here 100 single calls between HIP events per entry",
            "npoints": rows}


def cpu_baseline(fx):
    """Reference CPU algorithm (per-batch-element KD-tree build + 1-NN queries, serial;
    src/metrics/pcloud.jl:54-70) as restated in oracle/flux3d_oracle.c, 1 core, on the same
    workload.  The oracle is the checker/baseline only -- never on the product path."""
    import numpy as np
    from oracle import oracle
    x = fx.synth.uniform_cloud(fx.synth.SEED_A, DIM, NPTS, B_PER_GPU)
    y = fx.synth.uniform_cloud(fx.synth.SEED_B, DIM, MPTS, B_PER_GPU)
    oracle.chamfer_distance(x[:, :, :1], y[:, :, :1], kdtree=True)  # page in
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        oracle.chamfer_distance(x, y, kdtree=True)
        ts.append(time.perf_counter() - t0)
    best, med = min(ts), float(np.median(ts))
    # forward + backward (benchmarks/metrics.jl:28-32): KD-tree forward with indices, then the adjoint
    tfb = []
    for _ in range(3):
        t0 = time.perf_counter()
        ix, iy = oracle.nn1(x, y, kdtree=True)[:2]   # the forward's KD-tree searches (its gather + mean add < 1 %)
        oracle.chamfer_bwd(x, y, ix, iy)
        tfb.append(time.perf_counter() - t0)
    xb, yb = x[:, :, :4], y[:, :, :4]
    t0 = time.perf_counter()
    oracle.chamfer_distance(xb, yb)
    dt_bf = time.perf_counter() - t0
    pairs = B_PER_GPU * NPTS * MPTS
    oracle.nn1_allcores(x[:, :64, :1], y[:, :64, :1])  # spawn the OpenMP team outside the timed region
    dt_kd_mt = dt_bf_mt = None
    for _ in range(3):
        t0 = time.perf_counter()
        _, _, nthr = oracle.nn1_allcores(x, y, kdtree=True)
        dt = time.perf_counter() - t0
        dt_kd_mt = dt if dt_kd_mt is None else min(dt_kd_mt, dt)
        t0 = time.perf_counter()
        oracle.nn1_allcores(x, y, kdtree=False)
        dt = time.perf_counter() - t0
        dt_bf_mt = dt if dt_bf_mt is None else min(dt_bf_mt, dt)
    return {"value": pairs / best, "unit": "pairs/s", "cores": 1, "kind": "port",
            "sample": f"full workload B={B_PER_GPU} N=M={NPTS}, KD-tree 1-NN both directions + gather + mean, min of 5 ({best:.3f} s)",
            "min_ms": best * 1e3, "median_ms": med * 1e3,
            "forward_backward_min_ms": min(tfb) * 1e3, "forward_backward_median_ms": float(np.median(tfb)) * 1e3,
            "bruteforce_1core_pairs_per_s": 4 * NPTS * MPTS / dt_bf,
            "bruteforce_sample": f"B=4 slice, exact fp32 all-pairs ({dt_bf:.3f} s)",
            "allcores_threads": nthr,
            "kdtree_allcores_pairs_per_s": pairs / dt_kd_mt,
            "bruteforce_allcores_pairs_per_s": pairs / dt_bf_mt,
            "allcores_sample": f"full workload, min of 3: KD-tree {dt_kd_mt:.4f} s (64 tasks), brute force {dt_bf_mt:.4f} s",
            "allcores_note": "NN searches only, OpenMP threads = usable cores (affinity capped by the cgroup quota); the reference itself is "
                             "single-threaded (src/metrics/pcloud.jl:57-58), so `value` stays the 1-core figure"}


if __name__ == "__main__":
    main()
