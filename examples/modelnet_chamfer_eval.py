#!/usr/bin/env python3
"""ModelNet Chamfer evaluation on real OFF files, batch-sharded over the GPUs (BASELINE config 5).

  python examples/modelnet_chamfer_eval.py [--root DIR] [--batch 256] [--points 4096] [--mode overlap|serial|deferred]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         examples/modelnet_chamfer_eval.py ...                 (one process per GPU; torchrun only exports RANK / WORLD_SIZE /
                                                                 MASTER_*: no torch is imported here)
  python examples/modelnet_chamfer_eval.py --single-process N  (ONE process, N devices: fx3d_comm_init_all /
                                                                 fx3d_chamfer_fwd_multi, the form a Julia host uses)

Pipeline (every step on the path of SURVEY.md 8):
  ModelNet10.zip / ModelNet40.zip (the reference's test assets, tests/golden/modelnet) -> listing as
  src/datasets/modelnet/base.jl:30-108 walks it -> load_off per file (base.jl:100-101) -> TriMesh batches of 32 ->
  sample_points(4096) (src/transforms/mesh_func.jl:21-58, device sampler) -> chamfer_distance of the pair clouds
  (src/metrics/pcloud.jl:39-70), the batch sharded 32 per GPU, the two Float64 partial sums all-reduced over RCCL per
  evaluation (overlapped with the next evaluation's kernel by default), loss finalised with the global batch size.
The evaluation set: pair k = (mesh k mod n, mesh (k div n + k + 1) mod n) of the n listed meshes -- every pair a "prediction
vs ground truth" stand-in --, sampled with seeds derived from the GLOBAL chunk index, so the clouds, and therefore the
mean loss, do not depend on how many GPUs share the work.  Prints one JSON line on rank 0."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CHUNK = 32  # clouds per GPU and evaluation (config 5: 256 sharded 32 / GPU)


def unit_sphere(v):
    """Vertices centred on their mean and scaled so that the farthest one has norm 1: the usual preprocessing of a ModelNet
    evaluation (the OFF files come in whatever unit their author used: extents from 20 to 1650 in these eight)."""
    c = v - v.mean(axis=1, keepdims=True, dtype=np.float64).astype(np.float32)
    r = np.float32(np.sqrt((c.astype(np.float64) ** 2).sum(axis=0).max()))
    return np.asfortranarray(c / (r if r > 0 else np.float32(1)))


def listing(root, normalise=False):
    """[(name, verts (3,V) f32, faces (3,F) u32)] of every OFF file of both archives, train and test (base.jl:30-108)."""
    from flux3d_jl_amd import datasets
    from flux3d_jl_amd.rep import load_off
    out = []
    for variant, cats in ((10, ["sofa", "table"]), (40, ["desk", "monitor"])):
        for train in (True, False):
            d = datasets.ModelNet(root, variant, train, cats)
            for category, path in d.datapaths:
                v, f = load_off(path)
                out.append((f"MN{variant}/{category}/{'train' if train else 'test'}/{os.path.basename(path)}",
                            unit_sphere(v) if normalise else v, f))
    return out


def pair_of(k, n):
    a = k % n
    b = (k // n + k + 1) % n
    return a, (b if b != a else (b + 1) % n)


def chunk_clouds(fx, meshes, chunk_index, points, seed):
    """The (3, points, CHUNK) device clouds of global chunk `chunk_index`: pairs [CHUNK * c, CHUNK * (c + 1))."""
    ia, ib = zip(*[pair_of(k, len(meshes)) for k in range(CHUNK * chunk_index, CHUNK * (chunk_index + 1))])
    ta = fx.gpu(fx.TriMesh([meshes[i][1] for i in ia], [meshes[i][2] for i in ia]))
    tb = fx.gpu(fx.TriMesh([meshes[i][1] for i in ib], [meshes[i][2] for i in ib]))
    pa = fx.sample_points(ta, points, seed=seed + 2 * chunk_index)
    pb = fx.sample_points(tb, points, seed=seed + 2 * chunk_index + 1)
    return pa, pb


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", default=None, help="directory holding ModelNet10.zip / ModelNet40.zip (default: a temp copy of tests/golden/modelnet)")
    ap.add_argument("--batch", type=int, default=256, help="global number of cloud pairs (a multiple of 32 x world size)")
    ap.add_argument("--points", type=int, default=4096)
    ap.add_argument("--seed", type=int, default=0x5EED0C5)
    ap.add_argument("--mode", choices=["overlap", "serial", "deferred"], default="overlap")
    ap.add_argument("--single-process", type=int, default=0, metavar="NDEV", help="one process driving NDEV devices")
    ap.add_argument("--repeat", type=int, default=1, help="evaluate the whole set this many times (timing)")
    ap.add_argument("--raw", action="store_true", help="keep the files' own coordinates (default: every mesh normalised to the unit sphere)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.single_process and world > 1:
        raise SystemExit("--single-process is for one process")
    import flux3d_jl_amd as fx
    from flux3d_jl_amd.distributed import (DeferredShardedChamfer, MultiDevice, NativeComm, NativeShardedChamfer,
                                           shard_bounds)
    nshare = args.single_process or world
    if args.batch % (CHUNK * nshare):
        raise SystemExit(f"--batch must be a multiple of {CHUNK} x {nshare}")

    tmp = None
    root = args.root
    if root is None:  # the reference's archives are data fixtures of this repo; unpack them outside the tree
        import shutil
        tmp = tempfile.mkdtemp(prefix="fx3d_modelnet_")
        for z in ("ModelNet10.zip", "ModelNet40.zip"):
            shutil.copy(os.path.join(ROOT, "tests", "golden", "modelnet", z), tmp)
        root = tmp
    meshes = listing(root, normalise=not args.raw)
    nchunks = args.batch // CHUNK
    per_share = nchunks // nshare            # evaluations: evaluation e = chunk e of every rank / device
    t_io = time.perf_counter()

    losses = []
    if args.single_process:
        md = MultiDevice(ndev=args.single_process)
        clouds = []
        for d in range(md.ndev):             # device d holds chunks [d * per_share, (d + 1) * per_share)
            fx.set_device(md.devices[d])
            clouds.append([chunk_clouds(fx, meshes, d * per_share + e, args.points, args.seed) for e in range(per_share)])
        fx.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.repeat):
            losses = [float(md.chamfer_distance([clouds[d][e][0] for d in range(md.ndev)],
                                                [clouds[d][e][1] for d in range(md.ndev)], CHUNK * md.ndev))
                      for e in range(per_share)]
        dt = time.perf_counter() - t0
        info = dict(md.info(), form="one process, fx3d_chamfer_fwd_multi")
        md.close()
    else:
        fx.set_device(local_rank)
        comm = NativeComm(rank, world)       # RCCL communicator of the library (TCP rendezvous under torchrun): no torch
        cinfo = comm.info()
        if cinfo["nranks"] != world:
            raise SystemExit(f"communicator has {cinfo['nranks']} ranks, WORLD_SIZE={world}")
        first, count = shard_bounds(nchunks, world, rank)
        assert count == per_share
        clouds = [chunk_clouds(fx, meshes, first + e, args.points, args.seed) for e in range(per_share)]
        fx.synchronize()
        comm.barrier()
        s = fx.Stream.create()
        Bg = CHUNK * world
        t0 = time.perf_counter()
        with fx.stream(s):
            for _ in range(args.repeat):
                if args.mode == "deferred":
                    sh = DeferredShardedChamfer(comm=comm, group=per_share)
                    for pa, pb in clouds:
                        sh(pa, pb, Bg)
                    sh.flush()
                    losses = [float(v) for v in sh.losses.to_host()[:per_share]]
                elif args.mode == "serial":  # kernel -> all-reduce -> finalise on one stream, the loss read back per evaluation
                    sh = NativeShardedChamfer(comm)
                    losses = [float(sh(pa, pb, Bg, sync=True)) for pa, pb in clouds]
                else:                        # the collective on a second stream: one result slot per evaluation in flight
                    sh = NativeShardedChamfer(comm, overlap=True, slots=max(4, per_share))
                    outs = [sh(pa, pb, Bg, sync=False) for pa, pb in clouds]
                    sh.synchronize()
                    s.synchronize()
                    losses = [float(o.to_host()[0]) for o in outs]
        dt = comm.max_over_ranks(time.perf_counter() - t0)
        info = dict(cinfo, form=f"one process per GPU, mode {args.mode}")
    if tmp:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    if rank == 0:
        import ctypes
        try:  # C stdio of the loaded libraries first (RCCL prints its version banner there): the JSON line goes last
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        pairs = args.batch * args.points * args.points * args.repeat
        print(json.dumps({
            "eval": "ModelNet chamfer (OFF -> TriMesh -> sample_points -> sharded chamfer_distance)",
            "meshes": [(m[0], int(m[1].shape[1]), int(m[2].shape[1])) for m in meshes],
            "normalised_to_unit_sphere": not args.raw, "pairs_of_clouds": args.batch, "points": args.points, "shares": nshare, "evaluations": per_share,
            "mean_chamfer": float(np.mean(losses)), "loss_per_evaluation": losses,
            "seconds": dt, "point_pairs_per_s": pairs / dt, "comm": info,
            "load_and_sample_seconds": t_io and (t0 - t_io)}), flush=True)


if __name__ == "__main__":
    main()
