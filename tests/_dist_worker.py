"""Worker for test_distributed_cpu.py: world_size-2 gloo run of the batch-sharded chamfer path.
The kernel cannot run without a GPU, so each rank's partial sums come from the oracle (tests may
use it as a stand-in checker); everything else -- shard bounds, the all-reduce of the two Float64
sums, the global-B finalisation -- is the product logic of flux3d.jl_amd/distributed.py."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flux3d_jl_amd as fx  # noqa: E402
from flux3d_jl_amd.distributed import loss_from_sums, shard_bounds  # noqa: E402
from oracle import oracle  # noqa: E402


def _mesh_losses(meshes):
    """oracle laplacian / edge loss of a list of (verts (3,V), faces0 (3,F)) as ONE packed batch."""
    off, vs, fs = 0, [], []
    for v, f in meshes:
        vs.append(v)
        fs.append(f + off)
        off += v.shape[1]
    v, f = np.concatenate(vs, axis=1), np.concatenate(fs, axis=1)
    edges = oracle.edges_packed(f, v.shape[1])
    rowptr, colind, vals = oracle.laplacian_csr(edges, v.shape[1])
    return (float(oracle.laplacian_loss(v, rowptr, colind, vals)), v.shape[1],
            float(oracle.edge_loss(v, edges, 0.05)), edges.shape[0])


def mesh_losses_sharded_by_mesh(rank, world):
    """laplacian_loss / edge_loss over a packed batch split by mesh -> all-reduce of
    (sum, count) -> global mean; compared with the whole batch evaluated in one piece."""
    from flux3d_jl_amd.distributed import ShardedMeshLoss
    gold = os.path.join(ROOT, "tests", "golden")
    tv, tf = fx.load_obj(os.path.join(gold, "teapot.obj"))
    sv, sf = fx.load_obj(os.path.join(gold, "sphere.obj"))
    batch = [(tv, tf.astype(np.int64) - 1), (sv, sf.astype(np.int64) - 1), (tv * np.float32(1.5), tf.astype(np.int64) - 1)]
    for nmesh in (3, 1):  # 1 mesh on 2 ranks: the idle rank contributes zeros
        start, count = shard_bounds(nmesh, world, rank)
        red = ShardedMeshLoss()
        if count > 0:
            lap, nv, edge, ne = _mesh_losses(batch[start:start + count])
        else:
            lap, nv, edge, ne = 0.0, 0, 0.0, 0
        g_lap, g_edge = red.combine(lap, nv), red.combine(edge, ne)
        f_lap, _, f_edge, _ = _mesh_losses(batch[:nmesh])
        assert np.isclose(g_lap, f_lap, rtol=1e-6, atol=0), (rank, nmesh, g_lap, f_lap)
        assert np.isclose(g_edge, f_edge, rtol=1e-6, atol=0), (rank, nmesh, g_edge, f_edge)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    for (B, N, M) in [(5, 64, 48), (1, 33, 70), (8, 100, 100)]:
        start, count = shard_bounds(B, world, rank)
        sums = torch.zeros(2, dtype=torch.float64)
        if count > 0:  # B < world: idle ranks contribute zeros
            xs = fx.synth.uniform_cloud(fx.synth.SEED_A, 3, N, count, batch_offset=start)
            ys = fx.synth.uniform_cloud(fx.synth.SEED_B, 3, M, count, batch_offset=start)
            _, _, _, s = oracle.chamfer_distance(xs, ys, return_all=True)
            sums = torch.from_numpy(np.asarray(s, dtype=np.float64).copy())
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        loss = loss_from_sums(sums.numpy(), N, M, B, 3, 0.75, 1.25)
        x = fx.synth.uniform_cloud(fx.synth.SEED_A, 3, N, B)
        y = fx.synth.uniform_cloud(fx.synth.SEED_B, 3, M, B)
        full = oracle.chamfer_distance(x, y, 0.75, 1.25)
        assert np.isclose(loss, full, rtol=1e-6, atol=0), (rank, B, loss, full)
    mesh_losses_sharded_by_mesh(rank, world)
    dist.barrier()
    if rank == 0:
        print("DIST_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
