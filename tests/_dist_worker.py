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
    dist.barrier()
    if rank == 0:
        print("DIST_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
