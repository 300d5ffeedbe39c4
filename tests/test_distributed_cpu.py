"""N>1 path on CPU: world_size 2, gloo.  Shard bounds, all-reduce of the two partial
sums, finalisation with the GLOBAL batch size."""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT


def test_shard_bounds_cover_batch(fx):
    from flux3d_jl_amd.distributed import shard_bounds
    for B in (1, 2, 7, 32, 256, 257):
        for W in (1, 2, 4, 8):
            spans = [shard_bounds(B, W, r) for r in range(W)]
            assert sum(c for _, c in spans) == B
            pos = 0
            for s, c in spans:
                assert s == pos and c >= 0
                pos += c
            counts = [c for _, c in spans]
            assert max(counts) - min(counts) <= 1
    assert [shard_bounds(256, 8, r) for r in range(8)] == [(32 * r, 32) for r in range(8)]


def test_loss_from_sums_matches_oracle(fx, oracle):
    from flux3d_jl_amd.distributed import loss_from_sums
    x = fx.synth.uniform_cloud(1, 3, 50, 3)
    y = fx.synth.uniform_cloud(2, 3, 40, 3)
    loss, _, _, sums = oracle.chamfer_distance(x, y, 0.5, 2.0, return_all=True)
    assert loss_from_sums(sums, 50, 40, 3, 3, 0.5, 2.0) == loss
    # D = 2 keeps the hard-coded *3
    x2, y2 = fx.synth.uniform_cloud(3, 2, 30, 2), fx.synth.uniform_cloud(4, 2, 20, 2)
    l2, _, _, s2 = oracle.chamfer_distance(x2, y2, return_all=True)
    assert loss_from_sums(s2, 30, 20, 2, 2) == l2


def test_two_rank_gloo_allreduce():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29517",
           os.path.join(ROOT, "tests", "_dist_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "DIST_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
