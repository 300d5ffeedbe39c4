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


def _exchange(rendezvous, nranks=3):
    """fx3d_comm_exchange_id from `nranks` threads (ctypes releases the GIL): rank 0's 128 bytes reach every rank."""
    import ctypes as C
    import threading
    from flux3d_jl_amd import _lib
    lib = _lib.load()
    secret = bytes((7 * i + 3) % 256 for i in range(128))
    bufs = [(C.c_uint8 * 128)(*(secret if r == 0 else [0] * 128)) for r in range(nranks)]
    rcs = [None] * nranks

    def run(r):
        if r:  # the other ranks usually come up before rank 0 listens / publishes: they must retry
            rcs[r] = lib.fx3d_comm_exchange_id(bufs[r], nranks, r, rendezvous.encode())
        else:
            import time
            time.sleep(0.3)
            rcs[r] = lib.fx3d_comm_exchange_id(bufs[r], nranks, r, rendezvous.encode())

    ts = [threading.Thread(target=run, args=(r,)) for r in range(nranks)]
    [t.start() for t in ts]
    [t.join(60) for t in ts]
    assert rcs == [0] * nranks, (rcs, _lib.last_error())
    assert all(bytes(b) == secret for b in bufs)


def test_unique_id_rendezvous_tcp_without_torch(fx):
    """the RCCL unique id used to travel through torch.distributed; fx3d_comm_bootstrap now does the
    rendezvous itself.  The exchange half needs neither a GPU nor RCCL, so it runs here with 3 'ranks'."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    _exchange(f"tcp://127.0.0.1:{port}")


def test_unique_id_rendezvous_file_leaves_nothing_behind(fx, tmp_path):
    path = tmp_path / "fx3d_uid"
    _exchange(f"file://{path}", nranks=4)
    assert list(tmp_path.iterdir()) == []   # rank 0 removed the id and the readers' markers
    # a second job on the same path starts clean
    _exchange(f"file://{path}", nranks=2)


def test_rendezvous_argument_errors(fx):
    import ctypes as C
    from flux3d_jl_amd import _lib
    lib = _lib.load()
    buf = (C.c_uint8 * 128)()
    assert lib.fx3d_comm_exchange_id(buf, 2, 0, b"carrier-pigeon://x") == -1
    assert lib.fx3d_comm_exchange_id(buf, 2, 5, b"tcp://127.0.0.1:1") == -1
    assert lib.fx3d_comm_exchange_id(buf, 1, 0, b"tcp://127.0.0.1:1") == 0    # world size 1: nothing to exchange
    h = C.c_void_p()
    assert lib.fx3d_comm_bootstrap(C.byref(h), 0, 0, b"tcp://127.0.0.1:1") == -1
    v = C.c_int32(0)
    assert lib.fx3d_comm_info(None, None, None, C.byref(v)) == 0 and v.value > 20000   # RCCL version code, no communicator needed
