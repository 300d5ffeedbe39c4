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


_RDV_PROC = r"""
import ctypes as C, sys
sys.path.insert(0, {root!r})
from flux3d_jl_amd import _lib
lib = _lib.load()
rank, n, rdv = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
secret = bytes((11 * i + 5) % 256 for i in range(128))
buf = (C.c_uint8 * 128)(*(secret if rank == 0 else [0] * 128))
rc = lib.fx3d_comm_exchange_id(buf, n, rank, rdv.encode())
assert rc == 0, (rank, _lib.last_error())
assert bytes(buf) == secret, rank
print("RDV_OK", rank)
"""


def _exchange_processes(rendezvous, nranks, delay_rank0=0.0):
    """The rendezvous from `nranks` separate PROCESSES (what torchrun starts), rank 0 optionally last."""
    import time
    code = _RDV_PROC.format(root=ROOT)
    procs = []
    for r in list(range(1, nranks)) + [0]:
        if r == 0 and delay_rank0:
            time.sleep(delay_rank0)
        procs.append((r, subprocess.Popen([sys.executable, "-c", code, str(r), str(nranks), rendezvous],
                                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT)))
    for r, p in procs:
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0 and f"RDV_OK {r}" in out, (r, out, err[-2000:])


def test_file_rendezvous_eight_processes_with_stale_leftovers(fx, tmp_path):
    """a crashed earlier job with the same path used to poison the next one (readers took its id at once, a
    stale marker made rank 0 unlink the id before its reader had read it).  Now: an OLD payload is refused by the
    readers, a FRESH foreign one (a job that died seconds ago) is replaced by rank 0 and its readers start over, stale
    markers / acknowledgements are removed, and nothing is left behind -- with 8 concurrent processes (config 5's size)."""
    import struct
    import time
    path = tmp_path / "fx3d_uid"
    magic = 0x3144495544335846
    # leftovers of a job that died long ago: payload older than the staleness window, a marker and an acknowledgement
    path.write_bytes(struct.pack("<QQqii", magic, 0xDEAD, int(time.time()) - 100000, 8, 0) + bytes(128))
    (tmp_path / "fx3d_uid.3").write_bytes(struct.pack("<Q", 0xDEAD))
    (tmp_path / "fx3d_uid.5.ok").write_bytes(struct.pack("<Q", 0xDEAD))
    _exchange_processes(f"file://{path}", 8, delay_rank0=0.5)
    assert list(tmp_path.iterdir()) == []
    # leftovers of a job that died a moment ago (fresh timestamp, same world size): the readers confirm it first, rank 0
    # then wipes it and publishes its own -- every rank must still end with rank 0's id
    path.write_bytes(struct.pack("<QQqii", magic, 0xBEEF, int(time.time()), 8, 0) + bytes([0xEE] * 128))
    _exchange_processes(f"file://{path}", 8, delay_rank0=1.0)
    assert list(tmp_path.iterdir()) == []


def test_file_rendezvous_refuses_symlinks(fx, tmp_path):
    """The payload is created with O_EXCL | O_NOFOLLOW: a symlink planted at the predictable temp name is replaced, never
    written through."""
    target = tmp_path / "victim"
    target.write_bytes(b"precious")
    (tmp_path / "fx3d_uid.tmp").symlink_to(target)
    _exchange(f"file://{tmp_path / 'fx3d_uid'}", nranks=2)
    assert target.read_bytes() == b"precious"


def test_tcp_rendezvous_survives_foreign_connections(fx):
    """rank 0 used to abort the bootstrap on the first connection with a bad hand-shake (a port scanner, a
    health probe) and counted a rank that connected twice as two.  Now such connections are dropped and it keeps
    accepting until every rank has been served once."""
    import socket
    import threading
    import time
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    stop = threading.Event()

    def pest():  # garbage, silence and an immediate close, again and again
        while not stop.is_set():
            for payload in (b"GET / HTTP/1.0\r\n\r\n", b"", b"\x00" * 24):
                try:
                    with socket.create_connection(("127.0.0.1", port), timeout=0.2) as c:
                        if payload:
                            c.sendall(payload)
                        time.sleep(0.02)
                except OSError:
                    pass
            time.sleep(0.01)

    t = threading.Thread(target=pest, daemon=True)
    t.start()
    try:
        _exchange(f"tcp://127.0.0.1:{port}", nranks=4)
    finally:
        stop.set()
        t.join(5)


def test_default_rendezvous_prefers_tcp_under_a_launcher(fx, monkeypatch):
    from flux3d_jl_amd.distributed import default_rendezvous
    monkeypatch.delenv("FX3D_COMM_RENDEZVOUS", raising=False)
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29500")
    assert default_rendezvous() == "tcp://127.0.0.1:29501"
    monkeypatch.delenv("MASTER_ADDR")
    assert default_rendezvous().startswith("file://")
    monkeypatch.setenv("FX3D_COMM_RENDEZVOUS", "tcp://10.0.0.1:7")
    assert default_rendezvous() == "tcp://10.0.0.1:7"


def test_tcp_rendezvous_eight_processes(fx):
    """What bench.py --gpus 8 does under torchrun before the first launch: eight separate processes meet at
    tcp://127.0.0.1:(MASTER_PORT + 1), rank 0 last (it is the slowest to come up when it also prints)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    _exchange_processes(f"tcp://127.0.0.1:{port}", 8, delay_rank0=0.7)
