"""The fit_mesh objective of the reference's tutorial (examples/fit_mesh.jl:78-110), loss AND gradient
w.r.t. the vertex offsets, entirely on the device:

    loss_dolphin(x, src, tgt) = chamfer_distance(offset(src, x), tgt, 5000)
                                + 0.1 * laplacian_loss(offset(src, x)) + edge_loss(offset(src, x))

The reference differentiates it with Zygote; here the same chain is spelled out with the adjoint
kernels of the C ABI (chamfer_bwd -> sample_points_bwd -> padded->packed, laplacian/edge bwd)."""
import numpy as np

from . import _lib
from .device import DeviceArray, Graph, Stream, current_stream, stream
from .metrics import (MeshReg, _chamfer_points, chamfer_distance_grad, chamfer_sampled_grad, mesh_losses, mesh_losses_grad,
                      sampling_adjoint_is_ordered)
from .transforms import lincomb, offset, sample_points_grad, sample_points_pair


def loss_dolphin(x, src, tgt, num_samples=5000, seed=None, with_grad=False, w_lap=0.1, w_edge=1.0, sync=True,
                 seed_dev=None, m=None, step=None, ordered=True, fold=False):
    """Returns the Float32 loss (and the gradient w.r.t. x, device (3,sumV), when with_grad).
    ``sync=False``: the loss stays a 1-element device array (the three terms are combined by fx3d_lincomb in
    the reference's order) and the call enqueues without a single host round trip.  ``seed_dev``: device uint64
    added to both sampling seeds by the kernels (FitStepGraph advances it between replays).  ``step``: see
    :func:`chamfer_sampled_grad` (single source mesh): the optimiser step rides in the last adjoint's launch.
    ``ordered`` (default): the sampling adjoint without float atomics -- the gradient is bit-reproducible (the oracle's);
    ``ordered=False``: the scatter with float atomics, ~12 us per call faster on one mesh of 5000 draws (slower at eight).
    ``fold`` (with ``step``, ``sync=False``, ``with_grad``): the two regularisers ride in the sampling launches
    (:class:`MeshReg`: forward in the draw launch, adjoint in the launch of the chamfer adjoint's rows) -- the same bits, two
    launches less per iteration."""
    if m is None:  # (FitStepGraph passes the offset mesh the previous iteration's optimiser step already wrote)
        m = offset(src, x)
    s1 = None if seed is None else seed
    s2 = None if seed is None else seed + 1
    # (the source's fresh CDF, then BOTH draws in one launch; the target keeps its CDF while its vertices do not change)
    fold = bool(fold) and step is not None and with_grad and not sync and m.verts_aliased
    reg = MeshReg(m, 0.0, w_lap, w_edge) if fold else None
    A, Bp, fa, r1, r2 = sample_points_pair(m, tgt, num_samples, seed_a=s1, seed_b=s2, seed_dev=seed_dev, return_draws_a=True, reg=reg)
    loss1, ix, iy = _chamfer_points(A, Bp, 1.0, 1.0, return_indices=True, sync=sync)
    if fold:
        reg.set_base(loss1)
        g = DeviceArray.empty((3, m.V * m.N), np.float32)
        chamfer_sampled_grad(A, Bp, ix, iy, mesh_a=m, draws_a=(fa, r1, r2), out_a=g.reshape(3, m.V, m.N), step=step, ordered=ordered, reg=reg)
        return reg.total, g
    # both regularisers and the tutorial's sum fl(fl(l1 + fl(w_lap*l2)) + fl(w_edge*l3)) in ONE launch (unfused Float32)
    if sync:
        loss2, loss3, _ = mesh_losses(m, 0.0, w_lap, w_edge, sync=True)
        loss = np.float32(np.float32(loss1 + np.float32(w_lap) * loss2) + np.float32(w_edge) * loss3)
    else:
        _, _, loss = mesh_losses(m, 0.0, w_lap, w_edge, base=loss1, sync=False)
    if not with_grad:
        return loss
    if m.verts_aliased:
        # one mesh (or meshes of equal vertex counts): padded == packed.  Both mesh-loss adjoints in ONE gather launch (no float atomics, reusing the forward's
        # unit rows) WRITE the buffer; the chamfer adjoint and the sampling adjoint are ONE launch that scatter-adds on top
        # (the target's half of the chamfer adjoint is not needed and not computed): no memset node in the iteration
        g = mesh_losses_grad(m, 0.0, w_lap, w_edge, reuse_forward=True)
        # (round 6: ordered -- bit-reproducible -- and, when the caller hands over the optimiser's state, with its step in the
        #  same launch: the thread that finishes a vertex's gradient row applies Momentum + offset to it)
        chamfer_sampled_grad(A, Bp, ix, iy, mesh_a=m, draws_a=(fa, r1, r2), out_a=g.reshape(3, m.V, m.N), step=step, ordered=ordered)
        return loss, g
    gA, _ = chamfer_distance_grad(A, Bp, ix, iy)
    gpad = sample_points_grad(m, fa, r1, r2, gA, ordered=ordered)   # (3,Vmax,B)
    g = m.padded_to_packed_dev(gpad)                        # adjoint of _packed_to_padded
    mesh_losses_grad(m, 0.0, w_lap, w_edge, out=g, reuse_forward=True)
    return loss, g


class Momentum:
    """Flux.Optimise.Momentum(eta, rho): v = rho*v - eta*g ; x += v  (examples/fit_mesh.jl:87-88,110)."""

    def __init__(self, eta=1.0, rho=0.9):
        self.eta, self.rho, self.v = eta, rho, None

    def update(self, x, g):
        if self.v is None:
            self.v = DeviceArray.zeros(x.shape, np.float32)
        _lib.call("fx3d_momentum_step", x.size, float(self.rho), float(self.eta), g.ptr, self.v.ptr, x.ptr,
                  current_stream().handle)
        return x

    def step_args(self, x, base, out, counter=None, inc=0):
        """The ``step`` tuple of :func:`chamfer_sampled_grad`: this optimiser's update of x (and out = base + x, counter += inc)
        applied inside the adjoint's launch instead of by :meth:`update_offset` afterwards."""
        if self.v is None:
            self.v = DeviceArray.zeros(x.shape, np.float32)
        return (self.rho, self.eta, self.v, x, base, out, counter, inc)

    def update_offset(self, x, g, base, out, counter=None, inc=0):
        """update(x, g) and, in the same launch, out = base + x (the next iteration's offset mesh) and counter += inc."""
        if self.v is None:
            self.v = DeviceArray.zeros(x.shape, np.float32)
        _lib.call("fx3d_momentum_step_offset", x.size, float(self.rho), float(self.eta), g.ptr, self.v.ptr, x.ptr, base.ptr,
                  out.ptr, counter.ptr if counter is not None else None, int(inc), current_stream().handle)
        return x


class FitStepGraph:
    """One iteration of the fit_mesh loop (examples/fit_mesh.jl:98-110: loss, gradient, Momentum update) captured
    as a hipGraph: five launch-bound kernels (no memset, no copy; seven with ``fold=False``) replayed with one launch per iteration.

    The sampling seeds recorded in the graph are ``seed`` and ``seed + 1`` plus a device counter that the graph
    itself advances by two per replay, so every iteration draws fresh samples (the reference draws from the global
    RNG).  The constructor runs iteration 1 eagerly (``first_loss``) and records iteration 2; ``loss`` is the
    1-element device array every replay writes -- read it whenever a host value is wanted
    (``float(step.loss.item())`` after ``synchronize()``): the only host round trip.  ``x`` belongs to the graph while
    it is in use: after an external write to it call :meth:`resync`.

    ``ordered`` (round 6, the default): the sampling adjoint without float atomics -- every vertex's sum in a fixed order, the
    gradient the oracle's bit for bit and the same on every replay (as the reference's CPU adjoint is), with the optimiser step in
    the gather's launch (``step_in_launch``).  ``ordered=False``: the rows are scattered with float atomics while they are formed
    (sums in arrival order; bench.py ``graph_replay`` / ``graph_replay_scatter``).  ``fold`` (with the ordered form and the step in
    its launch): the two regularisers ride in the sampling launches (:class:`MeshReg`) -- five launches per iteration instead of
    seven, the same bits."""

    def __init__(self, x, src, tgt, opt, num_samples=5000, seed=0x5EED0C3, w_lap=0.1, w_edge=1.0, ordered=True, step_in_launch=True, fold=True):
        self.x, self.opt = x, opt
        self.stream = Stream.create()
        self.counter = DeviceArray.zeros((1,), np.uint64)
        self.iterations = 0

        # one mesh and an optimiser that offers it: the Momentum launch also writes the next iteration's offset mesh and advances
        # the seed counter (two launches less per iteration); the buffer is re-wrapped per body so that nothing derived from
        # the vertices (padded form, sampling CDF) is cached across iterations
        fused = src.verts_aliased and hasattr(opt, "update_offset")
        self._src_verts = src.dev("verts_packed")
        self.mverts = lincomb(1.0, self._src_verts, 1.0, x) if fused else None

        ordered = bool(ordered) and sampling_adjoint_is_ordered(src, num_samples)  # (more draws than the ordered form stages: the scatter)
        in_launch = fused and hasattr(opt, "step_args") and ordered and step_in_launch  # (round 6) ... and the step itself rides in the last adjoint's launch

        def body():
            m = src.with_verts_packed(self.mverts) if fused else None
            st = opt.step_args(x, src.dev("verts_packed"), self.mverts, self.counter, 2) if in_launch else None
            loss, g = loss_dolphin(x, src, tgt, num_samples, seed=seed, with_grad=True, w_lap=w_lap, w_edge=w_edge,
                                   sync=False, seed_dev=self.counter, m=m, step=st, ordered=ordered, fold=fold and in_launch)
            if in_launch:
                pass
            elif fused:
                opt.update_offset(x, g, src.dev("verts_packed"), self.mverts, self.counter, 2)
            else:
                opt.update(x, g)
                _lib.call("fx3d_counter_add", self.counter.ptr, 2, current_stream().handle)
            return loss

        current_stream().synchronize()  # x / optimiser state may still be in flight on the caller's stream
        with stream(self.stream):
            self.first_loss = body()  # eager once on the capture stream: workspaces, caches, optimiser state
            self.stream.synchronize()    # (a real step: iteration 1)
        self.iterations = 1
        self.graph = Graph()
        with self.graph.capture(self.stream):
            self.loss = body()

    def step(self):
        self.graph.launch(self.stream)
        self.iterations += 1
        return self.loss

    def resync(self):
        """Call after writing ``x`` from outside (a reset, a projection, a checkpoint restore): the fused graph keeps the
        offset mesh ``src + x`` in a buffer of its own that only its Momentum launch updates, so an external write to x
        would otherwise leave the replays optimising from the old vertex positions.  Recomputes that buffer on
        the graph's stream; a no-op for the unfused graph, whose body recomputes the offset every iteration."""
        if self.mverts is not None:
            current_stream().synchronize()  # the caller's write to x
            with stream(self.stream):
                lincomb(1.0, self._src_verts, 1.0, self.x, out=self.mverts)

    def synchronize(self):
        self.stream.synchronize()
