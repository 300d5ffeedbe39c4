# Day-one validation of the Julia binding (Flux3DHip.jl) on an MI355X box with Julia, Flux3D.jl and libflux3d_hip.so:
#
#     FLUX3D_HIP_LIB=/path/to/libflux3d_hip.so julia --project=<Flux3D.jl checkout> flux3d.jl_amd/julia/runtests.jl
#
# It mirrors, on `hip(...)` data, the tests the reference runs on its own storage types -- test/metrics.jl:75-114 (edge loss,
# chamfer against the dense formula, gradients), test/cuda/metrics.jl:87-129 (the same on the GPU storage, `sample_points` on the
# unit sphere), test/transforms/mesh_func.jl:4-14 (radius of the samples) -- plus what only this binding can check: bit identity of
# the nearest-neighbour indices with the reference's CPU method (src/metrics/pcloud.jl:54-70) and the README's known answer
# `laplacian_loss(teapot) = 0.05888283f0` (README.md:111-112).
#
# `julia` is not part of the build image (the shim is checked lexically by tests/test_julia_shim.py: every @ccall against the header);
# this file is therefore UNEXECUTED there.  It uses nothing but Test, Statistics, Zygote (Flux3D's own dependencies) and the shim.
using Test, Statistics
using Flux3D
using Flux3D: chamfer_distance, laplacian_loss, edge_loss, sample_points, load_trimesh, get_verts_packed, get_edges_packed,
              get_verts_padded, TriMesh, PointCloud
using Zygote: gradient

include(joinpath(@__DIR__, "Flux3DHip.jl"))
using .Flux3DHip: hip, unhip, HipArray, use_hip, knn_graph

const ASSETS = get(ENV, "FLUX3D_TEST_ASSETS", joinpath(dirname(pathof(Flux3D)), "..", "test", "assets"))

# the dense formula of test/metrics.jl:94-108, on host arrays
function naive_chamfer(x, y)
    xx = sum(x .^ 2, dims = 1)
    yy = sum(y .^ 2, dims = 1)
    zz = Flux3D.Flux.batched_mul(permutedims(x, (2, 1, 3)), y)
    P = (reshape(xx, size(xx, 2), 1, :) .+ reshape(yy, 1, size(yy, 2), :)) .- (2 .* zz)
    return mean(mean(minimum(P; dims = 2); dims = 2)) + mean(mean(minimum(P; dims = 1); dims = 2))
end

@testset "Flux3DHip on $(Flux3DHip.device_count()) device(s)" begin
    @test Flux3DHip.device_count() >= 1
    @test use_hip[]          # the library was found and a device is visible (the analogue of Flux3D.use_cuda)

    @testset "storage" begin
        a = rand(Float32, 3, 17, 2)
        d = hip(a)
        @test d isa HipArray{Float32,3}
        @test size(d) == size(a)
        @test unhip(d) == a
        @test unhip(copy(d)) == a
    end

    @testset "chamfer_distance on clouds (test/metrics.jl:109-114, test/cuda/metrics.jl:123-128)" begin
        x = rand(Float32, 3, 1000, 2)
        y = rand(Float32, 3, 500, 2)
        dx, dy = hip(x), hip(y)
        @test isapprox(chamfer_distance(dx, dy), naive_chamfer(x, y))
        @test isapprox(chamfer_distance(dx, dy), chamfer_distance(x, y); rtol = 1.0f-5)      # the reference's CPU method
        @test isapprox(chamfer_distance(dx, dy; w1 = 0.7, w2 = 1.3), chamfer_distance(x, y; w1 = 0.7, w2 = 1.3); rtol = 1.0f-5)
        @test chamfer_distance(dx, dx) == 0
        # rank-2 lift (src/metrics/pcloud.jl:28-37) and the PointCloud front door
        @test isapprox(chamfer_distance(hip(x[:, :, 1]), hip(y[:, :, 1])), chamfer_distance(x[:, :, 1], y[:, :, 1]); rtol = 1.0f-5)
        @test isapprox(chamfer_distance(PointCloud(dx), PointCloud(dy)), chamfer_distance(x, y); rtol = 1.0f-5)
        # gradients: against the dense formula at the reference's own tolerance, against the CPU method tightly
        g1 = gradient(a -> chamfer_distance(a, dy), dx)[1]
        g2 = gradient(a -> naive_chamfer(a, y), x)[1]
        g3 = gradient(a -> chamfer_distance(a, y), x)[1]
        @test isapprox(unhip(g1), g2, atol = 1e-2, rtol = 1e-3)
        @test isapprox(unhip(g1), g3, atol = 1e-7, rtol = 1e-5)
        # a second pullback call gets arrays of its own (ADVICE r5)
        l, back = Flux3D.Zygote.pullback((a, b) -> Flux3D._chamfer_distance(a, b, 1.0f0, 1.0f0), dx, dy)
        ga1, = back(1.0f0); ga2, = back(1.0f0)
        @test ga1.ptr != ga2.ptr && unhip(ga1) == unhip(ga2)
    end

    @testset "nearest neighbours: the CPU method's indices, bit for bit (src/metrics/pcloud.jl:54-70)" begin
        x = rand(Float32, 3, 4096, 3)
        y = rand(Float32, 3, 4000, 3)
        nx, ny = Flux3D._nearest_neighbors(x, y)
        hx, hy = Flux3D._nearest_neighbors(hip(x), hip(y))
        @test hx == nx && hy == ny
    end

    @testset "kNN graph (src/models/dgcnn.jl:3-9)" begin
        X = rand(Float32, 3, 256, 2)
        G = knn_graph(hip(X), 20)
        @test size(G) == (3, 20, 256, 2)
        ref = cat([Flux3D.CreateSingleKNNGraph(X[:, :, b], 20) for b = 1:2]...; dims = 4)
        @test unhip(G) == ref
    end

    @testset "meshes (test/metrics.jl:44-93, test/transforms/mesh_func.jl:4-14, README.md:103-112)" begin
        teapot = load_trimesh(joinpath(ASSETS, "teapot.obj"))
        @test laplacian_loss(hip(teapot)) == 0.05888283f0
        m = load_trimesh(joinpath(ASSETS, "teapot.obj"), joinpath(ASSETS, "sphere.obj"))
        dm = hip(m)
        @test isapprox(laplacian_loss(dm), laplacian_loss(m); rtol = 1.0f-5)
        @test isapprox(edge_loss(dm), edge_loss(m); rtol = 1.0f-5)
        @test isapprox(edge_loss(dm, 0.1), edge_loss(m, 0.1); rtol = 1.0f-5)
        verts, edges = get_verts_packed(m), get_edges_packed(m)
        @test isapprox(edge_loss(dm), mean((Flux3D._norm(verts[:, edges[:, 1]] - verts[:, edges[:, 2]]; dims = 1)) .^ 2); rtol = 1.0f-5)
        @test gradient(x -> edge_loss(x), dm) isa Tuple
        @test gradient(x -> laplacian_loss(x), dm) isa Tuple
        # sample_points: the type of the storage comes back, the samples of the unit sphere lie on it
        s2 = hip(load_trimesh(joinpath(ASSETS, "sphere.obj"), joinpath(ASSETS, "sphere.obj")))
        samples = sample_points(s2, 1000)
        @test samples isa HipArray{Float32,3} && size(samples) == (3, 1000, 2)
        radius = sqrt.(sum(unhip(samples) .^ 2; dims = 1))
        @test all(isapprox.(radius, 1.0, rtol = 1e-2, atol = 1e-5))
        @test gradient(x -> sum(sample_points(x, 1000)), s2) isa Tuple
        # chamfer_distance(mesh, mesh, n) (src/metrics/mesh.jl:34-44)
        loss = chamfer_distance(dm, dm)
        @test all(isapprox.(loss, 0, rtol = 1e-5, atol = 1e-2))
        @test gradient(x -> chamfer_distance(x, x), dm) isa Tuple
    end

    @testset "the tutorial's iteration in five launches (examples/fit_mesh.jl:78-110): passengers = separate calls, bit for bit" begin
        src = hip(load_trimesh(joinpath(ASSETS, "sphere.obj"))); tgt = hip(load_trimesh(joinpath(ASSETS, "teapot.obj")))
        nv = size(get_verts_packed(src), 2)
        state() = (hip(zeros(Float32, 3, nv)), hip(zeros(Float32, 3, nv)), hip(zeros(Float32, 3, nv)), hip(zeros(Float32, 3, nv)))
        base = get_verts_packed(src)
        # one after the other: draws, chamfer, both regularisers + the objective's sum, their adjoint, the sampling adjoint + Momentum
        ws = Flux3DHip.mesh_losses_workspace(src); loss1 = HipArray{Float32}(undef, 1)
        A, B = Flux3DHip.sample_points_pair(src, tgt, 5000; seed1 = UInt64(21), seed2 = UInt64(22))
        out_a = HipArray{Float32}(undef, 3)
        reg_a = Flux3DHip.mesh_reg(src, ws, out_a; base = loss1)
        A2, B2, draws = Flux3DHip.sample_points_pair_reg(src, tgt, reg_a, 5000; seed1 = UInt64(21), seed2 = UInt64(22))
        @test unhip(A2) == unhip(A) && unhip(B2) == unhip(B)               # the passengers leave the draws alone
        ix, iy = Flux3DHip.chamfer_fwd_dev!(loss1, A, B)
        sep = Flux3DHip.mesh_losses(src, ws; base = loss1)
        g1 = Flux3DHip.mesh_losses_grad(src, ws; reuse_forward = true)
        x1, v1, _, o1 = state()
        Flux3DHip.chamfer_sampled_grad_step!(g1, A, B, ix, iy, src, draws, x1, v1, base, o1; eta = 0.7, rho = 0.9)
        # as passengers (the forward rode with the draws above)
        g2 = HipArray{Float32}(undef, 3, nv); x2, v2, _, o2 = state()
        Flux3DHip.chamfer_sampled_grad_step_reg!(g2, A, B, ix, iy, src, draws, reg_a, x2, v2, base, o2; eta = 0.7, rho = 0.9)
        @test unhip(out_a) == unhip(sep)
        @test unhip(g2) == unhip(g1) && unhip(x2) == unhip(x1) && unhip(v2) == unhip(v1) && unhip(o2) == unhip(o1)
    end
end
