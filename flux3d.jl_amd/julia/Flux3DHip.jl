# Flux3DHip.jl -- thin @ccall shim that puts libflux3d_hip.so behind Flux3D.jl's own API.
#
# NOT EXECUTED IN THIS REPOSITORY'S CI: the build image has no `julia`.  It is kept declarative --
# one @ccall per C-ABI entry point (include/flux3d_hip.h), no logic beyond shape bookkeeping.  What IS
# checked mechanically (tests/test_julia_shim.py, CPU suite): every `@ccall LIB.fx3d_*` in this file
# against the header prototype (arity, every argument's C type, the return type), every FX3D_API symbol
# bound here, block structure (`function`/`end` balance), and the checklist of reference generic functions
# a HipArray-backed TriMesh touches (INTEGRATION.md section 2a) against the methods defined below.  The same
# call sequence is exercised by the Python twin of this layer (flux3d.jl_amd/*.py) on the GPU.
#
# Usage: this file is `include`d INSIDE `module Flux3D` (INTEGRATION.md section 2), after the rep / metrics /
# transforms includes, hence the relative `using ..Flux3D`.  (As a stand-alone package that depends on
# Flux3D, replace the two `..Flux3D` below by `Flux3D`; nothing else changes.)
#     include("Flux3DHip.jl"); using .Flux3DHip
#     A = hip(PointCloud(rand(Float32, 3, 4096, 32)));  B = hip(...)
#     chamfer_distance(A, B)                 # dispatches to the HipArray methods below
#     m = hip(load_trimesh("teapot.obj"));   laplacian_loss(m); sample_points(m, 5000)
#
# No CUDA.jl, no AMDGPU.jl, no Triton: device memory is owned by the library (fx3d_malloc) and
# wrapped in a HipArray with a finalizer.
module Flux3DHip

using ..Flux3D
import ..Flux3D: chamfer_distance, _chamfer_distance, _nearest_neighbors, sample_points,
                 laplacian_loss, edge_loss, TriMesh, PointCloud,
                 get_verts_packed, get_verts_padded, get_verts_list, get_faces_packed, get_faces_padded,
                 get_faces_list, get_edges_packed, get_laplacian_packed,
                 compute_faces_areas_packed, compute_faces_areas_padded,
                 _list_to_packed, _list_to_padded, _packed_to_padded, _packed_to_list,
                 _padded_to_list, _padded_to_packed, offset!
using SparseArrays: SparseMatrixCSC, findnz
import Zygote

export HipArray, hip, unhip, use_hip, knn_graph, edgeconv_graph, pointcloud_to_voxel

const LIB = get(ENV, "FLUX3D_HIP_LIB", joinpath(@__DIR__, "..", "lib", "libflux3d_hip.so"))
const Stream = Ptr{Cvoid}
const DEFAULT_STREAM = Stream(C_NULL)
const Event = Ptr{Cvoid}

# ---- status handling (include/flux3d_hip.h: every call returns fx3d_status) -------------------
function last_error()
    buf = Vector{UInt8}(undef, 512)
    @ccall LIB.fx3d_last_error(buf::Ptr{UInt8}, 512::Csize_t)::Csize_t
    return unsafe_string(pointer(buf))
end
@inline check(rc::Int32) = rc == 0 ? nothing : error("flux3d_hip [$rc]: " * last_error())

function device_count()
    n = Ref{Int32}(0)
    rc = @ccall LIB.fx3d_device_count(n::Ref{Int32})::Int32
    return rc == 0 ? Int(n[]) : 0
end
# analogue of Flux3D.use_cuda (src/Flux3D.jl:52-61)
const use_hip = Ref(false)
__init__() = (use_hip[] = isfile(LIB) && device_count() > 0)

# ---- device array: the `S` storage type of TriMesh{T,R,S} / PointCloud.points -----------------
# An AbstractArray without scalar indexing: everything the reference's generic rep code does to the storage type
# (constructors S{T,N}(undef, ...), similar, reshape, hcat, range getindex, T.(x), +, copy, deepcopy, fill!) has a
# method here that stays on the device; the checklist is INTEGRATION.md section 2a.
mutable struct HipArray{T,N} <: AbstractArray{T,N}
    ptr::Ptr{Cvoid}
    dims::NTuple{N,Int}
    owner::Any            # parent HipArray for views (keeps the allocation alive)
end
Base.size(a::HipArray) = a.dims
Base.sizeof(a::HipArray{T}) where {T} = prod(a.dims) * sizeof(T)
Base.IndexStyle(::Type{<:HipArray}) = IndexLinear()
Base.getindex(a::HipArray, i::Int) = error("scalar indexing of a HipArray is not supported; use unhip(a)")
Base.setindex!(a::HipArray, v, i::Int) = error("scalar indexing of a HipArray is not supported; use hip(x)")
Base.show(io::IO, a::HipArray{T,N}) where {T,N} = print(io, "HipArray{$T,$N}", size(a))
Base.show(io::IO, ::MIME"text/plain", a::HipArray) = show(io, a)

_free(a::HipArray) = (@ccall LIB.fx3d_free(a.ptr::Ptr{Cvoid})::Int32; nothing)
# `S{T,2}(undef, 3, n)` / `S{T,3}(undef, 3, V, N)` in the reference's TriMesh constructor (src/rep/mesh.jl:151-152)
function HipArray{T,N}(::UndefInitializer, dims::NTuple{N,Int}) where {T,N}
    p = Ref{Ptr{Cvoid}}(C_NULL)
    check(@ccall LIB.fx3d_malloc(p::Ref{Ptr{Cvoid}}, max(prod(dims) * sizeof(T), 1)::Csize_t)::Int32)
    a = HipArray{T,N}(p[], dims, nothing)
    finalizer(_free, a)
    return a
end
HipArray{T,N}(::UndefInitializer, dims::Vararg{Integer,N}) where {T,N} = HipArray{T,N}(undef, map(Int, dims))
HipArray{T}(::UndefInitializer, dims::NTuple{N,Integer}) where {T,N} = HipArray{T,N}(undef, map(Int, dims))
HipArray{T}(::UndefInitializer, dims::Vararg{Integer,N}) where {T,N} = HipArray{T,N}(undef, map(Int, dims))
# `similar(packed, D, M, N)` / `similar(verts_padded, 3, n, B)` (src/rep/utils.jl:129, src/transforms/mesh_func.jl:40)
Base.similar(a::HipArray, ::Type{T}, dims::Dims{N}) where {T,N} = HipArray{T,N}(undef, dims)

function hip(x::Array{T,N}) where {T,N}
    a = HipArray{T,N}(undef, size(x))
    check(@ccall LIB.fx3d_memcpy_h2d(a.ptr::Ptr{Cvoid}, x::Ptr{T}, sizeof(x)::Csize_t, DEFAULT_STREAM::Stream)::Int32)
    return a
end
hip(a::HipArray) = a
function unhip(a::HipArray{T,N}) where {T,N}
    x = Array{T,N}(undef, a.dims)
    check(@ccall LIB.fx3d_memcpy_d2h(x::Ptr{T}, a.ptr::Ptr{Cvoid}, sizeof(x)::Csize_t, DEFAULT_STREAM::Stream)::Int32)
    return x
end
unhip(x::Array) = x
Base.Array(a::HipArray) = unhip(a)
Base.collect(a::HipArray) = unhip(a)

# reshape shares the allocation (a view with an owner); one Colon is resolved as Base does
function _reshape(a::HipArray{T}, full::Dims{N}) where {T,N}
    prod(full) == length(a) ||
        throw(DimensionMismatch("new dimensions $(full) must be consistent with array size $(length(a))"))
    return HipArray{T,N}(a.ptr, full, a)
end
function _uncolon(a::HipArray, dims::Tuple{Vararg{Union{Int,Colon}}})
    count(d -> d isa Colon, dims) == 1 || throw(DimensionMismatch("new dimensions $(dims) may have at most one omitted dimension"))
    known = 1
    for d in dims
        d isa Int && (known *= d)
    end
    rest = known == 0 ? 0 : div(length(a), known)
    return map(d -> d isa Colon ? rest : d, dims)
end
Base.reshape(a::HipArray, dims::Dims) = _reshape(a, dims)
Base.reshape(a::HipArray, dims::Tuple{Vararg{Union{Int,Colon}}}) = _reshape(a, _uncolon(a, dims))
Base.dropdims(a::HipArray; dims) = _reshape(a, Tuple(s for (i, s) in enumerate(size(a)) if !(i in dims)))

# device-to-device copy of `n` elements, offsets in elements (0-based)
function copyto_d2d!(dst::HipArray{T}, doff::Int, src::HipArray{T}, soff::Int, n::Int) where {T}
    n == 0 && return dst
    (0 <= doff && doff + n <= length(dst) && 0 <= soff && soff + n <= length(src)) || throw(BoundsError())
    check(@ccall LIB.fx3d_memcpy_d2d((dst.ptr + doff * sizeof(T))::Ptr{Cvoid}, (src.ptr + soff * sizeof(T))::Ptr{Cvoid},
                                     (n * sizeof(T))::Csize_t, DEFAULT_STREAM::Stream)::Int32)
    return dst
end
function Base.copy(a::HipArray{T,N}) where {T,N}
    out = HipArray{T,N}(undef, size(a))
    return copyto_d2d!(out, 0, a, 0, length(a))
end
# deepcopy(m::TriMesh) in `offset` (src/transforms/mesh_func.jl:436): a bitwise copy would alias the allocation
Base.deepcopy_internal(a::HipArray, dict::IdDict) = get!(() -> copy(a), dict, a)
function Base.fill!(a::HipArray{T}, z::Number) where {T}
    iszero(z) || error("HipArray fill!: only zero is supported (the reference pads vertices with 0, src/rep/mesh.jl:855)")
    check(@ccall LIB.fx3d_memset(a.ptr::Ptr{Cvoid}, 0::Int32, sizeof(a)::Csize_t, DEFAULT_STREAM::Stream)::Int32)
    return a
end
# `T.(v)` ("To remove lazy wrappers", src/rep/mesh.jl:129; PointCloud(points) src/rep/pcloud.jl:45-51): identity
Base.Broadcast.broadcasted(::Type{T}, a::HipArray{T}) where {T} = a
# contiguous column ranges: `packed[:, i:j]`, `padded[:, 1:len, b]`, `points[:, :, b]` are device copies
function Base.getindex(a::HipArray{T,2}, ::Colon, r::UnitRange{Int}) where {T}
    D = size(a, 1)
    out = HipArray{T,2}(undef, D, length(r))
    return copyto_d2d!(out, 0, a, (first(r) - 1) * D, D * length(r))
end
function Base.getindex(a::HipArray{T,3}, ::Colon, r::UnitRange{Int}, b::Int) where {T}
    D, M, _ = size(a)
    out = HipArray{T,2}(undef, D, length(r))
    return copyto_d2d!(out, 0, a, ((b - 1) * M + first(r) - 1) * D, D * length(r))
end
Base.getindex(a::HipArray{T,3}, ::Colon, ::Colon, b::Int) where {T} = a[:, 1:size(a, 2), b]
# `hcat(list...)` in _list_to_packed (src/rep/utils.jl:98)
function Base.hcat(list::HipArray{T,2}...) where {T}
    D = size(list[1], 1)
    all(size(x, 1) == D for x in list) || throw(DimensionMismatch("hcat: mismatched leading dimension"))
    packed = HipArray{T,2}(undef, D, sum(size(x, 2) for x in list))
    off = 0
    for x in list
        copyto_d2d!(packed, off, x, 0, length(x)); off += length(x)
    end
    return packed
end

hip(p::PointCloud) = PointCloud(hip(p.points), p.normals === nothing ? nothing : hip(p.normals))
unhip(p::PointCloud) = PointCloud(unhip(p.points), p.normals === nothing ? nothing : unhip(p.normals))

# ---- TriMesh on HipArray storage ---------------------------------------------------------------------------------
# The reference only has constructor methods for Vector{<:CuArray} and Vector{<:Array} (src/rep/mesh.jl:107-117; the
# generic one is commented out, :100-105).  This is the third: same body as the S-typed constructor (:119-177) with
# S = HipArray; the functor (src/rep/mesh.jl:189-190) rebuilds through it, so `hip(m)` / `unhip(m)` / fmap work.
function TriMesh(verts::Vector{<:HipArray{T,2}}, faces::Vector{<:AbstractArray{R,2}};
                 offset::Number = -1) where {T<:AbstractFloat,R<:Integer}
    length(verts) == length(faces) ||
        error("batch size of verts and faces should match, $(length(verts)) != $(length(faces))")
    all(size(v, 1) == 3 for v in verts) || error("verts must be (3, V) arrays")
    verts_list = HipArray{T,2}[v for v in verts]
    faces_list = Array{R,2}[Array{R,2}(f) for f in faces]
    _verts_len = Int64[size(v, 2) for v in verts_list]
    _faces_len = Int64[size(f, 2) for f in faces_list]
    N = length(verts_list); V = maximum(_verts_len); F = maximum(_faces_len)
    equalised = all(_verts_len .== V) && all(_faces_len .== F)
    valid = _faces_len .> 0
    return TriMesh{T,R,HipArray}(
        N, V, F, equalised, valid, Int8(offset), _verts_len, _faces_len,
        HipArray{T,2}(undef, 3, sum(_verts_len)), HipArray{T,3}(undef, 3, V, N), verts_list, false, false, true,
        Array{R,2}(undef, 3, sum(_faces_len)), Array{R,3}(undef, 3, F, N), faces_list, false, false,
        nothing, nothing, nothing, nothing)
end
# functor(::TriMesh) moves only the verts (src/rep/mesh.jl:189-190)
hip(m::TriMesh) = TriMesh([hip(v) for v in get_verts_list(m)], get_faces_list(m); offset = m.offset)
unhip(m::TriMesh) = TriMesh([unhip(v) for v in get_verts_list(m)], get_faces_list(m); offset = m.offset)

# Converters between the list / packed / padded forms (src/rep/utils.jl:58-206) for device storage.  The accessors
# get_verts_packed / get_verts_padded / get_verts_list (src/rep/mesh.jl:344-394) and the lazy cache logic
# (_compute_verts_*, :838-882; setproperty!, :208-231) are the reference's own code: they only reach the storage
# through these six functions, `size`, and `convert(S, x)`.
_list_to_packed(list::Vector{<:HipArray{T,2}}) where {T<:Number} = hcat(list...)

function _packed_to_padded(packed::HipArray{T,2}, items_len::AbstractArray{<:Number,1}, pad_value::Number) where {T<:Number}
    iszero(pad_value) || error("HipArray _packed_to_padded: only pad_value = 0 is supported")
    lens = Int64[l for l in items_len]
    D = size(packed, 1); M = maximum(lens); B = length(lens)
    sum(lens) == size(packed, 2) || error("items_len does not add up to the packed size")
    padded = HipArray{T,3}(undef, D, M, B)
    if T === Float32 && D == 3
        check(@ccall LIB.fx3d_packed_to_padded(packed.ptr::Ptr{Cvoid}, lens::Ptr{Int64}, B::Int32, M::Int32,
                                               padded.ptr::Ptr{Cvoid}, DEFAULT_STREAM::Stream)::Int32)
    else
        fill!(padded, 0)
        cur = 0
        for (i, len) in enumerate(lens)
            copyto_d2d!(padded, (i - 1) * M * D, packed, cur * D, len * D); cur += len
        end
    end
    return padded
end
Zygote.@adjoint function _packed_to_padded(packed::HipArray{T,2}, items_len::AbstractArray{<:Number,1}, pad_value::Number) where {T<:Number}
    return _packed_to_padded(packed, items_len, pad_value), g -> (_padded_to_packed(g, items_len), nothing, nothing)
end

function _packed_to_list(packed::HipArray{T,2}, items_len::AbstractArray{<:Number,1}) where {T<:Number}
    list = HipArray{T,2}[]
    cur = 1
    for len in items_len
        push!(list, packed[:, cur:cur+Int(len)-1]); cur += Int(len)
    end
    return list
end

function _padded_to_list(padded::HipArray{T,3}, items_len::Union{Nothing,AbstractArray{<:Number,1}}) where {T<:Number}
    lens = items_len === nothing ? fill(size(padded, 2), size(padded, 3)) : items_len
    length(lens) == size(padded, 3) || error("items_len length should match the last dimension of padded array")
    return HipArray{T,2}[padded[:, 1:Int(len), i] for (i, len) in enumerate(lens)]
end

function _padded_to_packed(padded::HipArray{T,3}, items_len::Union{Nothing,AbstractArray{<:Number,1}} = nothing,
                           pad_value::Union{Nothing,Number} = nothing) where {T<:Number}
    (pad_value === nothing || items_len === nothing) || error("pad_value and items_len both should not be given")
    items_len === nothing && error("HipArray _padded_to_packed needs items_len")
    lens = Int64[l for l in items_len]
    D, M, B = size(padded)
    length(lens) == B || error("items_len length should match the last dimension of padded array")
    packed = HipArray{T,2}(undef, D, sum(lens))
    if T === Float32 && D == 3
        check(@ccall LIB.fx3d_padded_to_packed(padded.ptr::Ptr{Cvoid}, lens::Ptr{Int64}, B::Int32, M::Int32,
                                               packed.ptr::Ptr{Cvoid}, DEFAULT_STREAM::Stream)::Int32)
    else
        cur = 0
        for (i, len) in enumerate(lens)
            copyto_d2d!(packed, cur * D, padded, (i - 1) * M * D, len * D); cur += len
        end
    end
    return packed
end

function _list_to_padded(list::Vector{<:HipArray{T,2}}, pad_value::Number, pad_size::Union{Nothing,Tuple} = nothing) where {T<:Number}
    pad_size === nothing || length(pad_size) == 2 || error("pad_size should be a tuple of length 2")
    padded = _packed_to_padded(_list_to_packed(list), Int64[size(x, 2) for x in list], pad_value)
    pad_size === nothing || (size(padded, 1), size(padded, 2)) == pad_size ||
        error("HipArray _list_to_padded: pad_size must equal the largest item")
    return padded
end

# `verts_packed += offset_verts_packed` in offset! (src/transforms/mesh_func.jl:409-416)
Base.:+(a::HipArray{Float32,N}, b::HipArray{Float32,N}) where {N} = lincomb(1, a, 1, b)
Base.:-(a::HipArray{Float32,N}, b::HipArray{Float32,N}) where {N} = lincomb(1, a, -1, b)
offset!(m::TriMesh{Float32,R,HipArray}, offset_verts_packed::Array{Float32,2}) where {R} = offset!(m, hip(offset_verts_packed))

# scratch: caller-provided by ABI contract; one grow-only buffer per task is enough here
const _ws = Ref{Union{Nothing,HipArray{UInt8,1}}}(nothing)
function workspace(nbytes::Integer)
    if _ws[] === nothing || length(_ws[]) < nbytes
        _ws[] = HipArray{UInt8}(undef, max(Int(nbytes), 4096))
    end
    return _ws[]
end

# ---- nearest neighbours / chamfer: replaces src/metrics/pcloud.jl:72-86 (CuArray method) -------
function _nearest_neighbors(x::HipArray{Float32,3}, y::HipArray{Float32,3})
    D, N, B = size(x); _, M, _ = size(y)
    ix = HipArray{Int32}(undef, N, B); iy = HipArray{Int32}(undef, M, B)
    check(@ccall LIB.fx3d_nn1(x.ptr::Ptr{Cvoid}, N::Int32, y.ptr::Ptr{Cvoid}, M::Int32, B::Int32, D::Int32,
                              ix.ptr::Ptr{Cvoid}, iy.ptr::Ptr{Cvoid}, C_NULL::Ptr{Cvoid}, C_NULL::Ptr{Cvoid},
                              DEFAULT_STREAM::Stream)::Int32)
    # the reference returns Matrix{CartesianIndex{2}} (point, batch), 1-based (:83-84)
    hx, hy = unhip(ix), unhip(iy)
    nn_for_x = [CartesianIndex(Int(hx[i, b]) + 1, b) for i in 1:N, b in 1:B]
    nn_for_y = [CartesianIndex(Int(hy[j, b]) + 1, b) for j in 1:M, b in 1:B]
    return nn_for_x, nn_for_y
end

# which launch plan the library takes for a problem size (text; tools / bug reports)
function nn1_plan_describe(N::Integer, M::Integer, B::Integer, D::Integer = 3)
    buf = Vector{UInt8}(undef, 256)
    check(@ccall LIB.fx3d_nn1_plan_describe(N::Int32, M::Int32, B::Int32, D::Int32, buf::Ptr{UInt8}, length(buf)::Csize_t)::Int32)
    return unsafe_string(pointer(buf))
end

# fused forward (never materialises indices on the host): replaces :39-52 for HipArray storage
function _chamfer_fwd(A::HipArray{Float32,3}, B::HipArray{Float32,3}, w1::Float32, w2::Float32; indices::Bool = false)
    D, N, Bn = size(A); _, M, _ = size(B)
    nb = Ref{Csize_t}(0)
    check(@ccall LIB.fx3d_chamfer_workspace_bytes(N::Int32, M::Int32, Bn::Int32, D::Int32, nb::Ref{Csize_t})::Int32)
    ws = workspace(nb[])
    loss_dev = HipArray{Float32}(undef, 1); loss = Ref{Float32}(0)
    ix = indices ? HipArray{Int32}(undef, N, Bn) : nothing
    iy = indices ? HipArray{Int32}(undef, M, Bn) : nothing
    check(@ccall LIB.fx3d_chamfer_fwd(A.ptr::Ptr{Cvoid}, N::Int32, B.ptr::Ptr{Cvoid}, M::Int32, Bn::Int32, D::Int32,
                                      w1::Float32, w2::Float32, loss_dev.ptr::Ptr{Cvoid}, loss::Ref{Float32},
                                      (indices ? ix.ptr : C_NULL)::Ptr{Cvoid}, (indices ? iy.ptr : C_NULL)::Ptr{Cvoid},
                                      ws.ptr::Ptr{Cvoid}, length(ws)::Csize_t, DEFAULT_STREAM::Stream)::Int32)
    return loss[], ix, iy
end
_chamfer_distance(A::HipArray{Float32,3}, B::HipArray{Float32,3}, w1::Float32 = 1.0f0, w2::Float32 = 1.0f0) =
    _chamfer_fwd(A, B, w1, w2)[1]
# the same with the loss LEFT ON THE DEVICE (no host round trip: the shape of a captured fit iteration, where the regularisers' sum reads it
# as its base -- mesh_reg(...; base = loss_dev)); returns the neighbour indices
function chamfer_fwd_dev!(loss_dev::HipArray{Float32}, A::HipArray{Float32,3}, B::HipArray{Float32,3}; w1::Number = 1.0, w2::Number = 1.0)
    D, N, Bn = size(A); _, M, _ = size(B)
    nb = Ref{Csize_t}(0)
    check(@ccall LIB.fx3d_chamfer_workspace_bytes(N::Int32, M::Int32, Bn::Int32, D::Int32, nb::Ref{Csize_t})::Int32)
    ws = workspace(nb[])
    ix = HipArray{Int32}(undef, N, Bn); iy = HipArray{Int32}(undef, M, Bn)
    check(@ccall LIB.fx3d_chamfer_fwd(A.ptr::Ptr{Cvoid}, N::Int32, B.ptr::Ptr{Cvoid}, M::Int32, Bn::Int32, D::Int32,
                                      Float32(w1)::Float32, Float32(w2)::Float32, loss_dev.ptr::Ptr{Cvoid}, C_NULL::Ptr{Float32},
                                      ix.ptr::Ptr{Cvoid}, iy.ptr::Ptr{Cvoid}, ws.ptr::Ptr{Cvoid}, length(ws)::Csize_t,
                                      DEFAULT_STREAM::Stream)::Int32)
    return ix, iy
end

# value and gradient in ONE ABI call (what `Zygote.withgradient(chamfer_distance, A, B)` runs through the adjoint below; what
# benchmarks/metrics.jl:24-38 times as "total" and examples/fit_mesh.jl:106-110 runs per iteration): forward with indices +
# adjoint queued back to back.  Returns (loss, gA, gB, ix, iy); ix / iy are `nothing` unless `indices` (they then stay in the
# scratch).  `B_global`: the batch size the mean divides by (a shard of a larger batch passes the global one; default Bn).
# `out = (gA, gB)`: caller-owned gradient arrays, overwritten.
function chamfer_value_and_grad(A::HipArray{Float32,3}, B::HipArray{Float32,3}, w1::Float32 = 1.0f0, w2::Float32 = 1.0f0;
                                gout::Float32 = 1.0f0, B_global::Integer = size(A, 3), out = nothing, indices::Bool = false)
    D, N, Bn = size(A); _, M, _ = size(B)
    nb = Ref{Csize_t}(0)
    check(@ccall LIB.fx3d_chamfer_fwd_bwd_workspace_bytes(N::Int32, M::Int32, Bn::Int32, D::Int32, nb::Ref{Csize_t})::Int32)
    ws = workspace(nb[])
    loss_dev = HipArray{Float32}(undef, 1); loss = Ref{Float32}(0)
    gA = out === nothing ? HipArray{Float32}(undef, D, N, Bn) : out[1]
    gB = out === nothing ? HipArray{Float32}(undef, D, M, Bn) : out[2]
    size(gA) == (D, N, Bn) && size(gB) == (D, M, Bn) || throw(DimensionMismatch("chamfer_value_and_grad: out = (gA, gB) must match A and B"))
    ix = indices ? HipArray{Int32}(undef, N, Bn) : nothing
    iy = indices ? HipArray{Int32}(undef, M, Bn) : nothing
    check(@ccall LIB.fx3d_chamfer_fwd_bwd(A.ptr::Ptr{Cvoid}, N::Int32, B.ptr::Ptr{Cvoid}, M::Int32, Bn::Int32, D::Int32,
                                          w1::Float32, w2::Float32, gout::Float32, Int64(B_global)::Int64, loss_dev.ptr::Ptr{Cvoid},
                                          loss::Ref{Float32}, gA.ptr::Ptr{Cvoid}, gB.ptr::Ptr{Cvoid},
                                          (indices ? ix.ptr : C_NULL)::Ptr{Cvoid}, (indices ? iy.ptr : C_NULL)::Ptr{Cvoid},
                                          ws.ptr::Ptr{Cvoid}, length(ws)::Csize_t, DEFAULT_STREAM::Stream)::Int32)
    return loss[], gA, gB, ix, iy
end

# adjoint: indices are constants (`@ignore`, :45).  Zygote calls this method only when a gradient is wanted, so the forward
# already runs the fused value + gradient call with a cotangent of 1 -- `gradient(() -> chamfer_distance(A, B), ...)` and
# `withgradient` are ONE ABI call (ADVICE r4) -- and the pullback hands those arrays out when the incoming cotangent is
# exactly 1 (a plain gradient, the chamfer term of fit_mesh's weighted sum).  Any other cotangent re-runs the adjoint kernel
# with it on the kept indices: the same bits as the two-call form, never a rescaling of the unit gradient.
Zygote.@adjoint function _chamfer_distance(A::HipArray{Float32,3}, B::HipArray{Float32,3}, w1::Float32, w2::Float32)
    loss, gA, gB, ix, iy = chamfer_value_and_grad(A, B, w1, w2; indices = true)
    handed_out = Ref(false)
    function back(g)
        if Float32(g) == 1.0f0
            # the first pullback hands out the arrays of the forward; a second one (a jacobian, a caller that mutated the
            # first result in place) gets copies of its own (ADVICE r5)
            handed_out[] && return (copy(gA), copy(gB), nothing, nothing)
            handed_out[] = true
            return (gA, gB, nothing, nothing)
        end
        D, N, Bn = size(A); _, M, _ = size(B)
        hA = HipArray{Float32}(undef, D, N, Bn); hB = HipArray{Float32}(undef, D, M, Bn)
        check(@ccall LIB.fx3d_chamfer_bwd(A.ptr::Ptr{Cvoid}, N::Int32, B.ptr::Ptr{Cvoid}, M::Int32, Bn::Int32, D::Int32,
                                          ix.ptr::Ptr{Cvoid}, iy.ptr::Ptr{Cvoid}, w1::Float32, w2::Float32,
                                          Float32(g)::Float32, Bn::Int64, hA.ptr::Ptr{Cvoid}, hB.ptr::Ptr{Cvoid},
                                          DEFAULT_STREAM::Stream)::Int32)
        return (hA, hB, nothing, nothing)
    end
    return loss, back
end

# adjoint of chamfer_distance(m1::TriMesh, m2::TriMesh, n) (src/metrics/mesh.jl:34-44) w.r.t. the padded vertices of m1 and / or
# m2 in one launch: A / B = the forward's samples, ix / iy its neighbour indices, draws_* = (face, r1, r2) of the sampler.
# `nothing` for a mesh skips its side.  Returns (gverts1, gverts2), each (3, V, N) or nothing.
# The loss with the reference's own Float32 pairwise `mean` (src/metrics/pcloud.jl:47-50), from the forward's indices
# (0-based Int32 device arrays as _chamfer_fwd(...; indices = true) returns them): identical bits to the CPU path's `mean`.
function chamfer_loss_pairwise(A::HipArray{Float32,3}, B::HipArray{Float32,3}, ix::HipArray{Int32,2}, iy::HipArray{Int32,2},
                               w1::Float32 = 1.0f0, w2::Float32 = 1.0f0)
    D, N, Bn = size(A); M = size(B, 2)
    nb = Ref{Csize_t}(0)
    check(@ccall LIB.fx3d_chamfer_pairwise_workspace_bytes(N::Int32, M::Int32, Bn::Int32, D::Int32, nb::Ref{Csize_t})::Int32)
    ws = workspace(nb[])
    loss_dev = HipArray{Float32,1}(undef, (1,)); host = Ref{Float32}(0)
    check(@ccall LIB.fx3d_chamfer_loss_pairwise_f32(A.ptr::Ptr{Cvoid}, N::Int32, B.ptr::Ptr{Cvoid}, M::Int32, Bn::Int32, D::Int32,
                                                    ix.ptr::Ptr{Cvoid}, iy.ptr::Ptr{Cvoid}, w1::Float32, w2::Float32,
                                                    loss_dev.ptr::Ptr{Cvoid}, host::Ref{Float32}, ws.ptr::Ptr{Cvoid},
                                                    length(ws)::Csize_t, DEFAULT_STREAM::Stream)::Int32)
    return host[]
end

function chamfer_sampled_grad(A::HipArray{Float32,3}, B::HipArray{Float32,3}, ix::HipArray{Int32,2}, iy::HipArray{Int32,2},
                              m1, draws1, m2, draws2; w1::Number = 1.0, w2::Number = 1.0, gout::Number = 1)
    _, N, Bn = size(A); _, M, _ = size(B)
    side(m, d) = m === nothing ? (C_NULL, Int32(0), Int32(0), C_NULL, C_NULL, C_NULL, nothing, C_NULL, C_NULL) :
        (faces_padded_dev(m).ptr, Int32(m.V), Int32(m.F), d[1].ptr, d[2].ptr, d[3].ptr, HipArray{Float32}(undef, 3, m.V, m.N),
         vertex_faces_dev(m)[1].ptr, vertex_faces_dev(m)[2].ptr)
    f1, V1, F1, fi1, ra1, rb1, g1, vr1, ve1 = side(m1, draws1)
    f2, V2, F2, fi2, ra2, rb2, g2, vr2, ve2 = side(m2, draws2)
    nb = Ref{Csize_t}(0)   # the ordered form's scratch (rows of both sides + the per-entry sums); the library falls back to the
    check(@ccall LIB.fx3d_chamfer_sampled_bwd_workspace_bytes(N::Int32, M::Int32, Bn::Int32, nb::Ref{Csize_t})::Int32)   # float-atomic
                                                                                                                             # scatter beyond its limits
    ws = workspace(nb[])
    check(@ccall LIB.fx3d_chamfer_sampled_bwd(A.ptr::Ptr{Cvoid}, N::Int32, B.ptr::Ptr{Cvoid}, M::Int32, Bn::Int32,
                                              ix.ptr::Ptr{Cvoid}, iy.ptr::Ptr{Cvoid}, Float32(w1)::Float32, Float32(w2)::Float32,
                                              Float32(gout)::Float32, Bn::Int64, f1::Ptr{Cvoid}, V1::Int32, F1::Int32,
                                              fi1::Ptr{Cvoid}, ra1::Ptr{Cvoid}, rb1::Ptr{Cvoid},
                                              (g1 === nothing ? C_NULL : g1.ptr)::Ptr{Cvoid}, f2::Ptr{Cvoid}, V2::Int32, F2::Int32,
                                              fi2::Ptr{Cvoid}, ra2::Ptr{Cvoid}, rb2::Ptr{Cvoid},
                                              (g2 === nothing ? C_NULL : g2.ptr)::Ptr{Cvoid}, 0::Int32, vr1::Ptr{Cvoid},
                                              ve1::Ptr{Cvoid}, vr2::Ptr{Cvoid}, ve2::Ptr{Cvoid}, ws.ptr::Ptr{Cvoid},
                                              length(ws)::Csize_t, DEFAULT_STREAM::Stream)::Int32)
    return g1, g2
end

# The last launch of a fit_mesh iteration (examples/fit_mesh.jl:106-110) for source meshes of equal vertex counts (one mesh: always): the pullback of
# chamfer_distance(offset(src, x), tgt, n) onto the source's vertices, added to `g` (the regularisers' gradient), and
# Flux.Optimise.Momentum(eta, rho) + offset applied to every finished row by the thread that holds it:
#   vel = rho vel - eta g;  x += vel;  out = base + x;  counter += inc
function chamfer_sampled_grad_step!(g::HipArray{Float32}, A::HipArray{Float32,3}, B::HipArray{Float32,3}, ix::HipArray{Int32,2},
                                    iy::HipArray{Int32,2}, m, draws, x::HipArray{Float32}, vel::HipArray{Float32},
                                    base::HipArray{Float32}, out::HipArray{Float32}; eta = 1.0, rho = 0.9, w1::Number = 1.0,
                                    w2::Number = 1.0, gout::Number = 1, counter = C_NULL, inc::Integer = 0)
    _, N, Bn = size(A); _, M, _ = size(B)
    vfr, vfe = vertex_faces_dev(m)
    nb = Ref{Csize_t}(0)
    check(@ccall LIB.fx3d_chamfer_sampled_bwd_workspace_bytes(N::Int32, M::Int32, Bn::Int32, nb::Ref{Csize_t})::Int32)
    ws = workspace(nb[])
    check(@ccall LIB.fx3d_chamfer_sampled_bwd_step(A.ptr::Ptr{Cvoid}, N::Int32, B.ptr::Ptr{Cvoid}, M::Int32, Bn::Int32, ix.ptr::Ptr{Cvoid},
                                                   iy.ptr::Ptr{Cvoid}, Float32(w1)::Float32, Float32(w2)::Float32,
                                                   Float32(gout)::Float32, faces_padded_dev(m).ptr::Ptr{Cvoid}, m.V::Int32,
                                                   m.F::Int32, draws[1].ptr::Ptr{Cvoid}, draws[2].ptr::Ptr{Cvoid},
                                                   draws[3].ptr::Ptr{Cvoid}, g.ptr::Ptr{Cvoid}, 1::Int32, vfr.ptr::Ptr{Cvoid},
                                                   vfe.ptr::Ptr{Cvoid}, Float32(rho)::Float32, Float32(eta)::Float32,
                                                   vel.ptr::Ptr{Cvoid}, x.ptr::Ptr{Cvoid}, base.ptr::Ptr{Cvoid}, out.ptr::Ptr{Cvoid},
                                                   (counter isa HipArray ? counter.ptr : counter)::Ptr{Cvoid}, UInt64(inc)::UInt64,
                                                   ws.ptr::Ptr{Cvoid}, length(ws)::Csize_t, DEFAULT_STREAM::Stream)::Int32)
    return out
end

# ---- the fit iteration's regularisers as passengers of its sampling launches (include/flux3d_hip.h: fx3d_mesh_reg) ----
# examples/fit_mesh.jl:80-83: 0.1 laplacian_loss(m) + edge_loss(m).  Field for field the C struct (isbits: passed by reference).
struct MeshRegC
    verts::Ptr{Cvoid}; V::Int64
    rowptr::Ptr{Cvoid}; colind::Ptr{Cvoid}; vals::Ptr{Cvoid}
    edges::Ptr{Cvoid}; E::Int64
    target::Float32; w_lap::Float32; w_edge::Float32
    base_dev::Ptr{Cvoid}; loss_lap_dev::Ptr{Cvoid}; loss_edge_dev::Ptr{Cvoid}; total_dev::Ptr{Cvoid}
    ws::Ptr{Cvoid}; ws_bytes::Csize_t
end
# `out` (3 Float32 on the device) receives laplacian_loss, edge_loss and ((base + w_lap lap) + w_edge edge); `ws`: mesh_losses_workspace(m)
function mesh_reg(m::TriMesh{Float32,R,HipArray}, ws::HipArray{UInt8}, out::HipArray{Float32}; target::Number = 0, w_lap::Number = 0.1,
                  w_edge::Number = 1.0, base::Union{Nothing,HipArray{Float32}} = nothing) where {R}
    verts = get_verts_packed(m)::HipArray{Float32,2}
    rowptr, colind, vals = laplacian_csr_dev(m); edges = edges_dev(m)
    return MeshRegC(verts.ptr, size(verts, 2), rowptr.ptr, colind.ptr, vals.ptr, edges.ptr, size(edges, 1), Float32(target),
                    Float32(w_lap), Float32(w_edge), base === nothing ? C_NULL : base.ptr, out.ptr, out.ptr + 4, out.ptr + 8, ws.ptr,
                    length(ws))
end
# the draws of chamfer_distance(m1, m2, n) with the regularisers' forward of m1 riding in the launch; also returns m1's draws
function sample_points_pair_reg(m1::TriMesh{Float32,R1,HipArray}, m2::TriMesh{Float32,R2,HipArray}, reg::MeshRegC, n::Int = 5000;
                                eps::Number = Flux3D.EPS, seed1::UInt64 = rand(UInt64), seed2::UInt64 = seed1 + 1) where {R1,R2}
    v1 = get_verts_padded(m1); v2 = get_verts_padded(m2)
    f1 = faces_padded_dev(m1); l1 = faces_len_dev(m1); f2 = faces_padded_dev(m2); l2 = faces_len_dev(m2)
    nb1 = Ref{Csize_t}(0); nb2 = Ref{Csize_t}(0)
    check(@ccall LIB.fx3d_sample_points_workspace_bytes(m1.F::Int32, m1.N::Int32, nb1::Ref{Csize_t})::Int32)
    check(@ccall LIB.fx3d_sample_points_workspace_bytes(m2.F::Int32, m2.N::Int32, nb2::Ref{Csize_t})::Int32)
    w1 = HipArray{UInt8,1}(undef, (Int(nb1[]),)); w2 = HipArray{UInt8,1}(undef, (Int(nb2[]),))
    check(@ccall LIB.fx3d_sample_points_cdf_pair(v1.ptr::Ptr{Cvoid}, m1.V::Int32, f1.ptr::Ptr{Cvoid}, m1.F::Int32, l1.ptr::Ptr{Cvoid},
                                                 m1.N::Int32, w1.ptr::Ptr{Cvoid}, length(w1)::Csize_t, v2.ptr::Ptr{Cvoid}, m2.V::Int32,
                                                 f2.ptr::Ptr{Cvoid}, m2.F::Int32, l2.ptr::Ptr{Cvoid}, m2.N::Int32, w2.ptr::Ptr{Cvoid},
                                                 length(w2)::Csize_t, Float64(eps)::Float64, DEFAULT_STREAM::Stream)::Int32)
    o1 = HipArray{Float32,3}(undef, (3, n, m1.N)); o2 = HipArray{Float32,3}(undef, (3, n, m2.N))
    fi = HipArray{Int32,2}(undef, (n, m1.N)); ra = HipArray{Float32,2}(undef, (n, m1.N)); rb = HipArray{Float32,2}(undef, (n, m1.N))
    check(@ccall LIB.fx3d_sample_points_draw_pair_reg(v1.ptr::Ptr{Cvoid}, m1.V::Int32, f1.ptr::Ptr{Cvoid}, m1.F::Int32, l1.ptr::Ptr{Cvoid},
                                                      m1.N::Int32, n::Int32, seed1::UInt64, w1.ptr::Ptr{Cvoid}, length(w1)::Csize_t,
                                                      o1.ptr::Ptr{Cvoid}, fi.ptr::Ptr{Cvoid}, ra.ptr::Ptr{Cvoid}, rb.ptr::Ptr{Cvoid},
                                                      v2.ptr::Ptr{Cvoid}, m2.V::Int32, f2.ptr::Ptr{Cvoid}, m2.F::Int32, l2.ptr::Ptr{Cvoid},
                                                      m2.N::Int32, n::Int32, seed2::UInt64, w2.ptr::Ptr{Cvoid}, length(w2)::Csize_t,
                                                      o2.ptr::Ptr{Cvoid}, C_NULL::Ptr{Cvoid}, C_NULL::Ptr{Cvoid}, C_NULL::Ptr{Cvoid},
                                                      C_NULL::Ptr{Cvoid}, Ref(reg)::Ref{MeshRegC}, DEFAULT_STREAM::Stream)::Int32)
    return o1, o2, (fi, ra, rb)
end
# chamfer_sampled_grad_step! with the regularisers' adjoint (and the objective's sum) riding in its first launch: g is overwritten
function chamfer_sampled_grad_step_reg!(g::HipArray{Float32}, A::HipArray{Float32,3}, B::HipArray{Float32,3}, ix::HipArray{Int32,2},
                                        iy::HipArray{Int32,2}, m, draws, reg::MeshRegC, x::HipArray{Float32}, vel::HipArray{Float32},
                                        base::HipArray{Float32}, out::HipArray{Float32}; eta = 1.0, rho = 0.9, w1::Number = 1.0,
                                        w2::Number = 1.0, gout::Number = 1, counter = C_NULL, inc::Integer = 0)
    _, N, Bn = size(A); _, M, _ = size(B)
    vfr, vfe = vertex_faces_dev(m)
    nb = Ref{Csize_t}(0)
    check(@ccall LIB.fx3d_chamfer_sampled_bwd_workspace_bytes(N::Int32, M::Int32, Bn::Int32, nb::Ref{Csize_t})::Int32)
    ws = workspace(nb[])
    check(@ccall LIB.fx3d_chamfer_sampled_bwd_step_reg(A.ptr::Ptr{Cvoid}, N::Int32, B.ptr::Ptr{Cvoid}, M::Int32, Bn::Int32, ix.ptr::Ptr{Cvoid},
                                                       iy.ptr::Ptr{Cvoid}, Float32(w1)::Float32, Float32(w2)::Float32,
                                                       Float32(gout)::Float32, faces_padded_dev(m).ptr::Ptr{Cvoid}, m.V::Int32,
                                                       m.F::Int32, draws[1].ptr::Ptr{Cvoid}, draws[2].ptr::Ptr{Cvoid},
                                                       draws[3].ptr::Ptr{Cvoid}, g.ptr::Ptr{Cvoid}, 0::Int32, vfr.ptr::Ptr{Cvoid},
                                                       vfe.ptr::Ptr{Cvoid}, Float32(rho)::Float32, Float32(eta)::Float32,
                                                       vel.ptr::Ptr{Cvoid}, x.ptr::Ptr{Cvoid}, base.ptr::Ptr{Cvoid}, out.ptr::Ptr{Cvoid},
                                                       (counter isa HipArray ? counter.ptr : counter)::Ptr{Cvoid}, UInt64(inc)::UInt64,
                                                       ws.ptr::Ptr{Cvoid}, length(ws)::Csize_t, Ref(reg)::Ref{MeshRegC},
                                                       DEFAULT_STREAM::Stream)::Int32)
    return out
end

# ---- k-NN graph: replaces CreateSingleKNNGraph + the per-batch loop (src/models/dgcnn.jl:3-7,36) --
function knn_graph(X::HipArray{Float32,3}, K::Int)
    F, N, B = size(X)
    idx = HipArray{Int32}(undef, K, N, B)
    nb = Ref{Csize_t}(0)
    check(@ccall LIB.fx3d_knn_workspace_bytes(N::Int32, N::Int32, B::Int32, F::Int32, K::Int32, 1::Int32, nb::Ref{Csize_t})::Int32)
    if nb[] > 0      # feature space: the candidate clouds' statistics + fp16 image are built once per cloud (pre-pass)
        ws = workspace(nb[])
        check(@ccall LIB.fx3d_knn_ws(X.ptr::Ptr{Cvoid}, N::Int32, X.ptr::Ptr{Cvoid}, N::Int32, B::Int32, F::Int32, K::Int32,
                                     1::Int32, idx.ptr::Ptr{Cvoid}, C_NULL::Ptr{Cvoid}, ws.ptr::Ptr{Cvoid}, length(ws)::Csize_t,
                                     DEFAULT_STREAM::Stream)::Int32)
    else
        check(@ccall LIB.fx3d_knn(X.ptr::Ptr{Cvoid}, N::Int32, X.ptr::Ptr{Cvoid}, N::Int32, B::Int32, F::Int32,
                                  K::Int32, 1::Int32, idx.ptr::Ptr{Cvoid}, C_NULL::Ptr{Cvoid}, DEFAULT_STREAM::Stream)::Int32)
    end
    out = HipArray{Float32}(undef, F, K, N, B)
    check(@ccall LIB.fx3d_knn_gather(X.ptr::Ptr{Cvoid}, N::Int32, B::Int32, F::Int32, K::Int32, idx.ptr::Ptr{Cvoid},
                                     out.ptr::Ptr{Cvoid}, DEFAULT_STREAM::Stream)::Int32)
    return out      # (F,K,N,B), what `cat([CreateSingleKNNGraph(X[:,:,i],K) ...]..., dims=4)` builds at :36
end

# EdgeConv's graph build up to the MLP input (src/models/dgcnn.jl:32-51): self-kNN + cat(X, KNNGraph - X) +
# PermutedDimsArray + reshape in one library call; layout 1 = (K*N, 2F, B), layout 0 = (2F, K, N, B)
function edgeconv_graph(X::HipArray{Float32,3}, K::Int; layout::Int = 1)
    F, N, B = size(X)
    idx = HipArray{Int32}(undef, K, N, B)
    out = layout == 1 ? HipArray{Float32}(undef, K * N, 2F, B) : HipArray{Float32}(undef, 2F, K, N, B)
    check(@ccall LIB.fx3d_edgeconv_graph(X.ptr::Ptr{Cvoid}, N::Int32, B::Int32, F::Int32, K::Int32, layout::Int32,
                                         idx.ptr::Ptr{Cvoid}, out.ptr::Ptr{Cvoid}, DEFAULT_STREAM::Stream)::Int32)
    return out
end
# the graph is @nograd upstream (src/models/dgcnn.jl:9): only the repeated-X terms carry gradient
Zygote.@adjoint function edgeconv_graph(X::HipArray{Float32,3}, K::Int; layout::Int = 1)
    F, N, B = size(X)
    out = edgeconv_graph(X, K; layout = layout)
    function back(g)
        gx = HipArray{Float32}(undef, F, N, B)
        check(@ccall LIB.fx3d_edge_features_bwd(g.ptr::Ptr{Cvoid}, N::Int32, B::Int32, F::Int32, K::Int32,
                                                layout::Int32, gx.ptr::Ptr{Cvoid}, DEFAULT_STREAM::Stream)::Int32)
        return (gx, nothing)
    end
    return out, back
end

# pointcloud_to_voxel (src/conversions.jl:91-131) for device clouds -> (res,res,res,B) Float32 0/1
function pointcloud_to_voxel(p::PointCloud, res::Int = 32)
    pts = p.points::HipArray{Float32,3}
    _, N, B = size(pts)
    nb = Ref{Csize_t}(0); check(@ccall LIB.fx3d_voxel_workspace_bytes(B::Int32, nb::Ref{Csize_t})::Int32)
    ws = workspace(nb[]); vox = HipArray{Float32}(undef, res, res, res, B)
    check(@ccall LIB.fx3d_pointcloud_to_voxel(pts.ptr::Ptr{Cvoid}, N::Int32, B::Int32, res::Int32, vox.ptr::Ptr{Cvoid},
                                              ws.ptr::Ptr{Cvoid}, length(ws)::Csize_t, DEFAULT_STREAM::Stream)::Int32)
    return vox
end

# ---- TriMesh device mirrors: int32 0-based copies of the host integer data (cached per mesh) ---
const _mirror = WeakKeyDict{Any,Dict{Symbol,Any}}()
mirror(m::TriMesh) = get!(() -> Dict{Symbol,Any}(), _mirror, m)
# The reference's index arrays cross the ABI AS THEY ARE (`R` in {UInt32, Int64}, 1-based: src/rep/mesh.jl:70-98): the library
# stages the bytes and converts to its device form (Int32, 0-based) with one small kernel -- no host-side `Int32.(a .- 1)` pass.
# `clamp_pad`: the 0 padding of faces_padded becomes 0 (never dereferenced); `limit`: values outside [0, limit) are an error.
const IDX_I32 = Int32(0); const IDX_U32 = Int32(1); const IDX_I64 = Int32(2)
index_type(::Type{Int32}) = IDX_I32; index_type(::Type{UInt32}) = IDX_U32; index_type(::Type{Int64}) = IDX_I64
function index_upload(a::Array{R}; base::Integer = 1, clamp_pad::Bool = false, limit::Integer = 0) where {R<:Union{Int32,UInt32,Int64}}
    out = HipArray{Int32}(undef, size(a)...)
    isempty(a) && return out
    nb = Ref{Csize_t}(0)
    check(@ccall LIB.fx3d_index_upload_workspace_bytes(index_type(R)::Int32, length(a)::Int64, nb::Ref{Csize_t})::Int32)
    ws = HipArray{UInt8}(undef, Int(nb[])); bad = fill!(HipArray{UInt32}(undef, 1), 0)
    check(@ccall LIB.fx3d_index_upload(a::Ptr{Cvoid}, index_type(R)::Int32, Int32(base)::Int32, length(a)::Int64, Int32(clamp_pad)::Int32,
                                       Int64(limit)::Int64, out.ptr::Ptr{Cvoid}, bad.ptr::Ptr{Cvoid}, ws.ptr::Ptr{Cvoid},
                                       length(ws)::Csize_t, DEFAULT_STREAM::Stream)::Int32)
    nbad = unhip(bad)[1]
    nbad == 0 || throw(ArgumentError("index_upload: $nbad indices outside [0, $limit) after subtracting $base"))
    return out
end
index_upload(a::AbstractArray{<:Integer}; kw...) = index_upload(Array{Int64}(a); kw...)
# the same conversion for an index array that already lives on the device (e.g. faces a HipArray pipeline produced)
function index_convert(a::HipArray{R}; base::Integer = 1, clamp_pad::Bool = false, limit::Integer = 0) where {R<:Union{Int32,UInt32,Int64}}
    out = HipArray{Int32}(undef, size(a)...)
    isempty(a) && return out
    bad = fill!(HipArray{UInt32}(undef, 1), 0)   # out-of-range indices are counted, never silently rewritten (ADVICE r5)
    check(@ccall LIB.fx3d_index_convert(a.ptr::Ptr{Cvoid}, index_type(R)::Int32, Int32(base)::Int32, length(a)::Int64,
                                        Int32(clamp_pad)::Int32, Int64(limit)::Int64, out.ptr::Ptr{Cvoid}, bad.ptr::Ptr{Cvoid},
                                        DEFAULT_STREAM::Stream)::Int32)
    nbad = unhip(bad)[1]
    nbad == 0 || throw(ArgumentError("index_convert: $nbad indices outside the Int32 / [0, $limit) range after subtracting $base"))
    return out
end
faces_padded_dev(m) = get!(() -> index_upload(get_faces_padded(m); clamp_pad = true, limit = m.V), mirror(m), :faces_padded)
faces_len_dev(m) = get!(() -> hip(Int32.(m._faces_len)), mirror(m), :faces_len)
faces_packed_dev(m) = get!(() -> index_upload(get_faces_packed(m); limit = sum(m._verts_len)), mirror(m), :faces_packed)
edges_dev(m) = get!(() -> index_upload(get_edges_packed(m); limit = sum(m._verts_len)), mirror(m), :edges)          # (E,2) column-major
# vertex -> (face, corner) table of the padded batch (fx3d_build_vertex_faces): what the ordered -- atomic-free, bit-reproducible --
# adjoint of sample_points walks; built once per topology like the edge list (src/rep/mesh.jl:87-97)
function vertex_faces_dev(m)
    get!(mirror(m), :vertex_faces) do
        fp = Int32.(max.(Int64.(get_faces_padded(m)) .- 1, 0))       # (3, Fmax, B) 0-based, padding clamped (never read)
        fl = Int32.(m._faces_len)
        rowptr = Matrix{Int32}(undef, m.V + 1, m.N); ent = Matrix{Int32}(undef, 3 * m.F, m.N)
        check(@ccall LIB.fx3d_build_vertex_faces(fp::Ptr{Int32}, fl::Ptr{Int32}, m.V::Int32, m.F::Int32, m.N::Int32,
                                                 rowptr::Ptr{Int32}, ent::Ptr{Int32})::Int32)
        (hip(rowptr), hip(ent))
    end
end
function laplacian_csr_dev(m)
    get!(mirror(m), :lap) do
        # CSR of L == CSC of L' ; build from the reference's own cached SparseMatrixCSC (src/rep/mesh.jl:559-565)
        Lt = SparseMatrixCSC(transpose(get_laplacian_packed(m)))
        (hip(Int32.(Lt.colptr .- 1)), hip(Int32.(Lt.rowval .- 1)), hip(Float32.(Lt.nzval)))
    end
end

# ---- sample_points: replaces src/transforms/mesh_func.jl:21-58 for HipArray-backed meshes -------
function sample_points(m::TriMesh{Float32,R,HipArray}, num_samples::Int = 5000; eps::Number = Flux3D.EPS,
                       seed::UInt64 = rand(UInt64)) where {R}
    verts = get_verts_padded(m)::HipArray{Float32,3}
    nb = Ref{Csize_t}(0)
    check(@ccall LIB.fx3d_sample_points_workspace_bytes(m.F::Int32, m.N::Int32, nb::Ref{Csize_t})::Int32)
    ws = workspace(nb[])
    out = HipArray{Float32}(undef, 3, num_samples, m.N)
    check(@ccall LIB.fx3d_sample_points(verts.ptr::Ptr{Cvoid}, m.V::Int32, faces_padded_dev(m).ptr::Ptr{Cvoid}, m.F::Int32,
                                        faces_len_dev(m).ptr::Ptr{Cvoid}, m.N::Int32, num_samples::Int32,
                                        Float64(eps)::Float64, seed::UInt64, out.ptr::Ptr{Cvoid}, C_NULL::Ptr{Cvoid},
                                        C_NULL::Ptr{Cvoid}, C_NULL::Ptr{Cvoid}, ws.ptr::Ptr{Cvoid}, length(ws)::Csize_t,
                                        DEFAULT_STREAM::Stream)::Int32)
    return out
end

# ---- mesh losses: replace src/metrics/mesh.jl:9-32 (no `cpu(transpose(verts))` round trip) -------
# chamfer_distance(m1, m2, n) (src/metrics/mesh.jl:34-44) draws from both meshes: the two CDF builds in one launch, the two
# draws in one launch (identical samples to two sample_points calls with the same seeds)
function sample_points_pair(m1::TriMesh{Float32,R1,HipArray}, m2::TriMesh{Float32,R2,HipArray}, n::Int = 5000;
                            eps::Number = Flux3D.EPS, seed1::UInt64 = rand(UInt64), seed2::UInt64 = seed1 + 1) where {R1,R2}
    v1 = get_verts_padded(m1); v2 = get_verts_padded(m2)
    f1 = faces_padded_dev(m1); l1 = faces_len_dev(m1); f2 = faces_padded_dev(m2); l2 = faces_len_dev(m2)
    nb1 = Ref{Csize_t}(0); nb2 = Ref{Csize_t}(0)
    check(@ccall LIB.fx3d_sample_points_workspace_bytes(m1.F::Int32, m1.N::Int32, nb1::Ref{Csize_t})::Int32)
    check(@ccall LIB.fx3d_sample_points_workspace_bytes(m2.F::Int32, m2.N::Int32, nb2::Ref{Csize_t})::Int32)
    w1 = HipArray{UInt8,1}(undef, (Int(nb1[]),)); w2 = HipArray{UInt8,1}(undef, (Int(nb2[]),))
    check(@ccall LIB.fx3d_sample_points_cdf_pair(v1.ptr::Ptr{Cvoid}, m1.V::Int32, f1.ptr::Ptr{Cvoid}, m1.F::Int32, l1.ptr::Ptr{Cvoid},
                                                 m1.N::Int32, w1.ptr::Ptr{Cvoid}, length(w1)::Csize_t, v2.ptr::Ptr{Cvoid}, m2.V::Int32,
                                                 f2.ptr::Ptr{Cvoid}, m2.F::Int32, l2.ptr::Ptr{Cvoid}, m2.N::Int32, w2.ptr::Ptr{Cvoid},
                                                 length(w2)::Csize_t, Float64(eps)::Float64, DEFAULT_STREAM::Stream)::Int32)
    o1 = HipArray{Float32,3}(undef, (3, n, m1.N)); o2 = HipArray{Float32,3}(undef, (3, n, m2.N))
    check(@ccall LIB.fx3d_sample_points_draw_pair(v1.ptr::Ptr{Cvoid}, m1.V::Int32, f1.ptr::Ptr{Cvoid}, m1.F::Int32, l1.ptr::Ptr{Cvoid},
                                                  m1.N::Int32, n::Int32, seed1::UInt64, w1.ptr::Ptr{Cvoid}, length(w1)::Csize_t,
                                                  o1.ptr::Ptr{Cvoid}, C_NULL::Ptr{Cvoid}, C_NULL::Ptr{Cvoid}, C_NULL::Ptr{Cvoid},
                                                  v2.ptr::Ptr{Cvoid}, m2.V::Int32, f2.ptr::Ptr{Cvoid}, m2.F::Int32, l2.ptr::Ptr{Cvoid},
                                                  m2.N::Int32, n::Int32, seed2::UInt64, w2.ptr::Ptr{Cvoid}, length(w2)::Csize_t,
                                                  o2.ptr::Ptr{Cvoid}, C_NULL::Ptr{Cvoid}, C_NULL::Ptr{Cvoid}, C_NULL::Ptr{Cvoid},
                                                  C_NULL::Ptr{Cvoid}, DEFAULT_STREAM::Stream)::Int32)
    return o1, o2
end

function laplacian_loss(m::TriMesh{Float32,R,HipArray}) where {R}
    verts = get_verts_packed(m)::HipArray{Float32,2}
    rowptr, colind, vals = laplacian_csr_dev(m)
    nb = Ref{Csize_t}(0); check(@ccall LIB.fx3d_mesh_loss_workspace_bytes(size(verts, 2)::Int64, nb::Ref{Csize_t})::Int32)
    ws = workspace(nb[]); loss_dev = HipArray{Float32}(undef, 1); loss = Ref{Float32}(0)
    check(@ccall LIB.fx3d_laplacian_loss(verts.ptr::Ptr{Cvoid}, size(verts, 2)::Int64, rowptr.ptr::Ptr{Cvoid},
                                         colind.ptr::Ptr{Cvoid}, vals.ptr::Ptr{Cvoid}, loss_dev.ptr::Ptr{Cvoid},
                                         loss::Ref{Float32}, ws.ptr::Ptr{Cvoid}, length(ws)::Csize_t,
                                         DEFAULT_STREAM::Stream)::Int32)
    return loss[]
end

function edge_loss(m::TriMesh{Float32,R,HipArray}, target_length::Number = 0.0) where {R}
    verts = get_verts_packed(m)::HipArray{Float32,2}
    edges = edges_dev(m)
    nb = Ref{Csize_t}(0); check(@ccall LIB.fx3d_mesh_loss_workspace_bytes(size(edges, 1)::Int64, nb::Ref{Csize_t})::Int32)
    ws = workspace(nb[]); loss_dev = HipArray{Float32}(undef, 1); loss = Ref{Float32}(0)
    check(@ccall LIB.fx3d_edge_loss(verts.ptr::Ptr{Cvoid}, size(verts, 2)::Int64, edges.ptr::Ptr{Cvoid},
                                    size(edges, 1)::Int64, Float32(target_length)::Float32, loss_dev.ptr::Ptr{Cvoid},
                                    loss::Ref{Float32}, ws.ptr::Ptr{Cvoid}, length(ws)::Csize_t,
                                    DEFAULT_STREAM::Stream)::Int32)
    return loss[]
end

# ---- fit_mesh chain (examples/fit_mesh.jl:78-110): offset, packed<->padded and the three adjoints ----------
# verts + delta on device; `Flux3D.offset(m, delta)` (src/transforms/mesh_func.jl:409-438) for HipArray meshes
function lincomb(a::Number, x::HipArray{Float32}, b::Number, y::HipArray{Float32}, c::Number = 0,
                 z::Union{Nothing,HipArray{Float32}} = nothing)
    out = HipArray{Float32}(undef, size(x)...)
    check(@ccall LIB.fx3d_lincomb(length(x)::Int64, Float32(a)::Float32, x.ptr::Ptr{Cvoid}, Float32(b)::Float32,
                                  y.ptr::Ptr{Cvoid}, Float32(c)::Float32,
                                  (z === nothing ? C_NULL : z.ptr)::Ptr{Cvoid}, out.ptr::Ptr{Cvoid},
                                  DEFAULT_STREAM::Stream)::Int32)
    return out
end
offset_verts(m::TriMesh{Float32,R,HipArray}, delta::HipArray{Float32,2}) where {R} =
    lincomb(1, get_verts_packed(m)::HipArray{Float32,2}, 1, delta)
Zygote.@adjoint offset_verts(m, delta) = offset_verts(m, delta), g -> (nothing, g)

# Flux.Optimise.Momentum(eta, rho) on device arrays in one launch (examples/fit_mesh.jl:87-88,110)
function momentum_step!(x::HipArray{Float32}, v::HipArray{Float32}, g::HipArray{Float32}; eta = 1.0, rho = 0.9)
    check(@ccall LIB.fx3d_momentum_step(length(x)::Int64, Float32(rho)::Float32, Float32(eta)::Float32, g.ptr::Ptr{Cvoid},
                                        v.ptr::Ptr{Cvoid}, x.ptr::Ptr{Cvoid}, DEFAULT_STREAM::Stream)::Int32)
    return x
end

# ... fused with the next iteration's first steps: out = base + x (the offset mesh's packed vertices) and the seed counter
function momentum_step_offset!(x::HipArray{Float32}, v::HipArray{Float32}, g::HipArray{Float32}, base::HipArray{Float32},
                               out::HipArray{Float32}; eta = 1.0, rho = 0.9, counter = C_NULL, inc::Integer = 0)
    check(@ccall LIB.fx3d_momentum_step_offset(length(x)::Int64, Float32(rho)::Float32, Float32(eta)::Float32, g.ptr::Ptr{Cvoid},
                                               v.ptr::Ptr{Cvoid}, x.ptr::Ptr{Cvoid}, base.ptr::Ptr{Cvoid}, out.ptr::Ptr{Cvoid},
                                               (counter isa HipArray ? counter.ptr : counter)::Ptr{Cvoid}, UInt64(inc)::UInt64,
                                               DEFAULT_STREAM::Stream)::Int32)
    return out
end

# Stream capture: record the loop body once, replay it with one launch (hipGraph).  `f()` must only enqueue on
# `stream` (no host copies); run it once eagerly before capturing.
struct HipGraph; handle::Ptr{Cvoid}; end
function capture(f, stream::Stream)
    check(@ccall LIB.fx3d_graph_begin_capture(stream::Stream)::Int32)
    local h = Ref{Ptr{Cvoid}}(C_NULL)
    try
        f()
    finally
        check(@ccall LIB.fx3d_graph_end_capture(stream::Stream, h::Ref{Ptr{Cvoid}})::Int32)
    end
    return HipGraph(h[])
end
launch(g::HipGraph, stream::Stream) = check(@ccall LIB.fx3d_graph_launch(g.handle::Ptr{Cvoid}, stream::Stream)::Int32)
destroy(g::HipGraph) = check(@ccall LIB.fx3d_graph_destroy(g.handle::Ptr{Cvoid})::Int32)
# the per-replay part of a sampling seed: a device UInt64 the recorded graph advances itself
counter_add!(ctr::HipArray{UInt64}, inc::Integer, stream::Stream = DEFAULT_STREAM) =
    check(@ccall LIB.fx3d_counter_add(ctr.ptr::Ptr{Cvoid}, UInt64(inc)::UInt64, stream::Stream)::Int32)

# The two halves of sample_points: the CDF depends only on the mesh (keep it while the vertices do not change),
# the draw on (cdf, seed + seed_dev[]).
function sample_cdf(m::TriMesh{Float32,R,HipArray}, verts::HipArray{Float32,3}, eps = EPS) where {R}
    nb = Ref{Csize_t}(0)
    check(@ccall LIB.fx3d_sample_points_workspace_bytes(m.F::Int32, m.N::Int32, nb::Ref{Csize_t})::Int32)
    ws = HipArray{UInt8}(undef, nb[])
    check(@ccall LIB.fx3d_sample_points_cdf(verts.ptr::Ptr{Cvoid}, m.V::Int32, faces_padded_dev(m).ptr::Ptr{Cvoid}, m.F::Int32,
                                            faces_len_dev(m).ptr::Ptr{Cvoid}, m.N::Int32, Float64(eps)::Float64,
                                            ws.ptr::Ptr{Cvoid}, length(ws)::Csize_t, DEFAULT_STREAM::Stream)::Int32)
    return ws
end
function sample_draw(m::TriMesh{Float32,R,HipArray}, verts::HipArray{Float32,3}, cdf::HipArray{UInt8}, n::Int, seed::UInt64;
                     seed_dev::Union{Nothing,HipArray{UInt64}} = nothing) where {R}
    out = HipArray{Float32}(undef, 3, n, m.N); face = HipArray{Int32}(undef, n, m.N)
    r1 = HipArray{Float32}(undef, n, m.N); r2 = HipArray{Float32}(undef, n, m.N)
    check(@ccall LIB.fx3d_sample_points_draw(verts.ptr::Ptr{Cvoid}, m.V::Int32, faces_padded_dev(m).ptr::Ptr{Cvoid}, m.F::Int32,
                                             faces_len_dev(m).ptr::Ptr{Cvoid}, m.N::Int32, n::Int32, seed::UInt64,
                                             (seed_dev === nothing ? C_NULL : seed_dev.ptr)::Ptr{Cvoid}, cdf.ptr::Ptr{Cvoid},
                                             length(cdf)::Csize_t, out.ptr::Ptr{Cvoid}, face.ptr::Ptr{Cvoid},
                                             r1.ptr::Ptr{Cvoid}, r2.ptr::Ptr{Cvoid}, DEFAULT_STREAM::Stream)::Int32)
    return out, (face, r1, r2)
end

# src/rep/utils.jl:119-181 for (3,*) vertex arrays, device to device
function packed_to_padded(packed::HipArray{Float32,2}, verts_len::Vector{Int64}, Vmax::Int)
    out = HipArray{Float32}(undef, 3, Vmax, length(verts_len))
    check(@ccall LIB.fx3d_packed_to_padded(packed.ptr::Ptr{Cvoid}, verts_len::Ptr{Int64}, length(verts_len)::Int32,
                                           Vmax::Int32, out.ptr::Ptr{Cvoid}, DEFAULT_STREAM::Stream)::Int32)
    return out
end
function padded_to_packed(padded::HipArray{Float32,3}, verts_len::Vector{Int64})
    out = HipArray{Float32}(undef, 3, sum(verts_len))
    check(@ccall LIB.fx3d_padded_to_packed(padded.ptr::Ptr{Cvoid}, verts_len::Ptr{Int64}, length(verts_len)::Int32,
                                           size(padded, 2)::Int32, out.ptr::Ptr{Cvoid}, DEFAULT_STREAM::Stream)::Int32)
    return out
end
Zygote.@adjoint packed_to_padded(p, len, Vmax) = packed_to_padded(p, len, Vmax), g -> (padded_to_packed(g, len), nothing, nothing)

# sample_points keeping its draws, and the adjoint w.r.t. the padded vertices for the same draws
function sample_points_with_draws(m::TriMesh{Float32,R,HipArray}, verts::HipArray{Float32,3}, n::Int, eps, seed::UInt64) where {R}
    nb = Ref{Csize_t}(0)
    check(@ccall LIB.fx3d_sample_points_workspace_bytes(m.F::Int32, m.N::Int32, nb::Ref{Csize_t})::Int32)
    ws = workspace(nb[])
    out = HipArray{Float32}(undef, 3, n, m.N); face = HipArray{Int32}(undef, n, m.N)
    r1 = HipArray{Float32}(undef, n, m.N); r2 = HipArray{Float32}(undef, n, m.N)
    check(@ccall LIB.fx3d_sample_points(verts.ptr::Ptr{Cvoid}, m.V::Int32, faces_padded_dev(m).ptr::Ptr{Cvoid}, m.F::Int32,
                                        faces_len_dev(m).ptr::Ptr{Cvoid}, m.N::Int32, n::Int32, Float64(eps)::Float64,
                                        seed::UInt64, out.ptr::Ptr{Cvoid}, face.ptr::Ptr{Cvoid}, r1.ptr::Ptr{Cvoid},
                                        r2.ptr::Ptr{Cvoid}, ws.ptr::Ptr{Cvoid}, length(ws)::Csize_t,
                                        DEFAULT_STREAM::Stream)::Int32)
    return out, (face, r1, r2)
end
Zygote.@adjoint function sample_points_with_draws(m, verts, n, eps, seed)
    out, (face, r1, r2) = sample_points_with_draws(m, verts, n, eps, seed)
    function back(g)
        gv = HipArray{Float32}(undef, 3, m.V, m.N)
        # ordered form: every vertex's sum in a fixed order (no float atomics); meshes beyond its limits keep the scatter
        fits = Ref{Int32}(0)
        check(@ccall LIB.fx3d_sample_points_bwd_ordered(m.F::Int32, n::Int32, fits::Ref{Int32})::Int32)
        ordered = fits[] != 0
        vfr, vfe = ordered ? vertex_faces_dev(m) : (nothing, nothing)
        check(@ccall LIB.fx3d_sample_points_bwd(faces_padded_dev(m).ptr::Ptr{Cvoid}, m.V::Int32, m.F::Int32, m.N::Int32,
                                                n::Int32, face.ptr::Ptr{Cvoid}, r1.ptr::Ptr{Cvoid}, r2.ptr::Ptr{Cvoid},
                                                g[1].ptr::Ptr{Cvoid}, gv.ptr::Ptr{Cvoid}, 0::Int32,
                                                (ordered ? vfr.ptr : C_NULL)::Ptr{Cvoid}, (ordered ? vfe.ptr : C_NULL)::Ptr{Cvoid},
                                                DEFAULT_STREAM::Stream)::Int32)
        return (nothing, gv, nothing, nothing, nothing)
    end
    return (out, (face, r1, r2)), back
end

# d laplacian_loss / d verts_packed and d edge_loss / d verts_packed (Zygote through src/metrics/mesh.jl:9-32)
function laplacian_loss_grad(m::TriMesh{Float32,R,HipArray}, verts::HipArray{Float32,2}, gout::Number = 1) where {R}
    rowptr, colind, vals = laplacian_csr_dev(m)
    g = HipArray{Float32}(undef, size(verts)...)
    # the mesh's L is the Laplacian of an undirected edge list (src/rep/mesh.jl:957-1002): structurally symmetric, so the
    # atomic-free gather form applies; fx3d_laplacian_loss_bwd (any CSR, scatter) is bound below for raw matrices
    check(@ccall LIB.fx3d_laplacian_loss_bwd_sym(verts.ptr::Ptr{Cvoid}, size(verts, 2)::Int64, rowptr.ptr::Ptr{Cvoid},
                                                 colind.ptr::Ptr{Cvoid}, vals.ptr::Ptr{Cvoid}, Float32(gout)::Float32,
                                                 g.ptr::Ptr{Cvoid}, 0::Int32, C_NULL::Ptr{Cvoid},
                                                 DEFAULT_STREAM::Stream)::Int32)
    return g
end
# the same adjoint for ANY CSR (asymmetric / pruned / directed): row-by-row scatter with float atomics
function laplacian_loss_grad_csr(verts::HipArray{Float32,2}, rowptr::HipArray{Int32,1}, colind::HipArray{Int32,1},
                                 vals::HipArray{Float32,1}, gout::Number = 1)
    g = HipArray{Float32}(undef, size(verts)...)
    check(@ccall LIB.fx3d_laplacian_loss_bwd(verts.ptr::Ptr{Cvoid}, size(verts, 2)::Int64, rowptr.ptr::Ptr{Cvoid},
                                             colind.ptr::Ptr{Cvoid}, vals.ptr::Ptr{Cvoid}, Float32(gout)::Float32,
                                             g.ptr::Ptr{Cvoid}, 0::Int32, DEFAULT_STREAM::Stream)::Int32)
    return g
end
function edge_loss_grad(m::TriMesh{Float32,R,HipArray}, verts::HipArray{Float32,2}, target::Number = 0, gout::Number = 1) where {R}
    # gather over the vertex adjacency (the Laplacian's rowptr / colind of the same edge list): one launch, no float atomics
    rowptr, colind, _ = laplacian_csr_dev(m)
    g = HipArray{Float32}(undef, size(verts)...)
    check(@ccall LIB.fx3d_edge_loss_bwd_adj(verts.ptr::Ptr{Cvoid}, size(verts, 2)::Int64, rowptr.ptr::Ptr{Cvoid},
                                            colind.ptr::Ptr{Cvoid}, size(get_edges_packed(m), 1)::Int64,
                                            Float32(target)::Float32, Float32(gout)::Float32, g.ptr::Ptr{Cvoid}, 0::Int32,
                                            DEFAULT_STREAM::Stream)::Int32)
    return g
end
# the same adjoint from a bare (E,2) edge list on the device (scatter, float atomics)
function edge_loss_grad_edges(verts::HipArray{Float32,2}, edges::HipArray{Int32,2}, target::Number = 0, gout::Number = 1)
    g = HipArray{Float32}(undef, size(verts)...)
    check(@ccall LIB.fx3d_edge_loss_bwd(verts.ptr::Ptr{Cvoid}, size(verts, 2)::Int64, edges.ptr::Ptr{Cvoid},
                                        size(edges, 1)::Int64, Float32(target)::Float32, Float32(gout)::Float32,
                                        g.ptr::Ptr{Cvoid}, 0::Int32, DEFAULT_STREAM::Stream)::Int32)
    return g
end

# Both regularisers of the fit_mesh objective in ONE launch and both adjoints in ONE gather launch (no float atomics:
# gradients are bit-reproducible).  `ws` keeps the Laplacian's unit rows between the two calls (same mesh, same vertices).
function mesh_losses_workspace(m::TriMesh{Float32,R,HipArray}) where {R}
    nb = Ref{Csize_t}(0)
    check(@ccall LIB.fx3d_mesh_losses_workspace_bytes(size(get_verts_packed(m), 2)::Int64, size(get_edges_packed(m), 1)::Int64,
                                                      nb::Ref{Csize_t})::Int32)
    return HipArray{UInt8}(undef, Int(nb[]))
end
function mesh_losses(m::TriMesh{Float32,R,HipArray}, ws::HipArray{UInt8}; target::Number = 0, w_lap::Number = 0.1,
                     w_edge::Number = 1.0, base::Union{Nothing,HipArray{Float32}} = nothing) where {R}
    verts = get_verts_packed(m)::HipArray{Float32,2}
    rowptr, colind, vals = laplacian_csr_dev(m); edges = edges_dev(m)
    out = HipArray{Float32}(undef, 3)
    check(@ccall LIB.fx3d_mesh_losses(verts.ptr::Ptr{Cvoid}, size(verts, 2)::Int64, rowptr.ptr::Ptr{Cvoid}, colind.ptr::Ptr{Cvoid},
                                      vals.ptr::Ptr{Cvoid}, edges.ptr::Ptr{Cvoid}, size(edges, 1)::Int64, Float32(target)::Float32,
                                      Float32(w_lap)::Float32, Float32(w_edge)::Float32,
                                      (base === nothing ? C_NULL : base.ptr)::Ptr{Cvoid}, out.ptr::Ptr{Cvoid},
                                      (out.ptr + 4)::Ptr{Cvoid}, (out.ptr + 8)::Ptr{Cvoid}, ws.ptr::Ptr{Cvoid}, length(ws)::Csize_t,
                                      DEFAULT_STREAM::Stream)::Int32)
    return out      # (laplacian_loss, edge_loss, (base + w_lap*lap) + w_edge*edge) on the device
end
function mesh_losses_grad(m::TriMesh{Float32,R,HipArray}, ws::HipArray{UInt8}; target::Number = 0, g_lap::Number = 0.1,
                          g_edge::Number = 1.0, reuse_forward::Bool = false,
                          out::Union{Nothing,HipArray{Float32,2}} = nothing) where {R}
    verts = get_verts_packed(m)::HipArray{Float32,2}
    rowptr, colind, vals = laplacian_csr_dev(m)
    g = out === nothing ? HipArray{Float32}(undef, size(verts)...) : out
    check(@ccall LIB.fx3d_mesh_losses_bwd(verts.ptr::Ptr{Cvoid}, size(verts, 2)::Int64, rowptr.ptr::Ptr{Cvoid}, colind.ptr::Ptr{Cvoid},
                                          vals.ptr::Ptr{Cvoid}, size(get_edges_packed(m), 1)::Int64, Float32(target)::Float32,
                                          Float32(g_lap)::Float32, Float32(g_edge)::Float32, (reuse_forward ? 1 : 0)::Int32,
                                          g.ptr::Ptr{Cvoid}, (out === nothing ? 0 : 1)::Int32, ws.ptr::Ptr{Cvoid},
                                          length(ws)::Csize_t, DEFAULT_STREAM::Stream)::Int32)
    return g
end

# ---- multi-GPU (one Julia process per GPU): RCCL through the C ABI, SURVEY.md 8e ----------------------
# rank 0: id = comm_unique_id(); ship the 128 bytes to the other ranks (Distributed.jl, MPI.jl, a file);
# every rank: comm = comm_init(nranks, id, rank) after fx3d_set_device(local_rank).
function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    check(@ccall LIB.fx3d_comm_unique_id(id::Ptr{UInt8})::Int32)
    return id
end
function comm_init(nranks::Integer, id::Vector{UInt8}, rank::Integer)
    c = Ref{Ptr{Cvoid}}(C_NULL)
    check(@ccall LIB.fx3d_comm_init_rank(c::Ref{Ptr{Cvoid}}, nranks::Int32, id::Ptr{UInt8}, rank::Int32)::Int32)
    return c[]
end
# the whole rendezvous inside the library (no MPI.jl / Distributed.jl): "tcp://host:port" or "file://path"
function comm_bootstrap(nranks::Integer, rank::Integer, rendezvous::AbstractString)
    c = Ref{Ptr{Cvoid}}(C_NULL)
    check(@ccall LIB.fx3d_comm_bootstrap(c::Ref{Ptr{Cvoid}}, nranks::Int32, rank::Int32, rendezvous::Cstring)::Int32)
    return c[]
end
# the rendezvous alone: rank 0's id arrives in every rank's buffer
function comm_exchange_id!(id::Vector{UInt8}, nranks::Integer, rank::Integer, rendezvous::AbstractString)
    length(id) == 128 || error("the RCCL unique id has 128 bytes")
    check(@ccall LIB.fx3d_comm_exchange_id(id::Ptr{UInt8}, nranks::Int32, rank::Int32, rendezvous::Cstring)::Int32)
    return id
end
function comm_info(c)
    n = Ref{Int32}(0); r = Ref{Int32}(0); v = Ref{Int32}(0)
    check(@ccall LIB.fx3d_comm_info(c::Ptr{Cvoid}, n::Ref{Int32}, r::Ref{Int32}, v::Ref{Int32})::Int32)
    return (nranks = Int(n[]), rank = Int(r[]), rccl_version = Int(v[]))
end
comm_destroy(c) = check(@ccall LIB.fx3d_comm_destroy(c::Ptr{Cvoid})::Int32)
# all-reduce(max) of a Float64 device buffer: the control plane of a timing harness (max over ranks; any buffer: a barrier)
comm_allreduce_max!(c, buf::HipArray{Float64}; stream::Stream = DEFAULT_STREAM) =
    check(@ccall LIB.fx3d_comm_allreduce_max_f64(c::Ptr{Cvoid}, buf.ptr::Ptr{Cvoid}, length(buf)::Int64, stream::Stream)::Int32)

# ---- ONE Julia process, several GPUs (the reference is one process: src/metrics/pcloud.jl:54-70) --------------------
# m = comm_init_all(8); shards: set_device(d - 1) before hip(...) so that slab d lives on device d - 1;
# loss = chamfer_distance_multi(m, As, Bs, B_global): every device runs kernel -> all-reduce(2 x Float64) -> finalise on a
# worker thread + stream of the library; the call returns the global loss (read back from the first device).
struct MultiDevice; handle::Ptr{Cvoid}; ndev::Int; end
function comm_init_all(ndev::Integer, devices::Union{Nothing,Vector{Int32}} = nothing)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    devs = devices === nothing ? Ptr{Int32}(C_NULL) : pointer(devices)
    GC.@preserve devices check(@ccall LIB.fx3d_comm_init_all(h::Ref{Ptr{Cvoid}}, ndev::Int32, devs::Ptr{Int32})::Int32)
    return MultiDevice(h[], Int(ndev))
end
multi_destroy(m::MultiDevice) = check(@ccall LIB.fx3d_multi_destroy(m.handle::Ptr{Cvoid})::Int32)
multi_synchronize(m::MultiDevice) = check(@ccall LIB.fx3d_multi_sync(m.handle::Ptr{Cvoid})::Int32)
function multi_info(m::MultiDevice)
    n = Ref{Int32}(0); v = Ref{Int32}(0); devs = Vector{Int32}(undef, m.ndev)
    check(@ccall LIB.fx3d_multi_info(m.handle::Ptr{Cvoid}, n::Ref{Int32}, devs::Ptr{Int32}, v::Ref{Int32})::Int32)
    return (ndev = Int(n[]), devices = Int.(devs), rccl_version = Int(v[]))
end
function chamfer_distance_multi(m::MultiDevice, As::Vector{<:Union{Nothing,HipArray{Float32,3}}},
                                Bs::Vector{<:Union{Nothing,HipArray{Float32,3}}}, B_global::Integer;
                                w1::Number = 1.0, w2::Number = 1.0)
    length(As) == m.ndev && length(Bs) == m.ndev || error("one shard (or nothing) per device")
    k = findfirst(!isnothing, As)
    k === nothing && error("no shard at all")
    D, N, _ = size(As[k]); M = size(Bs[k], 2)
    xs = Ptr{Cvoid}[a === nothing ? C_NULL : a.ptr for a in As]
    ys = Ptr{Cvoid}[b === nothing ? C_NULL : b.ptr for b in Bs]
    bl = Int32[a === nothing ? 0 : size(a, 3) for a in As]
    loss = Ref{Float32}(0)
    GC.@preserve As Bs check(@ccall LIB.fx3d_chamfer_fwd_multi(m.handle::Ptr{Cvoid}, xs::Ptr{Ptr{Cvoid}}, N::Int32, ys::Ptr{Ptr{Cvoid}},
                                                               M::Int32, bl::Ptr{Int32}, D::Int32, B_global::Int64, Float32(w1)::Float32,
                                                               Float32(w2)::Float32, loss::Ref{Float32}, C_NULL::Ptr{Ptr{Cvoid}})::Int32)
    return loss[]
end

# chamfer_distance of a batch whose slab [start, start+B_local) lives on this rank; every rank gets the
# global loss (kernel -> all-reduce of 2 Float64 -> finalise with B_global), cf. src/metrics/pcloud.jl:39-52
function chamfer_distance_sharded(comm, A::HipArray{Float32,3}, B::HipArray{Float32,3}, B_global::Integer;
                                  w1::Number = 1.0, w2::Number = 1.0)
    D, N, Bl = size(A); _, M, _ = size(B)
    nb = Ref{Csize_t}(0)
    check(@ccall LIB.fx3d_chamfer_workspace_bytes(N::Int32, M::Int32, max(Bl, 1)::Int32, D::Int32, nb::Ref{Csize_t})::Int32)
    ws = workspace(nb[])
    sums = HipArray{Float64}(undef, 2); loss_dev = HipArray{Float32}(undef, 1); loss = Ref{Float32}(0)
    check(@ccall LIB.fx3d_chamfer_fwd_sharded(comm::Ptr{Cvoid}, A.ptr::Ptr{Cvoid}, N::Int32, B.ptr::Ptr{Cvoid}, M::Int32,
                                              Bl::Int32, D::Int32, B_global::Int64, Float32(w1)::Float32,
                                              Float32(w2)::Float32, sums.ptr::Ptr{Cvoid}, loss_dev.ptr::Ptr{Cvoid},
                                              loss::Ref{Float32}, ws.ptr::Ptr{Cvoid}, length(ws)::Csize_t,
                                              DEFAULT_STREAM::Stream)::Int32)
    return loss[]
end

# Overlapped form: kernel on `stream`, all-reduce + finalise on `comm_stream`; `slot` = (sums, loss_dev, ready, done) of a
# small ring owned by the caller.  Returns at once; event_synchronize(slot.done) before reading slot.loss_dev.
function chamfer_distance_sharded_async(comm, A::HipArray{Float32,3}, B::HipArray{Float32,3}, B_global::Integer, slot,
                                        stream::Stream, comm_stream::Stream; w1::Number = 1.0, w2::Number = 1.0)
    D, N, Bl = size(A); _, M, _ = size(B)
    nb = Ref{Csize_t}(0)
    check(@ccall LIB.fx3d_chamfer_workspace_bytes(N::Int32, M::Int32, max(Bl, 1)::Int32, D::Int32, nb::Ref{Csize_t})::Int32)
    ws = workspace(nb[])
    check(@ccall LIB.fx3d_chamfer_fwd_sharded_async(comm::Ptr{Cvoid}, A.ptr::Ptr{Cvoid}, N::Int32, B.ptr::Ptr{Cvoid}, M::Int32,
                                                    Bl::Int32, D::Int32, B_global::Int64, Float32(w1)::Float32,
                                                    Float32(w2)::Float32, slot.sums.ptr::Ptr{Cvoid}, slot.loss_dev.ptr::Ptr{Cvoid},
                                                    ws.ptr::Ptr{Cvoid}, length(ws)::Csize_t, stream::Stream, comm_stream::Stream,
                                                    slot.ready::Event, slot.done::Event)::Int32)
    return slot
end

# Deferred form for evaluation loops: `sums` (2,count) Float64 holds one slot per sharded batch
# (fx3d_chamfer_sums writes a slot), one all-reduce carries them all, one kernel finalises `count` losses.
function chamfer_finalize_many(comm, sums::HipArray{Float64,2}, N::Integer, M::Integer, B_global::Integer, D::Integer;
                               w1::Number = 1.0, w2::Number = 1.0)
    count = size(sums, 2)
    check(@ccall LIB.fx3d_comm_allreduce_sum_f64(comm::Ptr{Cvoid}, sums.ptr::Ptr{Cvoid}, (2 * count)::Int64,
                                                 DEFAULT_STREAM::Stream)::Int32)
    losses = HipArray{Float32}(undef, count)
    check(@ccall LIB.fx3d_chamfer_finalize_many(sums.ptr::Ptr{Cvoid}, count::Int32, N::Int32, M::Int32, B_global::Int64,
                                                D::Int32, Float32(w1)::Float32, Float32(w2)::Float32,
                                                losses.ptr::Ptr{Cvoid}, DEFAULT_STREAM::Stream)::Int32)
    return losses
end

# The two halves of _chamfer_distance (src/metrics/pcloud.jl:47-50) for a caller that runs its own collective between
# them (MPI.jl, Distributed.jl): this rank's partial sums into `sums` (2 Float64: a slot of a (2,count) array), then the
# loss from globally reduced sums with the GLOBAL batch size.
function chamfer_sums!(sums::HipArray{Float64}, A::HipArray{Float32,3}, B::HipArray{Float32,3}; indices::Bool = false,
                       stream::Stream = DEFAULT_STREAM)
    D, N, Bn = size(A); _, M, _ = size(B)
    nb = Ref{Csize_t}(0)
    check(@ccall LIB.fx3d_chamfer_workspace_bytes(N::Int32, M::Int32, max(Bn, 1)::Int32, D::Int32, nb::Ref{Csize_t})::Int32)
    ws = workspace(nb[])
    ix = indices ? HipArray{Int32}(undef, N, Bn) : nothing
    iy = indices ? HipArray{Int32}(undef, M, Bn) : nothing
    check(@ccall LIB.fx3d_chamfer_sums(A.ptr::Ptr{Cvoid}, N::Int32, B.ptr::Ptr{Cvoid}, M::Int32, Bn::Int32, D::Int32,
                                       sums.ptr::Ptr{Cvoid}, (indices ? ix.ptr : C_NULL)::Ptr{Cvoid},
                                       (indices ? iy.ptr : C_NULL)::Ptr{Cvoid}, ws.ptr::Ptr{Cvoid}, length(ws)::Csize_t,
                                       stream::Stream)::Int32)
    return sums, ix, iy
end
function chamfer_finalize(sums::HipArray{Float64}, N::Integer, M::Integer, B_global::Integer, D::Integer;
                          w1::Number = 1.0, w2::Number = 1.0, stream::Stream = DEFAULT_STREAM)
    loss_dev = HipArray{Float32}(undef, 1)
    check(@ccall LIB.fx3d_chamfer_finalize(sums.ptr::Ptr{Cvoid}, N::Int32, M::Int32, B_global::Int64, D::Int32,
                                           Float32(w1)::Float32, Float32(w2)::Float32, loss_dev.ptr::Ptr{Cvoid},
                                           stream::Stream)::Int32)
    return unhip(loss_dev)[1]
end
# cat(X, KNNGraph - X; dims = 1) for given neighbour indices (src/models/dgcnn.jl:36-51); idx (K,N,B) Int32 0-based
function edge_features(X::HipArray{Float32,3}, idx::HipArray{Int32,3}; layout::Int = 1)
    F, N, B = size(X); K = size(idx, 1)
    out = layout == 1 ? HipArray{Float32}(undef, K * N, 2F, B) : HipArray{Float32}(undef, 2F, K, N, B)
    check(@ccall LIB.fx3d_edge_features(X.ptr::Ptr{Cvoid}, N::Int32, B::Int32, F::Int32, K::Int32, idx.ptr::Ptr{Cvoid},
                                        layout::Int32, out.ptr::Ptr{Cvoid}, DEFAULT_STREAM::Stream)::Int32)
    return out
end
set_device(dev::Integer) = check(@ccall LIB.fx3d_set_device(dev::Int32)::Int32)

# ---- host-array convenience methods (SURVEY 8b): plain `Array`s in, host results out; the device path runs underneath ----
# (for callers that have not moved their data to HipArray: same results as the HipArray methods, PCIe copies included)
function chamfer_distance_host(A::Array{Float32,3}, B::Array{Float32,3}; w1::Number = 1.0, w2::Number = 1.0)
    D, N, Bn = size(A); M = size(B, 2)
    loss = Ref{Float32}(0)
    check(@ccall LIB.fx3d_chamfer_distance_host(A::Ptr{Float32}, N::Int32, B::Ptr{Float32}, M::Int32, Bn::Int32, D::Int32,
                                                Float32(w1)::Float32, Float32(w2)::Float32, loss::Ref{Float32},
                                                C_NULL::Ptr{Int32}, C_NULL::Ptr{Int32})::Int32)
    return loss[]
end
function knn_host(X::Array{Float32,3}, K::Int; drop_first::Bool = true)
    D, N, Bn = size(X)
    idx = Array{Int32,3}(undef, K, N, Bn)
    check(@ccall LIB.fx3d_knn_host(X::Ptr{Float32}, N::Int32, C_NULL::Ptr{Float32}, N::Int32, Bn::Int32, D::Int32, K::Int32,
                                   Int32(drop_first)::Int32, idx::Ptr{Int32}, C_NULL::Ptr{Float32})::Int32)
    return idx .+ Int32(1)
end
function sample_points_host(verts_padded::Array{Float32,3}, faces_padded0::Array{Int32,3}, faces_len::Vector{Int32}, n::Int;
                            eps::Number = 1e-6, seed::UInt64 = rand(UInt64))
    _, V, Bn = size(verts_padded); F = size(faces_padded0, 2)
    out = Array{Float32,3}(undef, 3, n, Bn)
    check(@ccall LIB.fx3d_sample_points_host(verts_padded::Ptr{Float32}, V::Int32, faces_padded0::Ptr{Int32}, F::Int32,
                                             faces_len::Ptr{Int32}, Bn::Int32, n::Int32, Float64(eps)::Float64, seed::UInt64,
                                             out::Ptr{Float32})::Int32)
    return out
end
function edge_loss_host(verts::Array{Float32,2}, edges0::Array{Int32,2}, target::Number = 0)
    loss = Ref{Float32}(0)
    check(@ccall LIB.fx3d_edge_loss_host(verts::Ptr{Float32}, size(verts, 2)::Int64, edges0::Ptr{Int32}, size(edges0, 1)::Int64,
                                         Float32(target)::Float32, loss::Ref{Float32})::Int32)
    return loss[]
end
function laplacian_loss_host(verts::Array{Float32,2}, rowptr::Vector{Int32}, colind::Vector{Int32}, vals::Vector{Float32})
    loss = Ref{Float32}(0)
    check(@ccall LIB.fx3d_laplacian_loss_host(verts::Ptr{Float32}, size(verts, 2)::Int64, rowptr::Ptr{Int32}, colind::Ptr{Int32},
                                              vals::Ptr{Float32}, loss::Ref{Float32})::Int32)
    return loss[]
end

# ---- the rest of the ABI: device / stream / event utilities, explicit-draw sampling, face areas, host topology ----
version() = unsafe_string(@ccall LIB.fx3d_version()::Cstring)
# variant switches (include/flux3d_hip.h): named integer options instead of environment reads on the launch path
set_option(name::AbstractString, value::Integer) = check(@ccall LIB.fx3d_set_option(name::Cstring, value::Int32)::Int32)
function get_option(name::AbstractString)
    v = Ref{Int32}(0); check(@ccall LIB.fx3d_get_option(name::Cstring, v::Ref{Int32})::Int32); return Int(v[])
end
function options()
    n = @ccall LIB.fx3d_option_count()::Int32
    names = [unsafe_string(@ccall LIB.fx3d_option_name(i::Int32)::Cstring) for i in 0:n-1]
    return Dict(k => get_option(k) for k in names)
end
function current_device()
    d = Ref{Int32}(0); check(@ccall LIB.fx3d_get_device(d::Ref{Int32})::Int32); return Int(d[])
end
function device_name(dev::Integer = current_device())
    buf = Vector{UInt8}(undef, 256)
    check(@ccall LIB.fx3d_device_name(dev::Int32, buf::Ptr{UInt8}, length(buf)::Csize_t)::Int32)
    return unsafe_string(pointer(buf))
end
# (PCI bus id, 16 UUID bytes) of a device: what a multi-process run gathers per rank to prove N ranks sat on N devices
function device_identity(dev::Integer = current_device())
    buf = Vector{UInt8}(undef, 64); uuid = Vector{UInt8}(undef, 16)
    check(@ccall LIB.fx3d_device_identity(dev::Int32, buf::Ptr{UInt8}, length(buf)::Csize_t, uuid::Ptr{UInt8})::Int32)
    return unsafe_string(pointer(buf)), uuid
end
device_synchronize() = check(@ccall LIB.fx3d_device_sync()::Int32)
function stream_create()
    s = Ref{Stream}(C_NULL); check(@ccall LIB.fx3d_stream_create(s::Ref{Stream})::Int32); return s[]
end
stream_destroy(s::Stream) = check(@ccall LIB.fx3d_stream_destroy(s::Stream)::Int32)
stream_synchronize(s::Stream = DEFAULT_STREAM) = check(@ccall LIB.fx3d_stream_sync(s::Stream)::Int32)
function event_create()
    e = Ref{Event}(C_NULL); check(@ccall LIB.fx3d_event_create(e::Ref{Event})::Int32); return e[]
end
function event_create_sync()  # ordering only (stream_wait_event / event_synchronize): no timestamps, no system-scope fence
    e = Ref{Event}(C_NULL); check(@ccall LIB.fx3d_event_create_sync(e::Ref{Event})::Int32); return e[]
end
event_destroy(e::Event) = check(@ccall LIB.fx3d_event_destroy(e::Event)::Int32)
event_record(e::Event, s::Stream = DEFAULT_STREAM) = check(@ccall LIB.fx3d_event_record(e::Event, s::Stream)::Int32)
event_synchronize(e::Event) = check(@ccall LIB.fx3d_event_sync(e::Event)::Int32)
stream_wait_event(s::Stream, e::Event) = check(@ccall LIB.fx3d_stream_wait_event(s::Stream, e::Event)::Int32)  # device-side ordering
function event_elapsed_ms(a::Event, b::Event)
    ms = Ref{Float32}(0); check(@ccall LIB.fx3d_event_elapsed_ms(a::Event, b::Event, ms::Ref{Float32})::Int32); return ms[]
end
# per-kernel HIP-event timing inside the library (bench.py's `roofline` object)
profile_enable(every_nth::Integer) = check(@ccall LIB.fx3d_profile_enable(every_nth::Int32)::Int32)
function profile_kernel_stats(name::AbstractString)
    avg = Ref{Float64}(0); mn = Ref{Float64}(0); mx = Ref{Float64}(0); cnt = Ref{Int64}(0)
    check(@ccall LIB.fx3d_profile_kernel_stats(name::Cstring, avg::Ref{Float64}, mn::Ref{Float64}, mx::Ref{Float64}, cnt::Ref{Int64})::Int32)
    return (avg_ms = avg[], min_ms = mn[], max_ms = mx[], count = cnt[])
end

# compute_faces_areas_packed / _padded (src/rep/mesh.jl:765-808) on HipArray meshes
function compute_faces_areas_packed(m::TriMesh{Float32,R,HipArray}; eps::Number = 1e-6) where {R}
    verts = get_verts_packed(m)::HipArray{Float32,2}; faces = faces_packed_dev(m)
    out = HipArray{Float32}(undef, size(faces, 2))
    check(@ccall LIB.fx3d_faces_areas_packed(verts.ptr::Ptr{Cvoid}, size(verts, 2)::Int64, faces.ptr::Ptr{Cvoid},
                                             size(faces, 2)::Int64, out.ptr::Ptr{Cvoid}, DEFAULT_STREAM::Stream)::Int32)
    return out
end
function compute_faces_areas_padded(m::TriMesh{Float32,R,HipArray}; eps::Number = 1e-6) where {R}
    verts = get_verts_padded(m)::HipArray{Float32,3}
    out = HipArray{Float32}(undef, 1, m.F, m.N)
    check(@ccall LIB.fx3d_faces_areas_padded(verts.ptr::Ptr{Cvoid}, m.V::Int32, faces_padded_dev(m).ptr::Ptr{Cvoid}, m.F::Int32,
                                             faces_len_dev(m).ptr::Ptr{Cvoid}, m.N::Int32, out.ptr::Ptr{Cvoid},
                                             DEFAULT_STREAM::Stream)::Int32)
    return out
end
# _sample_points for given draws (src/transforms/mesh_func.jl:60-82): face_idx (n,B) Int32 0-based, r1, r2 (n,B)
function sample_points_explicit(m::TriMesh{Float32,R,HipArray}, verts::HipArray{Float32,3}, face_idx::HipArray{Int32,2},
                                r1::HipArray{Float32,2}, r2::HipArray{Float32,2}) where {R}
    n = size(face_idx, 1)
    out = HipArray{Float32}(undef, 3, n, m.N)
    check(@ccall LIB.fx3d_sample_points_explicit(verts.ptr::Ptr{Cvoid}, m.V::Int32, faces_padded_dev(m).ptr::Ptr{Cvoid}, m.F::Int32,
                                                 m.N::Int32, n::Int32, face_idx.ptr::Ptr{Cvoid}, r1.ptr::Ptr{Cvoid},
                                                 r2.ptr::Ptr{Cvoid}, out.ptr::Ptr{Cvoid}, DEFAULT_STREAM::Stream)::Int32)
    return out
end
# host topology builders (integer work of src/rep/mesh.jl:907-1002 in C++; the reference's own Julia versions stay valid)
function build_edges_packed(faces::Matrix{Int64}, V::Integer; index_base::Integer = 1)
    F = size(faces, 2)
    buf = Vector{Int64}(undef, 6F); f2e = Matrix{Int64}(undef, F, 3); E = Ref{Int64}(0)
    check(@ccall LIB.fx3d_build_edges_packed(faces::Ptr{Int64}, F::Int64, V::Int64, index_base::Int32, buf::Ptr{Int64},
                                             f2e::Ptr{Int64}, E::Ref{Int64})::Int32)
    return reshape(buf[1:2E[]], Int(E[]), 2), f2e
end
function build_laplacian_csr(edges::Matrix{Int64}, V::Integer; index_base::Integer = 1)
    E = size(edges, 1)
    rowptr = Vector{Int32}(undef, V + 1); colind = Vector{Int32}(undef, 2E + V); vals = Vector{Float32}(undef, 2E + V)
    nnz = Ref{Int64}(0)
    check(@ccall LIB.fx3d_build_laplacian_csr(edges::Ptr{Int64}, E::Int64, V::Int64, index_base::Int32, rowptr::Ptr{Int32},
                                              colind::Ptr{Int32}, vals::Ptr{Float32}, nnz::Ref{Int64})::Int32)
    return rowptr, colind[1:nnz[]], vals[1:nnz[]]
end

end # module
