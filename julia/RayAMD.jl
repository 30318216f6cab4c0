# RayAMD.jl -- Julia host shim over libray_amd.so (include/ray_amd.h).
#
# Keeps the reference's entry points so that the reconstruction loop, foliation and fidelity
# logic (NonObstacleReconstruction, NonObstacle.jl:283-441; ObstacleDemo.jl:138-293) run unchanged
# on top of the MI355X kernels:
#
#   ScatterRelation(c, ∇c, ns, nd, dt, dir)               NonObstacle.jl:54
#   ScatterRelationObstacle(c, ∇c, ob, ∇ob, ns, nd, dt)   Obstacle.jl:11
#   ScatterForwardOperator(c, m, ext, dt)                  NonObstacle.jl:189   (s and sparse M)
#   NonReflectionTrace(c, m, ext, dt)                      Obstacle.jl:275
#   NonReflectionPlot(c, m, ext, dt)                       Obstacle.jl:222      (the polylines; plotting stays with the caller)
#   ScatterRelationObstaclePlot(c, ∇c, ob, ∇ob, ns, nd, dt, rayIndex)   Obstacle.jl:98   (m and the polylines)
# plus what the loops around them need on the device: the gridded-medium ScatterRelation (BASELINE config 2's
# composition), classify_rays (Utility.jl:1-39, ObstacleDemo.jl:48-67), the passes over a device-resident Frechet
# matrix (NonObstacle.jl:340-367) and RayGroup, the node-wide fan-out that replaces addprocs / @parallel
# (NonObstacle.jl:59,236-240).
#
# Only `ccall`; no AMDGPU.jl, no CUDA shims, no CPU fallback (a missing library is an error).
# Closures cannot cross a C ABI onto a GPU: the medium / obstacle arguments are descriptor values
# (SinProd, Circle, Trefoil) instead of Julia functions; pass the same value in the gradient slot.
#
# STATUS: written for Julia >= 1.6.  Julia is not installed in the build image (and the reference
# itself only parses on Julia 0.6), so this file has been syntax-reviewed but NOT executed; the
# executed mirror of the same calls is the Python package ray_amd/ (ctypes), which the GPU tests
# drive.  INTEGRATION.md lists the three lines a maintainer changes in the demo scripts.
module RayAMD

using SparseArrays
using SharedArrays   # NonObstacleReconstruction requires m::SharedArray{Float64,2} (NonObstacle.jl:283)

export SinProd, RadialSineSum, smooth_alt, obstacle_alt, Circle, Trefoil, trefoil_demo, smooth_demo, obstacle_demo,
       ScatterRelation, ScatterRelationObstacle, ScatterForwardOperator, NonReflectionTrace,
       NonReflectionPlot, ScatterRelationObstaclePlot, classify_rays,
       Frechet, forward_operator_dev, dofs, matvec, rmatvec, normal_matvec, normal_diag, fidelity_update!,
       RayGroup, set_observation!,
       misfit, set_arith!, ARITH_EXACT, ARITH_FUSED, read_rayfix

const LIB = get(ENV, "RAY_AMD_LIB", joinpath(@__DIR__, "..", "ray_amd", "libray_amd.so"))

const ARITH_EXACT = Cint(0)
const ARITH_FUSED = Cint(1)
const TIME_SECONDS = Cint(0)
const TIME_STEPS = Cint(1)
const TRACE_SHORTEN_CEASE = Cint(0)
const TRACE_OVERSHOOT = Cint(1)
const RAY_OPT_FAN_DIRECTIONS = Cint(3)

# ---- descriptors ------------------------------------------------------------------------------------
"c = 1 + A sin(wx x) sin(wy y); ∇c = [gx cos(wx x) sin(wy y), gy sin(wx x) cos(wy y)]"
struct SinProd
    A::Float64; wx::Float64; wy::Float64; gx::Float64; gy::Float64
end
smooth_demo()   = SinProd(0.3, 1.5 * pi, 1.5 * pi, 0.45 * pi, 0.45 * pi)          # NonObstacleDemo.jl:12,23
obstacle_demo() = SinProd(0.2, 2 * pi, 1.0 * pi, 2 * (0.2 * pi), 0.2 * pi)         # ObstacleDemo.jl:12,18
(s::SinProd)(x, y) = 1 + s.A * sin(s.wx * x) * sin(s.wy * y)                       # waveSpeed(x, y)

"c = 1 + sum a_k sin(pi |x - x_k|): the commented alternate media (NonObstacleDemo.jl:7-11; ObstacleDemo.jl:9-11)"
struct RadialSineSum
    a::Vector{Float64}; x::Vector{Float64}; y::Vector{Float64}
end
smooth_alt()   = RadialSineSum([0.2, 0.4, 0.2, 0.3], [0.5, -0.4, 0.3, -0.2], [0.2, -0.3, -0.4, 0.1])
obstacle_alt() = RadialSineSum([0.4, 0.4], [0.5, -0.4], [0.2, -0.3])
(s::RadialSineSum)(x, y) = 1.0 + sum(s.a[k] * sin(pi * sqrt((x - s.x[k])^2 + (y - s.y[k])^2)) for k in 1:length(s.a))
const Analytic = Union{SinProd,RadialSineSum}

"ob = sqrt(x^2+y^2) - r0 (ObstacleDemo.jl:20-34)"
struct Circle
    r0::Float64
end
"ob = r - (R - rho cos 3theta), normal with its own rho_n: the commented alternate of ObstacleDemo.jl:23-24,30-31"
struct Trefoil
    R::Float64; rho::Float64; rho_n::Float64
end
trefoil_demo() = Trefoil(0.4, 0.3, 0.2)

struct CObstacle            # ray_obstacle
    kind::Int32; pad::Int32; r0::Float64; reserved1::Float64; reserved2::Float64
end
c_obstacle(ob::Circle)  = CObstacle(1, 0, ob.r0, 0.0, 0.0)
c_obstacle(ob::Trefoil) = CObstacle(2, 0, ob.R, ob.rho, ob.rho_n)
struct CShootParams         # ray_shoot_params
    ns::Int32; nd::Int32; src_begin::Int32; src_end::Int32
    dir_lo::Float64; dir_hi::Float64; dt::Float64
    time_unit::Int32; max_steps::Int32
end

# ---- context (one per process / GPU) ------------------------------------------------------------------
const CTX = Ref{Ptr{Cvoid}}(C_NULL)

function check(rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:ray_last_error, LIB), Cstring, ()))
    error("libray_amd error $rc: $msg")
end

function context(device::Integer = 0)
    if CTX[] == C_NULL
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:ray_ctx_create, LIB), Cint, (Cint, Ref{Ptr{Cvoid}}), device, h))
        CTX[] = h[]
        atexit(() -> (empty!(GRID_MEDIA); ccall((:ray_ctx_destroy, LIB), Cint, (Ptr{Cvoid},), CTX[]); CTX[] = C_NULL))
    end
    CTX[]
end

set_arith!(a::Cint) = check(ccall((:ray_ctx_set_arith, LIB), Cint, (Ptr{Cvoid}, Cint), context(), a))

# ---- media ----------------------------------------------------------------------------------------------
function with_medium(f, c::SinProd)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:ray_medium_create_sinprod, LIB), Cint,
                (Ptr{Cvoid}, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Ref{Ptr{Cvoid}}),
                context(), c.A, c.wx, c.wy, c.gx, c.gy, h))
    try
        f(h[])
    finally
        ccall((:ray_medium_destroy, LIB), Cint, (Ptr{Cvoid},), h[])
    end
end

function with_medium(f, c::RadialSineSum)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:ray_medium_create_radial, LIB), Cint,
                (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Ptr{Cvoid}}),
                context(), length(c.a), c.a, c.x, c.y, h))
    try
        f(h[])
    finally
        ccall((:ray_medium_destroy, LIB), Cint, (Ptr{Cvoid},), h[])
    end
end

# Gridded media are kept per (N, ext): the reconstruction loop calls ScatterForwardOperator / the traces with a new c0 of
# the same size every iteration (NonObstacle.jl:331-333), so the handle (tables, hess records, their device memory)
# is created once and every later call re-uploads the nodal values with ray_medium_update_grid (one copy + one table
# build on the device) instead of four allocations and their release.  Freed with the context (ray_ctx_destroy
# releases every medium still alive on it).
const GRID_MEDIA = Dict{Tuple{Int,Float64},Ptr{Cvoid}}()
function with_medium(f, c::Matrix{Float64}, ext::Float64)
    N = size(c, 1)
    size(c, 2) == N || error("c must be N x N")
    ctx = context()
    h = get(GRID_MEDIA, (N, ext), C_NULL)
    if h == C_NULL
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:ray_medium_create_grid, LIB), Cint,
                    (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cdouble, Ref{Ptr{Cvoid}}), ctx, c, N, ext, r))
        h = r[]
        GRID_MEDIA[(N, ext)] = h
    else
        check(ccall((:ray_medium_update_grid, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), h, c))
    end
    f(h)
end

same_or_nothing(a, b, what) = (b === nothing || b === a || b == a) ||
    error("pass the $what descriptor itself (or nothing) in the gradient slot: closures cannot cross the C ABI")

# ---- NonObstacle.jl:54 ----------------------------------------------------------------------------------
function ScatterRelation(c::Analytic, ∇c, ns::Int, nd::Int, dt::Float64, dir = (0, pi))
    same_or_nothing(c, ∇c, "medium")
    m = SharedArray{Float64}((ns * nd, 9))
    prm = CShootParams(ns, nd, 0, ns, Float64(dir[1]), Float64(dir[2]), dt, TIME_SECONDS, 0)
    with_medium(c) do med
        check(ccall((:ray_shoot_to_exit, LIB), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{CObstacle}, Ref{CShootParams}, Ptr{Cdouble}, Int64,
                     Ptr{Int32}, Ptr{Int32}, Ptr{Int32}),
                    context(), med, C_NULL, prm, m, ns * nd, C_NULL, C_NULL, C_NULL))
    end
    m
end

# ---- Obstacle.jl:11 (column 9 in steps, the caller rescales: ObstacleDemo.jl:122,311) ---------------------
function ScatterRelationObstacle(c::Analytic, ∇c, ob::Union{Circle,Trefoil}, ∇ob, ns::Int, nd::Int, dt::Float64)
    same_or_nothing(c, ∇c, "medium"); same_or_nothing(ob, ∇ob, "obstacle")
    m = SharedArray{Float64}((ns * nd, 9))
    prm = CShootParams(ns, nd, 0, ns, 0.0, Float64(pi), dt, TIME_STEPS, 0)
    cob = Ref(c_obstacle(ob))
    with_medium(c) do med
        check(ccall((:ray_shoot_to_exit, LIB), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ref{CObstacle}, Ref{CShootParams}, Ptr{Cdouble}, Int64,
                     Ptr{Int32}, Ptr{Int32}, Ptr{Int32}),
                    context(), med, cob, prm, m, ns * nd, C_NULL, C_NULL, C_NULL))
    end
    m
end

# nd > 0: the rows of m are fans of nd directions per source (ScatterRelation's layout); a layout hint for the
# kernel (RAY_OPT_FAN_DIRECTIONS) for the length of this call -- without it the library looks for the fans in m
# itself.  The results do not depend on it.
function trace(c::Matrix{Float64}, m::AbstractMatrix{Float64}, ext::Float64, dt::Float64, mode::Cint; nd::Int = 0)
    num = size(m, 1)
    nd > 0 && check(ccall((:ray_ctx_set_option, LIB), Cint, (Ptr{Cvoid}, Cint, Cint), context(), RAY_OPT_FAN_DIRECTIONS, nd))
    mm = Matrix{Float64}(m)                       # dense column-major copy (m may be a view)
    X = Matrix{Float64}(undef, num, 4)
    try
        with_medium(c, ext) do med
            check(ccall((:ray_trace_fixed_time, LIB), Cint,
                        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Int64, Int64, Cdouble, Cint, Ptr{Cdouble}, Int64,
                         Ptr{Int32}, Ptr{Int32}),
                        context(), med, mm, num, max(num, 1), dt, mode, X, max(num, 1), C_NULL, C_NULL))
        end
    finally
        nd > 0 && ccall((:ray_ctx_set_option, LIB), Cint, (Ptr{Cvoid}, Cint, Cint), context(), RAY_OPT_FAN_DIRECTIONS, 0)
    end
    X
end

# ---- NonObstacle.jl:189: (M, s).  s[:,1:4], s[:,9] copied from m (:196), s[:,5:8] traced (:185); ------------
# M = droptol!(sparse(M), 1e-12) (:242), assembled sparse on the GPU and handed over as CSR, i.e. the CSC
# arrays of M' -- one transpose gives the reference's SparseMatrixCSC.
function ScatterForwardOperator(c::Matrix{Float64}, m, ext::Float64, dt::Float64; droptol::Float64 = 1e-12)
    num = size(m, 1); N = size(c, 1)
    mm = Matrix{Float64}(m)
    X = Matrix{Float64}(undef, num, 4)
    s = SharedArray{Float64}((num, 9))
    s[:, 1:4] = m[:, 1:4]; s[:, 9] = m[:, 9]
    local M
    with_medium(c, ext) do med
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:ray_forward_operator, LIB), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Int64, Int64, Cdouble, Cdouble, Ptr{Cdouble}, Int64,
                     Ptr{Cdouble}, Ptr{Int32}, Ptr{Int32}, Ref{Ptr{Cvoid}}),
                    context(), med, mm, num, max(num, 1), dt, droptol, X, max(num, 1), C_NULL, C_NULL, C_NULL, h))
        try
            nr = Ref{Int64}(0); nc = Ref{Int64}(0); nz = Ref{Int64}(0)
            check(ccall((:ray_frechet_shape, LIB), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ref{Int64}), h[], nr, nc, nz))
            indptr = Vector{Int64}(undef, nr[] + 1); indices = Vector{Int32}(undef, nz[]); data = Vector{Float64}(undef, nz[])
            check(ccall((:ray_frechet_export_csr, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int32}, Ptr{Cdouble}),
                        h[], indptr, indices, data))
            Mt = SparseMatrixCSC(Int(nc[]), Int(nr[]), indptr .+ 1, Int64.(indices) .+ 1, data)   # = M'
            M = copy(transpose(Mt))
        finally
            ccall((:ray_frechet_destroy, LIB), Cint, (Ptr{Cvoid},), h[])
        end
    end
    s[:, 5:8] = X
    M, s
end

# ---- Obstacle.jl:275 --------------------------------------------------------------------------------------
function NonReflectionTrace(c::Matrix{Float64}, m, ext::Float64, dt::Float64; nd::Int = 0)
    m[:, 5:8] = trace(c, m, ext, dt, TRACE_OVERSHOOT; nd = nd)
    m
end

# ---- the gridded-medium forms of the shooting entry points (BASELINE config 2 = a2's RHS inside a6's loop) ----
function ScatterRelation(c::Matrix{Float64}, ext::Float64, ns::Int, nd::Int, dt::Float64, dir = (0, pi))
    m = SharedArray{Float64}((ns * nd, 9))
    prm = CShootParams(ns, nd, 0, ns, Float64(dir[1]), Float64(dir[2]), dt, TIME_SECONDS, 0)
    with_medium(c, ext) do med
        check(ccall((:ray_shoot_to_exit, LIB), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{CObstacle}, Ref{CShootParams}, Ptr{Cdouble}, Int64,
                     Ptr{Int32}, Ptr{Int32}, Ptr{Int32}),
                    context(), med, C_NULL, prm, m, ns * nd, C_NULL, C_NULL, C_NULL))
    end
    m
end

function ScatterRelationObstacle(c::Matrix{Float64}, ext::Float64, ob::Union{Circle,Trefoil}, ns::Int, nd::Int, dt::Float64)
    m = SharedArray{Float64}((ns * nd, 9))
    prm = CShootParams(ns, nd, 0, ns, 0.0, Float64(pi), dt, TIME_STEPS, 0)
    cob = Ref(c_obstacle(ob))
    with_medium(c, ext) do med
        check(ccall((:ray_shoot_to_exit, LIB), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ref{CObstacle}, Ref{CShootParams}, Ptr{Cdouble}, Int64,
                     Ptr{Int32}, Ptr{Int32}, Ptr{Int32}),
                    context(), med, cob, prm, m, ns * nd, C_NULL, C_NULL, C_NULL))
    end
    m
end

# polylines [ray][point][x,y] -> one (npts x 2) matrix per ray
function unpack_paths(path::Array{Float64,3}, npts::Vector{Int32}, max_pts::Int)
    any(npts .> max_pts) && error("a ray produced $(maximum(npts)) points; raise max_pts (= $max_pts)")
    [permutedims(path[:, 1:npts[k], k]) for k in 1:length(npts)]
end

# ---- Obstacle.jl:222 without PyPlot: the fixed-time (overshooting) trace of every row plus its polyline `res` ----
function NonReflectionPlot(c::Matrix{Float64}, m::AbstractMatrix{Float64}, ext::Float64, dt::Float64; max_pts::Int = 4096)
    num = size(m, 1)
    mm = Matrix{Float64}(m)
    X = Matrix{Float64}(undef, num, 4)
    path = zeros(Float64, 2, max_pts, num)            # column-major [x,y | point | ray] = C's [ray][point][x,y]
    npts = zeros(Int32, num)
    with_medium(c, ext) do med
        check(ccall((:ray_trace_paths, LIB), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Int64, Int64, Cdouble, Cint, Cint, Ptr{Cdouble}, Ptr{Int32},
                     Ptr{Cdouble}, Int64, Ptr{Int32}, Ptr{Int32}),
                    context(), med, mm, num, max(num, 1), dt, TRACE_OVERSHOOT, max_pts, path, npts, X, max(num, 1), C_NULL, C_NULL))
    end
    unpack_paths(path, npts, max_pts)
end

# ---- Obstacle.jl:98 without PyPlot: m as the reference fills it (rows of rayIndex shot) and their polylines -----
function ScatterRelationObstaclePlot(c::Analytic, ∇c, ob::Union{Circle,Trefoil}, ∇ob, ns::Int, nd::Int, dt::Float64,
                                     rayIndex::AbstractVector{<:Integer}; max_pts::Int = 4096)
    m = ScatterRelationObstacle(c, ∇c, ob, ∇ob, ns, nd, dt)
    m[:, 5:9] .= 0.0
    sel = Matrix{Float64}(m[rayIndex, :])
    num = size(sel, 1)
    path = zeros(Float64, 2, max_pts, num)
    npts = zeros(Int32, num)
    cob = Ref(c_obstacle(ob))
    with_medium(c) do med
        check(ccall((:ray_shoot_paths, LIB), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ref{CObstacle}, Cdouble, Int32, Int32, Ptr{Cdouble}, Int64, Int64, Int32,
                     Ptr{Cdouble}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}),
                    context(), med, cob, dt, TIME_STEPS, 0, sel, num, max(num, 1), max_pts, path, npts, C_NULL, C_NULL, C_NULL))
    end
    m[rayIndex, :] = sel
    m, unpack_paths(path, npts, max_pts)
end

# ---- Utility.jl:1-39 + ObstacleDemo.jl:48-67 on the device -------------------------------------------------------
"returns (orthoIndex, lo, hi, arg): orthoIndex 1-based rows (ObstacleDemo.jl:48-49); lo / hi per source, 1-based as
derivativeCheck returns them (Utility.jl:15-39); arg the aligned exit angles (nd x ns)"
function classify_rays(m::AbstractMatrix{Float64}, ns::Int, nd::Int)
    num = ns * nd
    mm = Matrix{Float64}(m)
    ortho = zeros(UInt8, num); arg = zeros(num); lo = zeros(Int32, ns); hi = zeros(Int32, ns)
    check(ccall((:ray_classify, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Int64, Int64, Int64, Ptr{UInt8}, Ptr{Cdouble}, Ptr{Int32}, Ptr{Int32}),
                context(), mm, ns, nd, max(num, 1), ortho, arg, lo, hi))
    findall(!iszero, ortho), lo, hi, reshape(arg, nd, ns)
end

# ---- the device-resident Frechet matrix and the loop's passes over it (NonObstacle.jl:340-367) ---------------------
mutable struct Frechet
    h::Ptr{Cvoid}
    med::Ptr{Cvoid}
    nrows::Int; ncols::Int; nnz::Int
end

"ScatterForwardOperator with M left on the device: returns (F::Frechet, s).  close(F) releases it."
function forward_operator_dev(c::Matrix{Float64}, m, ext::Float64, dt::Float64; droptol::Float64 = 1e-12)
    num = size(m, 1); N = size(c, 1)
    mm = Matrix{Float64}(m)
    X = Matrix{Float64}(undef, num, 4)
    med = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:ray_medium_create_grid, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cdouble, Ref{Ptr{Cvoid}}), context(), c, N, ext, med))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:ray_forward_operator, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Int64, Int64, Cdouble, Cdouble, Ptr{Cdouble}, Int64,
                 Ptr{Cdouble}, Ptr{Int32}, Ptr{Int32}, Ref{Ptr{Cvoid}}),
                context(), med[], mm, num, max(num, 1), dt, droptol, X, max(num, 1), C_NULL, C_NULL, C_NULL, h))
    nr = Ref{Int64}(0); nc = Ref{Int64}(0); nz = Ref{Int64}(0)
    check(ccall((:ray_frechet_shape, LIB), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ref{Int64}), h[], nr, nc, nz))
    s = SharedArray{Float64}((num, 9))
    s[:, 1:4] = m[:, 1:4]; s[:, 9] = m[:, 9]; s[:, 5:8] = X
    Frechet(h[], med[], Int(nr[]), Int(nc[]), Int(nz[])), s
end

function Base.close(F::Frechet)
    F.h != C_NULL && ccall((:ray_frechet_destroy, LIB), Cint, (Ptr{Cvoid},), F.h)
    F.med != C_NULL && ccall((:ray_medium_destroy, LIB), Cint, (Ptr{Cvoid},), F.med)
    F.h = C_NULL; F.med = C_NULL
    nothing
end

maskptr(mask) = mask === nothing ? Ptr{UInt8}(C_NULL) : pointer(mask)

"dofs[j] = nnz(M[4j-3,:]) - sum(fidelity[find(M[4j-3,:])])   (NonObstacle.jl:340-342)"
function dofs(F::Frechet, fidelity::Vector{Float64})
    out = zeros(F.nrows ÷ 4)
    check(ccall((:ray_frechet_dofs, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}), F.h, fidelity, out))
    out
end

function matvec(F::Frechet, x::Vector{Float64})
    y = zeros(F.nrows)
    check(ccall((:ray_frechet_matvec, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}), F.h, x, y))
    y
end

"b = M[order,:]' * v[order] over all N^2 columns (NonObstacle.jl:354: take b[Idx]); mask::Vector{UInt8} per ray or nothing"
function rmatvec(F::Frechet, v::Vector{Float64}, mask = nothing)
    b = zeros(F.ncols)
    GC.@preserve mask check(ccall((:ray_frechet_matvec_t, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{UInt8}, Ptr{Cdouble}),
                                  F.h, v, maskptr(mask), b))
    b
end

"w = M[order,:]' * (M[order,:] * x): the action of MtM (NonObstacle.jl:353) without forming it"
function normal_matvec(F::Frechet, x::Vector{Float64}, mask = nothing)
    w = zeros(F.ncols)
    GC.@preserve mask check(ccall((:ray_frechet_normal_matvec, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{UInt8}, Ptr{Cdouble}),
                                  F.h, x, maskptr(mask), w))
    w
end

function normal_diag(F::Frechet, mask = nothing)
    d = zeros(F.ncols)
    GC.@preserve mask check(ccall((:ray_frechet_normal_diag, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Ptr{Cdouble}),
                                  F.h, maskptr(mask), d))
    d
end

"NonObstacle.jl:359-367 in place on `fidelity`; returns the residual (0 on unselected rows)"
function fidelity_update!(F::Frechet, correction::Vector{Float64}, mismatch::Vector{Float64}, fidelity::Vector{Float64},
                          rejection::Float64, decay::Float64, mask = nothing)
    residual = zeros(F.nrows)
    GC.@preserve mask check(ccall((:ray_frechet_fidelity_update, LIB), Cint,
                                  (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{UInt8}, Cdouble, Cdouble, Ptr{Cdouble}, Ptr{Cdouble}),
                                  F.h, correction, mismatch, maskptr(mask), rejection, decay, fidelity, residual))
    residual
end

# ---- node-wide fan-out: what addprocs(3) / @sync @parallel / remotecall_wait give the reference's callers -----------
# (NonObstacleDemo.jl:1; NonObstacle.jl:59,236-240).  One Julia process, every GPU of the node.
mutable struct RayGroup
    h::Ptr{Cvoid}
    n::Int
    rows::Int
end

function RayGroup(devices::Vector{<:Integer})
    dev = Vector{Cint}(devices)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:ray_group_create, LIB), Cint, (Ptr{Cint}, Cint, Ref{Ptr{Cvoid}}), dev, length(dev), h))
    RayGroup(h[], length(dev), 0)
end

function Base.close(g::RayGroup)
    g.h != C_NULL && ccall((:ray_group_destroy, LIB), Cint, (Ptr{Cvoid},), g.h)
    g.h = C_NULL
    nothing
end

"ScatterRelation on a gridded medium over all GPUs of the group: member k of G shoots the sources k, k+G, ..."
function ScatterRelation(g::RayGroup, c::Matrix{Float64}, ext::Float64, ns::Int, nd::Int, dt::Float64, dir = (0, pi))
    N = size(c, 1)
    m = SharedArray{Float64}((ns * nd, 9))
    prm = CShootParams(ns, nd, 0, ns, Float64(dir[1]), Float64(dir[2]), dt, TIME_SECONDS, 0)
    med = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:ray_group_medium_create_grid, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cdouble, Ref{Ptr{Cvoid}}), g.h, c, N, ext, med))
    try
        check(ccall((:ray_group_shoot_to_exit, LIB), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{CObstacle}, Ref{CShootParams}, Ptr{Cdouble}, Int64,
                     Ptr{Int32}, Ptr{Int32}, Ptr{Int32}),
                    g.h, med[], C_NULL, prm, m, ns * nd, C_NULL, C_NULL, C_NULL))
    finally
        ccall((:ray_group_medium_destroy, LIB), Cint, (Ptr{Cvoid},), med[])
    end
    g.rows = ns * nd
    m
end

"NonReflectionTrace (Obstacle.jl:275) over all GPUs of the group; the members keep the traced states for misfit(g)"
function NonReflectionTrace(g::RayGroup, c::Matrix{Float64}, m, ext::Float64, dt::Float64; nd::Int = 0,
                            mode::Cint = TRACE_OVERSHOOT)
    N = size(c, 1)
    num = size(m, 1)
    mm = Matrix{Float64}(m)
    X = Matrix{Float64}(undef, num, 4)
    med = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:ray_group_medium_create_grid, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cdouble, Ref{Ptr{Cvoid}}), g.h, c, N, ext, med))
    try
        check(ccall((:ray_group_trace_fixed_time, LIB), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Int64, Int64, Int32, Cdouble, Int32, Ptr{Cdouble}, Int64,
                     Ptr{Int32}, Ptr{Int32}),
                    g.h, med[], mm, num, max(num, 1), nd, dt, mode, X, max(num, 1), C_NULL, C_NULL))
    finally
        ccall((:ray_group_medium_destroy, LIB), Cint, (Ptr{Cvoid},), med[])
    end
    g.rows = num
    m[:, 5:8] = X
    m
end

"ScatterForwardOperator (NonObstacle.jl:189) over all GPUs of the group: (M, s) as the reference's master holds them"
function ScatterForwardOperator(g::RayGroup, c::Matrix{Float64}, m, ext::Float64, dt::Float64; nd::Int = 0,
                                droptol::Float64 = 1e-12)
    num = size(m, 1); N = size(c, 1)
    mm = Matrix{Float64}(m)
    X = Matrix{Float64}(undef, num, 4)
    s = SharedArray{Float64}((num, 9))
    s[:, 1:4] = m[:, 1:4]; s[:, 9] = m[:, 9]
    med = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:ray_group_medium_create_grid, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cdouble, Ref{Ptr{Cvoid}}), g.h, c, N, ext, med))
    local M
    try
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:ray_group_forward_operator, LIB), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Int64, Int64, Int32, Cdouble, Cdouble, Ptr{Cdouble}, Int64,
                     Ptr{Int32}, Ptr{Int32}, Ref{Ptr{Cvoid}}),
                    g.h, med[], mm, num, max(num, 1), nd, dt, droptol, X, max(num, 1), C_NULL, C_NULL, h))
        try
            nr = Ref{Int64}(0); nc = Ref{Int64}(0); nz = Ref{Int64}(0)
            check(ccall((:ray_group_frechet_shape, LIB), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ref{Int64}), h[], nr, nc, nz))
            indptr = Vector{Int64}(undef, nr[] + 1); indices = Vector{Int32}(undef, nz[]); data = Vector{Float64}(undef, nz[])
            check(ccall((:ray_group_frechet_export_csr, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int32}, Ptr{Cdouble}),
                        h[], indptr, indices, data))
            Mt = SparseMatrixCSC(Int(nc[]), Int(nr[]), indptr .+ 1, Int64.(indices) .+ 1, data)   # = M'
            M = copy(transpose(Mt))
        finally
            ccall((:ray_group_frechet_destroy, LIB), Cint, (Ptr{Cvoid},), h[])
        end
    finally
        ccall((:ray_group_medium_destroy, LIB), Cint, (Ptr{Cvoid},), med[])
    end
    s[:, 5:8] = X
    M, s
end

"the observation the group's misfit is taken against (rows as the last ScatterRelation(g, ...) laid them out)"
function set_observation!(g::RayGroup, observation::AbstractMatrix{Float64})
    o = Matrix{Float64}(observation)
    check(ccall((:ray_group_set_observation, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Int64, Int64),
                g.h, o, size(o, 1), max(size(o, 1), 1)))
end

"misfit of the slabs resident from the last ScatterRelation(g, ...): one RCCL all-reduce of the two sums"
function misfit(g::RayGroup, perc_thresh::Float64 = 0.2)
    out = zeros(2); mask = zeros(UInt8, g.rows); mism = zeros(4 * g.rows)
    check(ccall((:ray_group_misfit, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{UInt8}, Cdouble, Ptr{Cdouble}),
                g.h, out, mask, perc_thresh, mism))
    mism, sqrt(out[1] / out[2]), mask
end

# ---- NonObstacle.jl:336,372; ObstacleDemo.jl:219-226 --------------------------------------------------------
"returns (mismatch::Vector (4num, ray-major), objective, mask::Vector{UInt8})"
function misfit(m::AbstractMatrix{Float64}, observation::AbstractMatrix{Float64}, perc_thresh::Float64 = 0.2)
    num = size(m, 1)
    a = Matrix{Float64}(m); b = Matrix{Float64}(observation)
    out = zeros(2); mask = zeros(UInt8, num); mism = zeros(4 * num)
    check(ccall((:ray_misfit, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Int64, Ptr{Cdouble}, Ptr{UInt8}, Cdouble, Ptr{Cdouble}),
                context(), a, b, num, max(num, 1), out, mask, perc_thresh, mism))
    mism, sqrt(out[1] / out[2]), mask
end

# ---- RAYFIX1 reader (ray_amd/fixture.py): the golden vectors of tests/golden for a Julia-side check -----
# scripts/npz_to_rayfix.py writes them; arrays come back column-major, exactly as stored.
function read_rayfix(path::AbstractString)
    open(path, "r") do io
        readline(io) == "RAYFIX1" || error("$path: not a RAYFIX1 file")
        head = readline(io)
        body = read(io)
        out = Dict{String,Any}()
        # minimal parse of {"arrays": [{"name": "..", "dtype": "..", "shape": [..], "offset": n}, ...]}
        for mt in eachmatch(r"\{\"name\": \"([^\"]+)\", \"dtype\": \"(\w+)\", \"shape\": \[([^\]]*)\], \"offset\": (\d+)\}", head)
            T = Dict("f8" => Float64, "i4" => Int32, "i8" => Int64, "u1" => UInt8)[mt.captures[2]]
            shape = isempty(strip(mt.captures[3])) ? () : Tuple(parse.(Int, split(mt.captures[3], ",")))
            off = parse(Int, mt.captures[4])
            n = prod(shape)                                   # prod(()) == 1: a scalar
            v = collect(reinterpret(T, body[off + 1:off + n * sizeof(T)]))
            out[mt.captures[1]] = isempty(shape) ? v[1] : reshape(v, shape)
        end
        out
    end
end

end # module
