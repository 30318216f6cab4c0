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
       misfit, set_arith!, ARITH_EXACT, ARITH_FUSED, read_rayfix

const LIB = get(ENV, "RAY_AMD_LIB", joinpath(@__DIR__, "..", "ray_amd", "libray_amd.so"))

const ARITH_EXACT = Cint(0)
const ARITH_FUSED = Cint(1)
const TIME_SECONDS = Cint(0)
const TIME_STEPS = Cint(1)
const TRACE_SHORTEN_CEASE = Cint(0)
const TRACE_OVERSHOOT = Cint(1)

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
        atexit(() -> (ccall((:ray_ctx_destroy, LIB), Cint, (Ptr{Cvoid},), CTX[]); CTX[] = C_NULL))
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

function with_medium(f, c::Matrix{Float64}, ext::Float64)
    N = size(c, 1)
    size(c, 2) == N || error("c must be N x N")
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:ray_medium_create_grid, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cdouble, Ref{Ptr{Cvoid}}), context(), c, N, ext, h))
    try
        f(h[])
    finally
        ccall((:ray_medium_destroy, LIB), Cint, (Ptr{Cvoid},), h[])
    end
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

function trace(c::Matrix{Float64}, m::AbstractMatrix{Float64}, ext::Float64, dt::Float64, mode::Cint)
    num = size(m, 1)
    mm = Matrix{Float64}(m)                       # dense column-major copy (m may be a view)
    X = Matrix{Float64}(undef, num, 4)
    with_medium(c, ext) do med
        check(ccall((:ray_trace_fixed_time, LIB), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Int64, Int64, Cdouble, Cint, Ptr{Cdouble}, Int64,
                     Ptr{Int32}, Ptr{Int32}),
                    context(), med, mm, num, max(num, 1), dt, mode, X, max(num, 1), C_NULL, C_NULL))
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
function NonReflectionTrace(c::Matrix{Float64}, m, ext::Float64, dt::Float64)
    m[:, 5:8] = trace(c, m, ext, dt, TRACE_OVERSHOOT)
    m
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
