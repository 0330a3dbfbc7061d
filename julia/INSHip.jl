# INSHip.jl — glue that makes libinship.so a backend of IncompressibleNavierStokes.jl.
#
# NOT EXECUTED HERE: Julia is not installed in the build container (SURVEY.md §8b/§8c).  What can be checked without Julia is checked
# mechanically: tests/test_julia_glue.py parses every `ccall` of this file and compares symbol, arity and argument types with the
# prototypes of include/ins_hip.h, and checks that every reference function the glue claims to cover has a method here.
#
# Shape: a weak-dependency package extension in the style of ext/IncompressibleNavierStokesCUDSSExt.jl.  The user writes what they
# would write for any AMDGPU run of the reference,
#
#     using IncompressibleNavierStokes, AMDGPU, INSHip
#     setup = Setup(; x, boundary_conditions, Re, backend = ROCBackend())
#     state, outputs = solve_unsteady(; setup, ustart, tlims, Δt)          # default_psolver(setup) reaches the library too
#
# and dispatch does the rest: every `op!` method below is specialised on `ROCArray{Float64}` arguments, the setup-only constructors
# (`psolver_spectral`, `psolver_cg`, `ode_method_cache`) on a setup whose `backend` field is a `ROCBackend`, `psolver_direct` on the array
# type exactly like the CUDSS extension (pressure.jl:101-117).  AMDGPU.jl is used for device memory ONLY (ROCArray allocation and raw
# pointers); every kernel lives behind the C ABI of include/ins_hip.h.  No KernelAbstractions kernel of the reference runs.
module INSHip

using IncompressibleNavierStokes
using LinearAlgebra
using AMDGPU
using AMDGPU: ROCArray, ROCBackend, HIP
import IncompressibleNavierStokes:
    apply_bc_u!, apply_bc_p!, apply_bc_temp!, divergence!, scalewithvolume!, pressuregradient!, applypressure!, laplacian!,
    convection!, diffusion!, convectiondiffusion!, momentum!, project!, poisson!, psolver_spectral, psolver_cg, psolver_direct,
    timestep!, ode_method_cache, get_cfl_timestep!, kinetic_energy!, total_kinetic_energy,
    ExplicitRungeKuttaMethod, PeriodicBC, DirichletBC, SymmetricBC, PressureBC,
    vorticity!, interpolate_u_p!, interpolate_ω_p!, Qfield!, Dfield!, eig2field!, dissipation_from_strain!,
    convection_diffusion_temp!, dissipation!, gravity!, smagorinsky_closure

const lib = get(ENV, "INSHIP_LIB", "libinship.so")
const RA = ROCArray{Float64}

# A setup whose `backend` is a ROCBackend.  Field order of the NamedTuple: setup.jl:14-24
# (grid, boundary_conditions, Re, bodyforce, issteadybodyforce, closure_model, backend, workgroupsize, temperature[, dbodyforce]).
const ROCSetup = NamedTuple{N,<:Tuple{Any,Any,Any,Any,Any,Any,ROCBackend,Vararg{Any}}} where {N}

# ins_grid_desc_t (include/ins_hip.h) — field order and types must match the C struct exactly.
struct GridDesc
    D::Int32
    N::NTuple{3,Int32}
    dx::NTuple{3,Ptr{Float64}}
    dxu::NTuple{3,Ptr{Float64}}
    A1::NTuple{9,Ptr{Float64}}     # [α][β] row-major
    A2::NTuple{9,Ptr{Float64}}
    iu_lo::NTuple{9,Int32}
    iu_hi::NTuple{9,Int32}
    ip_lo::NTuple{3,Int32}
    ip_hi::NTuple{3,Int32}
    bc::NTuple{6,Int32}            # [β][side]
    bc_u::NTuple{18,Float64}       # [β][side][α]
end

check(rc) = rc == 0 || error("libinship: ", unsafe_string(ccall((:ins_last_error, lib), Cstring, ())))
stream() = Ptr{Cvoid}(HIP.stream().stream)   # the task-local HIP stream AMDGPU.jl is using
bccode(::PeriodicBC) = Int32(0)
bccode(::DirichletBC) = Int32(1)
bccode(::SymmetricBC) = Int32(2)
bccode(::PressureBC) = Int32(3)

"Constant Dirichlet data `bc.u::Tuple` of side (β, s) for component α (boundary_conditions.jl:347-350); 0 for no-slip / closures / other BCs."
bcconst(bc, α) = 0.0
bcconst(bc::DirichletBC, α) = bc.u isa Tuple ? Float64(bc.u[α]) : 0.0
"Does this side carry a closure `bc.u(α, x..., t)` (boundary_conditions.jl:351-357)?  Its values travel as plane buffers."
isclosure(bc) = bc isa DirichletBC && !(isnothing(bc.u) || bc.u isa Tuple)

"Device handle for `setup.grid`, built once per setup from the HOST copies of the 1-D metric vectors (grid.jl:177-248)."
function grid_handle(setup)
    g = setup.grid
    bcs = setup.boundary_conditions
    D = g.dimension()
    host(v) = Array(v)                                   # metrics are tiny; the library keeps its own device copy
    Δ, Δu = host.(g.Δ), host.(g.Δu)
    A = [(host(g.A[α][β][1]), host(g.A[α][β][2])) for α = 1:D, β = 1:D]
    pad3(f, T) = ntuple(i -> i <= D ? f(i) : zero(T), 3)
    ptr3(v) = ntuple(i -> i <= D ? pointer(v[i]) : Ptr{Float64}(0), 3)
    idx9(f, T) = ntuple(k -> (α = (k - 1) ÷ 3 + 1; β = (k - 1) % 3 + 1; α <= D && β <= D ? f(α, β) : zero(T)), 9)
    desc = GridDesc(
        D, pad3(i -> Int32(g.N[i]), Int32), ptr3(Δ), ptr3(Δu),
        idx9((α, β) -> pointer(A[α, β][1]), Ptr{Float64}), idx9((α, β) -> pointer(A[α, β][2]), Ptr{Float64}),
        idx9((α, β) -> Int32(first(g.Iu[α].indices[β]) - 1), Int32),    # Julia a:b -> [a-1, b)
        idx9((α, β) -> Int32(last(g.Iu[α].indices[β])), Int32),
        pad3(i -> Int32(first(g.Ip.indices[i]) - 1), Int32), pad3(i -> Int32(last(g.Ip.indices[i])), Int32),
        ntuple(k -> (β = (k - 1) ÷ 2 + 1; s = (k - 1) % 2 + 1; β <= D ? bccode(bcs[β][s]) : Int32(0)), 6),
        # constant Dirichlet data as [β][side][α] (a lid (1, 0.2, 0) on the upper y side sits at β = 2, side = 2)
        ntuple(k -> (β = (k - 1) ÷ 6 + 1; s = ((k - 1) ÷ 3) % 2 + 1; α = (k - 1) % 3 + 1; β <= D && α <= D ? bcconst(bcs[β][s], α) : 0.0), 18),
    )
    h = Ref{Ptr{Cvoid}}()
    GC.@preserve Δ Δu A check(ccall((:ins_grid_create, lib), Cint, (Ref{GridDesc}, Ref{Ptr{Cvoid}}), desc, h))
    h[]
end

# One cached handle per setup (setup is an immutable NamedTuple: key on objectid of its grid).
const HANDLES = Dict{UInt,Ptr{Cvoid}}()
handle(setup) = get!(() -> grid_handle(setup), HANDLES, objectid(setup.grid))

# ---- boundary conditions (boundary_conditions.jl) -----------------------------------------------------------
"Plane buffers for closure Dirichlet data: slot 3(2(β-1)+side-1)+α holds `bc.u(α, x..., t)` (or its time derivative) on `boundary(β, N, Iu[α], isright)`."
function bc_planes(u::RA, t, setup, dudt)
    g, bcs = setup.grid, setup.boundary_conditions
    D = g.dimension()
    any(isclosure, Iterators.flatten(bcs)) || return nothing, nothing
    planes, keep = fill(Ptr{Float64}(C_NULL), 18), Any[]
    for β = 1:D, (side, bc) in enumerate(bcs[β])
        isclosure(bc) || continue
        for α = 1:D
            I = IncompressibleNavierStokes.boundary(β, g.N, g.Iu[α], side == 2)
            xI = ntuple(γ -> reshape(Array(g.xu[α][γ])[I.indices[γ]], ntuple(Returns(1), γ - 1)..., :), D)
            vals = if dudt                                  # central difference in time, boundary_conditions.jl:351-357
                h = sqrt(eps(Float64)) * max(abs(t), one(t))
                (bc.u.(α, xI..., t + h) .- bc.u.(α, xI..., t - h)) ./ 2h
            else
                bc.u.(α, xI..., t)
            end
            buf = ROCArray(vec(Float64.(vals)))             # memory order of the plane: fastest direction first
            push!(keep, buf)
            planes[3 * (2 * (β - 1) + side - 1) + α] = pointer(buf)
        end
    end
    planes, keep
end
function apply_bc_u!(u::RA, t, setup; dudt = false, kwargs...)
    planes, keep = bc_planes(u, t, setup, dudt)
    GC.@preserve keep check(ccall((:ins_apply_bc_u_f64, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cint, Ptr{Ptr{Float64}}, Ptr{Cvoid}),
                                  handle(setup), pointer(u), dudt, isnothing(planes) ? C_NULL : planes, stream()))
    isnothing(keep) || AMDGPU.synchronize()                  # the plane buffers may be collected after this call
    u
end
function apply_bc_p!(p::RA, t, setup; kwargs...)
    check(ccall((:ins_apply_bc_p_f64, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Cvoid}), handle(setup), pointer(p), stream()))
    p
end

# ---- operators (operators.jl) -------------------------------------------------------------------------------
function divergence!(div::RA, u::RA, setup)
    check(ccall((:ins_divergence_f64, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
                handle(setup), pointer(u), pointer(div), stream()))
    div
end
function scalewithvolume!(p::RA, setup)
    check(ccall((:ins_scalewithvolume_f64, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Cvoid}), handle(setup), pointer(p), stream()))
    p
end
function pressuregradient!(G::RA, p::RA, setup)
    check(ccall((:ins_pressuregradient_f64, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
                handle(setup), pointer(p), pointer(G), stream()))
    G
end
function applypressure!(u::RA, p::RA, setup)
    check(ccall((:ins_applypressure_f64, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
                handle(setup), pointer(u), pointer(p), stream()))
    u
end
function laplacian!(L::RA, p::RA, setup)
    check(ccall((:ins_laplacian_f64, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
                handle(setup), pointer(p), pointer(L), stream()))
    L
end
function convection!(F::RA, u::RA, setup)                    # accumulates into F (operators.jl:378-415)
    check(ccall((:ins_convection_f64, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
                handle(setup), pointer(u), pointer(F), stream()))
    F
end
function diffusion!(F::RA, u::RA, setup; use_viscosity = true)   # accumulates into F (operators.jl:537-573)
    check(ccall((:ins_diffusion_f64, lib), Cint, (Ptr{Cvoid}, Cdouble, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
                handle(setup), use_viscosity ? 1 / setup.Re : 1.0, pointer(u), pointer(F), stream()))
    F
end
function convectiondiffusion!(F::RA, u::RA, setup)           # accumulates into F (operators.jl:634-690)
    check(ccall((:ins_convectiondiffusion_f64, lib), Cint, (Ptr{Cvoid}, Cdouble, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
                handle(setup), 1 / setup.Re, pointer(u), pointer(F), stream()))
    F
end
# momentum!(F, u, temp, t, setup): the fused fill + convection-diffusion pass, then body force and gravity (operators.jl:967-976)
function momentum!(F::RA, u::RA, temp, t, setup)
    check(ccall((:ins_momentum_f64, lib), Cint, (Ptr{Cvoid}, Cdouble, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
                handle(setup), 1 / setup.Re, pointer(u), pointer(F), stream()))
    isnothing(setup.bodyforce) || IncompressibleNavierStokes.applybodyforce!(F, u, t, setup)   # broadcast add of a ROCArray
    isnothing(temp) || gravity!(F, temp, setup)
    F
end
function kinetic_energy!(ke::RA, u::RA, setup; interpolate_first = false)
    check(ccall((:ins_kinetic_energy_f64, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Cint, Ptr{Cvoid}),
                handle(setup), pointer(u), pointer(ke), interpolate_first, stream()))
    ke
end
function total_kinetic_energy(u::RA, setup; interpolate_first = false)   # blocking: a scalar is read (operators.jl:1541-1556)
    e = Ref{Float64}()
    check(ccall((:ins_total_kinetic_energy_f64, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cint, Ref{Float64}, Ptr{Cvoid}),
                handle(setup), pointer(u), interpolate_first, e, stream()))
    e[]
end
function get_cfl_timestep!(buf, u::RA, setup)                # blocking: a scalar is read (solver.jl:101-125); `buf` is not needed
    dt = Ref{Float64}()
    check(ccall((:ins_cfl_timestep_f64, lib), Cint, (Ptr{Cvoid}, Cdouble, Ptr{Float64}, Ref{Float64}, Ptr{Cvoid}),
                handle(setup), setup.Re, pointer(u), dt, stream()))
    dt[]
end

# ---- pressure solvers (pressure.jl) --------------------------------------------------------------------------
mutable struct HipPSolver
    h::Ptr{Cvoid}
    setup::Any
    function HipPSolver(h, setup)
        s = new(h, setup)
        finalizer(x -> ccall((:ins_poisson_destroy, lib), Cint, (Ptr{Cvoid},), x.h), s)
    end
end
"Create spectral Poisson solver from setup (pressure.jl:289-351); `default_psolver(setup)` (pressure.jl:85-98) lands here for periodic uniform boxes."
function psolver_spectral(setup::ROCSetup)
    IncompressibleNavierStokes.assert_uniform_periodic(setup, "Spectral psolver")
    h = Ref{Ptr{Cvoid}}()
    check(ccall((:ins_poisson_spectral_create, lib), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), handle(setup), h))
    HipPSolver(h[], setup)
end
"Conjugate gradients iterative Poisson solver (pressure.jl:209-286), Jacobi preconditioner (`create_laplace_diag`, pressure.jl:188-206)."
function psolver_cg(setup::ROCSetup; abstol = 0.0, reltol = sqrt(eps(Float64)), maxiter = prod(setup.grid.Np), preconditioner = nothing)
    isnothing(preconditioner) || error("INSHip.psolver_cg: only the built-in Jacobi preconditioner runs on the device")
    h = Ref{Ptr{Cvoid}}()
    check(ccall((:ins_poisson_cg_create, lib), Cint, (Ptr{Cvoid}, Cdouble, Cdouble, Int64, Ref{Ptr{Cvoid}}),
                handle(setup), abstol, reltol, maxiter, h))
    HipPSolver(h[], setup)
end

"""
The 1-D factor `Tα` of `laplacian!` (operators.jl:328-350) on `Ip[α]`: `(p₊ - p)/Δu[I] - (p - p₋)/Δu[I-1]` with the reference's branch
order for the first / last volume; ghost pressures of the `else` branch follow `apply_bc_p!` (boundary_conditions.jl:186-215: periodic
wrap, Neumann copy for Dirichlet / Symmetric, zero for Pressure).  `laplacian_mat(setup)` (matrices.jl:484-492) is
`Σα Tα ⊗ (⊗β≠α Dβ)` with `Dβ = Diagonal(Δ[β][Ip[β]])`.
"""
function laplacian_1d(setup, α)
    g = setup.grid
    r = g.Ip.indices[α]
    n = length(r)
    Δu = Array(g.Δu[α])
    bl, br = setup.boundary_conditions[α]
    T = zeros(n, n)
    for i = 1:n
        I = r[i]
        cr, cl = 1 / Δu[I], 1 / Δu[I-1]
        isfirst, islast = i == 1, i == n
        right = left = true          # which one-sided differences survive
        ghost_r = ghost_l = 0        # column the ghost value aliases; 0 = the ghost pressure is zero (PressureBC)
        if isfirst && bl isa PressureBC
        elseif islast && br isa PressureBC
        elseif isfirst && bl isa DirichletBC
            left = false
        elseif islast && br isa DirichletBC
            right = false
        else
            isfirst && (ghost_l = bl isa PeriodicBC ? n : i)
            islast && (ghost_r = br isa PeriodicBC ? 1 : i)
        end
        if right
            T[i, i] -= cr
            j = islast ? ghost_r : i + 1
            j > 0 && (T[i, j] += cr)
        end
        if left
            T[i, i] -= cl
            j = isfirst ? ghost_l : i - 1
            j > 0 && (T[i, j] += cl)
        end
    end
    T
end

"""
Eigen-decomposition of the symmetric 1-D factor.  On a symmetric grid (cosine, tanh, uniform walls) the matrix commutes with the reflection
`i -> n+1-i`; its even and odd blocks `A ± B J` (A = M[1:h,1:h], B = M[1:h,h+1:n]) are decomposed separately, so every eigenvector is EXACTLY
even or odd (the plain solver mixes the nearly degenerate wall-mode pairs of a strongly stretched grid) and the library runs that direction as
two half-size GEMMs on folded data (csrc/ins_fdm.hip).  Same as `_eigh_even_odd` in incompressiblenavierstokes.jl_amd/pressure.py.
"""
function eigen_even_odd(M::Matrix{Float64})
    n = size(M, 1)
    R = reverse(reverse(M; dims = 1); dims = 2)
    centro = iseven(n) && n >= 8 && isapprox(M, R; rtol = 1e-12, atol = 1e-12 * maximum(abs, M))
    if !centro
        E = eigen(Symmetric(M))
        return E.values, E.vectors
    end
    h = n ÷ 2
    Ms = (M + M' + R + R') / 4
    A, BJ = Ms[1:h, 1:h], reverse(Ms[1:h, h+1:n]; dims = 2)
    Ee, Eo = eigen(Symmetric(A + BJ)), eigen(Symmetric(A - BJ))
    W = [Ee.vectors Eo.vectors; reverse(Ee.vectors; dims = 1) -reverse(Eo.vectors; dims = 1)] / sqrt(2)
    [Ee.values; Eo.values], W
end

"""
`psolver_direct` for `ROCArray`s (pressure.jl:101-154; same dispatch hook as ext/IncompressibleNavierStokesCUDSSExt.jl:18): fast
diagonalisation of the separable Laplacian.  The generalised eigenpairs `Tα v = λ Dα v` of the 1-D factors are computed here once and
handed to the library, which solves with six fp64 GEMMs on rocBLAS (Fourier passes in periodic uniform directions); singular systems
are solved in the reference's bordered form (pressure.jl:133-140).
"""
function psolver_direct(::ROCArray, setup)
    D = setup.grid.dimension()
    V, λ = Matrix{Float64}[], Vector{Float64}[]
    for α = 1:D
        d = Array(setup.grid.Δ[α])[setup.grid.Ip.indices[α]]
        T = laplacian_1d(setup, α)
        S = Diagonal(d .^ -0.5)
        vals, vecs = eigen_even_odd(Matrix(S * T * S))
        push!(V, S * vecs)           # column-major, V'DV = I
        push!(λ, vals)
    end
    Vp = [pointer(V[min(α, D)]) for α = 1:3]
    λp = [pointer(λ[min(α, D)]) for α = 1:3]
    h = Ref{Ptr{Cvoid}}()
    GC.@preserve V λ check(ccall((:ins_poisson_fdm_create, lib), Cint, (Ptr{Cvoid}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ref{Ptr{Cvoid}}),
                                 handle(setup), Vp, λp, h))
    HipPSolver(h[], setup)
end
(s::HipPSolver)(p::RA) =
    (check(ccall((:ins_poisson_solve_f64, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Cvoid}), s.h, pointer(p), stream())); p)
poisson!(psolver::HipPSolver, p::RA) = psolver(p)
function project!(u::RA, setup; psolver::HipPSolver, p::RA)
    check(ccall((:ins_project_f64, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
                handle(setup), psolver.h, pointer(u), pointer(p), stream()))
    u
end

# ---- step-adjacent operators (SURVEY §8f rows 2 and 4): with these methods the reference's own host-driven `timestep!`
# (closure model / temperature / body force present) runs entirely on libinship's kernels --------------------------------
vorticity!(ω::RA, u::RA, setup) =
    (check(ccall((:ins_vorticity_f64, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}), handle(setup), pointer(u), pointer(ω), stream())); ω)
interpolate_u_p!(up::RA, u::RA, setup) =
    (check(ccall((:ins_interpolate_u_p_f64, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}), handle(setup), pointer(u), pointer(up), stream())); up)
interpolate_ω_p!(ωp::RA, ω::RA, setup) =
    (check(ccall((:ins_interpolate_w_p_f64, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}), handle(setup), pointer(ω), pointer(ωp), stream())); ωp)
Qfield!(Q::RA, u::RA, setup) =
    (check(ccall((:ins_qfield_f64, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}), handle(setup), pointer(u), pointer(Q), stream())); Q)
Dfield!(d::RA, G::RA, p::RA, setup; ϵ = eps(Float64)) =
    (check(ccall((:ins_dfield_f64, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cdouble, Ptr{Cvoid}),
                 handle(setup), pointer(p), pointer(G), pointer(d), ϵ, stream())); d)
eig2field!(λ::RA, u::RA, setup) =
    (check(ccall((:ins_eig2field_f64, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}), handle(setup), pointer(u), pointer(λ), stream())); λ)
dissipation_from_strain!(ϵ::RA, u::RA, setup) =
    (check(ccall((:ins_dissipation_from_strain_f64, lib), Cint, (Ptr{Cvoid}, Cdouble, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
                 handle(setup), 1 / setup.Re, pointer(u), pointer(ϵ), stream())); ϵ)
convection_diffusion_temp!(c::RA, u::RA, temp::RA, setup) =
    (check(ccall((:ins_convection_diffusion_temp_f64, lib), Cint, (Ptr{Cvoid}, Cdouble, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
                 handle(setup), setup.temperature.α4, pointer(u), pointer(temp), pointer(c), stream())); c)
dissipation!(diss::RA, diff::RA, u::RA, setup) =
    (check(ccall((:ins_dissipation_f64, lib), Cint, (Ptr{Cvoid}, Cdouble, Cdouble, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
                 handle(setup), 1 / setup.Re, setup.Re * setup.temperature.α1 / setup.temperature.γ, pointer(u), pointer(diff), pointer(diss), stream())); diss)
gravity!(F::RA, temp::RA, setup) =
    (check(ccall((:ins_gravity_f64, lib), Cint, (Ptr{Cvoid}, Cint, Cdouble, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
                 handle(setup), setup.temperature.gdir - 1, setup.temperature.α2, pointer(temp), pointer(F), stream())); F)
# apply_bc_temp!: BC codes / Dirichlet constants per (β, side); closures `bc.u(x..., t)` are evaluated here into plane buffers
function apply_bc_temp!(temp::RA, t, setup; kwargs...)
    bcs = setup.temperature.boundary_conditions
    D = length(bcs)
    codes, vals = zeros(Int32, 6), zeros(Float64, 6)
    planes, keep = fill(Ptr{Float64}(C_NULL), 6), Any[]
    for β = 1:D, (side, bc) in enumerate(bcs[β])
        q = 2(β - 1) + side
        codes[q] = bccode(bc)
        bc isa DirichletBC || continue
        if bc.u isa Number
            vals[q] = bc.u
        elseif !isnothing(bc.u)
            I = IncompressibleNavierStokes.boundary(β, setup.grid.N, setup.grid.Ip, side == 2)
            xI = ntuple(α -> reshape(Array(setup.grid.xp[α])[I.indices[α]], ntuple(Returns(1), α - 1)..., :), D)
            buf = ROCArray(vec(bc.u.(xI..., t)))          # memory order of the plane: fastest direction first
            push!(keep, buf); planes[q] = pointer(buf)
        end
    end
    GC.@preserve keep check(ccall((:ins_apply_bc_temp_f64, lib), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Float64}, Ptr{Ptr{Float64}}, Ptr{Float64}, Ptr{Cvoid}),
                                  handle(setup), codes, vals, planes, pointer(temp), stream()))
    isempty(keep) || AMDGPU.synchronize()
    temp
end
# smagorinsky_closure(setup): σ as D(D+1)/2 scalar fields [xx, yy, (zz), xy, (xz, yz)].  A callable struct, so that `timestep!` can tell this
# closure from a user function and carry it inside the native stage loop (ins_rk_set_closure).
mutable struct HipSmagorinsky{S,B}
    setup::S
    σ::Any     # stress scratch of the three-kernel route, allocated when a call first needs it (the one-kernel route keeps σ in registers)
    s::B
end
smagorinsky_closure(setup::ROCSetup) = HipSmagorinsky(setup, nothing, IncompressibleNavierStokes.vectorfield(setup))
function (m::HipSmagorinsky)(u, θ)
    (; setup, s) = m
    # smagtensor! -> apply_bc_p!(σ) -> divoftensor! (operators.jl:1284-1300) behind one entry point: one kernel wherever the grid allows it (the
    # stress stays in registers), the three kernels with σ as scratch elsewhere
    σp = Ptr{Float64}(C_NULL)
    if ccall((:ins_smagorinsky_force_needs_sigma, lib), Cint, (Ptr{Cvoid},), handle(setup)) != 0
        if isnothing(m.σ)
            D = setup.grid.dimension()
            m.σ = similar(setup.grid.x[1], Float64, (setup.grid.N..., D * (D + 1) ÷ 2)); fill!(m.σ, 0)
        end
        σp = pointer(m.σ)
    end
    check(ccall((:ins_smagorinsky_force_f64, lib), Cint, (Ptr{Cvoid}, Cdouble, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
                handle(setup), θ, pointer(u), σp, pointer(s), stream()))
    s
end
# observespectrum: shells from spectral_stuff (host), everything else on the device
function spectrum_handle(setup; kwargs...)
    (; inds) = IncompressibleNavierStokes.spectral_stuff(setup; kwargs...)
    off = Int64[0; cumsum(length.(inds))]
    flat = Int64.(reduce(vcat, Array.(inds))) .- 1
    h = Ref{Ptr{Cvoid}}()
    check(ccall((:ins_spectrum_create, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Int64}, Ptr{Int64}, Ref{Ptr{Cvoid}}), handle(setup), length(inds), off, flat, h))
    h[]
end
spectrum!(ehat::RA, h::Ptr{Cvoid}, u::RA) =
    (check(ccall((:ins_spectrum_f64, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}), h, pointer(u), pointer(ehat), stream())); ehat)

# ---- explicit RK (step_explicit_runge_kutta.jl) -----------------------------------------------------------------
# `solve_unsteady` builds its cache with `ode_method_cache(method, setup)` (solver.jl:33) before it knows the pressure solver, so the native
# cache is created at the first `timestep!` from the stepper's psolver; `ref` is the reference's own cache (time_stepper_caches.jl:34-49)
# for the host-driven stage loop (closure model / temperature / unsteady body force), also created on demand.
mutable struct HipRKCache
    h::Ptr{Cvoid}
    psolver::Any
    ref::Any
end
ode_method_cache(method::ExplicitRungeKuttaMethod, setup::ROCSetup) = HipRKCache(C_NULL, nothing, nothing)
function native!(cache::HipRKCache, method, setup, psolver)
    cache.h != C_NULL && cache.psolver === psolver && return cache.h
    cache.h == C_NULL || ccall((:ins_rk_destroy, lib), Cint, (Ptr{Cvoid},), cache.h)
    ns = length(method.b)
    A = collect(transpose(Float64.(method.A)))   # row-major for C; method.A is already the SHIFTED tableau (methods.jl:231-236)
    c = Float64.(method.c)
    h = Ref{Ptr{Cvoid}}()
    check(ccall((:ins_rk_create, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Ref{Ptr{Cvoid}}),
                handle(setup), psolver.h, ns, A, c, h))
    cache.psolver = psolver
    cache.h = h[]
end
# ins_temperature_desc_t (include/ins_hip.h): the scalars of `temperature_equation` (setup.jl:48-87) the native stage loop needs
struct TempDesc
    a2::Float64
    a4::Float64
    diss_coef::Float64
    gdir::Int32
    dodissipation::Int32
    bc::NTuple{6,Int32}
    val::NTuple{6,Float64}
end
function tempdesc(setup)
    T = setup.temperature
    D = setup.grid.dimension()
    bcs = T.boundary_conditions
    code(β, s) = β <= D ? Int32(bccode(bcs[β][s])) : Int32(0)
    value(β, s) = (β <= D && bcs[β][s] isa DirichletBC && bcs[β][s].u isa Number) ? Float64(bcs[β][s].u) : 0.0
    TempDesc(T.α2, T.α4, setup.Re * T.α1 / T.γ, T.gdir - 1, T.dodissipation ? 1 : 0,
             ntuple(q -> code((q + 1) ÷ 2, 2 - q % 2), 6), ntuple(q -> value((q + 1) ÷ 2, 2 - q % 2), 6))
end
# A stepper (create_stepper: (; setup, psolver, u, temp, t, n), step_explicit_runge_kutta.jl:1-2) whose pressure solver is the library's
const HipStepper = NamedTuple{N,<:Tuple{Any,HipPSolver,Vararg{Any}}} where {N}
function timestep!(method::ExplicitRungeKuttaMethod, stepper::HipStepper, Δt; θ = nothing, cache::HipRKCache)
    (; setup, psolver, u, temp, t, n) = stepper
    # The fused native step is valid only without closure model / temperature / unsteady body force and with time-independent boundary
    # data (SURVEY.md §8b caveat); otherwise the reference's own stage loop runs, on the operator-level methods above, so user callbacks
    # can run between kernels.  A steady body force rides inside the native stage kernels (ins_rk_set_bodyforce).
    steady = (isnothing(setup.bodyforce) || setup.issteadybodyforce) && !any(isclosure, Iterators.flatten(setup.boundary_conditions))
    native = steady && isnothing(setup.closure_model) && isnothing(temp)
    # The temperature equation and this glue's Smagorinsky closure (constant θ) ride inside the native stage loop (ins_rk_step_ext_f64)
    ext = steady && !native && (isnothing(setup.closure_model) || (setup.closure_model isa HipSmagorinsky && θ isa Real)) &&
          (isnothing(temp) || !any(b -> b isa DirichletBC && !(isnothing(b.u) || b.u isa Number), Iterators.flatten(setup.temperature.boundary_conditions)))
    if ext
        h = native!(cache, method, setup, psolver)
        check(ccall((:ins_rk_set_bodyforce, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), h,
                    isnothing(setup.bodyforce) ? Ptr{Float64}(C_NULL) : pointer(setup.bodyforce)))
        check(ccall((:ins_rk_set_closure, lib), Cint, (Ptr{Cvoid}, Cint, Cdouble), h, isnothing(setup.closure_model) ? 0 : 1,
                    isnothing(setup.closure_model) ? 0.0 : Float64(θ)))
        if isnothing(temp)
            check(ccall((:ins_rk_set_temperature, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), h, C_NULL))
        else
            desc = Ref(tempdesc(setup))
            GC.@preserve desc check(ccall((:ins_rk_set_temperature, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), h, Base.unsafe_convert(Ptr{Cvoid}, desc)))
        end
        check(ccall((:ins_rk_step_ext_f64, lib), Cint, (Ptr{Cvoid}, Cdouble, Ptr{Float64}, Ptr{Float64}, Cdouble, Cdouble, Ptr{Cvoid}),
                    h, 1 / setup.Re, pointer(u), isnothing(temp) ? Ptr{Float64}(C_NULL) : pointer(temp), t, Δt, stream()))
        return IncompressibleNavierStokes.create_stepper(method; setup, psolver, u, temp, t = t + method.c[end] * Δt, n = n + 1)
    end
    # Callable Dirichlet data alone (boundary_conditions.jl:351-357): the ghost fills of the stage loop happen at t and t + c[i] Δt, so the closures are
    # evaluated for those nstage + 1 times now and the whole stage loop runs natively (ins_rk_step_bc_f64)
    if !native && isnothing(setup.closure_model) && isnothing(temp) && (isnothing(setup.bodyforce) || setup.issteadybodyforce)
        h = native!(cache, method, setup, psolver)
        check(ccall((:ins_rk_set_bodyforce, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), h,
                    isnothing(setup.bodyforce) ? Ptr{Float64}(C_NULL) : pointer(setup.bodyforce)))
        ns = length(method.b)
        sets = fill(Ptr{Float64}(C_NULL), 18 * (ns + 1))
        keeps = Any[]
        for q = 0:ns
            planes, keep = bc_planes(u, q == 0 ? t : t + method.c[q] * Δt, setup, false)
            push!(keeps, keep)
            isnothing(planes) || (sets[18q+1:18q+18] .= planes)
        end
        GC.@preserve keeps check(ccall((:ins_rk_step_bc_f64, lib), Cint, (Ptr{Cvoid}, Cdouble, Ptr{Float64}, Cdouble, Cdouble, Ptr{Ptr{Float64}}, Ptr{Cvoid}),
                                       h, 1 / setup.Re, pointer(u), t, Δt, sets, stream()))
        return IncompressibleNavierStokes.create_stepper(method; setup, psolver, u, temp, t = t + method.c[end] * Δt, n = n + 1)
    end
    if !native
        isnothing(cache.ref) && (cache.ref = invoke(ode_method_cache, Tuple{ExplicitRungeKuttaMethod,Any}, method, setup))
        return invoke(timestep!, Tuple{ExplicitRungeKuttaMethod,Any,Any}, method, stepper, Δt; θ, cache = cache.ref)
    end
    h = native!(cache, method, setup, psolver)
    check(ccall((:ins_rk_set_bodyforce, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), h,
                isnothing(setup.bodyforce) ? Ptr{Float64}(C_NULL) : pointer(setup.bodyforce)))
    check(ccall((:ins_rk_step_f64, lib), Cint, (Ptr{Cvoid}, Cdouble, Ptr{Float64}, Cdouble, Cdouble, Ptr{Ptr{Float64}}, Ptr{Cvoid}),
                h, 1 / setup.Re, pointer(u), t, Δt, C_NULL, stream()))
    IncompressibleNavierStokes.create_stepper(method; setup, psolver, u, temp, t = t + method.c[end] * Δt, n = n + 1)
end

# The fixed-Δt loop of solve_unsteady (solver.jl:74-83) when no processor looks at intermediate states: one native call; on the fused
# periodic path every step but the last leaves its final correction to the next step's first stage kernel.
function timesteps!(method::ExplicitRungeKuttaMethod, stepper::HipStepper, Δt, nstep; cache::HipRKCache)
    (; setup, psolver, u, temp, t, n) = stepper
    h = native!(cache, method, setup, psolver)
    check(ccall((:ins_rk_steps_f64, lib), Cint, (Ptr{Cvoid}, Cdouble, Ptr{Float64}, Cdouble, Cdouble, Cint, Ptr{Cvoid}),
                h, 1 / setup.Re, pointer(u), t, Δt, nstep, stream()))
    IncompressibleNavierStokes.create_stepper(method; setup, psolver, u, temp, t = t + nstep * method.c[end] * Δt, n = n + nstep)
end

# ---- run-time switches and multi-GPU communicators (include/ins_hip.h) -------------------------------------------------
set_option(name::AbstractString, value::Integer) = check(ccall((:ins_set_option, lib), Cint, (Cstring, Int64), name, value))
"One RCCL communicator per Julia process / GPU: rank 0 calls `comm_unique_id()`, ships the 128 bytes (MPI.jl, Distributed.jl, a file), every rank calls `comm_create`."
function comm_unique_id()
    id = zeros(UInt8, 128)
    check(ccall((:ins_comm_unique_id, lib), Cint, (Ptr{Cvoid},), id))
    id
end
function comm_create(nranks, rank, id::Vector{UInt8})
    h = Ref{Ptr{Cvoid}}()
    check(ccall((:ins_comm_create, lib), Cint, (Cint, Cint, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), nranks, rank, id, h))
    h[]
end
"z ghost planes of the local slab field `u` (slab setup: z sides HaloBC) exchanged with the neighbouring ranks, on the current stream."
halo_exchange!(comm::Ptr{Cvoid}, u::RA, slab_setup; comps = 0b111, down_only = false) =
    (check(ccall((:ins_halo_exchange_f64, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Cint, Cint, Ptr{Cvoid}),
                 comm, handle(slab_setup), pointer(u), comps, down_only, stream())); u)
allreduce!(comm::Ptr{Cvoid}, buf::RA, op::Integer) =
    (check(ccall((:ins_comm_allreduce_f64, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Cint, Ptr{Cvoid}), comm, pointer(buf), length(buf), op, stream())); buf)

end # module
