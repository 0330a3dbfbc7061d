# INSHip.jl — glue that makes libinship.so a backend of IncompressibleNavierStokes.jl.
#
# UNTESTED: Julia is not installed in the build container (SURVEY.md §8b/§8c), so this file has never been
# executed.  It shows the binding a maintainer would add (a weak-dependency package extension in the style of
# ext/IncompressibleNavierStokesCUDSSExt.jl): AMDGPU.jl is used for device memory ONLY (ROCArray allocation and
# raw pointers); every kernel lives behind the C ABI of include/ins_hip.h.  No KernelAbstractions kernels run.
module INSHip

using IncompressibleNavierStokes
using AMDGPU
using AMDGPU: ROCArray, ROCBackend, HIP
import IncompressibleNavierStokes:
    apply_bc_u!, apply_bc_p!, divergence!, scalewithvolume!, pressuregradient!, applypressure!, laplacian!,
    convection!, diffusion!, convectiondiffusion!, momentum!, project!, poisson!, psolver_spectral, psolver_cg,
    timestep!, ode_method_cache, ExplicitRungeKuttaMethod, PeriodicBC, DirichletBC, SymmetricBC, PressureBC,
    vorticity!, interpolate_u_p!, interpolate_ω_p!, Qfield!, Dfield!, eig2field!, dissipation_from_strain!,
    convection_diffusion_temp!, dissipation!, gravity!, apply_bc_temp!, smagorinsky_closure

const lib = get(ENV, "INSHIP_LIB", "libinship.so")

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
bccode(::PeriodicBC) = Int32(0); bccode(::DirichletBC) = Int32(1); bccode(::SymmetricBC) = Int32(2); bccode(::PressureBC) = Int32(3)

"Device handle for `setup.grid`, built once per setup from the HOST copies of the 1-D metric vectors (grid.jl:177-248)."
function grid_handle(setup)
    g = setup.grid
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
        ntuple(k -> (β = (k - 1) ÷ 2 + 1; s = (k - 1) % 2 + 1; β <= D ? bccode(setup.boundary_conditions[β][s]) : Int32(0)), 6),
        ntuple(_ -> 0.0, 18),   # constant Dirichlet data: fill from bc.u::Tuple as [β][side][α]
    )
    h = Ref{Ptr{Cvoid}}()
    GC.@preserve Δ Δu A check(ccall((:ins_grid_create, lib), Cint, (Ref{GridDesc}, Ref{Ptr{Cvoid}}), desc, h))
    h[]
end

# One cached handle per setup (setup is an immutable NamedTuple: key on objectid of its grid).
const HANDLES = Dict{UInt,Ptr{Cvoid}}()
handle(setup) = get!(() -> grid_handle(setup), HANDLES, objectid(setup.grid))
iship(setup) = setup.backend isa ROCBackend && setup.grid.x[1] isa ROCArray{Float64}

# ---- operators (operators.jl) -------------------------------------------------------------------------------
function divergence!(div::ROCArray{Float64}, u::ROCArray{Float64}, setup)
    check(ccall((:ins_divergence_f64, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
                handle(setup), pointer(u), pointer(div), stream()))
    div
end
function momentum!(F::ROCArray{Float64}, u::ROCArray{Float64}, temp::Nothing, t, setup)
    isnothing(setup.bodyforce) || error("bodyforce is outside the HIP hot path")
    check(ccall((:ins_momentum_f64, lib), Cint, (Ptr{Cvoid}, Cdouble, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
                handle(setup), 1 / setup.Re, pointer(u), pointer(F), stream()))
    F
end
function applypressure!(u::ROCArray{Float64}, p::ROCArray{Float64}, setup)
    check(ccall((:ins_applypressure_f64, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
                handle(setup), pointer(u), pointer(p), stream()))
    u
end
function apply_bc_u!(u::ROCArray{Float64}, t, setup; dudt = false, kwargs...)
    # closures bc.u(α, x..., t) are evaluated into plane buffers on the host side (see INTEGRATION.md); constants go through the descriptor
    check(ccall((:ins_apply_bc_u_f64, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cint, Ptr{Ptr{Float64}}, Ptr{Cvoid}),
                handle(setup), pointer(u), dudt, C_NULL, stream()))
    u
end
# scalewithvolume!, pressuregradient!, laplacian!, convection!, diffusion!, convectiondiffusion!, apply_bc_p!
# follow the same three-line pattern with ins_<name>_f64.

# ---- pressure solvers (pressure.jl) --------------------------------------------------------------------------
struct HipPSolver
    h::Ptr{Cvoid}
    setup::Any
end
function psolver_spectral(setup, ::Val{:hip})
    h = Ref{Ptr{Cvoid}}()
    check(ccall((:ins_poisson_spectral_create, lib), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), handle(setup), h))
    HipPSolver(h[], setup)
end
# psolver_direct (pressure.jl:101-154): fast diagonalisation of the separable Laplacian — the generalised eigenpairs of the 1-D factors
# (T_α v = λ D_α v; T_α = the 1-D factor of laplacian! with its boundary branches, operators.jl:328-350, D_α = Diagonal(Δ[α][Ip[α]]))
# are computed here once with LinearAlgebra.eigen and handed to the library, which solves with six fp64 GEMMs (rocBLAS).
# `laplacian_1d(setup, α)` is the 20-line assembly of T_α; see incompressiblenavierstokes.jl_amd/pressure.py:_laplacian_1d.
function psolver_direct(setup, ::Val{:hip})
    D = setup.grid.dimension()
    V, λ = Matrix{Float64}[], Vector{Float64}[]
    for α = 1:D
        d = setup.grid.Δ[α][setup.grid.Ip.indices[α]] |> Array
        T = laplacian_1d(setup, α)
        E = eigen(Symmetric(Diagonal(d .^ -0.5) * T * Diagonal(d .^ -0.5)))
        push!(V, Diagonal(d .^ -0.5) * E.vectors)   # column-major, V'DV = I
        push!(λ, E.values)
    end
    h = Ref{Ptr{Cvoid}}()
    GC.@preserve V λ check(ccall((:ins_poisson_fdm_create, lib), Cint, (Ptr{Cvoid}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ref{Ptr{Cvoid}}),
                                 handle(setup), pointer.(V), pointer.(λ), h))
    HipPSolver(h[], setup)
end
(s::HipPSolver)(p::ROCArray{Float64}) =
    (check(ccall((:ins_poisson_solve_f64, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Cvoid}), s.h, pointer(p), stream())); p)
function project!(u::ROCArray{Float64}, setup; psolver::HipPSolver, p::ROCArray{Float64})
    check(ccall((:ins_project_f64, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
                handle(setup), psolver.h, pointer(u), pointer(p), stream()))
    u
end

# ---- step-adjacent operators (SURVEY §8f rows 2 and 4): with these methods the reference's own host-driven `timestep!`
# (closure model / temperature / body force present) runs entirely on libinship's kernels --------------------------------
const RA = ROCArray{Float64}
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
# momentum!(F, u, temp, t, setup): the fused fill + convection-diffusion pass, then body force and gravity (operators.jl:967-976)
function momentum!(F::RA, u::RA, temp, t, setup)
    check(ccall((:ins_momentum_f64, lib), Cint, (Ptr{Cvoid}, Cdouble, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}), handle(setup), 1 / setup.Re, pointer(u), pointer(F), stream()))
    isnothing(setup.bodyforce) || IncompressibleNavierStokes.applybodyforce!(F, u, t, setup)   # broadcast add of a ROCArray
    isnothing(temp) || gravity!(F, temp, setup)
    F
end
# smagorinsky_closure(setup): σ as D(D+1)/2 scalar fields [xx, yy, (zz), xy, (xz, yz)]
function smagorinsky_closure(setup, ::Val{:hip})
    D = setup.grid.dimension()
    ns = D * (D + 1) ÷ 2
    σ = similar(setup.grid.x[1], Float64, (setup.grid.N..., ns)); fill!(σ, 0)
    s = IncompressibleNavierStokes.vectorfield(setup)
    ncell = prod(setup.grid.N)
    function closure(u, θ)
        check(ccall((:ins_smagtensor_f64, lib), Cint, (Ptr{Cvoid}, Cdouble, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}), handle(setup), θ, pointer(u), pointer(σ), stream()))
        for q = 0:ns-1   # apply_bc_p!(σ, 0, setup) component by component (operators.jl:1296)
            check(ccall((:ins_apply_bc_p_f64, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Cvoid}), handle(setup), pointer(σ) + 8 * ncell * q, stream()))
        end
        check(ccall((:ins_divoftensor_f64, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}), handle(setup), pointer(σ), pointer(s), stream()))
        s
    end
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
struct HipRKCache
    h::Ptr{Cvoid}
end
function ode_method_cache(method::ExplicitRungeKuttaMethod, setup, psolver::HipPSolver)
    ns = length(method.b)
    A = collect(transpose(method.A))        # row-major for C; method.A is already the SHIFTED tableau (methods.jl:231-236)
    h = Ref{Ptr{Cvoid}}()
    check(ccall((:ins_rk_create, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Ref{Ptr{Cvoid}}),
                handle(setup), psolver.h, ns, A, method.c, h))
    HipRKCache(h[])
end
function timestep!(method::ExplicitRungeKuttaMethod, stepper, Δt; θ = nothing, cache::HipRKCache)
    (; setup, psolver, u, temp, t, n) = stepper
    # The fused native step is valid only without closure model / temperature / unsteady body force (SURVEY.md §8b caveat);
    # otherwise fall back to the operator-level methods above so user callbacks can run between kernels.
    # A steady body force rides inside the native stage kernels (ins_rk_set_bodyforce: one more term of the stage combination).
    (isnothing(setup.closure_model) && isnothing(temp) && (isnothing(setup.bodyforce) || setup.issteadybodyforce)) ||
        return invoke(timestep!, Tuple{ExplicitRungeKuttaMethod,Any,Any}, method, stepper, Δt; θ, cache)
    check(ccall((:ins_rk_set_bodyforce, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), cache.h,
                isnothing(setup.bodyforce) ? Ptr{Float64}(C_NULL) : pointer(setup.bodyforce)))
    check(ccall((:ins_rk_step_f64, lib), Cint, (Ptr{Cvoid}, Cdouble, Ptr{Float64}, Cdouble, Cdouble, Ptr{Ptr{Float64}}, Ptr{Cvoid}),
                cache.h, 1 / setup.Re, pointer(u), t, Δt, C_NULL, stream()))
    IncompressibleNavierStokes.create_stepper(method; setup, psolver, u, temp, t = t + method.c[end] * Δt, n = n + 1)
end

# The fixed-Δt loop of solve_unsteady (solver.jl:74-83) when no processor looks at intermediate states: one native call; on the fused
# periodic path every step but the last leaves its final correction to the next step's first stage kernel.
function timesteps!(method::ExplicitRungeKuttaMethod, stepper, Δt, nstep; cache::HipRKCache)
    (; setup, psolver, u, temp, t, n) = stepper
    check(ccall((:ins_rk_steps_f64, lib), Cint, (Ptr{Cvoid}, Cdouble, Ptr{Float64}, Cdouble, Cdouble, Cint, Ptr{Cvoid}),
                cache.h, 1 / setup.Re, pointer(u), t, Δt, nstep, stream()))
    IncompressibleNavierStokes.create_stepper(method; setup, psolver, u, temp, t = t + nstep * method.c[end] * Δt, n = n + nstep)
end

end # module
