/*
 * ins_hip.h — C ABI of libinship.so: the MI355X (gfx950) hot path of IncompressibleNavierStokes.jl.
 *
 * The reference has no FFI; its extension seam is Julia multiple dispatch on `setup.backend` and on the
 * array type of `setup.grid.x[1]` (precedent: ext/IncompressibleNavierStokesCUDSSExt.jl:18).
 * Every entry point below replaces the body of one reference function and is what a Julia `ccall`
 * (julia/INSHip.jl) or the Python ctypes host (incompressiblenavierstokes.jl_amd/_lib.py) binds.
 *
 * Conventions
 *   - Plain C types only.  Field pointers are DEVICE pointers owned by the caller, laid out exactly as
 *     the reference's arrays: vector field `u` = Float64 (N1,N2[,N3],D) column-major, x fastest, component
 *     slowest (initializers.jl:5-6); scalar field `p` = Float64 (N1,N2[,N3]) (initializers.jl:2).
 *     N includes the ghost volumes (grid.jl:121).
 *   - Index ranges are 0-based half-open [lo, hi):  Julia `a:b`  ->  lo = a-1, hi = b.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  All work is enqueued on it and
 *     the call returns immediately; only the functions documented as "blocking" synchronise — the same
 *     contract as the reference, which blocks only when a scalar is read (solver.jl:112, pressure.jl:244).
 *   - Return value: 0 on success, negative on failure (INS_ERR_*).  No exceptions cross the boundary.
 *     `ins_last_error()` returns a thread-local human-readable message for the last failure.
 *   - The library owns only the handles it creates.
 */
#ifndef INS_HIP_H
#define INS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define INS_OK 0
#define INS_ERR_INVALID (-1)     /* bad argument (NULL handle, unsupported D, shape mismatch) */
#define INS_ERR_HIP (-2)         /* a HIP runtime call failed */
#define INS_ERR_FFT (-3)         /* a hipFFT/rocFFT call failed */
#define INS_ERR_UNSUPPORTED (-4) /* valid request that this build does not implement */
#define INS_ERR_NOCONV (-5)      /* iterative solver hit maxiter */
#define INS_ERR_COMM (-6)        /* an RCCL call failed */

/* Boundary-condition codes, boundary_conditions.jl:2-36.  INS_BC_HALO marks a side whose ghost plane is
 * filled by the caller (the z-faces of a slab in the multi-GPU decomposition). */
#define INS_BC_PERIODIC 0
#define INS_BC_DIRICHLET 1
#define INS_BC_SYMMETRIC 2
#define INS_BC_PRESSURE 3
#define INS_BC_HALO 4

typedef struct ins_grid ins_grid_t;       /* device copy of `setup.grid` metrics   (grid.jl:100-276)   */
typedef struct ins_poisson ins_poisson_t; /* a `psolver` closure                   (pressure.jl:85-351) */
typedef struct ins_rk ins_rk_t;           /* `ode_method_cache` + stepper state    (time_stepper_caches.jl:34-49) */
typedef struct ins_poisson32 ins_poisson32_t; /* `psolver_spectral` closure for T = Float32 */
typedef struct ins_rk32 ins_rk32_t;           /* `ode_method_cache` for T = Float32 */
typedef struct ins_comm ins_comm_t;       /* one rank of an RCCL communicator (multi-GPU z-slabs; no reference counterpart) */

/* Host-side description of `setup.grid` + `setup.boundary_conditions`; all pointers are HOST pointers
 * to the reference's 1-D metric vectors, copied (and turned into reciprocal tables) by ins_grid_create. */
typedef struct ins_grid_desc {
  int32_t D;                /* 2 or 3                                           grid.jl:105          */
  int32_t N[3];             /* volumes per direction incl. ghosts               grid.jl:121          */
  const double* dx[3];      /* Δ[α],  length N[α]                               grid.jl:177-181      */
  const double* dxu[3];     /* Δu[α], length N[α]                               grid.jl:183-187      */
  const double* A1[3][3];   /* A[α][β][1], length N[β]                          grid.jl:227-248      */
  const double* A2[3][3];   /* A[α][β][2], length N[β]                                               */
  int32_t iu_lo[3][3];      /* Iu[α] range in direction β, 0-based half-open    grid.jl:145-152      */
  int32_t iu_hi[3][3];
  int32_t ip_lo[3];         /* Ip                                               grid.jl:155-159      */
  int32_t ip_hi[3];
  int32_t bc[3][2];         /* INS_BC_* per direction and side                  setup.jl:16          */
  double bc_u[3][2][3];     /* DirichletBC constants u[β][side][α]; 0 for no-slip  boundary_conditions.jl:347-350 */
} ins_grid_desc_t;

/* ---------------------------------------------------------------------------------- library / errors */
int ins_version(void);
/* How often the library reset rocFFT's process-wide caches (containment of a rocFFT plan-cache defect).  A reset invalidates
 * hipFFT plans held OUTSIDE the library (e.g. PyTorch's torch.fft plan cache): compare around solver creation and drop them when it moved. */
int ins_fft_reset_count(void);
const char* ins_last_error(void);
/* Run-time switches (A/B paths and tile shapes; DESIGN.md §5 lists them).  Every switch starts from the environment variable of the same
 * name and can be changed afterwards, so that one process can run e.g. the fused and the reference-order stage loop on the same arrays
 * ("INS_DISABLE_FUSED_RK", "INS_DISABLE_FLUX64", "INS_FLUX64_ZC", ...).  Unknown name: INS_ERR_INVALID.  Switches that shape a handle
 * (solver plans, uniformity classification) are read when the handle is created. */
int ins_set_option(const char* name, int64_t value);
int ins_get_option(const char* name, int64_t* value);
int ins_option_count(void);
const char* ins_option_name(int i);
/* Device the calling thread's handles live on (hipSetDevice).  Blocking. */
int ins_set_device(int device);
/* Blocking: wait for all work on `stream`. */
int ins_sync(void* stream);

/* ---------------------------------------------------------------------------------- Grid  (grid.jl:100) */
int ins_grid_create(const ins_grid_desc_t* desc, ins_grid_t** out);
int ins_grid_destroy(ins_grid_t* grid);
/* 1 when every metric record the fused kernels read is bitwise constant (uniform grid whose spacing and weights are exact);
 * the constant-record kernels and the in-kernel pressure correction require it. */
int ins_grid_is_uniform_exact(const ins_grid_t* grid);

/* ---------------------------------------------------------------------------------- ghost fill */
/* apply_bc_u!(u, t, setup; dudt)   boundary_conditions.jl:159-167 (+276-288, 344-375, 414-428, 472-482).
 * Dirichlet values are the constants in the grid descriptor, or, when `planes` is non-NULL, per-(β,side,α)
 * DEVICE plane buffers planes[(β*2+side)*3+α] (NULL entries fall back to the constant); a plane spans the
 * full padded extent of the other directions in memory order — this is how time-dependent closures
 * `bc.u(α, x..., t)` cross the ABI: the host evaluates them into plane buffers. */
int ins_apply_bc_u_f64(const ins_grid_t* grid, double* u, int dudt, const double* const* planes, void* stream);
/* apply_bc_p!(p, t, setup)         boundary_conditions.jl:197-206 (+306-318, 388, 445-453, 497-502) */
int ins_apply_bc_p_f64(const ins_grid_t* grid, double* p, void* stream);

/* ---------------------------------------------------------------------------------- operators.jl */
/* scalewithvolume!(p, setup)               operators.jl:81-95   (whole padded array) */
int ins_scalewithvolume_f64(const ins_grid_t* grid, double* p, void* stream);
/* divergence!(div, u, setup)               operators.jl:106-125 (writes Ip only) */
int ins_divergence_f64(const ins_grid_t* grid, const double* u, double* div, void* stream);
/* pressuregradient!(G, p, setup)           operators.jl:159-178 (writes Iu[α] only) */
int ins_pressuregradient_f64(const ins_grid_t* grid, const double* p, double* G, void* stream);
/* applypressure!(u, p, setup)              operators.jl:214-233 */
int ins_applypressure_f64(const ins_grid_t* grid, double* u, const double* p, void* stream);
/* laplacian!(L, p, setup)                  operators.jl:297-364 (zeroes L, then writes Ip) */
int ins_laplacian_f64(const ins_grid_t* grid, const double* p, double* L, void* stream);
/* convection!(F, u, setup)                 operators.jl:378-415 (F += ...) */
int ins_convection_f64(const ins_grid_t* grid, const double* u, double* F, void* stream);
/* diffusion!(F, u, setup; use_viscosity)   operators.jl:537-573 (F += visc * ...); pass visc = 1/Re or 1 */
int ins_diffusion_f64(const ins_grid_t* grid, double visc, const double* u, double* F, void* stream);
/* convectiondiffusion!(F, u, setup)        operators.jl:634-690 (F += ...); visc = 1/Re */
int ins_convectiondiffusion_f64(const ins_grid_t* grid, double visc, const double* u, double* F, void* stream);
/* momentum!(F, u, nothing, t, setup)       operators.jl:967-976 with bodyforce = temp = nothing:
 * fill!(F, 0) + convectiondiffusion! fused into one write-only pass over F. */
int ins_momentum_f64(const ins_grid_t* grid, double visc, const double* u, double* F, void* stream);
/* kinetic_energy!(ke, u, setup; interpolate_first)   operators.jl:1516-1545 */
int ins_kinetic_energy_f64(const ins_grid_t* grid, const double* u, double* ke, int interpolate_first, void* stream);
/* total_kinetic_energy(u, setup; interpolate_first)  operators.jl:1551-1556.  Blocking (returns a scalar). */
int ins_total_kinetic_energy_f64(const ins_grid_t* grid, const double* u, int interpolate_first, double* out, void* stream);
/* get_cfl_timestep!(buf, u, setup)         solver.jl:101-125.  Blocking.  `Re` as in setup.Re. */
int ins_cfl_timestep_f64(const ins_grid_t* grid, double Re, const double* u, double* out, void* stream);
/* maximum(abs, divergence(u)[Ip]) — the reference's implied invariant (methods.jl:177-182).  Blocking. */
int ins_max_abs_divergence_f64(const ins_grid_t* grid, const double* u, double* out, void* stream);

/* ---------------------------------------------------------------------------------- pressure.jl */
/* psolver_spectral(setup)                  pressure.jl:289-351: rocFFT (hipFFT API) D2Z/Z2D plans, symbol
 * vectors ahat, work buffers pI / phat.  Requires all-periodic, uniform, even N (utils.jl:1-13). */
int ins_poisson_spectral_create(const ins_grid_t* grid, ins_poisson_t** out);
/* psolver_cg(setup; abstol, reltol, maxiter)   pressure.jl:209-286 with the Jacobi preconditioner of
 * pressure.jl:188-206.  maxiter <= 0 selects prod(Np) as the reference does. */
int ins_poisson_cg_create(const ins_grid_t* grid, double abstol, double reltol, int64_t maxiter, ins_poisson_t** out);
/* CG only.  enable != 0: solve the bordered system [L e; e' 0][p; λ] = [f; 0] that psolver_direct factorises
 * when L is singular (pressure.jl:133-140), i.e. subtract mean(f[Ip]) before iterating and mean(p[Ip]) after.
 * This is what makes a non-solvable right-hand side (e.g. the lid of examples/LidDrivenCavity3D.jl:29, whose
 * normal component is non-zero) behave as it does under the reference's default (direct) solver. */
int ins_poisson_cg_bordered(ins_poisson_t* ps, int enable);
/* z-slab CG: with a communicator set, the solver (created on the rank's slab grid, z sides INS_BC_HALO) sums its two dots and its norm over
 * the ranks (one scalar all-reduce each, on the stream) and exchanges the ghost planes of the search direction before every laplacian!.
 * NULL restores the single-domain solver.  The scalars of the iteration (pressure.jl:244-280) live on the device either way: the host reads the
 * stopping flag once per batch of iterations (option INS_CG_BATCH, default 32) instead of three scalars per iteration. */
int ins_poisson_cg_set_comm(ins_poisson_t* ps, ins_comm_t* comm);
/* psolver_direct(setup)                    pressure.jl:101-154 (replaces the SuiteSparse / cuDSS factorisation of
 * `laplacian_mat`, matrices.jl:484-492).  Fast diagonalisation of the separable operator L = Σα Tα ⊗ (⊗β≠α Dβ):
 * V[α] is the Np[α] x Np[α] column-major matrix of Dα-orthonormal generalised eigenvectors (Tα Vα = Dα Vα Λα),
 * lam[α] its Np[α] eigenvalues; V[2], lam[2] are ignored in 2-D.  Host pointers, copied.  A solve is six fp64
 * GEMMs (rocBLAS) plus the 1/(λx+λy+λz) scaling; when no side is a PressureBC the bordered system of
 * pressure.jl:133-140 is solved (mean(f) removed, null mode dropped, mean(p[Ip]) = 0).  Asynchronous. */
int ins_poisson_fdm_create(const ins_grid_t* grid, const double* const* V, const double* const* lam, ins_poisson_t** out);
int ins_poisson_destroy(ins_poisson_t* ps);
/* Which transform engine a spectral solver was planned on: *engine = 1 when all of its passes are this library's own kernels
 * (power-of-two sides and 192 / 384 = 3 * 2^m, csrc/ins_fft.hip), 2 when rocFFT 2-D plans feed the own fused z pass, 0 when every transform
 * is a rocFFT plan (pressure.jl:316 leaves the choice to the FFT library).  Other solver kinds: *engine = -1. */
int ins_poisson_fft_engine(const ins_poisson_t* ps, int32_t* engine);
/* *partitions > 0: the solver runs FOUR passes per solve — the z direction of pressure.jl:326-341 as periodic tridiagonal systems (the circulant with the
 * eigenvalues âz) solved by the partition method inside the two y passes, `*partitions` z-partitions (csrc/ins_fft.hip k_yz_*) — instead of five with a
 * z-FFT pass; 0: five passes (the default: the four-pass route is opt-in, INS_YZ_FUSED=1 / INS_YZ_PARTITIONS=P read at creation — it moves fewer bytes
 * but measured slower).  Same linear system, results equal to rounding. */
int ins_poisson_yz_partitions(const ins_poisson_t* ps, int32_t* partitions);
/* poisson!(psolver, p) = psolver(p)        pressure.jl:22: solves L p = f in place on the padded array.
 * Spectral: asynchronous.  CG: blocking (the reference reads residuals on the host, pressure.jl:244,275). */
int ins_poisson_solve_f64(ins_poisson_t* ps, double* p, void* stream);
/* Iterations / final residual of the last CG solve (0 / 0 for spectral). */
int ins_poisson_last_info(const ins_poisson_t* ps, int64_t* iterations, double* residual);
/* project!(u, setup; psolver, p)           pressure.jl:69-82.  `p` is left ghost-filled as the reference
 * leaves it.  Spectral solver: fused divergence*Ω -> FFT -> symbol -> iFFT -> gradient-subtract path. */
int ins_project_f64(const ins_grid_t* grid, ins_poisson_t* ps, double* u, double* p, void* stream);

/* ---------------------------------------------------------------------------------- explicit Runge-Kutta */
/* ode_method_cache(method, setup) + create_stepper     time_stepper_caches.jl:34-49, step_explicit_runge_kutta.jl:1-2.
 * `A` is the SHIFTED nstage x nstage tableau of methods.jl:231-236, row-major; `c` the shifted nodes. */
int ins_rk_create(const ins_grid_t* grid, ins_poisson_t* ps, int nstage, const double* A, const double* c, ins_rk_t** out);
int ins_rk_destroy(ins_rk_t* rk);
/* timestep!(method, stepper, Δt; cache)    step_explicit_runge_kutta.jl:4-59 for closure_model = temp =
 * bodyforce = nothing.  `u` is updated in place (ghosts filled on return); `visc` = 1/Re.
 * `planes` as in ins_apply_bc_u_f64 (time-independent Dirichlet data only; otherwise drive the stage loop
 * from the host with the operator-level calls). */
int ins_rk_step_f64(ins_rk_t* rk, double visc, double* u, double t, double dt, const double* const* planes, void* stream);
/* The same with time-dependent Dirichlet data `bc.u(α, x..., t)` (boundary_conditions.jl:351-357): the ghost fills of the stage loop happen at tstart and at
 * tstart + c[i] Δt (step_explicit_runge_kutta.jl:19, 32, 48, 55), so the host evaluates its closures for those nstage + 1 times before the step and the whole
 * stage loop runs natively.  planes_by_time: (nstage + 1) sets of 18 device pointers each, laid out per set as `planes` of ins_apply_bc_u_f64;
 * set 0: t, set q: t + c[q-1] Δt (c as given to ins_rk_create, i.e. shifted: c[nstage-1] = 1).  Blocking at the end (the pointer table is freed). */
int ins_rk_step_bc_f64(ins_rk_t* rk, double visc, double* u, double t, double dt, const double* const* planes_by_time, void* stream);
/* The fixed-Δt loop of solve_unsteady (solver.jl:74-83 without processors): nsteps calls of timestep! with time-independent
 * boundary data.  `u` is valid on entry and on return.  On the fused periodic path the projection's gradient-subtract of every step
 * but the last is applied in registers by the next step's first stage kernel (same arithmetic per cell; the uncorrected intermediate
 * fields are never visible); elsewhere this is exactly nsteps calls of ins_rk_step_f64. */
int ins_rk_steps_f64(ins_rk_t* rk, double visc, double* u, double t, double dt, int nsteps, void* stream);
/* Device pointers into the cache (valid until ins_rk_destroy): the stage pressure `p` and `ku[i]`. */
/* K6 alone: out = base + Σ_q coefs[q]·ks[q] over whole vector fields, summed in index order — the stage-combination
 * broadcasts `u .= ustart; u .+= Δt A[i,j] ku[j]` (step_explicit_runge_kutta.jl:35-38) and LMWray3's state_copyto! /
 * state_axpy! (step_lmwray3.jl:44-54) in one pass.  out may alias base.  Used by host-driven stage loops. */
int ins_combine_f64(const ins_grid_t* grid, const double* base, double* out, int nterms, const double* coefs,
                    const double* const* ks, void* stream);
/* Measurement hook (bench.py `roofline`): while enabled, ins_rk_step_f64 brackets every momentum-RHS
 * kernel launch with hipEvents recorded on the step's stream.  ins_rk_profile_read is blocking: it waits
 * for the recorded events, returns the accumulated kernel milliseconds and launch count, and resets. */
int ins_rk_profile_enable(ins_rk_t* rk, int enable);
int ins_rk_profile_read(ins_rk_t* rk, double* momentum_ms, int64_t* momentum_launches);
/* Steady body force (setup.bodyforce with issteadybodyforce, setup.jl:25-32; operators.jl:873-880 `F .+= bodyforce`): a DEVICE vector
 * field the caller keeps alive, added to every stage force inside the stage kernels' combination (one more term: no extra pass).
 * NULL removes it.  With a force the steps of ins_rk_steps_f64 are not chained. */
int ins_rk_set_bodyforce(ins_rk_t* rk, const double* force);
int ins_rk_pressure(const ins_rk_t* rk, double** p);
/* The cache array ku[i] of ode_method_cache (time_steppers.jl).  It holds the stage force k_i only on the k-basis paths (INS_RK_KEEP_K=1
 * selects them everywhere); the fused stage loops work in the stage-velocity basis, where the periodic loop leaves the arrays untouched
 * and the wall-bounded loop uses them for its uncorrected stage velocities. */
int ins_rk_stage_force(const ins_rk_t* rk, int i, double** ku);

/* ---------------------------------------------------------------------------------- step-adjacent field operators
 * (SURVEY.md §8f rows 2 and 4).  Vector fields: D components of N cells, component slowest; scalar fields: N cells. */
/* vorticity!(ω, u, setup)                  operators.jl:985-1020  (2-D: scalar ω; 3-D: vector ω; written on 0 .. N-2) */
int ins_vorticity_f64(const ins_grid_t* grid, const double* u, double* w, void* stream);
/* interpolate_u_p!(up, u, setup)           operators.jl:1311-1326 (writes Ip) */
int ins_interpolate_u_p_f64(const ins_grid_t* grid, const double* u, double* up, void* stream);
/* interpolate_ω_p!(ωp, ω, setup)           operators.jl:1336-1370 (writes Ip) */
int ins_interpolate_w_p_f64(const ins_grid_t* grid, const double* w, double* wp, void* stream);
/* Dfield!(d, G, p, setup; ϵ)               operators.jl:1385-1422: pressuregradient!(G, p) then d = |G|/2/lap on Ip; G is scratch/out */
int ins_dfield_f64(const ins_grid_t* grid, const double* p, double* G, double* d, double eps, void* stream);
/* Qfield!(Q, u, setup)                     operators.jl:1440-1460 */
int ins_qfield_f64(const ins_grid_t* grid, const double* u, double* Q, void* stream);
/* dissipation_from_strain!(ϵ, u, setup)    operators.jl:836-854; visc = 1/Re */
int ins_dissipation_from_strain_f64(const ins_grid_t* grid, double visc, const double* u, double* eps_out, void* stream);
/* eig2field!(λ, u, setup)                  operators.jl:1472-1492 (3-D only; middle eigenvalue of S² + R² in closed form) */
int ins_eig2field_f64(const ins_grid_t* grid, const double* u, double* lam, void* stream);
/* apply_bc_temp!(temp, t, setup)           boundary_conditions.jl:236-246 (+338-339, 391-405, 466-467, 512-513).
 * bc[2β+side] = INS_BC_* of setup.temperature.boundary_conditions, val[2β+side] = Dirichlet constant; `planes` (nullable, and nullable
 * per entry) = DEVICE buffers of Dirichlet values over the full padded plane in memory order, the way closures cross the ABI. */
int ins_apply_bc_temp_f64(const ins_grid_t* grid, const int32_t* bc, const double* val, const double* const* planes, double* temp, void* stream);
/* convection_diffusion_temp!(c, u, temp, setup)   operators.jl:712-737 (c += ...); a4 = setup.temperature.α4 */
int ins_convection_diffusion_temp_f64(const ins_grid_t* grid, double a4, const double* u, const double* temp, double* c, void* stream);
/* dissipation!(diss, diff, u, setup)       operators.jl:791-814 (diss += ...; diff is scratch); visc = 1/Re, coef = Re·α1/γ */
int ins_dissipation_f64(const ins_grid_t* grid, double visc, double coef, const double* u, double* diff, double* diss, void* stream);
/* gravity!(F, temp, setup)                 operators.jl:914-931 (F[:, gdir] += α2 avg(temp)); gdir 0-based */
int ins_gravity_f64(const ins_grid_t* grid, int gdir, double a2, const double* temp, double* F, void* stream);
/* smagtensor!(σ, u, θ, setup)              operators.jl:1135-1150.  σ is symmetric: D(D+1)/2 scalar fields [xx, yy, (zz), xy, (xz, yz)]
 * (the reference stores D×D SMatrix per cell); ghost-fill each with ins_apply_bc_p_f64 as smagorinsky_closure does (:1296). */
int ins_smagtensor_f64(const ins_grid_t* grid, double theta, const double* u, double* sigma, void* stream);
/* divoftensor!(s, σ, setup)                operators.jl:1203-1236 (writes Iu[α]) */
int ins_divoftensor_f64(const ins_grid_t* grid, const double* sigma, double* s, void* stream);
/* smagorinsky_closure(setup)(u, θ) = divoftensor(apply_bc_p(smagtensor(u, θ)))   operators.jl:1284-1300, the three steps above as one call.
 * All-periodic uniform 3-D boxes: one kernel, the stress stays in registers (sigma may be NULL); other grids: the three kernels with `sigma`
 * (D(D+1)/2 scalar fields) as scratch.  Writes the degrees of freedom of s. */
int ins_smagorinsky_force_f64(const ins_grid_t* grid, double theta, const double* u, double* sigma, double* s, void* stream);
/* 1 if ins_smagorinsky_force_f64 needs the `sigma` scratch on this grid (three-kernel route), 0 if it may be NULL (a host then need not
 * allocate the D(D+1)/2 fields the reference's closure keeps: 6.5 GB at 512³). */
int ins_smagorinsky_force_needs_sigma(const ins_grid_t* grid);
/* tensorbasis!(B, V, u, setup)            tensorbasis.jl:16-72 (writes Ip): nb, nv = 3, 2 (2-D) or 11, 5 (3-D).  B is nb·D·D scalar fields,
 * element (a, b) of basis tensor ib at field index ib·D·D + a + D·b (the SMatrix' column-major order); V is nv scalar fields. */
int ins_tensorbasis_f64(const ins_grid_t* grid, const double* u, double* B, double* V, void* stream);
/* ins_combine_f64 for scalar fields (tempstart + Σ Δt A[i,j] ktemp[j], step_explicit_runge_kutta.jl:39-44) */
int ins_combine_scalar_f64(const ins_grid_t* grid, const double* base, double* out, int nterms, const double* coefs,
                           const double* const* ks, void* stream);

/* ---------------------------------------------------------------------------------- extended stage loop
 * timestep!(method, stepper, Δt) with the temperature equation and / or the Smagorinsky closure carried inside the native loop
 * (step_explicit_runge_kutta.jl:4-59 with `temp` and `closure_model`; boundary data time-independent).  The host mirror used to drive
 * this loop operator by operator; csrc/ins_rk_ext.hip runs it as one call and, on periodic uniform 3-D boxes with the spectral solver,
 * on the fused stage / projection kernels (closure term + gravity enter the stage kernel as one extra force field). */
typedef struct ins_temperature_desc {
  double a2;            /* gravity!: F[:, gdir] += α2 avg(temp)                        operators.jl:914-931 */
  double a4;            /* convection_diffusion_temp!: α4                               operators.jl:712-737 */
  double diss_coef;     /* dissipation!: Re·α1/γ                                        operators.jl:791-814 */
  int32_t gdir;         /* 0-based */
  int32_t dodissipation;
  int32_t bc[6];        /* bc[2β+side] = INS_BC_* of setup.temperature.boundary_conditions */
  double val[6];        /* Dirichlet constants */
} ins_temperature_desc_t;
/* NULL removes the temperature equation from the cache's loop. */
int ins_rk_set_temperature(ins_rk_t* rk, const ins_temperature_desc_t* desc);
/* kind 0: no closure; 1: smagorinsky_closure(setup) with constant θ (operators.jl:1294-1305). */
int ins_rk_set_closure(ins_rk_t* rk, int32_t kind, double theta);
/* One step; `temp` (scalar field, nullable exactly when no temperature equation is set) is advanced with u.  Asynchronous. */
int ins_rk_step_ext_f64(ins_rk_t* rk, double visc, double* u, double* temp, double t, double dt, void* stream);

/* observespectrum(state; setup, npoint, a)   processors.jl:303-332.  The index sets of spectral_stuff (utils.jl:49-108) are built
 * on the host: bin i sums the modes inds[offsets[i] .. offsets[i+1]), each a 0-based column-major position in the K = Np .÷ 2
 * array of retained non-negative wavenumbers.  ins_spectrum_f64 writes ehat[0 .. nbin) (DEVICE): per component one ghost strip,
 * one rocFFT real-to-complex transform and one shell-sum kernel; nothing leaves the device. */
typedef struct ins_spectrum ins_spectrum_t;
int ins_spectrum_create(const ins_grid_t* grid, int nbin, const int64_t* offsets, const int64_t* inds, ins_spectrum_t** out);
/* the same with a weight per index: bin i = Σ weights[q] |û[inds[q]]|² / (2 prod(Np)²) — get_scale_numbers' Σ E(k)/k (operators.jl:1591-1612) */
int ins_spectrum_create_weighted(const ins_grid_t* grid, int nbin, const int64_t* offsets, const int64_t* inds, const double* weights,
                                 ins_spectrum_t** out);
int ins_spectrum_destroy(ins_spectrum_t* spectrum);
int ins_spectrum_f64(ins_spectrum_t* spectrum, const double* u, double* ehat, void* stream);

/* ---------------------------------------------------------------------------------- multi-GPU z-slabs
 * One process per GPU.  Rank r owns nz/P interior z-planes (+1 ghost plane per side); its
 * `ins_grid_t` is created with bc[2] = {INS_BC_HALO, INS_BC_HALO} and the local slice of the z metrics.
 * These entry points do only the rank-LOCAL work; the host performs the exchanges between them (RCCL via
 * torch.distributed in this repo, `ncclSend/Recv` + all-to-all from Julia): u halo planes before the stencil,
 * the last w-plane before the divergence, two transposes inside the Poisson solve, the first p-plane before
 * the gradient.  The reference has no multi-device code; these replace the same reference lines as the
 * single-GPU entry points they mirror. */
typedef struct ins_slab_fft ins_slab_fft_t;
/* Stage kernel (K1 + K6): k_out = momentum!(u_in) (operators.jl:967-976) when k_out != NULL, and
 * ustar[interior] = ustart + Σ_q coefs[q]·ks[q] + coef_self·momentum(u_in)  (step_explicit_runge_kutta.jl:35-38);
 * ustart == NULL means ustart = u_in.  Needs valid ghosts in u_in; 3-D all-DOF grids (periodic box or slab). */
int ins_stage_momentum_f64(const ins_grid_t* grid, double visc, const double* u_in, double* k_out, const double* ustart,
                           double* ustar, int nterms, const double* coefs, const double* const* ks, double coef_self, void* stream);
/* Same stage kernel for stages >= 2 of a slab with the PREVIOUS stage's projection applied in registers
 * (applypressure! + periodic apply_bc_u!, operators.jl:225-233, boundary_conditions.jl:276-288): `ustar_prev` = that stage's
 * uncorrected velocity with valid z-ghost planes; `p_ext` = its pressure as [1 plane below | nz/P local planes | 2 planes above],
 * unpadded in x and y.  Removes ins_slab_applypressure_f64 from every stage but the last.  Exactly-uniform slabs only. */
int ins_stage_momentum_corr_f64(const ins_grid_t* grid, double visc, const double* ustar_prev, const double* p_ext, double* k_out,
                                const double* ustart, double* ustar, int nterms, const double* coefs, const double* const* ks,
                                double coef_self, void* stream);
/* The same stage in two launches so that a halo exchange can run beside the first: part = 1 runs the planes that read no ghost
 * plane of ustar_prev / p_ext, part = 2 the two boundary ranges (part = 0: everything).  The stage velocity is
 *   ustar = (1 + c0m1) ustart + Σ coefs[q] ks[q] + self_in ustar_prev + coef_self f ;
 * c0m1 = self_in = 0 with ks = stage forces is step_explicit_runge_kutta.jl:35-38; with ks = earlier UNCORRECTED stage velocities and
 * c0m1 = -(Σ coefs + self_in) it is the same combination in the stage-velocity basis (no stage force is stored or read,
 * csrc/ins_rk.hip).  ustart = NULL (chained first stage: ustar_prev / p_ext are the PREVIOUS STEP's uncorrected result and pressure): the
 * corrected input is the start field, nterms = 0, and it is also stored to ustart_out when given.  self_in != 0, ustart = NULL and
 * ustart_out need the 64-wide stage kernel (x >= 66 cells, exactly uniform: ins_grid_is_uniform_exact). */
int ins_stage_momentum_corr_part_f64(const ins_grid_t* grid, double visc, const double* ustar_prev, const double* p_ext, double* k_out,
                                     const double* ustart, double* ustar, int nterms, const double* coefs, const double* const* ks,
                                     double coef_self, double c0m1, double self_in, double* ustart_out, int part, void* stream);
/* pI[nx,ny,nzl] = Ω·divergence(u) on the slab interior (operators.jl:117-125, 81-95; pressure.jl:320): x, y via
 * periodic wrap, z via the ghost plane. */
int ins_slab_divergence_f64(const ins_grid_t* grid, const double* u, double* pI, void* stream);
/* u -= ∇p on the slab interior from the unpadded pI and `p_top` = the next rank's first pI plane
 * (operators.jl:225-233); writes the periodic x / y ghost images of u (boundary_conditions.jl:276-288). */
int ins_slab_applypressure_f64(const ins_grid_t* grid, double* u, const double* pI, const double* p_top, void* stream);
/* Distributed psolver_spectral (pressure.jl:289-351).  np = GLOBAL interior sizes, h = uniform spacings. */
int ins_slab_fft_create(const int32_t np[3], const double h[3], int rank, int nranks, ins_slab_fft_t** out);
int ins_slab_fft_destroy(ins_slab_fft_t* fft);
/* Element counts of the caller-provided buffers: pI (doubles) and each complex work/send/recv buffer (complex). */
int ins_slab_fft_sizes(const ins_slab_fft_t* fft, int64_t* real_elems, int64_t* complex_elems);
/* 2-D R2C per local plane, packed for the transpose: sendbuf = [dest q][kz_local][ky_local][kx] */
int ins_slab_fft_forward_xy(ins_slab_fft_t* fft, double* pI, double* work, double* sendbuf, void* stream);
/* after the all-to-all (buf = [kz][ky_local][kx]): z-FFT, symbol division (pressure.jl:326-341), inverse z-FFT */
int ins_slab_fft_solve_z(ins_slab_fft_t* fft, double* buf, void* stream);
/* after the all-to-all back (recvbuf = [src q][kz_local][ky_local][kx]): unpack, 2-D C2R -> pI */
int ins_slab_fft_inverse_xy(ins_slab_fft_t* fft, double* recvbuf, double* work, double* pI, void* stream);

/* kx-chunked variants of the three calls above: chunk [kx0, kx0+kxc) of the half spectrum is packed contiguously
 * ([dest q][kz_local][ky_local][kxc]), so the transposes of different chunks can overlap each other (full-duplex links,
 * two communicators) and the z solve.  Chunks need a power-of-two nz (ins_slab_fft_can_chunk). */
int ins_slab_fft_can_chunk(const ins_slab_fft_t* fft);
int ins_slab_fft_xy_forward_only(ins_slab_fft_t* fft, double* pI, double* work, void* stream);
int ins_slab_fft_pack_chunk(ins_slab_fft_t* fft, double* work, double* sendbuf, int kx0, int kxc, void* stream);
int ins_slab_fft_solve_z_chunk(ins_slab_fft_t* fft, double* buf, int kx0, int kxc, void* stream);
int ins_slab_fft_unpack_chunk(ins_slab_fft_t* fft, double* recvbuf, double* work, int kx0, int kxc, void* stream);
int ins_slab_fft_xy_inverse_only(ins_slab_fft_t* fft, double* work, double* pI, void* stream);
/* Power-of-two boxes (ins_slab_fft_is_own): the own y pass writes / reads the packed exchange buffer directly, with uniform
 * chunk width cw (chunk c = kx in [c*cw, min((c+1)*cw, nx/2+1))), so no pack / unpack pass exists; with from_u != 0 the
 * right-hand side Ω·div(u) is formed inside the x pass from the slab's velocity (`src` = u on `grid`, z via ghost planes). */
int ins_slab_fft_is_own(const ins_slab_fft_t* fft);
int ins_slab_fft_forward_packed(ins_slab_fft_t* fft, const ins_grid_t* grid, const double* src, int from_u, double* work, double* sendbuf,
                                int cw, void* stream);
int ins_slab_fft_inverse_packed(ins_slab_fft_t* fft, double* recvbuf, double* work, double* pI, int cw, void* stream);

/* Transpose-free distributed solve (csrc/ins_ztri.hip): after the local (x, y) transforms the z direction of
 * pressure.jl:326-341 is one periodic tridiagonal system per (kx, ky) line — the circulant matrix the reference's z-FFT
 * diagonalises — solved across ranks by the partition method; ranks exchange two complex numbers per line instead of the
 * two all-to-all transposes.  Sequence on every rank:  ztri_forward -> all-gather of `edge` (ztri_edge_elems doubles per
 * rank, rank-major) -> ztri_finish.  `from_u` != 0: src is the slab's u* and Ω·div(u*) is formed inside the x pass
 * (power-of-two boxes); else src is pI (divergence! + scalewithvolume! already applied).  Works with one rank as well. */
int ins_slab_ztri_edge_elems(const ins_slab_fft_t* S, int64_t* doubles);
int ins_slab_ztri_forward(ins_slab_fft_t* S, const ins_grid_t* grid, const double* src, int from_u, double* work, double* edge, void* stream);
int ins_slab_ztri_finish(ins_slab_fft_t* S, double* work, const double* edges_all, double* pI, void* stream);
/* The same solve with the kxn·ny lines cut into `nchunks` ranges, so that the gather of one range travels while the next range is swept:
 * ztri_transform -> for each c: ztri_sweep_forward(c) + gather of its edge buffer (ztri_chunk gives range and size) -> for each c:
 * ztri_sweep_backward(c) -> ztri_inverse.  forward / finish above are the one-range forms. */
int ins_slab_ztri_chunk(const ins_slab_fft_t* S, int c, int nchunks, int64_t* line_lo, int64_t* line_cnt, int64_t* edge_doubles);
int ins_slab_ztri_transform(ins_slab_fft_t* S, const ins_grid_t* grid, const double* src, int from_u, double* work, void* stream);
int ins_slab_ztri_sweep_forward(ins_slab_fft_t* S, double* work, double* edge, int c, int nchunks, void* stream);
int ins_slab_ztri_sweep_backward(ins_slab_fft_t* S, double* work, const double* edges_all, int c, int nchunks, void* stream);
int ins_slab_ztri_inverse(ins_slab_fft_t* S, double* work, double* pI, void* stream);
/* x pass of ins_slab_ztri_forward (from_u = 1) for local planes [kz0, kz0 + nkz) only; follow with ins_slab_ztri_forward(from_u = 2).
 * Planes >= 1 read no ghost plane of u, so they can run while the w plane below the slab is still in flight. */
int ins_slab_xfwd_planes(ins_slab_fft_t* S, const ins_grid_t* grid, const double* u, double* work, int kz0, int nkz, void* stream);

/* ---------------------------------------------------------------------------------- multi-GPU communication (RCCL over xGMI)
 * The reference is single-device; these entry points are the communication half of this library's z-slab decomposition
 * of its stage loop, so that a non-Python host can drive several GPUs through the boundary.  All exchanges are enqueued on
 * `stream` (stream-ordered with the kernels that produce / consume the planes), grouped (ncclGroupStart/End), fp64, non-blocking for the host.
 * librccl is loaded when the first communicator is created. */
#define INS_COMM_ID_BYTES 128
/* One process per GPU: rank 0 obtains an id, the host ships the bytes to every rank, each rank creates its communicator with its device current. */
int ins_comm_unique_id(void* id128);
int ins_comm_create(int nranks, int rank, const void* id128, ins_comm_t** out);
/* One process driving ngpu devices (devices == NULL: 0..ngpu-1): out[i] is the communicator of devices[i] (ncclCommInitAll). */
int ins_comm_create_local(int ngpu, const int* devices, ins_comm_t** out);
/* ngpu > 1 driven by ONE host thread: bracket every round of exchange calls over the communicators (comm 0, comm 1, ...) with group_begin / group_end
 * (ncclGroupStart / ncclGroupEnd; RCCL's single-thread multi-device rule — the groups the entry points open themselves nest inside); with one host thread
 * per device, or one process per GPU, they are not needed.  The ngpu > 1 local mode has never run on hardware (no multi-GPU box): unpinned. */
int ins_comm_group_begin(void);
int ins_comm_group_end(void);
int ins_comm_destroy(ins_comm_t* comm);
int ins_comm_rank(const ins_comm_t* comm, int* rank, int* nranks);
/* Generic grouped exchange; messages between one pair of ranks match in posting order. */
int ins_comm_sendrecv_f64(ins_comm_t* comm, int nsend, const double* const* sendbufs, const int64_t* sendcounts, const int32_t* dsts, int nrecv,
                          double* const* recvbufs, const int64_t* recvcounts, const int32_t* srcs, void* stream);
/* z ghost planes of a padded local vector field on a slab grid (z sides INS_BC_HALO): the periodic z part of apply_bc_u!
 * (boundary_conditions.jl:276-288) across ranks.  comp_mask bit c selects component c; down_only: only plane nzl -> next rank's plane 0
 * (all `divergence!` needs, operators.jl:122). */
int ins_halo_exchange_f64(ins_comm_t* comm, const ins_grid_t* slab_grid, double* u, int comp_mask, int down_only, void* stream);
/* The same for a padded local SCALAR field (periodic z part of apply_bc_p!, boundary_conditions.jl:306-318). */
int ins_halo_exchange_scalar_f64(ins_comm_t* comm, const ins_grid_t* slab_grid, double* p, void* stream);
/* Ghost planes of the extended pressure buffer [1 below | nzl local | 2 above] read by the correcting stage kernel (ins_stage_momentum_corr_f64). */
int ins_halo_exchange_p_f64(ins_comm_t* comm, double* p_ext, int64_t plane_elems, int nzl, void* stream);
/* Interface values of the distributed tridiagonal z solve (ins_slab_ztri_forward -> this -> ins_slab_ztri_finish): direct != 0 sends to every
 * peer over its own xGMI link in one group, 0 uses ncclAllGather. */
int ins_ztri_allgather_f64(ins_comm_t* comm, const double* edge, double* edges_all, int64_t count, int direct, void* stream);
/* In-place reductions of device scalars across ranks (CFL minimum: solver.jl:101-125, energies, norms): op 0 sum, 1 max, 2 min. */
int ins_comm_allreduce_f64(ins_comm_t* comm, double* buf, int64_t count, int op, void* stream);
/* Transposes around the z-FFT (the alternative Poisson route, ins_slab_fft_*): block r of `send` -> block `rank` of rank r's `recv`. */
int ins_comm_alltoall_f64(ins_comm_t* comm, const double* send, double* recv, int64_t count, void* stream);

/* ---------------------------------------------------------------------------------- the `_f32` family (T = Float32)
 * The reference is generic in the element type and recommends single precision on GPUs (docs/src/manual/precision.md:3-16;
 * examples/DecayingTurbulence3D.jl:16 runs T = Float32).  Fields are the reference layout with Float32 elements; the grid handle is the
 * fp64 one (`ins_grid_create`).
 *   all-periodic uniform boxes (2-D / 3-D; what that example runs): the spectral pressure solver (ins_poisson_spectral_create_f32), K1 on wide 3-D boxes
 *     as the 64-outputs-per-wavefront stage kernel instantiated for float (csrc/ins_flux64.hip), own float2 FFT passes or hipFFT R2C / C2R plans;
 *   every other grid of the fp64 family (Dirichlet / Symmetric / Pressure sides with constant boundary data, stretched spacings): float operator kernels
 *     (csrc/ins_f32g.hip) and a Float32 solver handle wrapped around an fp64 solver of any kind (ins_poisson_wrap_f32).
 * Slab (halo) grids and time-dependent boundary data return INS_ERR_UNSUPPORTED — there is no silent fp64 fallback. */
int ins_apply_bc_u_f32(const ins_grid_t* grid, float* u, void* stream);                                  /* boundary_conditions.jl:276-288 */
int ins_apply_bc_p_f32(const ins_grid_t* grid, float* p, void* stream);                                  /* boundary_conditions.jl:306-318 */
int ins_momentum_f32(const ins_grid_t* grid, float visc, const float* u, float* F, void* stream);        /* operators.jl:967-976 */
int ins_poisson_spectral_create_f32(const ins_grid_t* grid, ins_poisson32_t** out);                      /* pressure.jl:289-351 */
/* psolver_direct / psolver_cg / psolver_spectral with T = Float32 on any grid (pressure.jl:85-154, 209-351): a Float32 solver handle around the fp64
 * solver `ps64` of the same grid.  project! forms Ω·div(u) from the float field in double, `ps64` solves, the pressure is rounded to float once
 * (at least as accurate as the reference's Float32 factorisation).  `ps64` stays the caller's and must outlive the returned handle. */
int ins_poisson_wrap_f32(const ins_grid_t* grid, ins_poisson_t* ps64, ins_poisson32_t** out);
int ins_poisson_destroy_f32(ins_poisson32_t* ps);
int ins_poisson_solve_f32(ins_poisson32_t* ps, float* p, void* stream);                                  /* pressure.jl:318-350 */
int ins_project_f32(const ins_grid_t* grid, ins_poisson32_t* ps, float* u, float* p, void* stream);      /* pressure.jl:69-82 */
int ins_rk_create_f32(const ins_grid_t* grid, ins_poisson32_t* ps, int nstage, const double* A, const double* c, ins_rk32_t** out);
int ins_rk_destroy_f32(ins_rk32_t* rk);
int ins_rk_step_f32(ins_rk32_t* rk, float visc, float* u, float dt, void* stream);                       /* step_explicit_runge_kutta.jl:4-59 */
/* nsteps steps of size dt as one call (the fixed-Δt loop of solve_unsteady, solver.jl:74-83, T = Float32), chained like ins_rk_steps_f64 where the
 * in-register correction runs on float2 spectra; u is valid before and after the call. */
int ins_rk_steps_f32(ins_rk32_t* rk, float visc, float* u, float dt, int nsteps, void* stream);
/* maximum(abs, divergence(u)) over Ip; blocking.  operators.jl:106-125 */
int ins_max_abs_divergence_f32(const ins_grid_t* grid, ins_poisson32_t* ps, const float* u, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* INS_HIP_H */
