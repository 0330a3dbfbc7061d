// Native stage loop that carries the temperature equation and the Smagorinsky closure:
// timestep!(method::ExplicitRungeKuttaMethod, stepper, Δt) with `temp` and / or `closure_model` set
// (step_explicit_runge_kutta.jl:4-59), boundary data time-independent.  Per stage i
//   apply_bc_u!, apply_bc_temp!                                                       :19-20
//   ku[i] = momentum(u, temp) = convection-diffusion + body force + gravity(temp)      :21   (operators.jl:967-976)
//   ktemp[i] = convection_diffusion_temp(u, temp) + dissipation(u)                     :23-27
//   ku[i] += closure(u, θ) = divoftensor(apply_bc_p(smagtensor(u, θ)))                 :31-34 (operators.jl:1294-1305)
//   u = ustart + Σ_j Δt A[i,j] ku[j],  temp = tempstart + Σ_j Δt A[i,j] ktemp[j]        :35-44
//   apply_bc_u!, project!                                                              :48-49
// and apply_bc_u!, apply_bc_temp! at the end (:55-56).  The host mirror drove exactly this from Python with one C-ABI call per
// operator; here the loop is one call, and on periodic uniform 3-D boxes with the spectral solver it runs on the fused kernels of the
// plain path: the closure term and gravity go into ONE vector field E_i that the stage kernel adds to its force in registers
// (RkEpi::extra: k_i = F_i + E_i is what is combined and stored — no axpy pass, no combine pass, no snapshot copy), the divergence is
// formed inside the solver's x pass and the gradient-subtract fills the ghost volumes (ins_k_project_periodic_fused).  The in-register
// pressure correction of the plain path does not apply: closure and temperature kernels need the corrected stage velocity in memory.
#include <cstring>
#include <vector>

#include "ins_internal.h"

int ins_k_momentum_generic(const ins_grid* G, double visc, const double* u, double* F, hipStream_t s);
int ins_k_momentum_fast3d_opts(const ins_grid* G, double visc, const double* u, double* F, bool zero_shell, hipStream_t s);
int ins_k_momentum_rk_fused(const ins_grid* G, double visc, const double* u_in, double* k_out, const RkEpi& epi, hipStream_t s);
int ins_k_project_periodic_fused(const ins_grid* G, ins_poisson* ps, double* u, double* p, bool keep_p, hipStream_t s, double* uout = nullptr);
int ins_k_project(const ins_grid* G, ins_poisson* ps, double* u, double* p, hipStream_t s);
bool ins_fast3d_supported(const ins_grid* G);
bool ins_flux64_supported(const ins_grid* G);
int ins_k_temp_stage(const ins_grid* G, double a4, double coef, const double* u, const double* temp, const double* w, const double* tempstart, int n,
                     const double* coefs, const double* const* ks, double c_self, double* ktemp_out, double* temp_out, hipStream_t s, const double* pI,
                     const double* diff = nullptr);
int ins_k_diffusion_overwrite(const ins_grid* G, double visc, const double* u, double* F, hipStream_t s);
bool ins_flux64m_supported(const ins_grid* G);
bool ins_smagforce_supported(const ins_grid* G);
int ins_k_smagforce(const ins_grid* G, double theta, const double* u, const double* pI, double* sout, hipStream_t s);
int ins_k_diffusion_flux3d(const ins_grid* G, double visc, const double* u, double* F, bool zero_shell, hipStream_t s);
int ins_k_momentum_rk_fused_corr(const ins_grid* G, double visc, const double* ustar_prev, const double* pI, double* k_out, const RkEpi& epi, hipStream_t s);
int ins_k_project_periodic_solve_only(const ins_grid* G, ins_poisson* ps, const double* u, hipStream_t s);
int ins_k_smagtensor_corr(const ins_grid* G, double theta, const double* ustar, const double* pI, double* sig, hipStream_t s);

struct ins_rk_ext {
  int closure = 0;  // 0 none, 1 Smagorinsky
  double theta = 0.0;
  bool temp_on = false;
  ins_temperature_desc_t td;
  double* sigma = nullptr;      // D(D+1)/2 scalar fields
  double* E = nullptr;          // closure term + gravity (vector field)
  double* diff = nullptr;       // scratch of dissipation! (vector field)
  double* tempstart = nullptr;  // scalar
  double* w = nullptr;          // u · diffusion(u) (vector field, written by the stage kernel; fused path)
  double* tb[2] = {nullptr, nullptr};  // stage temperatures (fused path)
  std::vector<double*> ktemp;   // nstage scalars
};

void ins_rk_ext_free(ins_rk_ext* e) {
  if (!e) return;
  for (double* p : {e->sigma, e->E, e->diff, e->tempstart, e->w, e->tb[0], e->tb[1]})
    if (p) (void)hipFree(p);
  for (double* p : e->ktemp)
    if (p) (void)hipFree(p);
  delete e;
}

static ins_rk_ext* ext_of(ins_rk* rk) {
  if (!rk->ext) rk->ext = new ins_rk_ext();
  return rk->ext;
}

extern "C" int ins_rk_set_closure(ins_rk_t* rk, int32_t kind, double theta) {
  INS_REQUIRE(rk, "null argument");
  INS_REQUIRE(kind == 0 || kind == 1, "closure kind: 0 (none) or 1 (Smagorinsky)");
  ins_rk_ext* e = ext_of(rk);
  e->closure = kind;
  e->theta = theta;
  return INS_OK;
}

extern "C" int ins_rk_set_temperature(ins_rk_t* rk, const ins_temperature_desc_t* desc) {
  INS_REQUIRE(rk, "null argument");
  ins_rk_ext* e = ext_of(rk);
  e->temp_on = desc != nullptr;
  if (desc) {
    INS_REQUIRE(desc->gdir >= 0 && desc->gdir < rk->grid->g.D, "gdir out of range");
    e->td = *desc;
  }
  return INS_OK;
}

static long long g_fused_steps = 0, g_tiled_steps = 0;
extern "C" long long ins_dbg_ext_tiled_steps(void) { return g_tiled_steps; }
// test hook: how many steps took the fused path (periodic uniform 3-D boxes, spectral solver)
extern "C" long long ins_dbg_ext_fused_steps(void) { return g_fused_steps; }

static int zalloc(double** p, size_t bytes, hipStream_t s) {
  if (*p) return INS_OK;
  INS_HIP_TRY(hipMalloc(p, bytes));
  INS_HIP_TRY(hipMemsetAsync(*p, 0, bytes, s));
  return INS_OK;
}

extern "C" int ins_rk_step_ext_f64(ins_rk_t* rk, double visc, double* u, double* temp, double t, double dt, void* stream) {
  INS_REQUIRE(rk && u, "null argument");
  ins_rk_ext* e = rk->ext;
  const bool closure = e && e->closure == 1, with_temp = e && e->temp_on;
  INS_REQUIRE(with_temp == (temp != nullptr), "a temperature field needs ins_rk_set_temperature and vice versa");
  if (!closure && !with_temp) return ins_rk_step_f64(rk, visc, u, t, dt, nullptr, stream);
  const ins_grid* G = rk->grid;
  const GridDev& g = G->g;
  hipStream_t s = as_stream(stream);
  const int ns = rk->nstage, D = g.D;
  const size_t sbytes = (size_t)G->ncell * sizeof(double), vbytes = sbytes * D;
  int rc;
  // the stress scratch (D(D+1)/2 scalar fields: 6.5 GB at 512^3) exists only where the three-kernel closure runs (the one-kernel form keeps σ in registers)
  auto need_sigma = [&]() { return zalloc(&e->sigma, sbytes * (D * (D + 1) / 2), s); };
  if (closure && (rc = zalloc(&e->E, vbytes, s))) return rc;
  if (with_temp) {
    if ((rc = zalloc(&e->tempstart, sbytes, s))) return rc;
    if (e->td.dodissipation && (rc = zalloc(&e->diff, vbytes, s))) return rc;
    if ((int)e->ktemp.size() < ns) e->ktemp.resize(ns, nullptr);
    for (int i = 0; i < ns; ++i)
      if ((rc = zalloc(&e->ktemp[i], sbytes, s))) return rc;
    INS_HIP_TRY(hipMemcpyAsync(e->tempstart, temp, sbytes, hipMemcpyDeviceToDevice, s));  // state_copyto!(xstart, x)   :14
  }
  const ins_temperature_desc_t& td = e->td;
  auto bc_temp = [&]() { return with_temp ? ins_apply_bc_temp_f64(G, td.bc, td.val, nullptr, temp, stream) : INS_OK; };
  auto temp_rhs = [&](const double* v, int i) -> int {
    if (!with_temp) return INS_OK;
    int r;
    INS_HIP_TRY(hipMemsetAsync(e->ktemp[i], 0, sbytes, s));
    if ((r = ins_convection_diffusion_temp_f64(G, td.a4, v, temp, e->ktemp[i], stream))) return r;
    if (td.dodissipation && (r = ins_dissipation_f64(G, visc, td.diss_coef, v, e->diff, e->ktemp[i], stream))) return r;
    return INS_OK;
  };
  auto temp_combine = [&](int i) -> int {
    if (!with_temp) return INS_OK;
    double coefs[INS_MAX_STAGES];
    const double* ks[INS_MAX_STAGES];
    int n = 0;
    for (int j = 0; j <= i; ++j) {
      coefs[n] = dt * rk->A[i * ns + j];
      ks[n] = e->ktemp[j];
      ++n;
    }
    return ins_combine_scalar_f64(G, e->tempstart, temp, n, coefs, ks, stream);
  };

  bool fused = !ins_opt(OPT_INS_DISABLE_FUSED_RK) && !ins_opt(OPT_INS_DISABLE_EXT_FUSED) && D == 3 && G->all_periodic && G->all_dof &&
               rk->ps->kind == POISSON_SPECTRAL && ins_fast3d_supported(G) && ins_flux64_supported(G);
  for (int a = 0; fused && a < 3; ++a) fused = rk->ps->np[a] >= 2;

  if (fused) {
    // Periodic uniform 3-D box, spectral solver.  Per stage: [closure kernels ->] stage kernel (convection-diffusion + closure field + gravity
    // from temp in registers, RK combination, and w = u · diffusion(u) as a by-product) -> one temperature kernel (its right-hand side and its
    // RK combination) -> the five solver passes -> gradient-subtract with ghost fill -> temperature ghost fill.
    ++g_fused_steps;
    for (int b = 0; b < 2; ++b)
      if ((rc = zalloc(&rk->ub[b], vbytes, s))) return rc;
    if (with_temp) {
      for (int b = 0; b < 2; ++b)
        if ((rc = zalloc(&e->tb[b], sbytes, s))) return rc;
      if (td.dodissipation && (rc = zalloc(&e->w, vbytes, s))) return rc;
    }
    if ((rc = ins_k_apply_bc_u(G, u, 0, nullptr, s))) return rc;
    if ((rc = bc_temp())) return rc;
    // Stage-velocity basis (rk_step_fused_periodic): the projection writes the corrected field to another array, so the uncorrected stage
    // velocities V_m stay in memory (in the ku arrays) and no k_j is stored or read; INS_RK_KEEP_K=1 restores the k-basis.
    bool vbasis = ns > 1 && !ins_opt(OPT_INS_RK_KEEP_K);
    for (int i = 0; vbasis && i < ns; ++i) vbasis = rk->A[i * ns + i] != 0.0;
    // Every consumer of the stage velocity can correct it on the fly (the stage kernel as on the plain path, the split temperature kernel for
    // its two face velocities, the stress-tensor kernel through periodic images), so the gradient-subtract pass runs for the last stage only —
    // except when the temperature stage rides inside the stage kernel (it has no correcting variant).
    // The temperature stage itself rides inside the stage kernel (TempEpi: T as a fourth register component, the lower-face dissipation terms
    // from the halo column / halo row / previous plane); INS_EXT_TEMP_SPLIT=1 keeps it as a kernel of its own (then without a closure the
    // gradient-subtract passes between the stages can go, see above).
    const bool tin_kernel = with_temp && !ins_opt(OPT_INS_EXT_TEMP_SPLIT);
    // (with a closure and no temperature equation the stress-tensor kernel corrects on the fly as well: ins_k_smagtensor_corr)
    // The closure force as one kernel (ins_smagforce.hip: the stress never leaves the registers; it corrects on the fly like the stress-tensor
    // kernel, from the uncorrected stage velocity and its pressure)
    const bool smag1 = closure && ins_smagforce_supported(G);
    const bool incorr = vbasis && !tin_kernel && !ins_opt(OPT_INS_DISABLE_INKERNEL_CORR) && g.N[0] >= 34 && g.N[1] >= 8 && g.N[2] >= 8;
    const double* in = u;
    const double* tin = temp;
    for (int i = 0; i < ns; ++i) {
      const bool last = i == ns - 1 && ns > 1;
      double* out = last ? u : (vbasis ? rk->ku[i] : rk->ub[i & 1]);      // what the stage kernel writes (uncorrected)
      double* corrected = (last || incorr) ? out : rk->ub[i & 1];          // what the projection leaves (K4 route: with ghost volumes)
      double* tout = with_temp ? (last ? temp : e->tb[i & 1]) : nullptr;
      const bool corr_in = incorr && i > 0;  // `in` is the uncorrected V_{i-1}, ps->pI its pressure
      if (smag1) {
        if ((rc = ins_k_smagforce(G, e->theta, in, corr_in ? rk->ps->pI : nullptr, e->E, s))) return rc;
      } else if (closure) {
        if ((rc = need_sigma())) return rc;
        rc = corr_in ? ins_k_smagtensor_corr(G, e->theta, in, rk->ps->pI, e->sigma, s) : ins_smagtensor_f64(G, e->theta, in, e->sigma, stream);
        if (rc) return rc;
        if ((rc = ins_k_apply_bc_p_fields(G, e->sigma, D * (D + 1) / 2, s))) return rc;  // apply_bc_p!(σ, 0, setup)   operators.jl:1302
        if ((rc = ins_divoftensor_f64(G, e->sigma, e->E, stream))) return rc;
      }
      RkEpi epi;
      memset(&epi, 0, sizeof(epi));
      if (vbasis) {
        double beta[INS_MAX_STAGES];
        for (int m = i - 1; m >= 0; --m) {  // β_i · A[0:i,0:i] = A[i,0:i], A lower triangular
          double v = rk->A[i * ns + m];
          for (int j = m + 1; j < i; ++j) v -= beta[j] * rk->A[j * ns + m];
          beta[m] = v / rk->A[m * ns + m];
        }
        for (int m = 0; m < i; ++m) {
          if (beta[m] == 0.0) continue;
          epi.c0m1 -= beta[m];
          if (corr_in && m == i - 1) {  // V_{i-1} is this stage's stencil input: its uncorrected value is taken from registers
            epi.self_in = beta[m];
            continue;
          }
          epi.coef[epi.n] = beta[m];
          epi.k[epi.n] = rk->ku[m];
          ++epi.n;
        }
      } else {
        for (int j = 0; j < i; ++j) {
          const double coef = dt * rk->A[i * ns + j];
          if (coef == 0.0) continue;
          epi.coef[epi.n] = coef;
          epi.k[epi.n] = rk->ku[j];
          ++epi.n;
        }
        for (int i2 = i + 1; i2 < ns; ++i2)
          if (rk->A[i2 * ns + i] != 0.0) epi.write_k = 1;
      }
      if (rk->force) {  // the steady force f is not part of what is stored: Δt A[i,i] f on top of the V_m, Δt Σ_{j<=i} A[i,j] f on top of the k_j
        double cf = dt * rk->A[i * ns + i];
        if (!vbasis)
          for (int j = 0; j < i; ++j) cf += dt * rk->A[i * ns + j];
        epi.coef[epi.n] = cf;
        epi.k[epi.n] = rk->force;
        ++epi.n;
      }
      epi.coef_self = dt * rk->A[i * ns + i];
      epi.ustart = (i == 0) ? nullptr : u;
      epi.ustar = out;
      epi.extra = closure ? e->E : nullptr;
      TempEpi te;
      if (with_temp) {
        epi.gtemp = tin;
        epi.ga2 = td.a2;
        epi.gdir = td.gdir;
        if (tin_kernel) {
          memset(&te, 0, sizeof(te));
          te.temp = tin;
          te.tempstart = e->tempstart;
          te.temp_out = tout;
          bool later = false;
          for (int i2 = i + 1; i2 < ns; ++i2) later = later || rk->A[i2 * ns + i] != 0.0;
          te.ktemp_out = later ? e->ktemp[i] : nullptr;
          for (int j = 0; j < i; ++j) {
            const double c = dt * rk->A[i * ns + j];
            if (c == 0.0) continue;
            te.coef[te.n] = c;
            te.k[te.n] = e->ktemp[j];
            ++te.n;
          }
          te.c_self = dt * rk->A[i * ns + i];
          te.a4 = td.a4;
          te.dcoef = td.dodissipation ? td.diss_coef : 0.0;
          epi.tstage = &te;
        } else {
          epi.wout = td.dodissipation ? e->w : nullptr;
        }
      }
      rc = corr_in ? ins_k_momentum_rk_fused_corr(G, visc, in, rk->ps->pI, rk->ku[i], epi, s) : ins_k_momentum_rk_fused(G, visc, in, rk->ku[i], epi, s);
      if (rc) return rc;
      if (with_temp && !tin_kernel) {
        double coefs[INS_MAX_STAGES];
        const double* ks[INS_MAX_STAGES];
        int n = 0;
        for (int j = 0; j < i; ++j) {
          coefs[n] = dt * rk->A[i * ns + j];
          ks[n] = e->ktemp[j];
          ++n;
        }
        bool later = false;
        for (int i2 = i + 1; i2 < ns; ++i2) later = later || rk->A[i2 * ns + i] != 0.0;
        if ((rc = ins_k_temp_stage(G, td.a4, td.diss_coef, in, tin, td.dodissipation ? e->w : nullptr, e->tempstart, n, coefs, ks, dt * rk->A[i * ns + i],
                                   later ? e->ktemp[i] : nullptr, tout, s, corr_in ? rk->ps->pI : nullptr)))
          return rc;
      }
      if (incorr && i < ns - 1)
        rc = ins_k_project_periodic_solve_only(G, rk->ps, out, s);
      else
        rc = ins_k_project_periodic_fused(G, rk->ps, out, rk->p, i == ns - 1, s, corrected == out ? nullptr : corrected);
      if (rc) return rc;
      if (with_temp && (rc = ins_apply_bc_temp_f64(G, td.bc, td.val, nullptr, tout, stream))) return rc;
      in = corrected;
      tin = tout;
    }
    if (ns == 1) {
      INS_HIP_TRY(hipMemcpyAsync(u, rk->ub[0], vbytes, hipMemcpyDeviceToDevice, s));
      if (with_temp) INS_HIP_TRY(hipMemcpyAsync(temp, e->tb[0], sbytes, hipMemcpyDeviceToDevice, s));
    }
    return INS_OK;
  }

  // 3-D grids the tiled stage kernels take (walls, stretched grids, any solver): K1 + K6 + the extra force field in one stage kernel (the caller's u
  // is ustart for the whole step, stage velocities ping-pong in two library buffers: no snapshot copy, no combination pass, no axpy), one
  // temperature kernel per stage (right-hand side from (u, temp, diffusion(u)) + its RK combination), full projection after every stage.
  if (!ins_opt(OPT_INS_DISABLE_FUSED_RK) && !ins_opt(OPT_INS_DISABLE_EXT_FUSED) && D == 3 && ins_fast3d_supported(G)) {
    for (int b = 0; b < 2; ++b)
      if (!rk->ub[b]) {
        INS_HIP_TRY(hipMalloc(&rk->ub[b], vbytes));
        INS_HIP_TRY(hipMemcpyAsync(rk->ub[b], u, vbytes, hipMemcpyDeviceToDevice, s));  // once: volumes no kernel ever writes
      }
    if (with_temp) {
      for (int b = 0; b < 2; ++b)
        if (!e->tb[b]) {
          INS_HIP_TRY(hipMalloc(&e->tb[b], sbytes));
          INS_HIP_TRY(hipMemcpyAsync(e->tb[b], temp, sbytes, hipMemcpyDeviceToDevice, s));
        }
    }
    // The 64-wide masked stage kernel also leaves w = u·diffusion(u) (the dissipation term of the temperature equation) from
    // the diffusive parts of the fluxes it has in registers: no diffusion pass of its own.
    const bool w_from_stage = with_temp && td.dodissipation && ins_flux64m_supported(G);
    if (w_from_stage && (rc = zalloc(&e->w, vbytes, s))) return rc;  // ghost volumes stay zero (the reference's fill!(diff, 0))
    ++g_tiled_steps;
    double* cur = u;
    double* tin = temp;
    for (int i = 0; i < ns; ++i) {
      const bool last = i == ns - 1 && ns > 1;
      double* out = last ? u : rk->ub[i & 1];
      double* tout = with_temp ? (last ? temp : e->tb[i & 1]) : nullptr;
      if ((rc = ins_k_apply_bc_u(G, cur, 0, nullptr, s))) return rc;                                         // :19
      if (with_temp && (rc = ins_apply_bc_temp_f64(G, td.bc, td.val, nullptr, tin, stream))) return rc;      // :20
      if (closure && ins_smagforce_supported(G)) {  // one kernel, also with walls and on stretched grids (ins_smagforce.hip, GEN)
        if ((rc = ins_k_smagforce(G, e->theta, cur, nullptr, e->E, s))) return rc;
      } else if (closure) {
        if ((rc = need_sigma())) return rc;
        if ((rc = ins_smagtensor_f64(G, e->theta, cur, e->sigma, stream))) return rc;
        if ((rc = ins_k_apply_bc_p_fields(G, e->sigma, D * (D + 1) / 2, s))) return rc;
        if ((rc = ins_divoftensor_f64(G, e->sigma, e->E, stream))) return rc;
      }
      if (with_temp && td.dodissipation && !w_from_stage && (rc = ins_k_diffusion_flux3d(G, visc, cur, e->diff, false, s)))
        return rc;  // e->diff: shell zero since its allocation
      RkEpi epi;
      memset(&epi, 0, sizeof(epi));
      for (int j = 0; j < i; ++j) {
        const double coef = dt * rk->A[i * ns + j];
        if (coef == 0.0) continue;
        epi.coef[epi.n] = coef;
        epi.k[epi.n] = rk->ku[j];
        ++epi.n;
      }
      for (int i2 = i + 1; i2 < ns; ++i2)
        if (rk->A[i2 * ns + i] != 0.0) epi.write_k = 1;
      if (rk->force) {
        double cf = 0.0;
        for (int j = 0; j <= i; ++j) cf += dt * rk->A[i * ns + j];
        epi.coef[epi.n] = cf;
        epi.k[epi.n] = rk->force;
        ++epi.n;
      }
      epi.coef_self = dt * rk->A[i * ns + i];
      epi.ustart = (i == 0) ? nullptr : u;
      epi.ustar = out;
      epi.extra = closure ? e->E : nullptr;
      if (with_temp) {  // gravity inside the stage kernel (two loads of temp per volume)
        epi.gtemp = tin;
        epi.ga2 = td.a2;
        epi.gdir = td.gdir;
        if (w_from_stage) epi.wout = e->w;
      }
      if ((rc = ins_k_momentum_rk_fused(G, visc, cur, rk->ku[i], epi, s))) return rc;
      if (with_temp) {
        double coefs[INS_MAX_STAGES];
        const double* ks[INS_MAX_STAGES];
        int n = 0;
        for (int j = 0; j < i; ++j) {
          coefs[n] = dt * rk->A[i * ns + j];
          ks[n] = e->ktemp[j];
          ++n;
        }
        bool later = false;
        for (int i2 = i + 1; i2 < ns; ++i2) later = later || rk->A[i2 * ns + i] != 0.0;
        if ((rc = ins_k_temp_stage(G, td.a4, td.diss_coef, cur, tin, w_from_stage ? e->w : nullptr, e->tempstart, n, coefs, ks, dt * rk->A[i * ns + i],
                                   later ? e->ktemp[i] : nullptr, tout, s, nullptr, td.dodissipation && !w_from_stage ? e->diff : nullptr)))
          return rc;
      }
      if ((rc = ins_k_apply_bc_u(G, out, 0, nullptr, s))) return rc;                                          // :48
      if ((rc = ins_k_project(G, rk->ps, out, rk->p, s))) return rc;                                            // :49
      cur = out;
      tin = tout;
    }
    if (ns == 1) {
      INS_HIP_TRY(hipMemcpyAsync(u, rk->ub[0], vbytes, hipMemcpyDeviceToDevice, s));
      if (with_temp) INS_HIP_TRY(hipMemcpyAsync(temp, e->tb[0], sbytes, hipMemcpyDeviceToDevice, s));
    }
    if ((rc = ins_k_apply_bc_u(G, u, 0, nullptr, s))) return rc;                                               // :55
    return bc_temp();                                                                                          // :56
  }

  // any grid: the reference's kernel sequence
  INS_HIP_TRY(hipMemcpyAsync(rk->ustart, u, vbytes, hipMemcpyDeviceToDevice, s));
  for (int i = 0; i < ns; ++i) {
    if ((rc = ins_k_apply_bc_u(G, u, 0, nullptr, s))) return rc;
    if ((rc = bc_temp())) return rc;
    rc = ins_fast3d_supported(G) ? ins_k_momentum_fast3d_opts(G, visc, u, rk->ku[i], false, s) : ins_k_momentum_generic(G, visc, u, rk->ku[i], s);
    if (rc) return rc;
    if (with_temp && (rc = ins_gravity_f64(G, td.gdir, td.a2, temp, rk->ku[i], stream))) return rc;
    if ((rc = temp_rhs(u, i))) return rc;
    if (closure) {
      if ((rc = need_sigma())) return rc;
      if ((rc = ins_smagtensor_f64(G, e->theta, u, e->sigma, stream))) return rc;
      if ((rc = ins_k_apply_bc_p_fields(G, e->sigma, D * (D + 1) / 2, s))) return rc;
      if ((rc = ins_divoftensor_f64(G, e->sigma, e->E, stream))) return rc;
      const double one = 1.0;
      const double* ks[1] = {e->E};
      if ((rc = ins_combine_f64(G, rk->ku[i], rk->ku[i], 1, &one, ks, stream))) return rc;  // ku[i] += m(u, θ)
    }
    double coefs[INS_MAX_STAGES + 1];
    const double* ks[INS_MAX_STAGES + 1];
    int n = 0;
    double cf = 0.0;
    for (int j = 0; j <= i; ++j) {
      const double c = dt * rk->A[i * ns + j];
      cf += c;
      if (c == 0.0) continue;
      coefs[n] = c;
      ks[n] = rk->ku[j];
      ++n;
    }
    if (rk->force) {
      coefs[n] = cf;
      ks[n] = rk->force;
      ++n;
    }
    if ((rc = ins_combine_f64(G, rk->ustart, u, n, coefs, ks, stream))) return rc;
    if ((rc = temp_combine(i))) return rc;
    if ((rc = ins_k_apply_bc_u(G, u, 0, nullptr, s))) return rc;
    if ((rc = ins_k_project(G, rk->ps, u, rk->p, s))) return rc;
  }
  if ((rc = ins_k_apply_bc_u(G, u, 0, nullptr, s))) return rc;
  return bc_temp();
}
