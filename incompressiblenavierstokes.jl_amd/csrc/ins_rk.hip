// Explicit Runge-Kutta stage loop (step_explicit_runge_kutta.jl:4-59) and its cache
// (time_stepper_caches.jl:34-49).
#include <cstdlib>

#include "ins_internal.h"


int ins_k_momentum_generic(const ins_grid* G, double visc, const double* u, double* F, hipStream_t s);
int ins_k_momentum_fast3d(const ins_grid* G, double visc, const double* u, double* F, hipStream_t s);
int ins_k_momentum_fast3d_opts(const ins_grid* G, double visc, const double* u, double* F, bool zero_shell, hipStream_t s);
bool ins_fast3d_supported(const ins_grid* G);
bool ins_flux64_supported(const ins_grid* G);

int ins_k_momentum_rk_fused(const ins_grid* G, double visc, const double* u_in, double* k_out, const RkEpi& epi, hipStream_t s);
int ins_k_project_periodic_fused(const ins_grid* G, ins_poisson* ps, double* u, double* p, bool keep_p, hipStream_t s, double* uout = nullptr);

int ins_k_momentum(const ins_grid* G, double visc, const double* u, double* F, hipStream_t s) {
  if (ins_fast3d_supported(G)) return ins_k_momentum_fast3d(G, visc, u, F, s);
  return ins_k_momentum_generic(G, visc, u, F, s);
}

namespace {

struct Combine {
  int n;
  double coef[INS_MAX_STAGES + 1];  // + 1: the steady body force is one more term (ins_rk_set_bodyforce)
  const double* k[INS_MAX_STAGES + 1];
};

// K6: u = ustart + Σ_j (Δt A[i,j]) ku[j]  in ONE pass (the reference does 1 copy + i axpy passes and does
// not skip tableau zeros, step_explicit_runge_kutta.jl:35-38).  Summation order as in the reference.
__global__ __launch_bounds__(256) void k_combine(long long n, const double* ustart, double* u, Combine cb) {
  const long long stride = (long long)gridDim.x * 256 * 2;
  for (long long t = ((long long)blockIdx.x * 256 + threadIdx.x) * 2; t < n; t += stride) {
    if (t + 1 < n) {
      double2 v = *reinterpret_cast<const double2*>(ustart + t);
      for (int j = 0; j < cb.n; ++j) {
        const double2 kv = *reinterpret_cast<const double2*>(cb.k[j] + t);
        v.x += cb.coef[j] * kv.x;
        v.y += cb.coef[j] * kv.y;
      }
      *reinterpret_cast<double2*>(u + t) = v;
    } else {
      double v = ustart[t];
      for (int j = 0; j < cb.n; ++j) v += cb.coef[j] * cb.k[j][t];
      u[t] = v;
    }
  }
}

}  // namespace

static int combine_n(long long nvec, const double* base, double* out, int nterms, const double* coefs, const double* const* ks, void* stream);

extern "C" int ins_combine_f64(const ins_grid_t* G, const double* base, double* out, int nterms, const double* coefs,
                               const double* const* ks, void* stream) {
  INS_REQUIRE(G && base && out, "null argument");
  return combine_n(G->ncell * G->g.D, base, out, nterms, coefs, ks, stream);
}

extern "C" int ins_combine_scalar_f64(const ins_grid_t* G, const double* base, double* out, int nterms, const double* coefs,
                                      const double* const* ks, void* stream) {
  INS_REQUIRE(G && base && out, "null argument");
  return combine_n(G->ncell, base, out, nterms, coefs, ks, stream);
}

static int combine_n(long long nvec, const double* base, double* out, int nterms, const double* coefs, const double* const* ks, void* stream) {
  INS_REQUIRE(nterms >= 0 && nterms <= INS_MAX_STAGES + 1 && (nterms == 0 || (coefs && ks)), "bad stage terms");
  Combine cb;
  cb.n = 0;
  for (int q = 0; q < nterms; ++q) {
    if (coefs[q] == 0.0) continue;
    INS_REQUIRE(ks[q], "null stage field");
    cb.coef[cb.n] = coefs[q];
    cb.k[cb.n] = ks[q];
    ++cb.n;
  }
  const unsigned nblk = (unsigned)std::min<long long>((nvec / 2 + 255) / 256, 8192);
  hipLaunchKernelGGL(k_combine, dim3(nblk), dim3(256), 0, as_stream(stream), nvec, base, out, cb);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

static void step_graph_free(ins_rk* rk);
bool ins_k_spectral_own3d(const ins_poisson* ps);  // every pass of the solver is one of the library's own kernels (ins_poisson.hip)

extern "C" int ins_rk_create(const ins_grid_t* G, ins_poisson_t* ps, int nstage, const double* A, const double* c, ins_rk_t** out) {
  INS_REQUIRE(G && ps && A && c && out, "null argument");
  INS_REQUIRE(ps->grid == G, "psolver was created for a different grid");
  INS_REQUIRE(nstage >= 1 && nstage <= INS_MAX_STAGES, "unsupported number of stages");
  for (int i = 0; i < nstage; ++i)
    for (int j = i + 1; j < nstage; ++j) INS_REQUIRE(A[i * nstage + j] == 0.0, "shifted tableau must be lower triangular (explicit method)");
  ins_rk* rk = new ins_rk();
  rk->grid = G;
  rk->ps = ps;
  rk->nstage = nstage;
  rk->A.assign(A, A + nstage * nstage);
  rk->c.assign(c, c + nstage);
  const size_t vbytes = (size_t)G->ncell * G->g.D * sizeof(double);
  bool ok = hipMalloc(&rk->ustart, vbytes) == hipSuccess && hipMalloc(&rk->p, G->ncell * sizeof(double)) == hipSuccess;
  rk->ku.assign(nstage, nullptr);
  for (int i = 0; ok && i < nstage; ++i) ok = hipMalloc(&rk->ku[i], vbytes) == hipSuccess;
  if (ok) ok = hipMemset(rk->p, 0, G->ncell * sizeof(double)) == hipSuccess && hipMemset(rk->ustart, 0, vbytes) == hipSuccess;
  for (int i = 0; ok && i < nstage; ++i) ok = hipMemset(rk->ku[i], 0, vbytes) == hipSuccess;
  if (!ok) {
    ins_set_error("ins_rk_create: device allocation of %d vector fields failed", nstage + 1);
    ins_rk_destroy(rk);
    return INS_ERR_HIP;
  }
  *out = rk;
  return INS_OK;
}

extern "C" int ins_rk_destroy(ins_rk_t* rk) {
  if (!rk) return INS_OK;
  if (rk->ustart) (void)hipFree(rk->ustart);
  if (rk->p) (void)hipFree(rk->p);
  for (double* k : rk->ku)
    if (k) (void)hipFree(k);
  for (double* v : rk->vb)
    if (v && v != rk->ub[0] && v != rk->ub[1]) (void)hipFree(v);
  for (double* b : rk->ub)
    if (b) (void)hipFree(b);
  for (hipEvent_t e : rk->prof_events) (void)hipEventDestroy(e);
  ins_rk_ext_free(rk->ext);
  step_graph_free(rk);
  delete rk;
  return INS_OK;
}

extern "C" int ins_rk_profile_enable(ins_rk_t* rk, int enable) {
  INS_REQUIRE(rk, "null argument");
  rk->profiling = enable != 0;
  return INS_OK;
}

extern "C" int ins_rk_profile_read(ins_rk_t* rk, double* momentum_ms, int64_t* momentum_launches) {
  INS_REQUIRE(rk && momentum_ms && momentum_launches, "null argument");
  double total = 0.0;
  const size_t n = rk->prof_events.size() / 2;
  for (size_t i = 0; i < n; ++i) {
    float ms = 0.f;
    INS_HIP_TRY(hipEventSynchronize(rk->prof_events[2 * i + 1]));
    INS_HIP_TRY(hipEventElapsedTime(&ms, rk->prof_events[2 * i], rk->prof_events[2 * i + 1]));
    total += ms;
  }
  for (hipEvent_t e : rk->prof_events) (void)hipEventDestroy(e);
  rk->prof_events.clear();
  *momentum_ms = total;
  *momentum_launches = (int64_t)n;
  return INS_OK;
}

extern "C" int ins_rk_pressure(const ins_rk_t* rk, double** p) {
  INS_REQUIRE(rk && p, "null argument");
  *p = rk->p;
  return INS_OK;
}

extern "C" int ins_rk_set_bodyforce(ins_rk_t* rk, const double* force) {
  INS_REQUIRE(rk, "null argument");
  rk->force = force;
  return INS_OK;
}

extern "C" int ins_rk_stage_force(const ins_rk_t* rk, int i, double** ku) {
  INS_REQUIRE(rk && ku, "null argument");
  INS_REQUIRE(i >= 0 && i < rk->nstage, "stage index out of range");
  *ku = rk->ku[i];
  return INS_OK;
}

// Fused periodic path (3-D, all-periodic, spectral solver): per stage
//   K1+K6  k_i = momentum(u_in), u* = ustart + Σ Δt a_ij k_j      (one pass; u* goes to a ping-pong buffer)
//   K2     pI = Ω div(u*) with periodic wrap                      (no ghost fill of u* needed)
//   rocFFT forward, K3 symbol, rocFFT inverse
//   K4     u* -= ∇p on the interior + its periodic ghost images    (no apply_bc_u! launch)
// Same arithmetic, in the same order, as the reference stage loop (step_explicit_runge_kutta.jl:17-50);
// `ustart` is the caller's `u`, which stays untouched until the last stage writes the result into it.
// chain: 0 = a whole step (u valid in, valid out); bit 1 = `u` holds the previous step's UNCORRECTED result and ps->pI its pressure
// (the first stage corrects in registers and stores the corrected field as ustart); bit 2 = leave this step's result uncorrected in `u`.
static int rk_step_fused_periodic(ins_rk* rk, double visc, double* u, double dt, hipStream_t s, int chain = 0) {
  const ins_grid* G = rk->grid;
  const int ns = rk->nstage;
  const size_t vbytes = (size_t)G->ncell * 3 * sizeof(double);
  for (int b = 0; b < 2; ++b)
    if (!rk->ub[b]) {
      INS_HIP_TRY(hipMalloc(&rk->ub[b], vbytes));
      INS_HIP_TRY(hipMemsetAsync(rk->ub[b], 0, vbytes, s));
    }
  int rc;
  const bool raw_in = chain & 1, raw_out = chain & 2;
  if (!raw_in && (rc = ins_k_apply_bc_u(G, u, 0, nullptr, s))) return rc;  // :19 (first stage; later ghosts come from K4)
  // On exactly-uniform grids stages >= 2 read the previous stage's UNCORRECTED u* plus its pressure and apply
  // the projection's gradient-subtract in registers (k_momentum_flux<..., CORR>), so K4 runs for the last stage only.
  const bool no_corr = ins_opt(OPT_INS_DISABLE_INKERNEL_CORR) != 0;
  const bool inkernel = !no_corr && G->uniform_exact && ns > 1 && G->g.N[0] >= 8 && G->g.N[1] >= 8 && G->g.N[2] >= 8;
  // Stage-velocity basis.  With in-kernel correction the UNCORRECTED stage velocities V_m = ustart + Δt Σ_{j<=m} A[m,j] k_j stay in
  // memory anyway (they are the next stencil's input), and when every A[m,m] != 0 they span the same space as {ustart, k_j}:
  //   V_i = (1 - Σ_m β_im) ustart + Σ_{m<i} β_im V_m + Δt A[i,i] k_i,     β_i · A[0:i,0:i] = A[i,0:i].
  // So no k_j is ever written or read: RK44 moves 336 instead of 432 B per cell and step through the stage kernels (β_3 = (1/3, 2/3, 1/3),
  // all other β = 0).  Algebraically the reference's combination (step_explicit_runge_kutta.jl:35-38); rounding differs at the 1e-16 level.
  // INS_RK_KEEP_K=1 restores the k-basis (and fills the ku cache arrays, which this basis leaves untouched).
  const bool keep_k = ins_opt(OPT_INS_RK_KEEP_K) != 0;
  bool vbasis = inkernel && !keep_k;
  for (int i = 0; vbasis && i < ns; ++i) vbasis = rk->A[i * ns + i] != 0.0;
  if (vbasis && (int)rk->vb.size() < ns - 1) {
    rk->vb.resize(ns - 1, nullptr);
    for (int m = 0; m < ns - 1; ++m)
      if (!rk->vb[m]) {
        if (m < 2 && rk->ub[m]) {
          rk->vb[m] = rk->ub[m];
          continue;
        }
        INS_HIP_TRY(hipMalloc(&rk->vb[m], vbytes));
        INS_HIP_TRY(hipMemsetAsync(rk->vb[m], 0, vbytes, s));
      }
  }
  const double* in = u;
  for (int i = 0; i < ns; ++i) {
    double* out = (i == ns - 1 && ns > 1) ? u : (vbasis ? rk->vb[i] : rk->ub[i & 1]);
    RkEpi epi;
    memset(&epi, 0, sizeof(epi));
    if (vbasis) {
      double beta[INS_MAX_STAGES];
      for (int m = i - 1; m >= 0; --m) {  // β_i · A[0:i,0:i] = A[i,0:i], A lower triangular
        double v = rk->A[i * ns + m];
        for (int j = m + 1; j < i; ++j) v -= beta[j] * rk->A[j * ns + m];
        beta[m] = v / rk->A[m * ns + m];
      }
      const bool flux64 = ins_flux64_supported(G);  // the 62-wide kernel has no register copy of the uncorrected input
      for (int m = 0; m < i; ++m) {
        if (beta[m] == 0.0) continue;
        epi.c0m1 -= beta[m];
        if (m == i - 1 && flux64) {  // V_{i-1} is this stage's stencil input
          epi.self_in = beta[m];
          continue;
        }
        epi.coef[epi.n] = beta[m];
        epi.k[epi.n] = rk->vb[m];
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
    if (rk->force) {  // steady body force (operators.jl:873-880): k_j = F_j + f, so f enters with Δt A[i,i] in the stage-velocity basis
      double cf = dt * rk->A[i * ns + i];  // (the V_m already hold their share) and with Δt Σ_{j<=i} A[i,j] in the k-basis (ku[j] = F_j)
      if (!vbasis)
        for (int j = 0; j < i; ++j) cf += dt * rk->A[i * ns + j];
      epi.coef[epi.n] = cf;
      epi.k[epi.n] = rk->force;
      ++epi.n;
    }
    epi.coef_self = dt * rk->A[i * ns + i];
    epi.ustart = (i == 0) ? nullptr : (raw_in ? rk->ustart : u);  // raw_in: the corrected start field lives in the cache array
    epi.ustar = out;
    if (i == 0 && raw_in) epi.ustart_out = rk->ustart;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (rk->profiling) {
      INS_HIP_TRY(hipEventCreate(&e0));
      INS_HIP_TRY(hipEventCreate(&e1));
      INS_HIP_TRY(hipEventRecord(e0, s));
    }
    rc = (inkernel && (i > 0 || raw_in)) ? ins_k_momentum_rk_fused_corr(G, visc, in, rk->ps->pI, rk->ku[i], epi, s)
                             : ins_k_momentum_rk_fused(G, visc, in, rk->ku[i], epi, s);
    if (rc) return rc;
    if (rk->profiling) {
      INS_HIP_TRY(hipEventRecord(e1, s));
      rk->prof_events.push_back(e0);
      rk->prof_events.push_back(e1);
    }
    rc = (inkernel && (i < ns - 1 || raw_out)) ? ins_k_project_periodic_solve_only(G, rk->ps, out, s)
                                               : ins_k_project_periodic_fused(G, rk->ps, out, rk->p, i == ns - 1, s);
    if (rc) return rc;
    in = out;
  }
  if (ns == 1) INS_HIP_TRY(hipMemcpyAsync(u, rk->ub[0], vbytes, hipMemcpyDeviceToDevice, s));
  return INS_OK;
}

// Fused periodic path in 2-D (uniform periodic power-of-two boxes): per stage the flux-form stage kernel (K1 + K6, ins_flux2d.hip) and the
// four-launch projection above — five launches per stage instead of ten, which is what a 128² .. 512² grid is bound by; with the in-register correction
// (stages >= 2, ins_flux2d.hip CORR) the projections between two stages only solve: four launches per stage, 17 per RK44 step.
// chain: as rk_step_fused_periodic (bit 1: `u` holds the previous step's uncorrected result and ps->pI its pressure; bit 2: leave this step's result uncorrected).
static int rk_step_fused_periodic_2d(ins_rk* rk, double visc, double* u, double dt, hipStream_t s, int chain = 0) {
  const ins_grid* G = rk->grid;
  const int ns = rk->nstage;
  const size_t vbytes = (size_t)G->ncell * 2 * sizeof(double);
  for (int b = 0; b < 2; ++b)
    if (!rk->ub[b]) {
      INS_HIP_TRY(hipMalloc(&rk->ub[b], vbytes));
      INS_HIP_TRY(hipMemsetAsync(rk->ub[b], 0, vbytes, s));
    }
  int rc;
  const bool raw_in = chain & 1, raw_out = chain & 2;
  if (!raw_in && (rc = ins_k_apply_bc_u(G, u, 0, nullptr, s))) return rc;  // :19 (first stage; later ghosts come with the gradient-subtract)
  const bool incorr = ns > 1 && !ins_opt(OPT_INS_DISABLE_INKERNEL_CORR) && !ins_opt(OPT_INS_DISABLE_CORR2D) && G->g.N[0] >= 6 && G->g.N[1] >= 6;
  if (chain && !incorr) {
    ins_set_error("chained 2-D steps need the in-register correction");
    return INS_ERR_INVALID;
  }
  const double* in = u;
  for (int i = 0; i < ns; ++i) {
    double* out = (i == ns - 1 && ns > 1) ? u : rk->ub[i & 1];
    RkEpi epi;
    memset(&epi, 0, sizeof(epi));
    for (int j = 0; j < i; ++j) {
      const double coef = dt * rk->A[i * ns + j];
      if (coef == 0.0) continue;
      epi.coef[epi.n] = coef;
      epi.k[epi.n] = rk->ku[j];
      ++epi.n;
    }
    if (rk->force) {
      double cf = 0.0;
      for (int j = 0; j <= i; ++j) cf += dt * rk->A[i * ns + j];
      epi.coef[epi.n] = cf;
      epi.k[epi.n] = rk->force;
      ++epi.n;
    }
    for (int i2 = i + 1; i2 < ns; ++i2)
      if (rk->A[i2 * ns + i] != 0.0) epi.write_k = 1;
    epi.coef_self = dt * rk->A[i * ns + i];
    epi.ustart = (i == 0) ? nullptr : (raw_in ? rk->ustart : u);  // raw_in: the corrected start field lives in the cache array
    epi.ustar = out;
    if (i == 0 && raw_in) epi.ustart_out = rk->ustart;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (rk->profiling) {
      INS_HIP_TRY(hipEventCreate(&e0));
      INS_HIP_TRY(hipEventCreate(&e1));
      INS_HIP_TRY(hipEventRecord(e0, s));
    }
    // stages >= 2 read the previous stage's UNCORRECTED u* and its pressure and correct in registers (k_flux2d<…, CORR>): between two stages the projection
    // only solves (three launches instead of four, 48 B per volume less); the last stage's projection materialises u, p and their ghosts
    if ((rc = ins_k_flux2d(G, visc, in, rk->ku[i], &epi, s, (incorr && (i > 0 || raw_in)) ? rk->ps->pI : nullptr))) return rc;
    if (rk->profiling) {
      INS_HIP_TRY(hipEventRecord(e1, s));
      rk->prof_events.push_back(e0);
      rk->prof_events.push_back(e1);
    }
    rc = (incorr && (i < ns - 1 || raw_out)) ? ins_k_project_periodic_solve_only_2d(G, rk->ps, out, s) : ins_k_project_periodic_fused_2d(G, rk->ps, out, rk->p, i == ns - 1, s);
    if (rc) return rc;
    in = out;
  }
  if (ns == 1) INS_HIP_TRY(hipMemcpyAsync(u, rk->ub[0], vbytes, hipMemcpyDeviceToDevice, s));
  return INS_OK;
}

// ------------------------------------------------------------------------------------------------ a step as a hipGraph (opt-in: INS_STEP_GRAPH=1)
// A 32^3 .. 64^3 box (or a 2-D grid) runs ~24 dependent launches of a few microseconds each per RK44 step.  With INS_STEP_GRAPH=1 ins_rk_steps_f64 captures
// the launches of ONE step of its loop into a hipGraph the first time it sees (u, Δt, ν) and replays it for the following steps — the same kernels with the
// same arguments in the same order, so results are bitwise those of the plain loop (tests/test_gpu_step_graph.py).
// Measured (profiles/r03_step_graph_lab.txt, RK44): the HOST time to issue a step drops 0.09 -> 0.02 ms, but the device span GROWS — 32^3: 0.137 -> 0.168
// ms/step, 128^2: 0.113 -> 0.140 — because these steps are bound by the device's dependent-dispatch latency (~5.7 us per kernel), not by the host's issue
// rate, and a graph node costs ~1.2 us more than a stream launch on this runtime.  So the graph is NOT the default: it is for callers whose host thread has
// other work to do while the steps run (the reference's processors / a training loop), and the lever for small boxes is fewer launches per stage.
//   * all-periodic boxes on the spectral solver's own FFT passes only (no rocFFT / rocBLAS call, no host read inside the step);
//   * the caller's stream may be the legacy null stream, which cannot be captured: capture and replay run on a stream of the cache, ordered against
//     the caller's stream by events on both sides;
//   * the first step of a call always runs directly (lazy allocations and attribute settings happen there, none during the capture); a capture that fails
//     for any reason is dropped and the loop continues with direct launches.
struct StepGraph {
  hipStream_t gs = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
  hipGraphExec_t exec = nullptr;
  double* u = nullptr;
  double dt = 0.0, visc = 0.0;
  const double* force = nullptr;
  int kind = 0;
  long long epoch = -1;
  long long replays = 0;
  bool broken = false;  // a capture failed on this cache: do not try again
};

static void step_graph_free(ins_rk* rk) {
  StepGraph* g = static_cast<StepGraph*>(rk->step_graph);
  if (!g) return;
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->ev_in) (void)hipEventDestroy(g->ev_in);
  if (g->ev_out) (void)hipEventDestroy(g->ev_out);
  if (g->gs) (void)hipStreamDestroy(g->gs);
  delete g;
  rk->step_graph = nullptr;
}

// test / lab hook: how many steps this cache has run as graph replays
extern "C" long long ins_dbg_rk_graph_replays(const ins_rk_t* rk) { return (rk && rk->step_graph) ? static_cast<const StepGraph*>(rk->step_graph)->replays : 0; }

// kind 1: the chained middle step of the fused periodic 3-D loop; 2: a whole fused periodic step (3-D without the chain, or 2-D).  0: no graph.
static int step_graph_kind(const ins_rk* rk, bool chain_ok) {
  const ins_grid* G = rk->grid;
  if (!ins_opt(OPT_INS_STEP_GRAPH) || ins_opt(OPT_INS_DISABLE_STEP_GRAPH) || rk->profiling || rk->ext) return 0;
  if (ins_opt(OPT_INS_DISABLE_FUSED_RK) || !G->all_periodic || rk->ps->kind != POISSON_SPECTRAL) return 0;
  if (G->g.D == 3) {
    if (!(G->all_dof && ins_fast3d_supported(G) && ins_k_spectral_own3d(rk->ps))) return 0;
    for (int a = 0; a < 3; ++a)
      if (rk->ps->np[a] < 2) return 0;
    return chain_ok ? 1 : 2;
  }
  return (ins_poisson_own2d(rk->ps) && ins_flux2d_supported(G)) ? (chain_ok ? 1 : 2) : 0;
}

// one step of the chained loop (3-D or 2-D fused periodic path)
static int chain_step(ins_rk* rk, double visc, double* u, double dt, hipStream_t s, int chain) {
  return rk->grid->g.D == 2 ? rk_step_fused_periodic_2d(rk, visc, u, dt, s, chain) : rk_step_fused_periodic(rk, visc, u, dt, s, chain);
}

// Capture `enqueue(gs)` into g->exec.  Nothing executes here.  false: no graph (the cache is marked broken).
template <typename F>
static bool step_graph_capture(StepGraph* g, F&& enqueue) {
  if (g->exec) {
    (void)hipGraphExecDestroy(g->exec);
    g->exec = nullptr;
  }
  if (!g->gs && hipStreamCreateWithFlags(&g->gs, hipStreamNonBlocking) != hipSuccess) return false;
  if (!g->ev_in && hipEventCreateWithFlags(&g->ev_in, hipEventDisableTiming) != hipSuccess) return false;
  if (!g->ev_out && hipEventCreateWithFlags(&g->ev_out, hipEventDisableTiming) != hipSuccess) return false;
  if (hipStreamBeginCapture(g->gs, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  const int rc = enqueue(g->gs);
  hipGraph_t graph = nullptr;
  const hipError_t e = hipStreamEndCapture(g->gs, &graph);
  bool ok = rc == INS_OK && e == hipSuccess && graph != nullptr;
  if (ok) ok = hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0) == hipSuccess;
  if (graph) (void)hipGraphDestroy(graph);
  if (!ok) {
    (void)hipGetLastError();
    g->exec = nullptr;
  }
  return ok;
}

// nsteps steps of size dt with the final correction of every step but the last folded into the next step's first stage kernel
// (same arithmetic per cell; the uncorrected intermediate results never become visible).  Falls back to single steps elsewhere.
extern "C" int ins_rk_steps_f64(ins_rk_t* rk, double visc, double* u, double t, double dt, int nsteps, void* stream) {
  INS_REQUIRE(rk && u && nsteps >= 0, "bad argument");
  const ins_grid* G = rk->grid;
  hipStream_t s = as_stream(stream);
  const GridDev& g = G->g;
  const bool no_fuse = ins_opt(OPT_INS_DISABLE_FUSED_RK) != 0, no_corr = ins_opt(OPT_INS_DISABLE_INKERNEL_CORR) != 0,
             no_chain = ins_opt(OPT_INS_DISABLE_STEP_CHAIN) != 0;
  // (boxes too narrow for the 64-wide stage kernel chain on the 62-wide one: it stores the corrected start field too — round 3)
  bool ok = !no_fuse && !no_corr && !no_chain && !rk->force && g.D == 3 && G->all_periodic && G->all_dof && rk->ps->kind == POISSON_SPECTRAL && ins_fast3d_supported(G) &&
            G->uniform_exact && rk->nstage > 1 && g.N[0] >= 8 && g.N[1] >= 8 && g.N[2] >= 8;  // = in-kernel correction runs
  const bool ok2d = !no_fuse && !no_corr && !no_chain && !ins_opt(OPT_INS_DISABLE_CORR2D) && !rk->force && !rk->ext && g.D == 2 && rk->ps->kind == POISSON_SPECTRAL &&
                    ins_poisson_own2d(rk->ps) && ins_flux2d_supported(G) && rk->nstage > 1 && g.N[0] >= 6 && g.N[1] >= 6;
  const int gkind = step_graph_kind(rk, (ok || ok2d) && nsteps >= 2);
  if (gkind && nsteps >= 3) {
    StepGraph* sg = static_cast<StepGraph*>(rk->step_graph);
    if (!sg) rk->step_graph = sg = new StepGraph();
    int rc, done = 0;
    // first step: direct (kind 1: it leaves its result uncorrected for the chain)
    if ((rc = gkind == 1 ? chain_step(rk, visc, u, dt, s, 2) : ins_rk_step_f64(rk, visc, u, t, dt, nullptr, stream))) return rc;
    done = 1;
    const int last_direct = gkind == 1 ? 1 : 0;  // kind 1: the last step corrects (chain bit 1 only) and runs directly
    const int nreplay = nsteps - done - last_direct;
    const bool same = sg->exec && sg->u == u && sg->dt == dt && sg->visc == visc && sg->force == rk->force && sg->kind == gkind && sg->epoch == ins_opt_epoch();
    if (!same && !sg->broken) {
      const bool got = step_graph_capture(sg, [&](hipStream_t gs) {
        return gkind == 1 ? chain_step(rk, visc, u, dt, gs, 3) : ins_rk_step_f64(rk, visc, u, t, dt, nullptr, gs);
      });
      sg->u = u, sg->dt = dt, sg->visc = visc, sg->force = rk->force, sg->kind = gkind, sg->epoch = ins_opt_epoch();
      if (!got) sg->broken = true;
    }
    if (sg->exec && !sg->broken && nreplay > 0) {
      INS_HIP_TRY(hipEventRecord(sg->ev_in, s));
      INS_HIP_TRY(hipStreamWaitEvent(sg->gs, sg->ev_in, 0));
      for (int n = 0; n < nreplay; ++n) INS_HIP_TRY(hipGraphLaunch(sg->exec, sg->gs));
      INS_HIP_TRY(hipEventRecord(sg->ev_out, sg->gs));
      INS_HIP_TRY(hipStreamWaitEvent(s, sg->ev_out, 0));
      sg->replays += nreplay;
      done += nreplay;
    }
    for (int n = done; n < nsteps; ++n) {  // the last step of a chain, or everything when no graph exists
      if (gkind == 1)
        rc = chain_step(rk, visc, u, dt, s, 1 | (n < nsteps - 1 ? 2 : 0));
      else
        rc = ins_rk_step_f64(rk, visc, u, t + n * dt, dt, nullptr, stream);
      if (rc) return rc;
    }
    return INS_OK;
  }
  if (ok2d && nsteps >= 2) {  // 2-D fused path: the same chain (the final gradient-subtract / ghost pass of every step but the last goes into the next step's first stage kernel)
    for (int n = 0; n < nsteps; ++n) {
      int rc = chain_step(rk, visc, u, dt, s, (n > 0 ? 1 : 0) | (n < nsteps - 1 ? 2 : 0));
      if (rc) return rc;
    }
    return INS_OK;
  }
  if (!ok || nsteps < 2) {
    for (int n = 0; n < nsteps; ++n) {
      int rc = ins_rk_step_f64(rk, visc, u, t + n * dt, dt, nullptr, stream);
      if (rc) return rc;
    }
    return INS_OK;
  }
  for (int n = 0; n < nsteps; ++n) {
    int rc = rk_step_fused_periodic(rk, visc, u, dt, s, (n > 0 ? 1 : 0) | (n < nsteps - 1 ? 2 : 0));
    if (rc) return rc;
  }
  return INS_OK;
}

// planes: nsets x 18 device pointers.  nsets == 1: the same boundary data for every ghost fill of the step; nsets == nstage + 1: set q holds the data at
// time tstart (q = 0) / tstart + c[q-1] Δt — the fill before stage i's momentum! reads set i (the OLD stage time, step_explicit_runge_kutta.jl:19), the fill
// before its projection and the final one read set i + 1 (:32, :48, :55).
static int rk_step_any(ins_rk_t* rk, double visc, double* u, double dt, const double* const* planes, int nsets, void* stream) {
  INS_REQUIRE(rk && u, "null argument");
  const ins_grid* G = rk->grid;
  hipStream_t s = as_stream(stream);
  {
    const bool no_fuse = ins_opt(OPT_INS_DISABLE_FUSED_RK) != 0;
    const GridDev& g = G->g;
    bool ok = !no_fuse && !planes && g.D == 3 && G->all_periodic && G->all_dof && rk->ps->kind == POISSON_SPECTRAL && ins_fast3d_supported(G);
    for (int a = 0; ok && a < 3; ++a) ok = rk->ps->np[a] >= 2;
    if (ok) return rk_step_fused_periodic(rk, visc, u, dt, s);
    if (!no_fuse && !planes && g.D == 2 && ins_poisson_own2d(rk->ps) && ins_flux2d_supported(G)) return rk_step_fused_periodic_2d(rk, visc, u, dt, s);
  }
  const long long nvec = G->ncell * G->g.D;
  const double** dplanes = nullptr;
  struct Guard {
    const double** p;
    ~Guard() {
      if (p) (void)hipFree(p);
    }
  } guard{nullptr};
  if (planes) {
    INS_HIP_TRY(hipMalloc(&dplanes, (size_t)nsets * 18 * sizeof(double*)));
    guard.p = dplanes;
    INS_HIP_TRY(hipMemcpy(dplanes, planes, (size_t)nsets * 18 * sizeof(double*), hipMemcpyHostToDevice));
  }
  auto pset = [&](int q) -> const double** { return dplanes ? dplanes + 18 * (nsets > 1 ? q : 0) : nullptr; };
  int rc;
  const int ns = rk->nstage;
  // Time-independent boundary data: K6 runs as K1's epilogue (no k_combine pass, no
  // snapshot copy: the caller's u is ustart for the whole step and the stage velocities ping-pong in two library buffers,
  // as on the periodic path).  Every non-interior volume of a stage buffer is (re)written by apply_bc_u! before it is read.
  const bool no_fuse_np = ins_opt(OPT_INS_DISABLE_FUSED_RK) != 0;
  const bool tiled = ins_fast3d_supported(G);  // else: the generic kernel with the same epilogue (2-D grids, tiny boxes)
  if (!no_fuse_np && !planes) {
    const size_t vbytes = (size_t)nvec * sizeof(double);
    for (int b = 0; b < 2; ++b)
      if (!rk->ub[b]) {
        INS_HIP_TRY(hipMalloc(&rk->ub[b], vbytes));
        INS_HIP_TRY(hipMemcpyAsync(rk->ub[b], u, vbytes, hipMemcpyDeviceToDevice, s));  // once: volumes no kernel ever writes
      }
    // Masked grids with Periodic / Dirichlet sides and the direct solver: stages >= 2 read the previous stage's UNCORRECTED u* and its
    // pressure and apply `u = u* - ∇p` in registers on the degrees of freedom (CORR = 3 of the 62-wide stage kernel), so between two
    // stages the projection only solves for p: no gradient-subtract pass over u, no second ghost fill.  Same arithmetic per volume as
    // project! + apply_bc_u! (boundary data is time-independent on this entry point); INS_DISABLE_INKERNEL_CORR restores them.
    const bool incorr = tiled && ns > 1 && !ins_opt(OPT_INS_DISABLE_INKERNEL_CORR) && ins_corr3_supported(G) && ins_k_project_fdm_fused(rk->ps);
    // Stage-velocity basis, as on the periodic path (rk_step_fused_periodic): with the in-kernel correction the uncorrected stage velocities
    // V_m (boundary data applied) stay in memory as the next stencil's input, so the stage combination is written in terms of them and no
    // k_j is stored or read (RK44: 360 instead of 432 B per cell and step through the stage kernels; on volumes that are no DOF every term
    // holds the same boundary value and the weights sum to one).  The ku arrays serve as the V_m buffers; INS_RK_KEEP_K=1 restores the k-basis.
    bool vbasis = incorr && !ins_opt(OPT_INS_RK_KEEP_K);
    for (int i = 0; vbasis && i < ns; ++i) vbasis = rk->A[i * ns + i] != 0.0;
    double* cur = u;
    for (int i = 0; i < ns; ++i) {
      const bool corr_in = incorr && i > 0;
      if (!corr_in && (rc = ins_k_apply_bc_u(G, cur, 0, nullptr, s))) return rc;  // :19 (corr_in: `cur` got its boundary data at :48 already)
      double* out = (i == ns - 1 && ns > 1) ? u : (vbasis ? rk->ku[i] : rk->ub[i & 1]);
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
      if (rk->force) {  // k_j = F_j + f: the V_m already hold their share of f (stage-velocity basis)
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
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (rk->profiling) {
        INS_HIP_TRY(hipEventCreate(&e0));
        INS_HIP_TRY(hipEventCreate(&e1));
        INS_HIP_TRY(hipEventRecord(e0, s));
      }
      if (corr_in)
        rc = ins_k_momentum_rk_fused_corr3(G, visc, cur, rk->p, rk->ku[i], epi, s);
      else
        rc = tiled ? ins_k_momentum_rk_fused(G, visc, cur, rk->ku[i], epi, s)
                   : (ins_flux2d_supported(G) ? ins_k_flux2d(G, visc, cur, rk->ku[i], &epi, s) : ins_k_momentum_rk_fused_generic(G, visc, cur, rk->ku[i], epi, s));
      if (rc) return rc;   // :21, :35-38
      if (rk->profiling) {
        INS_HIP_TRY(hipEventRecord(e1, s));
        rk->prof_events.push_back(e0);
        rk->prof_events.push_back(e1);
      }
      cur = out;
      if ((rc = ins_k_apply_bc_u(G, cur, 0, nullptr, s))) return rc;           // :48
      if (incorr && i < ns - 1)
        rc = ins_k_project_fdm_solve_only(G, rk->ps, cur, rk->p, s);            // :49 without its gradient-subtract
      else
        rc = ins_k_project(G, rk->ps, cur, rk->p, s);                           // :49
      if (rc) return rc;
    }
    if (ns == 1) INS_HIP_TRY(hipMemcpyAsync(u, rk->ub[0], vbytes, hipMemcpyDeviceToDevice, s));
    return ins_k_apply_bc_u(G, u, 0, nullptr, s);                              // :55
  }
  // copyto!(ustart, u)                                                   step_explicit_runge_kutta.jl:14
  INS_HIP_TRY(hipMemcpyAsync(rk->ustart, u, nvec * sizeof(double), hipMemcpyDeviceToDevice, s));
  for (int i = 0; i < ns; ++i) {
    if ((rc = ins_k_apply_bc_u(G, u, 0, pset(i), s))) return rc;             // :19
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (rk->profiling) {
      INS_HIP_TRY(hipEventCreate(&e0));
      INS_HIP_TRY(hipEventCreate(&e1));
      INS_HIP_TRY(hipEventRecord(e0, s));
    }
    // ku[i]'s ghost shell is zero from ins_rk_create and never written, so the fast path skips re-zeroing it
    rc = ins_fast3d_supported(G) ? ins_k_momentum_fast3d_opts(G, visc, u, rk->ku[i], false, s)
                                 : ins_k_momentum_generic(G, visc, u, rk->ku[i], s);   // :21
    if (rc) return rc;
    if (rk->profiling) {
      INS_HIP_TRY(hipEventRecord(e1, s));
      rk->prof_events.push_back(e0);
      rk->prof_events.push_back(e1);
    }
    Combine cb;                                                               // :35-38
    cb.n = 0;
    for (int j = 0; j <= i; ++j) {
      const double coef = dt * rk->A[i * ns + j];
      if (coef == 0.0) continue;
      cb.coef[cb.n] = coef;
      cb.k[cb.n] = rk->ku[j];
      ++cb.n;
    }
    if (rk->force) {
      double cf = 0.0;
      for (int j = 0; j <= i; ++j) cf += dt * rk->A[i * ns + j];
      cb.coef[cb.n] = cf;
      cb.k[cb.n] = rk->force;
      ++cb.n;
    }
    const unsigned nblk = (unsigned)std::min<long long>((nvec / 2 + 255) / 256, 8192);
    hipLaunchKernelGGL(k_combine, dim3(nblk), dim3(256), 0, s, nvec, rk->ustart, u, cb);
    INS_LAUNCH_CHECK();
    if ((rc = ins_k_apply_bc_u(G, u, 0, pset(i + 1), s))) return rc;         // :48
    if ((rc = ins_k_project(G, rk->ps, u, rk->p, s))) return rc;              // :49
  }
  if ((rc = ins_k_apply_bc_u(G, u, 0, pset(ns), s))) return rc;              // :55
  if (planes) INS_HIP_TRY(hipStreamSynchronize(s));                           // dplanes is freed on return
  return INS_OK;
}

extern "C" int ins_rk_step_f64(ins_rk_t* rk, double visc, double* u, double t, double dt, const double* const* planes, void* stream) {
  (void)t;  // boundary data is time-independent on this entry point (see header)
  return rk_step_any(rk, visc, u, dt, planes, 1, stream);
}

// timestep! with time-dependent Dirichlet data (boundary_conditions.jl:351-357: bc.u(α, x..., t)): closures cannot cross the ABI, but the times at which
// the stage loop fills the ghost volumes are known before the step — tstart and tstart + c[i] Δt —, so the host evaluates its closures on the boundary
// planes for those nstage + 1 times and the whole stage loop runs here (the reference's kernel sequence: apply_bc_u!, momentum!, the combination,
// apply_bc_u!, project!).  planes_by_time: (nstage + 1) x 18 device pointers, entry (q·18 + (β·2 + side)·3 + α) as in ins_apply_bc_u_f64 (NULL: constant data).
extern "C" int ins_rk_step_bc_f64(ins_rk_t* rk, double visc, double* u, double t, double dt, const double* const* planes_by_time, void* stream) {
  INS_REQUIRE(rk && u && planes_by_time, "null argument");
  (void)t;
  return rk_step_any(rk, visc, u, dt, planes_by_time, rk->nstage + 1, stream);
}
