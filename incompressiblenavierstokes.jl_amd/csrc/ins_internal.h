// Internal definitions shared by the HIP translation units of libinship.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "ins_hip.h"

#define INS_EPS 2.220446049250313e-16
#define INS_MAX_STAGES 16

// ------------------------------------------------------------------------------------------------
// Error plumbing: every extern "C" entry returns an INS_ERR_* code and records a message.
// ------------------------------------------------------------------------------------------------
void ins_set_error(const char* fmt, ...);

#define INS_HIP_TRY(expr)                                                                   \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      ins_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));   \
      return INS_ERR_HIP;                                                                   \
    }                                                                                       \
  } while (0)

#define INS_FFT_TRY(expr)                                                         \
  do {                                                                            \
    hipfftResult _r = (expr);                                                     \
    if (_r != HIPFFT_SUCCESS) {                                                   \
      ins_set_error("%s:%d: %s -> hipfftResult %d", __FILE__, __LINE__, #expr, (int)_r); \
      return INS_ERR_FFT;                                                         \
    }                                                                             \
  } while (0)

#define INS_REQUIRE(cond, msg)                                        \
  do {                                                                \
    if (!(cond)) {                                                    \
      ins_set_error("%s:%d: %s (%s)", __FILE__, __LINE__, msg, #cond); \
      return INS_ERR_INVALID;                                         \
    }                                                                 \
  } while (0)

#define INS_LAUNCH_CHECK() INS_HIP_TRY(hipGetLastError())

// ------------------------------------------------------------------------------------------------
// Run-time switches (ins_options.hip): name = environment variable that initialises it = name accepted by ins_set_option.
// ------------------------------------------------------------------------------------------------
#define INS_OPT_LIST(X)          \
  X(INS_DISABLE_FAST3D)          \
  X(INS_K1_LDS)                  \
  X(INS_FLUX_ROWS)               \
  X(INS_FLUX_ZC)                 \
  X(INS_FLUX_XW)                 \
  X(INS_FLUX_BAR)                \
  X(INS_FLUX64_62_FROM)          \
  X(INS_DISABLE_FLUX128)         \
  X(INS_DISABLE_YZ_FUSED)        \
  X(INS_YZ_FUSED)                \
  X(INS_YZ_SKEL)                 \
  X(INS_YZ_PARTITIONS)           \
  X(INS_F32_ONE_COLUMN)          \
  X(INS_FLUX128_CORR)            \
  X(INS_FLUX128_XW)              \
  X(INS_FLUX128_ROWS)            \
  X(INS_FLUX128_ROWS_CORR)       \
  X(INS_FLUX128_NW)              \
  X(INS_FLUX128_ZC)              \
  X(INS_DISABLE_FDM_ZDCT)        \
  X(INS_DISABLE_FDM_ZFFT)        \
  X(INS_DISABLE_FDM_XFFT)        \
  X(INS_DISABLE_FDM_XYFFT)       \
  X(INS_DISABLE_OWNFFT)          \
  X(INS_OWNFFT_POW2_ONLY)        \
  X(INS_FIELDS_NO_MARCH)         \
  X(INS_FIELDS_ROWS)             \
  X(INS_FIELDS_NOBAR)            \
  X(INS_FIELDS_ZC)               \
  X(INS_DISABLE_FLUX2D)          \
  X(INS_DISABLE_FLUX64M)         \
  X(INS_DISABLE_SMAGFORCE)       \
  X(INS_DISABLE_SMAGFORCE_GEN)   \
  X(INS_SMAGFORCE_FORCE_GEN)     \
  X(INS_SMAGFORCE_ZC)            \
  X(INS_SMAGFORCE_BAR)         \
  X(INS_FLUX64M_ZC)              \
  X(INS_FLUX64M_NOBAR)           \
  X(INS_FLUX64M_XW)              \
  X(INS_FLUX64M_ROWS)            \
  X(INS_FLUX64M_SKIP_FIRST)      \
  X(INS_DISABLE_FLUX64)          \
  X(INS_FLUX64_ROWS)             \
  X(INS_FLUX64_ROWS_CORR)        \
  X(INS_FLUX64_ZC)               \
  X(INS_FLUX64_ZC_CORR)          \
  X(INS_FLUX64_XW)               \
  X(INS_FLUX64_LDS)              \
  X(INS_FLUX64_SKEL)             \
  X(INS_FLUX64_NW)               \
  X(INS_FLUX64_NT)               \
  X(INS_FLUX64_NOBAR)            \
  X(INS_FLUX64_MINTILES)         \
  X(INS_FLUX64_XW_CORR)          \
  X(INS_UNIFORM_BITWISE)         \
  X(INS_PHAT_DENSE)              \
  X(INS_DISABLE_FDM_FUSED)       \
  X(INS_DISABLE_FDM_FOLD)        \
  X(INS_DISABLE_FDM_UNFOLD4)     \
  X(INS_DISABLE_FDM_FOLDFUSE)    \
  X(INS_DISABLE_INKERNEL_CORR)   \
  X(INS_RK_KEEP_K)               \
  X(INS_DISABLE_EXT_FUSED)       \
  X(INS_EXT_TEMP_SPLIT)          \
  X(INS_DISABLE_FUSED_RK)        \
  X(INS_DISABLE_STEP_CHAIN)      \
  X(INS_DISABLE_STEP_GRAPH)      \
  X(INS_DISABLE_LINE3)           \
  X(INS_DISABLE_XYFUSED)         \
  X(INS_DISABLE_CORR2D)          \
  X(INS_SPECTRUM_ROCFFT)         \
  X(INS_X_SKEL)                  \
  X(INS_LINE3_TK)                \
  X(INS_LINE3_WGS)               \
  X(INS_STEP_GRAPH)              \
  X(INS_ZSOLVE_SKEL)             \
  X(INS_ZSOLVE_TK)               \
  X(INS_ZSOLVE_RADIX4)           \
  X(INS_ZSOLVE_NT)               \
  X(INS_ZSOLVE_WGS)              \
  X(INS_DISABLE_ZSOLVE)          \
  X(INS_ZTRI_SKEL)               \
  X(INS_CG_HOSTSYNC)             \
  X(INS_CG_BATCH)                \
  X(INS_F32_CORR_ROWS)           \
  X(INS_F32_FP64_SPECTRA)        \
  X(INS_F32_HIPFFT_PROJECT)      \
  X(INS_F32_SPLIT_GRADIENT)      \
  X(INS_FFT_ALLOW_RESET)
#define INS_OPT_ENUM(id) OPT_##id,
enum InsOptId { INS_OPT_LIST(INS_OPT_ENUM) INS_OPT_COUNT };
#undef INS_OPT_ENUM
long long ins_opt(int id);  // current value (0 = off / default)
long long ins_opt_epoch();  // bumped by every ins_set_option: cached launch plans (step graphs) are rebuilt when it moves

// ------------------------------------------------------------------------------------------------
// Device view of the grid, passed to kernels by value (lives in the kernarg segment -> SGPR loads).
// Metric tables are tiny 1-D device vectors (<= 4 KB each), L1/L2/K$-resident.
//   rdx  = 1/Δ      rdxu = 1/Δu        (reciprocal tables: the reference's 27 fp64 divisions per cell
//   mdx  = Δ  > 2eps ? 1/Δ  : 0         become multiplies; <= 1 ulp per term, inside the 1e-12 tolerance)
//   mdxu = Δu > 2eps ? 1/Δu : 0        (the `(Δ > 2eps) * ∂` strong-zero masks of operators.jl:683-684)
// ------------------------------------------------------------------------------------------------
struct GridDev {
  int D;
  int N[3];
  long long sx[3];  // element strides of directions 0..2
  long long sc;     // component stride = prod(N)
  const double* dx[3];
  const double* dxu[3];
  const double* rdx[3];
  const double* rdxu[3];
  const double* mdx[3];
  const double* mdxu[3];
  const double* A1[3][3];
  const double* A2[3][3];
  int iu_lo[3][3], iu_hi[3][3];
  int ip_lo[3], ip_hi[3];
  int bc[3][2];
  double bc_u[3][2][3];
};

struct ins_grid {
  ins_grid_desc_t desc;  // host copy (metric pointers below are re-pointed to `host`)
  std::vector<double> host;
  double* dev = nullptr;  // one device slab holding every table
  size_t dev_count = 0;
  GridDev g;
  bool all_periodic = false;
  bool uniform = false;
  double h[3] = {0, 0, 0};  // Δx[α] of the first volume (uniform grids)
  long long ncell = 0;      // prod(N)
  bool all_dof = false;        // every interior volume is a DOF of every component (all-periodic)
  bool uniform_exact = false;  // all metric records bitwise identical over the used index range
  void* rec_dev = nullptr;     // flux-kernel metric records (ins_fast3d_flux.hip), built lazily
  double rec_visc = -1.0;
  void* rec_diff_dev = nullptr;  // the same records with zero interpolation weights: the flux kernels then evaluate diffusion!(F, u) alone
  double rec_diff_visc = -1.0;
  // scratch for blocking reductions
  double* red_dev = nullptr;
  double* red_host = nullptr;  // pinned
};

enum PoissonKind { POISSON_SPECTRAL = 0, POISSON_CG = 1, POISSON_FDM = 2 };

// fast-diagonalisation direct solver (ins_fdm.hip)
struct ins_fdm;
int ins_fdm_create(int D, const int n[3], const double* const V[3], const double* const lam[3], int singular, ins_fdm** out);
int ins_fdm_destroy(ins_fdm* F);
int ins_fdm_enable_zfft(ins_fdm* F, double hz, const double* lam_z_host);
int ins_fdm_enable_zdct(ins_fdm* F, double hz, const double* lam_z_host);
int ins_fdm_enable_xfft(ins_fdm* F, double hx, const double* lam_x_host);
int ins_fdm_enable_xyfft(ins_fdm* F, double hx, double hy, const double* lam_x_host, const double* lam_y_host);
int ins_fdm_solve(ins_fdm* F, hipStream_t s, const ins_grid* G = nullptr, const double* u = nullptr, bool folded_io = false);
int ins_fdm_fold_mask(const ins_fdm* F);
bool ins_fdm_takes_u(const ins_fdm* F);
double* ins_fdm_buffer(ins_fdm* F);
const double* ins_fdm_mean(ins_fdm* F);  // device scalar the consumer subtracts (singular systems), or nullptr

struct ins_poisson {
  PoissonKind kind;
  const ins_grid* grid;
  // spectral
  hipfftHandle plan_fwd = 0, plan_inv = 0;
  bool plans = false;
  double* pI = nullptr;            // real n^D
  hipfftDoubleComplex* phat = nullptr;  // (n/2+1) n [n]
  double* ahat[3] = {nullptr, nullptr, nullptr};
  int np[3] = {1, 1, 1};
  int kmax[3] = {1, 1, 1};
  void* work = nullptr;
  size_t work_bytes = 0;
  hipStream_t plan_stream = nullptr;
  bool zfused = false;      // 3-D: batched 2-D (x,y) plans + the fused z kernel (ins_zsolve.hip)
  bool ownfft = false;      // 3-D power-of-two box: all five passes are own LDS kernels (ins_fft.hip), no rocFFT
  bool y3 = false;          // ownfft, 3-D: the y passes run on the register passes (k_line3, ins_zsolve.hip); ahat[1] is in THEIR storage order
  int kxs = 0;              // ownfft: row stride of phat (kmax[0] rounded up to 8 complex = 128 B)
  double* tw = nullptr;     // z twiddles
  double* tw_x = nullptr;   // x / y twiddles (ownfft)
  double* tw_y = nullptr;
  int yz_P = 0;             // > 0: the z direction rides on the y passes (ins_fft.hip k_yz_*: four passes per solve), yz_P partitions
  double* yz_scratch = nullptr;
  // cg
  double abstol = 0, reltol = 0;
  long long maxiter = 0;
  double *r = nullptr, *L = nullptr, *q = nullptr, *dinv = nullptr;
  bool bordered = false;
  bool singular = true;  // no PressureBC side
  long long ndof = 0;
  long long last_iter = 0;
  double last_res = 0;
  double* cg_scal = nullptr;  // device scalars + block partials of the device-resident iteration
  double* cg_host = nullptr;  // pinned mirror of the scalars
  struct ins_comm* comm = nullptr;  // z-slab CG: scalar all-reduces + ghost planes of q (ins_poisson_cg_set_comm)
  // fdm (psolver_direct)
  ins_fdm* fdm = nullptr;
};

struct ins_rk_ext;  // temperature equation / closure state of the extended stage loop (ins_rk_ext.hip)
void ins_rk_ext_free(ins_rk_ext* e);
struct ins_rk {
  const ins_grid* grid;
  ins_poisson* ps;
  int nstage;
  std::vector<double> A, c;
  double* ustart = nullptr;
  std::vector<double*> ku;
  double* p = nullptr;
  double* ub[2] = {nullptr, nullptr};  // ping-pong stage velocities of the fused path
  std::vector<double*> vb;        // all uncorrected stage velocities V_0..V_{s-2} (stage-velocity basis, ins_rk.hip)
  const double* force = nullptr;  // steady body force field (caller-owned), ins_rk_set_bodyforce
  ins_rk_ext* ext = nullptr;
  bool profiling = false;
  std::vector<hipEvent_t> prof_events;  // (start, stop) pairs around momentum launches
  void* step_graph = nullptr;           // launch-bound boxes: one step of ins_rk_steps_f64 captured as a hipGraph (ins_rk.hip)
};

// Metric records of the flux-form stage kernels on stretched / masked grids (ins_fast3d_flux.hip builds them per viscosity: ins_flux3d_prepare;
// ins_flux64m.hip reads the same tables).  One record per (direction d, index idx); 16 doubles = 128 B so a record is one aligned scalar burst.
//   vs = ν·mdx[d][idx+1]   diffusion coefficient of the upper d-face for the d-component   (Δb, α == β)
//   vo = ν·mdxu[d][idx]    ... for the other components                                     (Δb, α != β)
//   a[β], b[β] = ½A₂[β][d][idx], ½A₁[β][d][idx+1]   half weights of component β read along d
//   rs = 1/Δu[d][idx], ro = 1/Δ[d][idx]              control-volume width reciprocals (α == β / α != β)
struct Rec {
  double vs, vo, a0, b0, a1, b1, a2, b2, rs, ro, pad[6];
};

// Runge-Kutta stage epilogue fused behind the stencil (K1 + K6): with f = momentum(u) still in registers,
//   u* = ustart + Σ_j (Δt A[i,j]) k_j + (Δt A[i,i]) f          (step_explicit_runge_kutta.jl:35-38, same order)
// is written in the same pass, and k_i = f is stored only when a later stage needs it.
struct RkEpi {
  int n;                    // previous-stage terms with non-zero coefficient
  int write_k;              // store k_i
  double coef[INS_MAX_STAGES + 1];  // + 1: the steady body force is one more term (ins_rk_set_bodyforce)
  const double* k[INS_MAX_STAGES + 1];
  double coef_self;         // Δt A[i,i]
  double self_in;           // coefficient of the stencil input itself (its uncorrected value, taken from registers: ins_flux64.hip)
  double c0m1;              // ustart enters as (1 + c0m1)·ustart; 0 in the k-basis, -Σ coef in the stage-velocity basis (ins_rk.hip)
  const double* ustart;     // nullptr: ustart is the stencil input itself (first stage)
  double* ustar;            // stage velocity out (interior volumes only)
  double* ustart_out;       // optional (first stage of a chained step, ustart == nullptr): the corrected stencil input is stored here
  const double* extra;      // optional vector field added to the stage force before it is used and stored (closure term: ins_rk_ext.hip)
  const double* gtemp;      // optional temperature field: gravity! is added to component gdir of the stage force (ga2 = α2)
  double ga2;
  int gdir;
  double* wout;             // optional: w_α = u_α · diffusion(u)_α of the stencil input is stored here (dissipation!, ins_rk_ext.hip)
  const struct TempEpi* tstage;  // optional (HOST pointer, read at launch): the temperature equation's stage inside the stage kernel
};

// One stage of the temperature equation carried by the 64-wide stage kernel (ins_flux64.hip, EXTRA; step_explicit_runge_kutta.jl:23-27, 39-44):
//   ktemp_i = convection_diffusion_temp(u, temp) + dissipation(u),   temp_out = tempstart + Σ_j coef_j k_j + c_self ktemp_i
struct TempEpi {
  const double* temp;       // T_i, padded, ghost volumes valid
  const double* tempstart;
  double* temp_out;         // T_{i+1}: another array than temp
  double* ktemp_out;        // nullable
  int n;
  double coef[INS_MAX_STAGES];
  const double* k[INS_MAX_STAGES];
  double c_self, a4, dcoef;  // Δt A[i,i];  α4;  Re·α1/γ (0: no dissipation term)
};


// ------------------------------------------------------------------------------------------------
// Internal launchers (stream-ordered, non-blocking) used across translation units.
// ------------------------------------------------------------------------------------------------
int ins_k_apply_bc_u(const ins_grid* grid, double* u, int dudt, const double* const* planes, hipStream_t s);
int ins_k_apply_bc_p(const ins_grid* grid, double* p, hipStream_t s);
int ins_k_apply_bc_p_fields(const ins_grid* grid, double* p, int nf, hipStream_t s);
int ins_k_momentum(const ins_grid* grid, double visc, const double* u, double* F, hipStream_t s);
int ins_k_divergence(const ins_grid* grid, const double* u, double* div, hipStream_t s);
int ins_k_diffusion_overwrite(const ins_grid* grid, double visc, const double* u, double* F, hipStream_t s);
int ins_k_scalewithvolume(const ins_grid* grid, double* p, hipStream_t s);
int ins_k_applypressure(const ins_grid* grid, double* u, const double* p, hipStream_t s);
int ins_k_laplacian(const ins_grid* grid, const double* p, double* L, hipStream_t s);
int ins_k_project(const ins_grid* grid, ins_poisson* ps, double* u, double* p, hipStream_t s);
int ins_k_momentum_rk_fused(const ins_grid* G, double visc, const double* u_in, double* k_out, const RkEpi& epi, hipStream_t s);
bool ins_flux2d_supported(const ins_grid* G);
int ins_k_flux2d(const ins_grid* G, double visc, const double* u, double* F, const RkEpi* epi, hipStream_t s, const double* pI = nullptr);
int ins_k_project_periodic_solve_only_2d(const ins_grid* G, ins_poisson* ps, const double* u, hipStream_t s);
int ins_k_momentum_rk_fused_generic(const ins_grid* G, double visc, const double* u_in, double* k_out, const RkEpi& epi, hipStream_t s);
int ins_k_momentum_rk_fused_corr(const ins_grid* G, double visc, const double* ustar_prev, const double* pI, double* k_out, const RkEpi& epi,
                                 hipStream_t s);
int ins_k_momentum_rk_fused_corr_slab(const ins_grid* G, double visc, const double* ustar_prev, const double* p_ext, double* k_out,
                                      const RkEpi& epi, hipStream_t s, int part = 0);
bool ins_corr3_supported(const ins_grid* G);
int ins_k_momentum_rk_fused_corr3(const ins_grid* G, double visc, const double* ustar_prev, const double* p_padded, double* k_out, const RkEpi& epi,
                                  hipStream_t s);
bool ins_k_project_fdm_fused(const ins_poisson* ps);
int ins_k_project_fdm_solve_only(const ins_grid* G, ins_poisson* ps, const double* u, double* p, hipStream_t s);
int ins_k_project_periodic_fused_2d(const ins_grid* G, ins_poisson* ps, double* u, double* p, bool keep_p, hipStream_t s);
bool ins_poisson_own2d(const ins_poisson* ps);
int ins_k_project_periodic_solve_only(const ins_grid* G, ins_poisson* ps, const double* u, hipStream_t s);
int ins_k_poisson_solve(ins_poisson* ps, double* p, hipStream_t s);
// blocking reductions over an index box of a scalar field; op: 0 sum(a*b), 1 max|a|, 2 min(a)
int ins_k_reduce(const ins_grid* grid, int op, const double* a, const double* b, const int lo[3], const int hi[3], double* out,
                 hipStream_t s);

int ins_validate_real_plans(hipfftHandle fwd, hipfftHandle inv, int rank, const int* n, int batch);
int ins_fft_make_real_plans(hipfftHandle* fwd, hipfftHandle* inv, int rank, int* n, int batch);
void ins_fft_solver_released();
bool ins_zsolve_supported(int nz);
bool ins_ownfft_supported(const int np[3]);
bool ins_ownfft_supported_slab(const int np[3]);
bool ins_ownfft_supported_mixed(const int np[3]);
void ins_ownfft_permute_symbol(int n, const double* ay, double* out);
// kxs: row stride of phat in complex elements (0 = dense, n0/2+1); a multiple of 8 keeps the y/z tiles 128-B aligned
// n2 planes starting at interior plane kz0
int ins_k_ownfft_xfwd(const ins_grid* G, const double* src, int from_u, double* phat, int n0, int n1, int n2, const double* tw, hipStream_t s,
                      int kxs = 0, int kz0 = 0);
int ins_k_ownfft_xinv(const double* phat, double* pI, int n0, int n1, int n2, const double* tw, hipStream_t s, int kxs = 0);
int ins_k_ownfft_y(double* phat, int kxn, int n1, int n2, const double* tw, bool inverse, hipStream_t s, int kxs = 0);
bool ins_ownfft_xy_supported(int n0, int n1);
int ins_k_ownfft_xy(const ins_grid* G, const double* src, int from_u, double* phat, double* pI, int n0, int n1, int n2, const double* twx, const double* twy,
                    bool inverse, hipStream_t s, int kxs);
int ins_k_ownfft_xysolve2d(const ins_grid* G, const double* src, int from_u, double* pI, int n0, int n1, const double* twx, const double* twy, const double* ax,
                           const double* ay, hipStream_t s);
bool ins_line3_supported(int n);
void ins_line3_permute_symbol(int n, const double* ay, double* out);
int ins_line3_pos_of_freq(int n, int k);
int ins_k_line3_z(double* phat, int kxn, int n1, int n2, const double* tw, hipStream_t s, int kxs);
int ins_k_line3_y(double* phat, int kxn, int n1, int n2, const double* tw, bool inverse, hipStream_t s, int kxs);
int ins_k_line3_y_f32(float* phat, int kxn, int n1, int n2, const float* tw, bool inverse, hipStream_t s, int kxs);
int ins_ownfft_yz_partitions(int kxn, int n1, int n2);
long long ins_ownfft_yz_scratch(int kxn, int n1, int n2, int P);
int ins_k_ownfft_yz_solve(double* phat, int kxn, int n1, int n2, int kxs, int P, const double* ax, const double* ay, double c, double scale, const double* tw_y,
                          double* scratch, hipStream_t s);
int ins_k_ownfft_y_packed(double* phat, double* packed, int kxn, int n1, int nzl, int nyl, int cw, const double* tw, bool inverse,
                          hipStream_t s);
int ins_zsolve_twiddles(int nz, double** out);
int ins_k_fdm_z(double* data, int n0, int n1, int nz, const double* lx, const double* ly, const double* lz, const double* ox, const double* oy,
                double h, double tol, int singular, const double* meanf, double* partial, const double* tw, int* nblk, hipStream_t s, const double* dct_w = nullptr);
int ins_k_zsolve(double* data, int nz, long long nl, const double* ax, int kxn, const double* ay, const double* az, const double* tw,
                 double inv_n, bool zero_mean, hipStream_t s, int kxs = 0);

// communication used inside the library (ins_comm.hip): op 0 sum, 1 max, 2 min; scalar-field z ghost planes on a slab grid
int ins_comm_allreduce_internal(struct ins_comm* c, double* buf, long long count, int op, hipStream_t s);
int ins_comm_halo_scalar_internal(struct ins_comm* c, const ins_grid* G, double* p, hipStream_t s);

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline unsigned cdiv(long long a, long long b) { return (unsigned)((a + b - 1) / b); }
