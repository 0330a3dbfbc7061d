"""ctypes binding of libinship.so (the C ABI declared in include/ins_hip.h).

There is NO fallback: if the HIP library is missing or fails to load, importing the operators raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("INS_HIP_LIB") or os.path.join(_HERE, "libinship.so")  # INS_HIP_LIB: A/B builds (tools/)

INS_BC_PERIODIC, INS_BC_DIRICHLET, INS_BC_SYMMETRIC, INS_BC_PRESSURE, INS_BC_HALO = range(5)

c_double_p = C.POINTER(C.c_double)
vp = C.c_void_p


class GridDesc(C.Structure):
    """`ins_grid_desc_t` (include/ins_hip.h)."""

    _fields_ = [
        ("D", C.c_int32),
        ("N", C.c_int32 * 3),
        ("dx", c_double_p * 3),
        ("dxu", c_double_p * 3),
        ("A1", (c_double_p * 3) * 3),
        ("A2", (c_double_p * 3) * 3),
        ("iu_lo", (C.c_int32 * 3) * 3),
        ("iu_hi", (C.c_int32 * 3) * 3),
        ("ip_lo", C.c_int32 * 3),
        ("ip_hi", C.c_int32 * 3),
        ("bc", (C.c_int32 * 2) * 3),
        ("bc_u", ((C.c_double * 3) * 2) * 3),
    ]


# name -> (restype, argtypes); must list EVERY symbol include/ins_hip.h declares (tests/test_abi.py checks)
SIGNATURES = {
    "ins_version": (C.c_int, []),
    "ins_fft_reset_count": (C.c_int, []),
    "ins_last_error": (C.c_char_p, []),
    "ins_set_option": (C.c_int, [C.c_char_p, C.c_int64]),
    "ins_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_int64)]),
    "ins_option_count": (C.c_int, []),
    "ins_option_name": (C.c_char_p, [C.c_int]),
    "ins_set_device": (C.c_int, [C.c_int]),
    "ins_sync": (C.c_int, [vp]),
    "ins_grid_create": (C.c_int, [C.POINTER(GridDesc), C.POINTER(vp)]),
    "ins_grid_destroy": (C.c_int, [vp]),
    "ins_grid_is_uniform_exact": (C.c_int, [vp]),
    "ins_apply_bc_u_f64": (C.c_int, [vp, vp, C.c_int, C.POINTER(vp), vp]),
    "ins_apply_bc_p_f64": (C.c_int, [vp, vp, vp]),
    "ins_scalewithvolume_f64": (C.c_int, [vp, vp, vp]),
    "ins_divergence_f64": (C.c_int, [vp, vp, vp, vp]),
    "ins_pressuregradient_f64": (C.c_int, [vp, vp, vp, vp]),
    "ins_applypressure_f64": (C.c_int, [vp, vp, vp, vp]),
    "ins_laplacian_f64": (C.c_int, [vp, vp, vp, vp]),
    "ins_convection_f64": (C.c_int, [vp, vp, vp, vp]),
    "ins_diffusion_f64": (C.c_int, [vp, C.c_double, vp, vp, vp]),
    "ins_convectiondiffusion_f64": (C.c_int, [vp, C.c_double, vp, vp, vp]),
    "ins_momentum_f64": (C.c_int, [vp, C.c_double, vp, vp, vp]),
    "ins_kinetic_energy_f64": (C.c_int, [vp, vp, vp, C.c_int, vp]),
    "ins_total_kinetic_energy_f64": (C.c_int, [vp, vp, C.c_int, c_double_p, vp]),
    "ins_cfl_timestep_f64": (C.c_int, [vp, C.c_double, vp, c_double_p, vp]),
    "ins_max_abs_divergence_f64": (C.c_int, [vp, vp, c_double_p, vp]),
    "ins_vorticity_f64": (C.c_int, [vp, vp, vp, vp]),
    "ins_interpolate_u_p_f64": (C.c_int, [vp, vp, vp, vp]),
    "ins_interpolate_w_p_f64": (C.c_int, [vp, vp, vp, vp]),
    "ins_dfield_f64": (C.c_int, [vp, vp, vp, vp, C.c_double, vp]),
    "ins_qfield_f64": (C.c_int, [vp, vp, vp, vp]),
    "ins_dissipation_from_strain_f64": (C.c_int, [vp, C.c_double, vp, vp, vp]),
    "ins_eig2field_f64": (C.c_int, [vp, vp, vp, vp]),
    "ins_apply_bc_temp_f64": (C.c_int, [vp, C.POINTER(C.c_int32), c_double_p, C.POINTER(vp), vp, vp]),
    "ins_convection_diffusion_temp_f64": (C.c_int, [vp, C.c_double, vp, vp, vp, vp]),
    "ins_dissipation_f64": (C.c_int, [vp, C.c_double, C.c_double, vp, vp, vp, vp]),
    "ins_gravity_f64": (C.c_int, [vp, C.c_int, C.c_double, vp, vp, vp]),
    "ins_smagtensor_f64": (C.c_int, [vp, C.c_double, vp, vp, vp]),
    "ins_divoftensor_f64": (C.c_int, [vp, vp, vp, vp]),
    "ins_smagorinsky_force_f64": (C.c_int, [vp, C.c_double, vp, vp, vp, vp]),
    "ins_smagorinsky_force_needs_sigma": (C.c_int, [vp]),
    "ins_tensorbasis_f64": (C.c_int, [vp, vp, vp, vp, vp]),
    "ins_combine_scalar_f64": (C.c_int, [vp, vp, vp, C.c_int, c_double_p, C.POINTER(vp), vp]),
    "ins_spectrum_create": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(vp)]),
    "ins_spectrum_create_weighted": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), c_double_p, C.POINTER(vp)]),
    "ins_spectrum_destroy": (C.c_int, [vp]),
    "ins_spectrum_f64": (C.c_int, [vp, vp, vp, vp]),
    "ins_poisson_spectral_create": (C.c_int, [vp, C.POINTER(vp)]),
    "ins_poisson_cg_create": (C.c_int, [vp, C.c_double, C.c_double, C.c_int64, C.POINTER(vp)]),
    "ins_poisson_cg_bordered": (C.c_int, [vp, C.c_int]),
    "ins_poisson_cg_set_comm": (C.c_int, [vp, vp]),
    "ins_poisson_fdm_create": (C.c_int, [vp, C.POINTER(c_double_p), C.POINTER(c_double_p), C.POINTER(vp)]),
    "ins_poisson_destroy": (C.c_int, [vp]),
    "ins_poisson_fft_engine": (C.c_int, [vp, C.POINTER(C.c_int32)]),
    "ins_poisson_yz_partitions": (C.c_int, [vp, C.POINTER(C.c_int32)]),
    "ins_poisson_solve_f64": (C.c_int, [vp, vp, vp]),
    "ins_poisson_last_info": (C.c_int, [vp, C.POINTER(C.c_int64), c_double_p]),
    "ins_project_f64": (C.c_int, [vp, vp, vp, vp, vp]),
    "ins_rk_create": (C.c_int, [vp, vp, C.c_int, c_double_p, c_double_p, C.POINTER(vp)]),
    "ins_rk_destroy": (C.c_int, [vp]),
    "ins_rk_step_f64": (C.c_int, [vp, C.c_double, vp, C.c_double, C.c_double, C.POINTER(vp), vp]),
    "ins_rk_step_bc_f64": (C.c_int, [vp, C.c_double, vp, C.c_double, C.c_double, C.POINTER(vp), vp]),
    "ins_rk_steps_f64": (C.c_int, [vp, C.c_double, vp, C.c_double, C.c_double, C.c_int, vp]),
    "ins_combine_f64": (C.c_int, [vp, vp, vp, C.c_int, c_double_p, C.POINTER(vp), vp]),
    "ins_rk_profile_enable": (C.c_int, [vp, C.c_int]),
    "ins_rk_profile_read": (C.c_int, [vp, c_double_p, C.POINTER(C.c_int64)]),
    "ins_rk_set_bodyforce": (C.c_int, [vp, vp]),
    "ins_rk_pressure": (C.c_int, [vp, C.POINTER(vp)]),
    "ins_rk_stage_force": (C.c_int, [vp, C.c_int, C.POINTER(vp)]),
    "ins_rk_set_temperature": (C.c_int, [vp, vp]),
    "ins_rk_set_closure": (C.c_int, [vp, C.c_int32, C.c_double]),
    "ins_rk_step_ext_f64": (C.c_int, [vp, C.c_double, vp, vp, C.c_double, C.c_double, vp]),
    "ins_stage_momentum_f64": (C.c_int, [vp, C.c_double, vp, vp, vp, vp, C.c_int, c_double_p, C.POINTER(vp), C.c_double, vp]),
    "ins_stage_momentum_corr_f64": (C.c_int, [vp, C.c_double, vp, vp, vp, vp, vp, C.c_int, c_double_p, C.POINTER(vp), C.c_double, vp]),
    "ins_stage_momentum_corr_part_f64": (C.c_int, [vp, C.c_double, vp, vp, vp, vp, vp, C.c_int, c_double_p, C.POINTER(vp), C.c_double, C.c_double, C.c_double, vp, C.c_int, vp]),
    "ins_slab_divergence_f64": (C.c_int, [vp, vp, vp, vp]),
    "ins_slab_applypressure_f64": (C.c_int, [vp, vp, vp, vp, vp]),
    "ins_slab_fft_create": (C.c_int, [C.POINTER(C.c_int32), c_double_p, C.c_int, C.c_int, C.POINTER(vp)]),
    "ins_slab_fft_destroy": (C.c_int, [vp]),
    "ins_slab_fft_sizes": (C.c_int, [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "ins_slab_fft_forward_xy": (C.c_int, [vp, vp, vp, vp, vp]),
    "ins_slab_fft_solve_z": (C.c_int, [vp, vp, vp]),
    "ins_slab_fft_inverse_xy": (C.c_int, [vp, vp, vp, vp, vp]),
    "ins_slab_fft_can_chunk": (C.c_int, [vp]),
    "ins_slab_fft_xy_forward_only": (C.c_int, [vp, vp, vp, vp]),
    "ins_slab_fft_pack_chunk": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp]),
    "ins_slab_fft_solve_z_chunk": (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
    "ins_slab_fft_unpack_chunk": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp]),
    "ins_slab_fft_xy_inverse_only": (C.c_int, [vp, vp, vp, vp]),
    "ins_slab_fft_is_own": (C.c_int, [vp]),
    "ins_slab_ztri_edge_elems": (C.c_int, [vp, C.POINTER(C.c_int64)]),
    "ins_slab_ztri_forward": (C.c_int, [vp, vp, vp, C.c_int, vp, vp, vp]),
    "ins_slab_ztri_finish": (C.c_int, [vp, vp, vp, vp, vp]),
    "ins_slab_xfwd_planes": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, vp]),
    "ins_slab_ztri_chunk": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "ins_slab_ztri_transform": (C.c_int, [vp, vp, vp, C.c_int, vp, vp]),
    "ins_slab_ztri_sweep_forward": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp]),
    "ins_slab_ztri_sweep_backward": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp]),
    "ins_slab_ztri_inverse": (C.c_int, [vp, vp, vp, vp]),
    "ins_slab_fft_forward_packed": (C.c_int, [vp, vp, vp, C.c_int, vp, vp, C.c_int, vp]),
    "ins_slab_fft_inverse_packed": (C.c_int, [vp, vp, vp, vp, C.c_int, vp]),
    "ins_apply_bc_u_f32": (C.c_int, [vp, vp, vp]),
    "ins_apply_bc_p_f32": (C.c_int, [vp, vp, vp]),
    "ins_momentum_f32": (C.c_int, [vp, C.c_float, vp, vp, vp]),
    "ins_poisson_spectral_create_f32": (C.c_int, [vp, C.POINTER(vp)]),
    "ins_poisson_wrap_f32": (C.c_int, [vp, vp, C.POINTER(vp)]),
    "ins_poisson_destroy_f32": (C.c_int, [vp]),
    "ins_poisson_solve_f32": (C.c_int, [vp, vp, vp]),
    "ins_project_f32": (C.c_int, [vp, vp, vp, vp, vp]),
    "ins_rk_create_f32": (C.c_int, [vp, vp, C.c_int, c_double_p, c_double_p, C.POINTER(vp)]),
    "ins_rk_destroy_f32": (C.c_int, [vp]),
    "ins_rk_step_f32": (C.c_int, [vp, C.c_float, vp, C.c_float, vp]),
    "ins_rk_steps_f32": (C.c_int, [vp, C.c_float, vp, C.c_float, C.c_int, vp]),
    "ins_max_abs_divergence_f32": (C.c_int, [vp, vp, vp, C.POINTER(C.c_float), vp]),
    "ins_comm_unique_id": (C.c_int, [vp]),
    "ins_comm_create": (C.c_int, [C.c_int, C.c_int, vp, C.POINTER(vp)]),
    "ins_comm_create_local": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(vp)]),
    "ins_comm_group_begin": (C.c_int, []),
    "ins_comm_group_end": (C.c_int, []),
    "ins_comm_destroy": (C.c_int, [vp]),
    "ins_comm_rank": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ins_comm_sendrecv_f64": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.c_int, C.POINTER(vp),
                                        C.POINTER(C.c_int64), C.POINTER(C.c_int32), vp]),
    "ins_halo_exchange_f64": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp]),
    "ins_halo_exchange_scalar_f64": (C.c_int, [vp, vp, vp, vp]),
    "ins_halo_exchange_p_f64": (C.c_int, [vp, vp, C.c_int64, C.c_int, vp]),
    "ins_ztri_allgather_f64": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int, vp]),
    "ins_comm_allreduce_f64": (C.c_int, [vp, vp, C.c_int64, C.c_int, vp]),
    "ins_comm_alltoall_f64": (C.c_int, [vp, vp, vp, C.c_int64, vp]),
}

_lib = None


class INSHipError(RuntimeError):
    pass


def load():
    """Load libinship.so (once).  Raises loudly if it is absent — the product has no CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise INSHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C incompressiblenavierstokes.jl_amd/csrc`).  There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().ins_last_error()
        raise INSHipError(f"libinship error {rc}: {msg.decode() if msg else ''}")


def call(name, *args):
    check(getattr(load(), name)(*args))


def set_option(name, value):
    """`ins_set_option`: flip a run-time switch of the library (same names as the environment variables)."""
    call("ins_set_option", name.encode(), int(value))


def get_option(name):
    v = C.c_int64()
    call("ins_get_option", name.encode(), C.byref(v))
    return int(v.value)


class options:
    """Context manager: `with options(INS_DISABLE_FUSED_RK=1): ...` — set switches, restore the previous values on exit."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: get_option(k) for k in self.kw}
        for k, v in self.kw.items():
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)
        return False


_fft_resets_seen = 0


def sync_fft_plan_caches():
    """The library resets rocFFT's process-wide state when it catches the plan-cache defect (csrc/ins_fftcheck.hip); plans that PyTorch
    cached for torch.fft are then stale.  Called after every solver / observer creation."""
    global _fft_resets_seen
    n = load().ins_fft_reset_count()
    if n != _fft_resets_seen:
        _fft_resets_seen = n
        import torch

        torch.cuda.synchronize()
        torch.backends.cuda.cufft_plan_cache.clear()
