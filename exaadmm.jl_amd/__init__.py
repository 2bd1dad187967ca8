"""exaadmm.jl_amd -- MI355X-native backend for ExaAdmm.jl's two-level ADMM ACOPF inner loop.

Host-side mirror, over the C ABI of libexaadmm_hip.so (include/exaadmm_hip.h), of the
reference's plugin surface for this path, same names and argument meaning:

    solve_acopf(case; kwargs) -> (env, mod)        /root/reference/src/interface/solve_acopf.jl:1-46
    admm_two_level(env, mod)                       /root/reference/src/algorithms/admm_two_level.jl:1-88
    AdmmEnv, ModelAcopf, Parameters, mod.info      /root/reference/src/utils/environment.jl
    admm_increment_outer ... admm_poststep         /root/reference/docs/src/dev.md:65-155

The reference's host language is Julia, which does not exist in the build image; the Julia
binding a maintainer would add is shipped as source in julia/ExaAdmmHip.jl.
All compute goes through the HIP library; if it is missing or no gfx950 device is visible the
calls raise -- there is no CPU path in this package.

The directory name contains a dot, so import it with `load_package()` from the repo-root
helper `exaadmm_loader.py` (or importlib, as that helper does).
"""
import ctypes as C
import math
import os
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EXAADMM_HIP_LIB", os.path.join(_HERE, "libexaadmm_hip.so"))

c_double_p = C.POINTER(C.c_double)
c_i64_p = C.POINTER(C.c_int64)


class ExaAdmmError(RuntimeError):
    pass


class Grid(C.Structure):
    """exaadmm_grid"""
    _F64 = ("pgmin", "pgmax", "qgmin", "qgmax", "c2", "c1", "c0",
            "YffR", "YffI", "YftR", "YftI", "YttR", "YttI", "YtfR", "YtfI",
            "FrVmBound", "ToVmBound", "FrVaBound", "ToVaBound", "rateA")
    _BUS = ("Pd", "Qd", "Vmin", "Vmax", "YshR", "YshI")
    _IDX = ("FrStart", "FrIdx", "ToStart", "ToIdx", "GenStart", "GenIdx")
    _fields_ = ([("ngen", C.c_int64), ("nline", C.c_int64), ("nbus", C.c_int64),
                 ("index_base", C.c_int32), ("baseMVA", C.c_double)] +
                [(n, c_double_p) for n in _F64] + [("brBusIdx", c_i64_p)] +
                [(n, c_double_p) for n in _BUS] + [(n, c_i64_p) for n in _IDX] +
                [("nline_owned", C.c_int64)])


class Parameters(C.Structure):
    """exaadmm_params == Parameters of the reference (utils/environment.jl:6-76)"""
    _fields_ = [("mu_max", C.c_double), ("max_auglag", C.c_int32), ("initial_beta", C.c_double),
                ("beta", C.c_double), ("inc_c", C.c_double), ("theta", C.c_double),
                ("outer_eps", C.c_double), ("MAX_MULTIPLIER", C.c_double),
                ("outer_iterlim", C.c_int32), ("inner_iterlim", C.c_int32), ("scale", C.c_double),
                ("verbose", C.c_int32), ("check_every", C.c_int32)]


class Info(C.Structure):
    """exaadmm_info == IterationInformation (utils/environment.jl:328-358)"""
    _fields_ = [("status_code", C.c_int32), ("inner", C.c_int32), ("outer", C.c_int32),
                ("cumul", C.c_int32), ("objval", C.c_double), ("primres", C.c_double),
                ("dualres", C.c_double), ("mismatch", C.c_double), ("norm_z_curr", C.c_double),
                ("norm_z_prev", C.c_double), ("eps_pri", C.c_double), ("time_overall", C.c_double),
                ("time_x_update", C.c_double), ("time_xbar_update", C.c_double),
                ("time_zl_update", C.c_double), ("avg_auglag_it", C.c_double),
                ("avg_minor_it", C.c_double)]

    @property
    def status(self):
        return "Solved" if self.status_code == 1 else "IterationLimit"


OWN_STREAM = C.c_void_p(-1).value   # EXAADMM_OWN_STREAM: (void *)~0

VECTORS = {"u_curr": 0, "v_curr": 1, "l_curr": 2, "rho": 3, "rd": 4, "rp": 5, "z_curr": 6,
           "z_prev": 7, "lz": 8, "Ax_plus_By": 9}

_LIB = None


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch wheels for ROCm carry a PRIVATE libamdhip64 / libhsa-runtime64 / librccl
    (torch/lib); libexaadmm_hip.so is linked against the system one (/opt/rocm/lib).  A process that loads both ends up
    with two HIP runtimes, and starting the second one fails now and then ("no HIP device visible" / "No HIP GPUs are
    available", observed in both orders on MI355X boxes).  If PyTorch is installed, its runtime is therefore mapped
    FIRST and globally, without importing torch: the library's hip* symbols then bind to it, `import torch` later finds
    it already loaded, and RCCL (dist_native.h) is taken from the same directory.  Hosts without PyTorch (a Julia or
    plain-C driver) use the system runtime.  EXAADMM_NO_TORCH_RUNTIME=1 disables the preload."""
    if os.environ.get("EXAADMM_NO_TORCH_RUNTIME"):
        return None
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return None
        path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if not os.path.exists(path):
            return None
        C.CDLL(path, mode=C.RTLD_GLOBAL)
        return path
    except Exception:
        return None


def bound_hip_runtime():
    """path and version of the HIP runtime the loaded library is bound to, and the HIP version it was compiled against"""
    buf = C.create_string_buffer(1024)
    library().exaadmm_hip_runtime_info(buf, 1024)
    return buf.value.decode()


def library():
    """Load libexaadmm_hip.so.  Raises if it has not been built: no fallback."""
    global _LIB
    if _LIB is not None:
        return _LIB
    _preload_torch_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise ExaAdmmError("HIP extension %s is missing; run __graft_entry__.build() "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    H = C.c_void_p
    L.exaadmm_version.restype = C.c_char_p
    ver = L.exaadmm_version().decode()
    if "-sgpr-regalloc=basic" not in ver:
        # ROCm 7.2 miscompiles the branch kernel with the default SGPR allocator (csrc/Makefile)
        raise ExaAdmmError("%s was not built through csrc/Makefile (version string %r lacks -sgpr-regalloc=basic): "
                           "its branch kernel cannot be trusted" % (LIB_PATH, ver))
    L.exaadmm_last_error.restype = C.c_char_p
    L.exaadmm_dist_rccl_path.restype = C.c_char_p
    # start the HIP runtime the library is bound to NOW, at a defined moment (with PyTorch installed that is torch's bundled
    # runtime, mapped above; otherwise the system one), and check that it is a runtime the library's code can talk to
    L.exaadmm_device_count.restype = C.c_int
    L.exaadmm_device_count()
    L.exaadmm_hip_runtime_info.argtypes = [C.c_char_p, C.c_int64]
    buf = C.create_string_buffer(1024)
    if L.exaadmm_hip_runtime_info(buf, 1024) != 0:
        raise ExaAdmmError("HIP runtime / library mismatch: %s; set EXAADMM_NO_TORCH_RUNTIME=1 to use the system runtime"
                           % buf.value.decode())
    L.exaadmm_io_error.restype = C.c_char_p
    L.exaadmm_default_params.argtypes = [C.POINTER(Parameters)]
    L.exaadmm_default_params.restype = None
    L.exaadmm_create.argtypes = [C.POINTER(Grid), C.POINTER(Parameters), C.c_int, C.POINTER(H)]
    L.exaadmm_destroy.argtypes = [H]
    L.exaadmm_set_params.argtypes = [H, C.POINTER(Parameters)]
    L.exaadmm_get_params.argtypes = [H, C.POINTER(Parameters)]
    L.exaadmm_nvar.argtypes = [H]
    L.exaadmm_nvar.restype = C.c_int64
    L.exaadmm_init_solution.argtypes = [H, C.c_double, C.c_double]
    L.exaadmm_upload_solution.argtypes = [H, C.c_int, c_double_p, C.c_int64]
    L.exaadmm_download_solution.argtypes = [H, C.c_int, c_double_p, C.c_int64]
    L.exaadmm_upload_membuf.argtypes = [H, c_double_p, C.c_int64, C.c_int64]
    L.exaadmm_download_membuf.argtypes = [H, c_double_p, C.c_int64, C.c_int64]
    L.exaadmm_set_load.argtypes = [H, c_double_p, c_double_p, C.c_int64]
    L.exaadmm_set_pg_bounds.argtypes = [H, c_double_p, c_double_p, C.c_int64]
    L.exaadmm_inner_prestep.argtypes = [H]
    L.exaadmm_update_x.argtypes = [H, C.c_int]
    L.exaadmm_update_x_gen.argtypes = [H]
    L.exaadmm_update_x_line.argtypes = [H, C.c_int]
    L.exaadmm_update_xbar.argtypes = [H]
    L.exaadmm_update_z.argtypes = [H, C.c_double]
    L.exaadmm_update_l.argtypes = [H, C.c_double]
    L.exaadmm_update_residual.argtypes = [H, c_double_p]
    L.exaadmm_update_lz.argtypes = [H, C.c_double]
    L.exaadmm_norm_z.argtypes = [H, c_double_p]
    L.exaadmm_objective.argtypes = [H, c_double_p]
    L.exaadmm_branch_stats.argtypes = [H, c_double_p, c_double_p]
    L.exaadmm_download_branch_counters.argtypes = [H, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                                   C.POINTER(C.c_int32), C.c_int64]
    L.exaadmm_branch_kernel_occupancy.argtypes = [H, C.POINTER(C.c_int)]
    L.exaadmm_download_tile_trips.argtypes = [H, C.POINTER(C.c_int32), C.c_int64]
    L.exaadmm_loop_kind.argtypes = [H, C.POINTER(C.c_int)]
    L.exaadmm_loop_downgrades.argtypes = [H, C.POINTER(C.c_int)]
    L.exaadmm_resident_form.argtypes = [H, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.exaadmm_resident_selfcheck_stats.argtypes = [H, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.exaadmm_set_option.argtypes = [H, C.c_char_p, C.c_int64]
    L.exaadmm_get_option.argtypes = [H, C.c_char_p, C.POINTER(C.c_int64)]
    L.exaadmm_dist_set_rccl_library.argtypes = [C.c_char_p]
    if hasattr(L, "exaadmm_test_hook"):   # the test build only (libexaadmm_hip_hooks.so, -DEXA_TEST_HOOKS)
        L.exaadmm_test_hook.argtypes = [H, C.c_char_p, C.c_int64]
    L.exaadmm_run_inner.argtypes = [H, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int), c_double_p]
    L.exaadmm_solve_two_level.argtypes = [H, C.POINTER(Info)]
    L.exaadmm_run_fixed.argtypes = [H, C.c_int, C.c_int, C.c_double, c_double_p, c_double_p, c_double_p]
    L.exaadmm_state_save.argtypes = [H]
    L.exaadmm_state_restore.argtypes = [H]
    L.exaadmm_set_stream.argtypes = [H, C.c_void_p]
    L.exaadmm_set_async.argtypes = [H, C.c_int]
    L.exaadmm_set_halo_index.argtypes = [H, C.c_int, C.POINTER(C.c_int32), c_i64_p, C.c_int64]
    L.exaadmm_halo_pack.argtypes = [H, C.c_int, C.c_void_p]
    L.exaadmm_halo_unpack.argtypes = [H, C.c_int, C.c_void_p]
    L.exaadmm_update_zl_partial.argtypes = [H, C.c_double, C.c_void_p]
    L.exaadmm_sumsq_z_partial.argtypes = [H, C.c_void_p]
    L.exaadmm_ctrl_begin.argtypes = [H, C.c_int, C.c_int, C.c_double, C.c_int64]
    L.exaadmm_ctrl_step.argtypes = [H, C.c_void_p]
    L.exaadmm_ctrl_read.argtypes = [H, C.POINTER(C.c_int), C.POINTER(C.c_int), c_double_p]
    L.exaadmm_ctrl_end.argtypes = [H]
    L.exaadmm_dist_unique_id.argtypes = [C.c_char_p, C.c_int64]
    L.exaadmm_dist_prepare.argtypes = [H, C.c_int, C.c_int, C.c_int64, c_i64_p, c_i64_p]
    L.exaadmm_dist_connect.argtypes = [H, C.c_char_p, C.c_int64]
    L.exaadmm_dist_attach.argtypes = [H, C.c_int, C.c_int, C.c_char_p, C.c_int64, C.c_int64, c_i64_p, c_i64_p]
    L.exaadmm_dist_exchange_count.argtypes = []
    L.exaadmm_dist_exchange_count.restype = C.c_int64
    L.exaadmm_dist_detach.argtypes = [H]
    L.exaadmm_dist_p2p_export.argtypes = [H, C.c_char_p, C.c_int64]
    L.exaadmm_dist_p2p_import.argtypes = [H, C.c_char_p, C.c_int64]
    L.exaadmm_dist_selftest.argtypes = [H, C.c_int64, c_double_p]
    L.exaadmm_dist_capture_selftest.argtypes = [H, C.c_int64, C.POINTER(C.c_int)]
    L.exaadmm_dist_stats.argtypes = [H, C.POINTER(C.c_int), C.POINTER(C.c_int), c_double_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                     C.POINTER(C.c_int)]
    L.exaadmm_mp_create.argtypes = [C.POINTER(Grid), C.POINTER(Parameters), C.c_int, c_double_p, c_double_p,
                                    C.c_double, C.c_int, C.POINTER(H)]
    L.exaadmm_mp_periods.argtypes = [H]
    L.exaadmm_mp_nvar.argtypes = [H]
    L.exaadmm_mp_nvar.restype = C.c_int64
    L.exaadmm_mp_update_x_gen.argtypes = [H, C.c_int]
    L.exaadmm_mp_download_ramp.argtypes = [H, C.c_int, c_double_p, C.c_int64]
    L.exaadmm_mp_upload_ramp.argtypes = [H, C.c_int, c_double_p, C.c_int64]
    L.exaadmm_load_matpower.argtypes = [C.c_char_p, C.c_double, C.c_char_p, C.POINTER(C.c_void_p)]
    L.exaadmm_case_grid.argtypes = [C.c_void_p, C.POINTER(Grid)]
    L.exaadmm_case_free.argtypes = [C.c_void_p]
    _LIB = L
    # a site's own communication library / the tests' in-process stand-in (the library itself reads no environment variable)
    if os.environ.get("EXAADMM_RCCL_LIB"):
        _check_rc(L, L.exaadmm_dist_set_rccl_library(os.fsencode(os.environ["EXAADMM_RCCL_LIB"])), "exaadmm_dist_set_rccl_library")
    return L


def _check_rc(L, rc, what):
    if rc != 0:
        raise ExaAdmmError("%s failed (status %d): %s" % (what, rc, L.exaadmm_last_error().decode()))


# Options of a handle (include/exaadmm_hip.h, exaadmm_set_option) taken from the ENVIRONMENT -- a convenience of this host-side mirror
# for A/B runs of the tests, the benchmark and the scripts (`EXAADMM_NO_RESIDENT=1 python bench.py`); the library itself reads none of
# them.  (variable, option, value when the variable is set -- None: int(the variable's value)).  EXAADMM_OPT_<KEY>=<int> sets any option.
_ENV_OPTIONS = (("EXAADMM_BUS_KERNEL", "bus_kernel_lane", lambda v: 1 if v == "lane" else 0), ("EXAADMM_NO_GRAPH", "graph", 0),
                ("EXAADMM_NO_SPEC", "spec", 0), ("EXAADMM_MP_NO_SPEC", "spec", 0), ("EXAADMM_NO_REC", "rec", 0),
                ("EXAADMM_NO_RESIDENT", "resident", 0), ("EXAADMM_RESIDENT", "resident", None), ("EXAADMM_RESIDENT_TILES", "resident_tiles", None),
                ("EXAADMM_NO_HEAVY", "heavy", 0), ("EXAADMM_HEAVY_MAX", "heavy_max", None), ("EXAADMM_HEAVY_SPLIT", "heavy_split", None), ("EXAADMM_HEAVY_SPLIT_PREV", "heavy_split_prev", None),
                ("EXAADMM_HEAVY_ADOPT", "heavy_adopt", None), ("EXAADMM_BLOCK_ORDER", "block_order", None),
                ("EXAADMM_DIST_OVERLAP", "dist_overlap", None), ("EXAADMM_DIST_P2P", "dist_p2p", None),
                ("EXAADMM_TRUST_RESIDENT", "trust_resident", None))
_OPTION_DEFAULTS = {"bus_kernel_lane": 0, "graph": 1, "spec": 1, "rec": 1, "resident": 1, "resident_tiles": 1, "heavy": 1, "heavy_max": 512,
                    "heavy_split": 48, "heavy_split_prev": 96, "heavy_adopt": 4, "block_order": 1, "dist_overlap": 1, "dist_p2p": 1, "trust_resident": 0}
# switches of the TEST build only (libexaadmm_hip_hooks.so, exaadmm_test_hook): a model created while one of these variables is set needs that build
_ENV_TEST_HOOKS = (("EXAADMM_RESIDENT_TEST_FAIL", "fail_resident_launch"), ("EXAADMM_RESIDENT_COOP", "resident_coop"),
                   ("EXAADMM_RESIDENT_CAP_LIMIT", "resident_cap_limit"), ("EXAADMM_HEAVY_SELECT_SMALL", "heavy_select_small"))
# diagnostic build only (libexaadmm_hip_prof.so): phase clocks of the resident loop's lanes
_ENV_PROF_OPTIONS = (("EXAADMM_PROF_RESIDENT", "prof_resident", 1),)


def options_from_environment(environ=None):
    """the options the environment asks for (everything else at its default)"""
    environ = os.environ if environ is None else environ
    want = dict(_OPTION_DEFAULTS)
    for var, key, val in _ENV_OPTIONS:
        v = environ.get(var)
        if v is None or v == "":
            continue
        want[key] = val(v) if callable(val) else (int(v) if val is None else val)
    for var, v in environ.items():
        if var.startswith("EXAADMM_OPT_") and v != "":
            want[var[len("EXAADMM_OPT_"):].lower()] = int(v)
    return want


class _HandleOptions:
    """mixin of the model classes: exaadmm_set_option / exaadmm_get_option, and the environment's switches applied to the handle
    when the model is CREATED (apply_env_options() re-reads them on demand; set_option is the direct way)"""
    _opt_applied = None

    def set_option(self, key, value):
        _check(library().exaadmm_set_option(self._h, key.encode(), int(value)), "exaadmm_set_option(%s)" % key)

    def get_option(self, key):
        v = C.c_int64()
        _check(library().exaadmm_get_option(self._h, key.encode(), C.byref(v)), "exaadmm_get_option(%s)" % key)
        return v.value

    def apply_env_options(self):
        want = options_from_environment()
        hooks = tuple((key, int(os.environ[var])) for var, key in _ENV_TEST_HOOKS if os.environ.get(var, "") != "")
        prof = tuple((key, val) for var, key, val in _ENV_PROF_OPTIONS if os.environ.get(var, "") != "")
        sig = (tuple(sorted(want.items())), hooks, prof)
        if sig == self._opt_applied:
            return
        for k, v in want.items():
            self.set_option(k, v)
        for k, v in prof:
            self.set_option(k, v)
        was = dict(self._opt_applied[1]) if self._opt_applied is not None else {}
        if hooks or was:
            L = library()
            if not hasattr(L, "exaadmm_test_hook"):
                raise ExaAdmmError("%s need(s) the test build of the library (libexaadmm_hip_hooks.so, EXAADMM_HIP_LIB)"
                                   % ", ".join(var for var, _ in _ENV_TEST_HOOKS if os.environ.get(var, "") != ""))
            now = dict(hooks)
            for key in set(now) | set(was):   # (a variable that is no longer set: back to the default)
                default = {"fail_resident_launch": 0, "resident_coop": 1, "resident_cap_limit": 0, "heavy_select_small": 0}[key]
                _check(L.exaadmm_test_hook(self._h, key.encode(), now.get(key, default)), "exaadmm_test_hook(%s)" % key)
        self._opt_applied = sig

    def test_hook(self, key, value):
        L = library()
        if not hasattr(L, "exaadmm_test_hook"):
            raise ExaAdmmError("test hooks exist only in the test build of the library (libexaadmm_hip_hooks.so, EXAADMM_HIP_LIB)")
        _check(L.exaadmm_test_hook(self._h, key.encode(), int(value)), "exaadmm_test_hook(%s)" % key)

    def resident_form(self):
        """what the last loop call launched, as the library decided it: dict(tiles, heavy, launches, wavefronts); tiles = 0: not the resident loop"""
        t, hv, n, w = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _check(library().exaadmm_resident_form(self._h, C.byref(t), C.byref(hv), C.byref(n), C.byref(w)), "exaadmm_resident_form")
        return {"tiles": t.value, "heavy": hv.value, "launches": n.value, "wavefronts": w.value}

    def resident_selfchecks(self):
        """(self checks made, self checks that found a difference) of the resident loops on this handle"""
        a, b = C.c_int(), C.c_int()
        _check(library().exaadmm_resident_selfcheck_stats(self._h, C.byref(a), C.byref(b)), "exaadmm_resident_selfcheck_stats")
        return a.value, b.value


def _check(rc, what):
    if rc != 0:
        L = library()
        msg = L.exaadmm_last_error().decode() or L.exaadmm_io_error().decode()
        raise ExaAdmmError("%s failed (status %d): %s" % (what, rc, msg))


def _dp(a):
    return a.ctypes.data_as(c_double_p)


# --------------------------------------------------------------------------- grid data
def _as_grid_arrays(g):
    """normalise a dict of arrays (zero-based indices) to contiguous f64 / i64 numpy arrays"""
    out = {}
    for k in Grid._F64 + Grid._BUS:
        out[k] = np.ascontiguousarray(g[k], dtype=np.float64)
    for k in Grid._IDX + ("brBusIdx",):
        out[k] = np.ascontiguousarray(g[k], dtype=np.int64)
    out["ngen"], out["nline"], out["nbus"] = int(g["ngen"]), int(g["nline"]), int(g["nbus"])
    out["baseMVA"] = float(g["baseMVA"])
    out["nline_owned"] = int(g.get("nline_owned", 0))
    return out


def opf_loaddata(case, tight_factor=1.0, case_format="matpower"):
    """MATPOWER file -> grid dict through the native loader (exaadmm_load_matpower).
    Counterpart of opf_loaddata + GridData (utils/opfdata.jl:408-415, utils/grid_data.jl:61-83)."""
    L = library()
    handle = C.c_void_p()
    _check(L.exaadmm_load_matpower(os.fsencode(case), float(tight_factor), case_format.encode(),
                                   C.byref(handle)), "exaadmm_load_matpower(%s)" % case)
    try:
        cg = Grid()
        _check(L.exaadmm_case_grid(handle, C.byref(cg)), "exaadmm_case_grid")
        G, Ln, B = cg.ngen, cg.nline, cg.nbus
        sizes = {k: G for k in ("pgmin", "pgmax", "qgmin", "qgmax", "c2", "c1", "c0")}
        sizes.update({k: Ln for k in ("YffR", "YffI", "YftR", "YftI", "YttR", "YttI", "YtfR", "YtfI", "rateA")})
        sizes.update({k: 2 * Ln for k in ("FrVmBound", "ToVmBound", "FrVaBound", "ToVaBound", "brBusIdx")})
        sizes.update({k: B for k in Grid._BUS})
        sizes.update({"FrStart": B + 1, "ToStart": B + 1, "GenStart": B + 1, "FrIdx": Ln, "ToIdx": Ln, "GenIdx": G})
        g = {"ngen": G, "nline": Ln, "nbus": B, "baseMVA": cg.baseMVA}
        for k, n in sizes.items():
            p = getattr(cg, k)
            g[k] = np.ctypeslib.as_array(p, shape=(n,)).copy() if n > 0 else np.zeros(
                0, dtype=np.int64 if k in Grid._IDX + ("brBusIdx",) else np.float64)
        return g
    finally:
        L.exaadmm_case_free(handle)


# --------------------------------------------------------------------------- env / model
class AdmmEnv:
    """AdmmEnv (utils/environment.jl:85-158): case data + ADMM parameters."""

    def __init__(self, case, rho_pq, rho_va, case_format="matpower", use_linelimit=True,
                 use_projection=False, tight_factor=1.0, gpu_no=0, verbose=1, grid=None):
        if not use_linelimit:
            raise ExaAdmmError("use_linelimit=false is a dead option in the reference "
                               "(polar_kernel_two_level_alternative is undefined there)")
        if use_projection:
            raise ExaAdmmError("use_projection=true is not part of the ACOPF path (pf_projection is disabled upstream)")
        self.case = case
        self.initial_rho_pq = float(rho_pq)
        self.initial_rho_va = float(rho_va)
        self.tight_factor = float(tight_factor)
        self.gpu_no = int(gpu_no)
        self.use_linelimit = True
        self.use_projection = False
        self.data = _as_grid_arrays(grid if grid is not None else
                                    opf_loaddata(case, tight_factor=tight_factor, case_format=case_format))
        self.params = Parameters()
        library().exaadmm_default_params(C.byref(self.params))
        self.params.verbose = int(verbose)


class Solution:
    """Solution (utils/environment.jl:177-226): attribute access downloads the device vector
    in the reference layout; assignment uploads."""

    def __init__(self, mod):
        object.__setattr__(self, "_mod", mod)

    def __getattr__(self, name):
        if name in VECTORS:
            return self._mod.download(name)
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in VECTORS:
            self._mod.upload(name, value)
        else:
            object.__setattr__(self, name, value)


class ModelAcopf(_HandleOptions):
    """ModelAcopf (models/acopf/acopf_model.jl:6-95) living on one MI355X."""

    def __init__(self, env):
        L = library()
        self.env = env
        g = env.data
        self.grid_data = g
        self.ngen, self.nline, self.nbus = g["ngen"], g["nline"], g["nbus"]
        self.n = 6
        self.nvar = 2 * self.ngen + 8 * self.nline
        self.gen_start, self.line_start = 1, 2 * self.ngen + 1   # 1-based like the reference
        cg = Grid()
        cg.ngen, cg.nline, cg.nbus, cg.index_base, cg.baseMVA = self.ngen, self.nline, self.nbus, 0, g["baseMVA"]
        cg.nline_owned = g.get("nline_owned", 0)
        for k in Grid._F64 + Grid._BUS:
            setattr(cg, k, _dp(g[k]))
        for k in Grid._IDX + ("brBusIdx",):
            setattr(cg, k, g[k].ctypes.data_as(c_i64_p))
        self._h = C.c_void_p()
        _check(L.exaadmm_create(C.byref(cg), C.byref(env.params), env.gpu_no, C.byref(self._h)), "exaadmm_create")
        self.apply_env_options()
        _check(L.exaadmm_init_solution(self._h, env.initial_rho_pq, env.initial_rho_va), "exaadmm_init_solution")
        self.info = Info()
        self.solution = Solution(self)
        self.pgmin_curr = g["pgmin"].copy()
        self.pgmax_curr = g["pgmax"].copy()

    # -- plumbing
    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            library().exaadmm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync_params(self):
        _check(library().exaadmm_set_params(self._h, C.byref(self.env.params)), "exaadmm_set_params")

    def download(self, name):
        out = np.empty(self.nvar)
        _check(library().exaadmm_download_solution(self._h, VECTORS[name], _dp(out), self.nvar), "download " + name)
        return out

    def upload(self, name, value):
        a = np.ascontiguousarray(value, dtype=np.float64)
        _check(library().exaadmm_upload_solution(self._h, VECTORS[name], _dp(a), a.size), "upload " + name)

    @property
    def membuf(self):
        out = np.zeros((self.nline, 31))
        _check(library().exaadmm_download_membuf(self._h, _dp(out), 31, self.nline), "download membuf")
        return out.T  # 31 x nline view like the reference

    @membuf.setter
    def membuf(self, value):
        a = np.ascontiguousarray(np.asarray(value, dtype=np.float64).T)
        _check(library().exaadmm_upload_membuf(self._h, _dp(a), 31, self.nline), "upload membuf")

    def set_load(self, Pd, Qd):
        Pd = np.ascontiguousarray(Pd, dtype=np.float64)
        Qd = np.ascontiguousarray(Qd, dtype=np.float64)
        _check(library().exaadmm_set_load(self._h, _dp(Pd), _dp(Qd), Pd.size), "exaadmm_set_load")

    def set_pg_bounds(self, lo, hi):
        lo = np.ascontiguousarray(lo, dtype=np.float64)
        hi = np.ascontiguousarray(hi, dtype=np.float64)
        _check(library().exaadmm_set_pg_bounds(self._h, _dp(lo), _dp(hi), lo.size), "exaadmm_set_pg_bounds")
        self.pgmin_curr, self.pgmax_curr = lo.copy(), hi.copy()

    def branch_stats(self):
        a, m = C.c_double(), C.c_double()
        _check(library().exaadmm_branch_stats(self._h, C.byref(a), C.byref(m)), "exaadmm_branch_stats")
        return a.value, m.value

    def branch_counters(self):
        """per-branch (auglag iterations, TRON minor iterations, solver-loop trips) of the last x-update"""
        a, m, t = (np.zeros(self.nline, dtype=np.int32) for _ in range(3))
        ip = C.POINTER(C.c_int32)
        _check(library().exaadmm_download_branch_counters(self._h, a.ctypes.data_as(ip), m.ctypes.data_as(ip),
                                                          t.ctypes.data_as(ip), self.nline), "branch counters")
        return a, m, t

    def tile_trips(self):
        """per tile of 64 branches in storage order (= per wavefront): largest solver-trip count of its lanes in the last x-update"""
        n = (self.nline + 63) // 64
        out = np.zeros(n, dtype=np.int32)
        _check(library().exaadmm_download_tile_trips(self._h, out.ctypes.data_as(C.POINTER(C.c_int32)), n), "exaadmm_download_tile_trips")
        return out

    def loop_kind(self):
        """0 = four launches per iteration, 1 = two (speculative loop), 2 = resident loop (one launch per batch of iterations)"""
        k = C.c_int()
        _check(library().exaadmm_loop_kind(self._h, C.byref(k)), "exaadmm_loop_kind")
        return k.value

    def loop_downgrades(self):
        """how often a resident batch of this handle was found not co-resident (shared GPU), restored and repeated with the
        two-launch loop (the handle then keeps that loop)"""
        k = C.c_int()
        _check(library().exaadmm_loop_downgrades(self._h, C.byref(k)), "exaadmm_loop_downgrades")
        return k.value

    def branch_kernel_occupancy(self):
        """resident wavefronts per SIMD allowed by the branch kernel's register footprint"""
        w = C.c_int()
        _check(library().exaadmm_branch_kernel_occupancy(self._h, C.byref(w)), "exaadmm_branch_kernel_occupancy")
        return w.value

    def run_fixed(self, iters, first_major_iter=2, beta=None, time_branch=False):
        """K inner iterations without convergence test; returns (norms, ms_total, ms_branch)."""
        out = np.zeros(4)
        ms, msb = C.c_double(), C.c_double()
        beta = self.env.params.beta if beta is None else beta
        _check(library().exaadmm_run_fixed(self._h, int(iters), int(first_major_iter), float(beta), _dp(out),
                                           C.byref(ms), C.byref(msb) if time_branch else None), "exaadmm_run_fixed")
        return out, ms.value, (msb.value if time_branch else None)

    def state_save(self):
        _check(library().exaadmm_state_save(self._h), "exaadmm_state_save")

    def state_restore(self):
        _check(library().exaadmm_state_restore(self._h), "exaadmm_state_restore")

    def run_inner(self, outer, max_inner, beta):
        out = np.zeros(4)
        done = C.c_int()
        _check(library().exaadmm_run_inner(self._h, int(outer), int(max_inner), float(beta), C.byref(done), _dp(out)),
               "exaadmm_run_inner")
        return done.value, out


# --------------------------------------------------------------------------- plugin steps
# signatures (env, mod, device=None) as in docs/src/dev.md:65-155
def admm_increment_outer(env, mod, device=None):
    mod.info.outer += 1


def admm_increment_reset_inner(env, mod, device=None):
    mod.info.inner = 0


def admm_increment_inner(env, mod, device=None):
    mod.info.inner += 1
    mod.info.cumul += 1


def admm_outer_prestep(env, mod, device=None):
    v = C.c_double()
    _check(library().exaadmm_norm_z(mod.handle, C.byref(v)), "exaadmm_norm_z")
    mod.info.norm_z_prev = v.value


def admm_inner_prestep(env, mod, device=None):
    _check(library().exaadmm_inner_prestep(mod.handle), "exaadmm_inner_prestep")


def admm_update_x(env, mod, device=None):
    mod.sync_params()
    _check(library().exaadmm_update_x(mod.handle, mod.info.inner), "exaadmm_update_x")


def admm_update_xbar(env, mod, device=None):
    _check(library().exaadmm_update_xbar(mod.handle), "exaadmm_update_xbar")


def admm_update_z(env, mod, device=None):
    _check(library().exaadmm_update_z(mod.handle, env.params.beta), "exaadmm_update_z")


def admm_update_l(env, mod, device=None):
    _check(library().exaadmm_update_l(mod.handle, env.params.beta), "exaadmm_update_l")


def admm_update_residual(env, mod, device=None):
    out = np.zeros(4)
    _check(library().exaadmm_update_residual(mod.handle, _dp(out)), "exaadmm_update_residual")
    mod.info.primres, mod.info.dualres, mod.info.norm_z_curr, mod.info.mismatch = out


def admm_update_lz(env, mod, device=None):
    mod.sync_params()
    _check(library().exaadmm_update_lz(mod.handle, env.params.beta), "exaadmm_update_lz")


def admm_poststep(env, mod, device=None):
    v = C.c_double()
    _check(library().exaadmm_objective(mod.handle, C.byref(v)), "exaadmm_objective")
    mod.info.objval = v.value


def admm_two_level(env, mod, device=None, fused=True):
    """admm_two_level (algorithms/admm_two_level.jl:1-88).
    fused=True  : the whole driver runs inside the library (exaadmm_solve_two_level), loop
                  control on the device.
    fused=False : this Python loop calls the step-wise entry points exactly in the
                  reference's order (what the Julia shim does)."""
    par, info = env.params, mod.info
    if fused:
        mod.sync_params()
        _check(library().exaadmm_solve_two_level(mod.handle, C.byref(info)), "exaadmm_solve_two_level")
        _check(library().exaadmm_get_params(mod.handle, C.byref(par)), "exaadmm_get_params")
        return
    sqrt_d = math.sqrt(mod.nvar)
    OUTER_TOL = sqrt_d * par.outer_eps
    C.memset(C.byref(info), 0, C.sizeof(info))
    info.mismatch = math.inf
    info.norm_z_prev = info.norm_z_curr = math.inf
    par.beta = par.initial_beta
    info.status_code = 0
    t0 = time.perf_counter()
    while info.outer < par.outer_iterlim:
        admm_increment_outer(env, mod, device)
        admm_outer_prestep(env, mod, device)
        admm_increment_reset_inner(env, mod, device)
        while info.inner < par.inner_iterlim:
            admm_increment_inner(env, mod)
            admm_inner_prestep(env, mod, device)
            admm_update_x(env, mod, device)
            admm_update_xbar(env, mod, device)
            admm_update_z(env, mod, device)
            admm_update_l(env, mod, device)
            admm_update_residual(env, mod, device)
            info.eps_pri = sqrt_d / (2500 * info.outer)
            if info.primres <= info.eps_pri:
                break
        if info.mismatch <= OUTER_TOL:
            info.status_code = 1
            break
        admm_update_lz(env, mod, device)
        if info.norm_z_curr > par.theta * info.norm_z_prev:
            par.beta = min(par.inc_c * par.beta, 1e24)
    info.time_overall = time.perf_counter() - t0
    admm_poststep(env, mod, device)


def print_statistics(env, mod, file=None):
    """print_statistics (utils/print_statistics.jl:1-21).  Step times of a fused solve are sampled
    device times (one iteration in eight); generators run inside the branch launch."""
    info = mod.info
    out = file or __import__("sys").stdout
    w = out.write
    w(" ** Statistics\n")
    w("Objective value  . . . . . . . . . %12.6e\n" % info.objval)
    w("Residual . . . . . . . . . . . . . %12.6e\n" % info.mismatch)
    w("Outer iterations . . . . . . . . . %12d\n" % info.outer)
    w("Cumulative iterations  . . . . . . %12d\n" % info.cumul)
    w("Time per iteration . . . . . . . . %12.6f (secs/iter)\n" % (info.time_overall / max(info.cumul, 1)))
    w("Overall time . . . . . . . . . . . %12.3f (secs)\n" % info.time_overall)
    w("Generator + branch time  . . . . . %12.3f (secs)\n" % info.time_x_update)
    w("Bus time . . . . . . . . . . . . . %12.3f (secs)\n" % info.time_xbar_update)
    w("z, l, residual time  . . . . . . . %12.3f (secs)\n" % info.time_zl_update)


def solve_acopf(case, case_format="matpower", outer_iterlim=20, inner_iterlim=1000, rho_pq=400.0,
                rho_va=40000.0, obj_scale=1.0, scale=1e-4, use_linelimit=True, use_projection=False,
                tight_factor=1.0, outer_eps=2e-4, gpu_no=0, verbose=0, grid=None, fused=True,
                check_every=1):
    """solve_acopf (interface/solve_acopf.jl:1-46), backend fixed to the MI355X library.
    obj_scale is accepted and, as in the reference (ModelAcopf is built before the parameter
    is assigned, solve_acopf.jl:33 vs :36), has no effect on the generator costs."""
    env = AdmmEnv(case, rho_pq, rho_va, case_format=case_format, use_linelimit=use_linelimit,
                  use_projection=use_projection, tight_factor=tight_factor, gpu_no=gpu_no,
                  verbose=verbose, grid=grid)
    mod = ModelAcopf(env)
    env.params.scale = scale
    env.params.outer_eps = outer_eps
    env.params.outer_iterlim = outer_iterlim
    env.params.inner_iterlim = inner_iterlim
    env.params.check_every = check_every
    admm_two_level(env, mod, fused=fused)
    if verbose > 0:
        print_statistics(env, mod)
    return env, mod


# --------------------------------------------------------------------------- rolling horizon
def get_load(load_prefix, load_scale=1.0):
    """get_load (utils/opfdata.jl:121-130): <prefix>.Pd / <prefix>.Qd, whitespace matrices nbus x T."""
    pd = np.atleast_2d(np.loadtxt(load_prefix + ".Pd")) * load_scale
    qd = np.atleast_2d(np.loadtxt(load_prefix + ".Qd")) * load_scale
    if pd.shape != qd.shape:
        raise ExaAdmmError("Pd / Qd load files differ in shape")
    return pd, qd


def update_real_power_current_bounds(mod, x_curr=None):
    """acopf_admm_rolling_cpu.jl:1-14: ramp-limited generation bounds around the last dispatch."""
    x = mod.solution.u_curr if x_curr is None else x_curr
    pg = x[0:2 * mod.ngen:2]
    lo = np.maximum(mod.grid_data["pgmin"], pg - mod.ramp_rate)
    hi = np.minimum(mod.grid_data["pgmax"], pg + mod.ramp_rate)
    mod.set_pg_bounds(lo, hi)


def admm_restart_rolling(env, mod, load, start_period=1, end_period=6, update_start_period_bounds=False,
                         fused=True, on_period=None):
    """admm_restart_rolling (acopf_admm_rolling_cpu.jl:16-48): per period swap Pd/Qd, warm-start
    admm_two_level on the same model (lz, z, l, u, v and the branch multipliers persist; beta is
    reset inside admm_two_level), then tighten the pg bounds by the ramp rate."""
    pd, qd = load
    if not (start_period >= 1 and end_period <= pd.shape[1]):
        raise ExaAdmmError("period range outside the load profile")
    if update_start_period_bounds:
        update_real_power_current_bounds(mod)
    results = []
    for t in range(start_period, end_period + 1):
        mod.set_load(pd[:, t - 1], qd[:, t - 1])
        admm_two_level(env, mod, fused=fused)
        results.append(dict(period=t, status=mod.info.status, objval=mod.info.objval, mismatch=mod.info.mismatch,
                            cumul=mod.info.cumul, outer=mod.info.outer, time=mod.info.time_overall))
        if on_period:
            on_period(results[-1])
        update_real_power_current_bounds(mod)
    return results


def solve_acopf_rolling(case, load_prefix=None, load=None, case_format="matpower", outer_iterlim=20,
                        inner_iterlim=1000, rho_pq=400.0, rho_va=40000.0, obj_scale=1.0, scale=1e-4,
                        tight_factor=0.99, outer_eps=2e-4, gpu_no=0, verbose=0, ramp_ratio=0.02,
                        start_period=1, end_period=6, grid=None, fused=True):
    """solve_acopf_rolling (interface/solve_acopf_rolling.jl:2-30).  `load` may be given directly as
    (Pd, Qd) matrices instead of a file prefix.  Returns (env, mod, per-period results)."""
    env = AdmmEnv(case, rho_pq, rho_va, case_format=case_format, tight_factor=tight_factor, gpu_no=gpu_no,
                  verbose=verbose, grid=grid)
    mod = ModelAcopf(env)
    mod.ramp_rate = ramp_ratio * mod.grid_data["pgmax"]     # acopf_model.jl:66-67
    env.params.scale = scale
    env.params.outer_eps = outer_eps
    env.params.outer_iterlim = outer_iterlim
    env.params.inner_iterlim = inner_iterlim
    if load is None:
        if load_prefix is None:
            raise ExaAdmmError("load profile required (load_prefix or load=(Pd, Qd))")
        load = get_load(load_prefix)
    results = admm_restart_rolling(env, mod, load, start_period=start_period, end_period=end_period, fused=fused)
    return env, mod, results


# --------------------------------------------------------------------------- multi-period ACOPF
RAMP_VECTORS = dict(VECTORS, s_curr=20, mu=21, xi=22)
RAMP_VECTORS.pop("v_curr", None)   # SolutionRamping.v_curr is never read on this path


class _PeriodSolution:
    """mod.models[i].solution of the reference: the period's slice of the time-expanded vectors,
    in the single-period reference layout [2 ngen | 8 nline]."""

    def __init__(self, mp, t):
        object.__setattr__(self, "_mp", mp)
        object.__setattr__(self, "_t", t)

    def __getattr__(self, name):
        if name in VECTORS:
            return self._mp.period_vector(name, self._t)
        raise AttributeError(name)


class _RampSolution:
    """mod.solution[i] of the reference (SolutionRamping, mpacopf_model.jl:1-38), ngen entries."""

    def __init__(self, mp, t):
        object.__setattr__(self, "_mp", mp)
        object.__setattr__(self, "_t", t)

    def __getattr__(self, name):
        if name in RAMP_VECTORS:
            G = self._mp.ngen
            return self._mp.download_ramp(name)[self._t * G:(self._t + 1) * G]
        raise AttributeError(name)


class _Period:
    def __init__(self, mp, t):
        self.solution = _PeriodSolution(mp, t)
        self.grid_data = mp.grid_data
        self.ngen, self.nline, self.nbus = mp.ngen, mp.nline, mp.nbus
        self.gen_start, self.line_start = 1, 2 * mp.ngen + 1


class ModelMpacopf(_HandleOptions):
    """ModelMpacopf (models/mpacopf/mpacopf_model.jl:40-112) on one MI355X: one time-expanded device
    model (include/exaadmm_hip.h, exaadmm_mp_create).  `load` = (pd, qd), nbus x T matrices as
    returned by get_load; periods start_period..end_period (1-based, inclusive) are used.
    All admm_* plugin steps and admm_two_level take this object unchanged."""

    def __init__(self, env, load, start_period=1, end_period=1, ramp_ratio=0.02):
        L = library()
        pd, qd = (np.asarray(a, dtype=np.float64) for a in load)
        if not (1 <= start_period <= end_period <= pd.shape[1]):
            raise ExaAdmmError("start_period / end_period out of range of the load profile")
        self.env = env
        g = env.data
        self.grid_data = g
        self.ngen, self.nline, self.nbus = g["ngen"], g["nline"], g["nbus"]
        self.len_horizon = T = end_period - start_period + 1
        self.ramp_ratio = float(ramp_ratio)
        self.n = 6
        cg = Grid()
        cg.ngen, cg.nline, cg.nbus, cg.index_base, cg.baseMVA = self.ngen, self.nline, self.nbus, 0, g["baseMVA"]
        for k in Grid._F64 + Grid._BUS:
            setattr(cg, k, _dp(g[k]))
        for k in Grid._IDX + ("brBusIdx",):
            setattr(cg, k, g[k].ctypes.data_as(c_i64_p))
        self._pd = np.ascontiguousarray(pd[:, start_period - 1:end_period].T)   # T x nbus, period-major
        self._qd = np.ascontiguousarray(qd[:, start_period - 1:end_period].T)
        self._h = C.c_void_p()
        _check(L.exaadmm_mp_create(C.byref(cg), C.byref(env.params), T, _dp(self._pd), _dp(self._qd), self.ramp_ratio,
                                   env.gpu_no, C.byref(self._h)), "exaadmm_mp_create")
        self.apply_env_options()
        _check(L.exaadmm_init_solution(self._h, env.initial_rho_pq, env.initial_rho_va), "exaadmm_init_solution")
        self.nvar = int(L.exaadmm_mp_nvar(self._h))              # mod.nvar of the reference
        self.nvar_expanded = int(L.exaadmm_nvar(self._h))        # length of the device vectors
        self.info = Info()
        self.models = [_Period(self, t) for t in range(T)]
        self.solution = [_RampSolution(self, t) for t in range(T)]

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            library().exaadmm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync_params(self):
        _check(library().exaadmm_set_params(self._h, C.byref(self.env.params)), "exaadmm_set_params")

    def init_solution(self):
        _check(library().exaadmm_init_solution(self._h, self.env.initial_rho_pq, self.env.initial_rho_va), "exaadmm_init_solution")

    def download(self, name):
        out = np.empty(self.nvar_expanded)
        _check(library().exaadmm_download_solution(self._h, VECTORS[name], _dp(out), out.size), "download " + name)
        return out

    def upload(self, name, value):
        a = np.ascontiguousarray(value, dtype=np.float64)
        _check(library().exaadmm_upload_solution(self._h, VECTORS[name], _dp(a), a.size), "upload " + name)

    def period_vector(self, name, t):
        """Period t (0-based) of a solution vector in the single-period reference layout."""
        x = self.download(name)
        G, Ln, T = self.ngen, self.nline, self.len_horizon
        return np.concatenate([x[2 * G * t:2 * G * (t + 1)], x[2 * G * T + 8 * Ln * t:2 * G * T + 8 * Ln * (t + 1)]])

    def download_ramp(self, name):
        out = np.empty(self.len_horizon * self.ngen)
        _check(library().exaadmm_mp_download_ramp(self._h, RAMP_VECTORS[name], _dp(out), out.size), "download ramp " + name)
        return out

    def upload_ramp(self, name, value):
        a = np.ascontiguousarray(value, dtype=np.float64)
        _check(library().exaadmm_mp_upload_ramp(self._h, RAMP_VECTORS[name], _dp(a), a.size), "upload ramp " + name)

    @property
    def membuf(self):
        n = self.len_horizon * self.nline
        out = np.zeros((n, 31))
        _check(library().exaadmm_download_membuf(self._h, _dp(out), 31, n), "download membuf")
        return out.T.copy()

    @membuf.setter
    def membuf(self, value):
        n = self.len_horizon * self.nline
        a = np.ascontiguousarray(np.asarray(value, dtype=np.float64).T)
        _check(library().exaadmm_upload_membuf(self._h, _dp(a), 31, n), "upload membuf")

    def state_save(self):
        _check(library().exaadmm_state_save(self._h), "exaadmm_state_save")

    def state_restore(self):
        _check(library().exaadmm_state_restore(self._h), "exaadmm_state_restore")

    def loop_kind(self):
        """0 = four launches per iteration, 1 = two (the z / l pass of an iteration rides in the next branch launch)"""
        k = C.c_int()
        _check(library().exaadmm_loop_kind(self._h, C.byref(k)), "exaadmm_loop_kind")
        return k.value

    def run_inner(self, outer, max_inner, beta):
        self.sync_params()
        done = C.c_int()
        out = np.zeros(4)
        _check(library().exaadmm_run_inner(self._h, outer, max_inner, beta, C.byref(done), _dp(out)), "exaadmm_run_inner")
        return done.value, out

    def run_fixed(self, iters, first_major_iter=2, beta=None, time_branch=False):
        self.sync_params()
        out = np.zeros(4)
        ms, msb = C.c_double(), C.c_double()
        beta = self.env.params.beta if beta is None else beta
        _check(library().exaadmm_run_fixed(self._h, iters, first_major_iter, beta, _dp(out), C.byref(ms),
                                           C.byref(msb) if time_branch else None), "exaadmm_run_fixed")
        return out, ms.value, msb.value


def solve_mpacopf(case, load_prefix=None, load=None, case_format="matpower", start_period=1, end_period=1,
                  outer_iterlim=20, inner_iterlim=1000, rho_pq=400.0, rho_va=40000.0, obj_scale=1.0, scale=1e-4,
                  use_linelimit=True, tight_factor=1.0, outer_eps=2e-4, gpu_no=0, verbose=0, ramp_ratio=0.02,
                  warm_start=True, grid=None, fused=True):
    """solve_mpacopf (interface/solve_mpacopf.jl:1-37).  `load` = (pd, qd) may be given instead of
    `load_prefix`.  warm_start: every period is first solved alone; as in the reference the
    following init_solution! resets all vectors, so what survives is membuf (the line-limit
    multipliers, rows 25-26) -- reproduced here by solving each period on a single-period handle
    and copying those rows into the multi-period model."""
    if load is None:
        if load_prefix is None:
            raise ExaAdmmError("solve_mpacopf needs load_prefix or load")
        load = get_load(load_prefix)
    env = AdmmEnv(case, rho_pq, rho_va, case_format=case_format, use_linelimit=use_linelimit,
                  tight_factor=tight_factor, gpu_no=gpu_no, verbose=verbose, grid=grid)
    mod = ModelMpacopf(env, load, start_period=start_period, end_period=end_period, ramp_ratio=ramp_ratio)
    env.params.scale = scale
    env.params.outer_eps = outer_eps
    env.params.outer_iterlim = outer_iterlim
    env.params.inner_iterlim = inner_iterlim
    if warm_start:
        mb = mod.membuf
        Ln = mod.nline
        single = ModelAcopf(env)
        mb0 = single.membuf.copy()   # membuf as constructed (zeros, row 29 = rateA): every models[i] starts from it
        for t in range(mod.len_horizon):
            _check(library().exaadmm_init_solution(single.handle, env.initial_rho_pq, env.initial_rho_va), "exaadmm_init_solution")
            single.membuf = mb0
            single.set_load(mod._pd[t], mod._qd[t])
            admm_two_level(env, single, fused=fused)
            mb[:, t * Ln:(t + 1) * Ln] = single.membuf
        single.close()
        mod.membuf = mb          # the periods' solved multipliers ...
        mod.init_solution()      # ... survive init_solution! (it resets the vectors only), as in the reference
    admm_two_level(env, mod, fused=fused)
    if verbose > 0:
        print_statistics(env, mod)
    return env, mod
