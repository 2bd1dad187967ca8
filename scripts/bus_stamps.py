import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, exaadmm_loader
pkg = exaadmm_loader.load_package(); syn = exaadmm_loader.load_submodule("synthetic")
L = pkg.library()
L.exaadmm_debug_wave_timeline.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int64]
g = syn.sized_like("case13659pegase", seed=0)
env = pkg.AdmmEnv("syn", 5e1, 5e3, verbose=0, grid=g); mod = pkg.ModelAcopf(env)
mod.run_fixed(150, first_major_iter=1, beta=1e3)
Lp = ((g["nline"] + 63) // 64) * 64
buf = np.zeros(2 * Lp, dtype=np.uint64)
pkg._check(L.exaadmm_debug_wave_timeline(mod.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size), "tl")
def report(tag):
    pkg._check(L.exaadmm_debug_wave_timeline(mod.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size), "tl")
    st = buf[:256 * 8].reshape(256, 8).astype(np.int64)
    ok = st[:, 4] > 0
    d = np.diff(st[ok, :5], axis=1)
    print("%s per block (clocks): entry->first loads landed median %.0f | ->records landed %.0f | ->leader published %.0f | ->stores landed %.0f | total %.0f (max %.0f)" % (
        tag, np.median(d[:, 0]), np.median(d[:, 1]), np.median(d[:, 2]), np.median(d[:, 3]), np.median(d.sum(axis=1)), d.sum(axis=1).max()))

pkg.admm_update_xbar(env, mod)            # step-wise entry point: k_bus_group_t<false> (gathers)
report("k_bus_group_t<false> (gathers)       ")
mod.run_fixed(2, first_major_iter=151, beta=1e3)   # fused loop: the last launch is k_bus_group_t<true> (pushed records)
report("k_bus_group_t<true>  (pushed records)")
