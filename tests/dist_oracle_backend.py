"""TEST-ONLY backend for exaadmm.jl_amd/distributed.py: the same interface as HipBackend, computed
by the CPU oracle on numpy arrays, so the partition / halo / all-reduce orchestration can run
under torch.distributed's gloo backend without a GPU.  Lives in tests/: the product has no CPU
path.  Same scheme as the HIP path: every local branch (natives and the replicas of the peers' cut
branches) is solved and updated here; only the owned entries count in the residual sums."""
import ctypes as C
import math

import numpy as np
import torch

from oracle import oracle as O

VEC_NAMES = {0: "u_curr", 1: "v_curr", 2: "l_curr", 3: "rho", 6: "z_curr", 7: "z_prev", 8: "lz"}


class OracleBackend:
    def __init__(self, part, rho_pq, rho_va, scale=1e-4):
        self.part = part
        self.m = O.OracleModel(part.grid, rho_pq, rho_va, scale=scale)
        self.L = O.lib()
        self.sums = torch.zeros(4, dtype=torch.float64)
        self.ctrl = None
        self.check_every = 1
        self.plans = (part.plan_send_v, part.plan_recv_v)

    # -- plumbing
    def set_check_every(self, k):
        self.check_every = int(k)

    def new_buffer(self, n):
        return torch.zeros(max(int(n), 1), dtype=torch.float64)

    def upload(self, name, vec):
        self.m.sol[name][:] = vec

    def download(self, name):
        return self.m.sol[name].copy()

    def _skip(self):
        return self.ctrl is not None and self.ctrl["done"]

    # -- loop control (mirror of Ctrl / k_ctrl_step)
    def ctrl_begin(self, outer, max_inner, beta):
        self.ctrl = dict(done=False, inner=0, max_inner=max_inner, beta=beta,
                         eps_pri=math.sqrt(self.part.nvar_global) / (2500.0 * outer), res=np.zeros(4))

    def ctrl_step(self):
        c = self.ctrl
        if c["done"]:
            return
        c["res"] = np.sqrt(self.sums.numpy().copy())
        c["inner"] += 1
        test = self.check_every <= 1 or c["inner"] % self.check_every == 0
        if (test and c["res"][0] <= c["eps_pri"]) or c["inner"] >= c["max_inner"]:
            c["done"] = True

    def ctrl_read(self):
        return self.ctrl["done"], self.ctrl["inner"], self.ctrl["res"]

    def ctrl_end(self):
        self.ctrl = None

    # -- steps
    def update_x(self, major_iter):
        if self._skip():
            return
        if self.ctrl is not None:
            major_iter = self.ctrl["inner"] + 1
        m = self.m
        self.L.oracle_generator_update(C.byref(m.cg), C.byref(m.cs))
        self.L.oracle_branch_update(C.byref(m.cg), C.byref(m.par), C.byref(m.cs), int(major_iter), C.byref(m.info))

    def update_xbar(self):
        if self._skip():
            return
        self.L.oracle_bus_update(C.byref(self.m.cg), C.byref(self.m.cs))

    def pack(self, slot, buf):
        which, idx, _ = self.plans[slot]
        out = buf.numpy()
        for w in np.unique(which):
            sel = which == w
            out[:len(idx)][sel] = self.m.sol[VEC_NAMES[int(w)]][idx[sel]]

    def unpack(self, slot, buf):
        which, idx, _ = self.plans[slot]
        src = buf.numpy()
        for w in np.unique(which):
            sel = which == w
            self.m.sol[VEC_NAMES[int(w)]][idx[sel]] = src[:len(idx)][sel]

    def update_zl_partial(self, beta):
        if self._skip():
            return self.sums
        if self.ctrl is not None:
            beta = self.ctrl["beta"]
        m, g = self.m, self.m.cg
        self.L.oracle_inner_prestep(C.byref(g), C.byref(m.cs))
        self.L.oracle_update_z(C.byref(g), C.byref(m.cs), beta)
        self.L.oracle_update_l(C.byref(g), C.byref(m.cs), beta)
        self.L.oracle_update_residual(C.byref(g), C.byref(m.cs), C.byref(m.info))
        n = self.part.nvar_owned
        s = m.sol
        self.sums[:] = torch.tensor([np.sum(s["rp"][:n] ** 2), np.sum(s["rd"][:n] ** 2),
                                     np.sum(s["z_curr"][:n] ** 2), np.sum(s["Ax_plus_By"][:n] ** 2)])
        return self.sums

    def sumsq_z_partial(self):
        n = self.part.nvar_owned
        self.sums[:] = 0.0
        self.sums[2] = float(np.sum(self.m.sol["z_curr"][:n] ** 2))
        return self.sums

    def update_lz(self, beta):
        self.L.oracle_update_lz(C.byref(self.m.cg), C.byref(self.m.cs), beta, self.m.par.MAX_MULTIPLIER)
