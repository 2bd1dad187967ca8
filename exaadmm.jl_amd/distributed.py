"""Multi-GPU two-level ADMM ACOPF: one process per GPU, grid sharded by bus/branch.

The reference has no working multi-process path (MPI is imported, only `MPI.Comm_size` is
called for padding: /root/reference/src/models/acopf/acopf_model.jl:49-53; the intended
per-rank branch offset `shift_lines` is vestigial: src/utils/environment.jl:23,58).
This module is new work, designed for 8 MI355X on one node (xGMI, point-to-point links):

  * buses are split into P contiguous ranges of a breadth-first bus order; a generator lives with its
    bus; a branch lives with the owner of its FROM bus ("native" there: it counts in that rank's share of
    the residual norms) and, when its TO bus belongs to another rank, is REPLICATED on that rank (a "ghost",
    stored after the natives);
  * both owners of a cut branch solve it and update its z, l, lz -- same inputs, same code, bit-identical
    results -- so u, z, l never travel.  Per inner iteration there is ONE exchange: after the bus update each
    side sends the 4 new v entries of its own end of every cut branch, together with its 4 partial
    residual sums; the sums are added in rank order on every rank (identical break decisions without an
    all-reduce);
  * in-library loop (default): csrc/dist_native.h issues that exchange to RCCL from C++ as one grouped
    ncclSend / ncclRecv per peer and keeps the fused loop's two launches per iteration; the step-wise
    loop below (torch.distributed collectives; gloo in the CPU tests) is the debugging twin.

The orchestration (DistributedAcopf) talks to a backend object; `HipBackend` drives the C ABI of
libexaadmm_hip.so.  The CPU/gloo tests plug a test-only backend into the same orchestration.
"""
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np

FROM_SIDE = (0, 1, 4, 6)  # pij, qij, wi, thetai of a branch: produced by the bus update of its from-bus
TO_SIDE = (2, 3, 5, 7)    # pji, qji, wj, thetaj: produced by the bus update of its to-bus
U, V, L_, RHO, Z, LZ = 0, 1, 2, 3, 6, 8   # exaadmm_vector ids


# --------------------------------------------------------------------------- model construction
def init_state(grid, rho_pq, rho_va):
    """init_solution! on the host for the GLOBAL grid (acopf_init_solution_cpu.jl:1-45,
    acopf_model.jl:87-89): returns dict of reference-layout vectors + membuf rows."""
    G, Ln = int(grid["ngen"]), int(grid["nline"])
    nvar = 2 * G + 8 * Ln
    v = np.zeros(nvar)
    rho = np.full(nvar, float(rho_pq))
    v[0:2 * G:2] = 0.5 * (grid["pgmin"] + grid["pgmax"])
    v[1:2 * G:2] = 0.5 * (grid["qgmin"] + grid["qgmax"])
    bb = np.asarray(grid["brBusIdx"]).reshape(Ln, 2)
    Vmax, Vmin = np.asarray(grid["Vmax"]), np.asarray(grid["Vmin"])
    wij0 = (Vmax[bb[:, 0]] ** 2 + Vmin[bb[:, 0]] ** 2) / 2
    wji0 = (Vmax[bb[:, 1]] ** 2 + Vmin[bb[:, 1]] ** 2) / 2
    wR0 = np.sqrt(wij0 * wji0)
    vb = v[2 * G:].reshape(Ln, 8)
    vb[:, 0] = grid["YffR"] * wij0 + grid["YftR"] * wR0
    vb[:, 1] = -grid["YffI"] * wij0 - grid["YftI"] * wR0
    vb[:, 2] = grid["YttR"] * wji0 + grid["YtfR"] * wR0
    vb[:, 3] = -grid["YttI"] * wji0 - grid["YtfI"] * wR0
    vb[:, 4], vb[:, 5] = wij0, wji0
    rho[2 * G:].reshape(Ln, 8)[:, 4:] = float(rho_va)
    return {"v_curr": v, "rho": rho, "nvar": nvar}


def bfs_bus_order(grid):
    """Breadth-first bus order from bus 0 (bandwidth-reducing relabel for contiguous partitions);
    every connected component is appended in turn."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import breadth_first_order
    B, Ln = int(grid["nbus"]), int(grid["nline"])
    bb = np.asarray(grid["brBusIdx"]).reshape(Ln, 2)
    adj = sp.coo_matrix((np.ones(2 * Ln, dtype=np.int8), (np.concatenate([bb[:, 0], bb[:, 1]]), np.concatenate([bb[:, 1], bb[:, 0]]))),
                        shape=(B, B)).tocsr()
    seen = np.zeros(B, dtype=bool)
    order = []
    for s0 in range(B):
        if seen[s0]:
            continue
        o = breadth_first_order(adj, s0, directed=False, return_predecessors=False)
        seen[o] = True
        order.append(o)
        if seen.all():
            break
    return np.concatenate(order).astype(np.int64)


class Partition:
    """Rank r's share of a global grid (all indices zero-based)."""

    def __init__(self, grid, world, rank, bus_owner=None, reorder=True):
        """bus_owner: explicit owner per bus.  Default: contiguous ranges of a breadth-first bus order
        (reorder=True), which keeps the cut small whatever numbering the case file uses; reorder=False cuts the
        loader's own bus order into ranges (only sensible for grids that are banded in that order)."""
        G, Ln, B = int(grid["ngen"]), int(grid["nline"]), int(grid["nbus"])
        self.world, self.rank = world, rank
        self.nvar_global = 2 * G + 8 * Ln
        if bus_owner is None:
            bounds = [(B * p) // world for p in range(world + 1)]
            order = bfs_bus_order(grid) if (reorder and world > 1) else np.arange(B, dtype=np.int64)
            bus_owner = np.zeros(B, dtype=np.int64)
            for p in range(world):
                bus_owner[order[bounds[p]:bounds[p + 1]]] = p
        self.bus_owner = bus_owner
        bb = np.asarray(grid["brBusIdx"]).reshape(Ln, 2)
        fr_own, to_own = bus_owner[bb[:, 0]], bus_owner[bb[:, 1]]
        gstart, gidx = np.asarray(grid["GenStart"]), np.asarray(grid["GenIdx"])
        gen_bus = np.empty(G, dtype=np.int64)
        gen_bus[gidx[:G]] = np.repeat(np.arange(B, dtype=np.int64), np.diff(gstart))

        self.buses = np.flatnonzero(bus_owner == rank)
        self.gens = np.flatnonzero(bus_owner[gen_bus] == rank)
        self.native = np.flatnonzero(fr_own == rank)
        self.ghost = np.flatnonzero((to_own == rank) & (fr_own != rank))
        self.n_native, self.n_ghost = len(self.native), len(self.ghost)
        self.branches = np.concatenate([self.native, self.ghost])
        Bl, Gl, Ll = len(self.buses), len(self.gens), len(self.branches)
        if Bl == 0 or self.n_native == 0:
            raise ValueError("rank %d got an empty partition (%d buses, %d branches)" % (rank, Bl, self.n_native))
        bus_l = np.full(B, -1, dtype=np.int64)
        bus_l[self.buses] = np.arange(Bl)
        br_l = np.full(Ln, -1, dtype=np.int64)
        br_l[self.branches] = np.arange(Ll)
        self.br_local = br_l

        # cut branches per peer, ordered by global branch id: both sides build the same list
        cut_out = self.native[to_own[self.native] != rank]          # natives whose to-bus is remote
        self.cut = {}
        for p in range(world):
            mine = np.concatenate([cut_out[to_own[cut_out] == p], self.ghost[fr_own[self.ghost] == p]])
            self.cut[p] = np.sort(mine)
        self.n_cut = len(cut_out)

        # local grid
        g = {"ngen": Gl, "nline": Ll, "nbus": Bl, "baseMVA": float(grid["baseMVA"]),
             "nline_owned": self.n_native}
        for k in ("pgmin", "pgmax", "qgmin", "qgmax", "c2", "c1", "c0"):
            g[k] = np.ascontiguousarray(np.asarray(grid[k])[self.gens])
        for k in ("YffR", "YffI", "YftR", "YftI", "YttR", "YttI", "YtfR", "YtfI", "rateA"):
            g[k] = np.ascontiguousarray(np.asarray(grid[k])[self.branches])
        for k in ("FrVmBound", "ToVmBound", "FrVaBound", "ToVaBound"):
            g[k] = np.ascontiguousarray(np.asarray(grid[k]).reshape(Ln, 2)[self.branches].ravel())
        for k in ("Pd", "Qd", "Vmin", "Vmax", "YshR", "YshI"):
            g[k] = np.ascontiguousarray(np.asarray(grid[k])[self.buses])
        lfr = bus_l[bb[self.branches, 0]]     # -1 where the end is remote
        lto = bus_l[bb[self.branches, 1]]
        # remote ends point at local bus 0: only the OWN end is used (bus-sorted storage order);
        # the values init_solution! derives from them are overridden by the uploaded global state
        g["brBusIdx"] = np.stack([np.maximum(lfr, 0), np.maximum(lto, 0)], 1).ravel()

        def csr(owner, n, gid=None):
            # within a bus keep the GLOBAL component order (ghosts interleave with natives), so the
            # bus update adds its terms in exactly the order of the unpartitioned grid
            keep = np.flatnonzero(owner >= 0)
            key2 = keep if gid is None else gid[keep]
            order = keep[np.lexsort((key2, owner[keep]))]
            start = np.zeros(n + 1, dtype=np.int64)
            np.add.at(start, owner[keep] + 1, 1)
            return np.cumsum(start), order.astype(np.int64)

        g["FrStart"], g["FrIdx"] = csr(lfr, Bl, self.branches)
        g["ToStart"], g["ToIdx"] = csr(lto, Bl, self.branches)
        g["GenStart"], g["GenIdx"] = csr(bus_l[gen_bus[self.gens]], Bl)
        self.grid = g
        self.Gl, self.Ll, self.Bl = Gl, Ll, Bl
        self.nvar_local = 2 * Gl + 8 * Ll
        self.nvar_owned = 2 * Gl + 8 * self.n_native
        self.G_global = G

        # the one exchange of an iteration, per peer and cut branch: the 4 v entries of the LOCAL end go out,
        # the 4 of the REMOTE end come in (step-wise twin of dist_native.h: exaadmm_set_halo_index slots 0 / 1)
        ls = 2 * Gl
        fs, ts = np.asarray(FROM_SIDE, dtype=np.int64), np.asarray(TO_SIDE, dtype=np.int64)

        def plan(local_end):
            idx, counts = [], []
            for p in range(world):
                b = self.cut[p]
                loc = br_l[b]
                native = fr_own[b] == rank
                side = np.where((native == local_end)[:, None], fs[None, :], ts[None, :])
                i = (ls + 8 * loc[:, None] + side).ravel()
                idx.append(i)
                counts.append(int(i.size))
            idx = np.concatenate(idx).astype(np.int64)
            return (np.full(idx.size, V, dtype=np.int32), idx, counts)

        self.plan_send_v = plan(True)     # slot 0: gather the local ends' v entries
        self.plan_recv_v = plan(False)    # slot 1: scatter the peers' into the remote ends
        self.cut_local = np.concatenate([br_l[self.cut[p]] for p in range(world)]).astype(np.int64)
        self.cut_counts = np.asarray([len(self.cut[p]) for p in range(world)], dtype=np.int64)

    # ---- global <-> local vectors (reference layout)
    def local_index_map(self):
        """global reference-layout index of every local entry"""
        G = self.G_global
        gi = np.empty(self.nvar_local, dtype=np.int64)
        gi[0:2 * self.Gl:2] = 2 * self.gens
        gi[1:2 * self.Gl:2] = 2 * self.gens + 1
        gi[2 * self.Gl:] = (2 * G + 8 * self.branches[:, None] + np.arange(8)[None, :]).ravel()
        return gi

    def slice_global(self, vec):
        return np.ascontiguousarray(np.asarray(vec)[self.local_index_map()])


# --------------------------------------------------------------------------- GPU backend
class HipBackend:
    """One partition on one MI355X through the C ABI; halo/reduction buffers are torch tensors."""

    def __init__(self, pkg, part, rho_pq, rho_va, params=None, device_index=0):
        import torch
        self.torch = torch
        self.pkg = pkg
        self.part = part
        self.dev = torch.device("cuda", device_index)
        torch.cuda.set_device(self.dev)
        env = pkg.AdmmEnv("partition", rho_pq, rho_va, gpu_no=device_index, verbose=0, grid=part.grid)
        if params:
            for k, v in params.items():
                setattr(env.params, k, v)
        self.env = env
        self.mod = pkg.ModelAcopf(env)
        self.lib = pkg.library()
        h = self.mod.handle
        for slot, (which, idx, _) in enumerate((part.plan_send_v, part.plan_recv_v)):
            pkg._check(self.lib.exaadmm_set_halo_index(h, slot, which.ctypes.data_as(C.POINTER(C.c_int32)),
                                                       idx.ctypes.data_as(pkg.c_i64_p), len(idx)), "exaadmm_set_halo_index")
        self.sums = torch.zeros(4, dtype=torch.float64, device=self.dev)
        self._use_torch_stream()

    def _use_torch_stream(self):
        """step-wise twin: the library's launches go onto torch's current stream, asynchronously, so that torch's
        collectives order themselves against them"""
        h = self.mod.handle
        stream = self.torch.cuda.current_stream(self.dev).cuda_stream
        self.pkg._check(self.lib.exaadmm_set_stream(h, C.c_void_p(stream)), "exaadmm_set_stream")
        self.pkg._check(self.lib.exaadmm_set_async(h, 1), "exaadmm_set_async")

    # ---- distributed loop inside the library (csrc/dist_native.h): RCCL called from C++ on the handle's own stream
    native = False

    def attach_native(self, dist, connect=True):
        """Collective over the ranks of `dist` (torch.distributed, used only to hand the 128-byte RCCL id around and
        to agree on failures): creates the library-side communicator and message buffers.  From here on run_inner /
        run_fixed / norm_z run inside the library.  Returns False -- on EVERY rank, with the handle back on torch's
        stream -- when any rank could not attach; the step-wise loop then serves as the fallback."""
        pkg, lib, h, p = self.pkg, self.lib, self.mod.handle, self.part
        world, rank = dist.get_world_size(), dist.get_rank()

        def all_ok(ok):
            if world == 1:
                return bool(ok)
            t = self.torch.tensor([1 if ok else 0], dtype=self.torch.int32, device=self.dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(int(t.cpu()[0]))

        def fail(where, err):
            if err is not None:
                sys.stderr.write("exaadmm: rank %d: %s failed (%s)\n" % (rank, where, err))
            try:
                lib.exaadmm_dist_detach(h)
            except Exception:
                pass
            self._use_torch_stream()
            self.native = False
            return False

        self.torch.cuda.synchronize(self.dev)
        pkg._check(lib.exaadmm_set_stream(h, C.c_void_p(pkg.OWN_STREAM)), "exaadmm_set_stream")
        pkg._check(lib.exaadmm_set_async(h, 0), "exaadmm_set_async")
        # phase 1, local: bind RCCL, check the cut lists against the grid, allocate the message buffers
        err = None
        try:
            pkg._check(lib.exaadmm_dist_prepare(h, world, rank, p.nvar_global, p.cut_counts.ctypes.data_as(pkg.c_i64_p),
                                                p.cut_local.ctypes.data_as(pkg.c_i64_p)), "exaadmm_dist_prepare")
        except pkg.ExaAdmmError as e:
            err = e
        if not all_ok(err is None):
            return fail("exaadmm_dist_prepare", err)
        one_rank_comm = world == 1 and os.environ.get("EXAADMM_DIST_ONE_RANK_COMM")   # tests: a real one-rank communicator
        # connect = False: no RCCL communicator at all -- the partition then lives on its peers' mailboxes alone (resident loop, sums
        # through the mailboxes) and has no two-launch loop to fall back to (tests: two PROCESSES on one GPU, where RCCL refuses to connect)
        if (world > 1 and connect) or one_rank_comm:
            # phase 2: the id (rank 0), agreed on before anybody enters ncclCommInitRank
            ident = [None]
            if rank == 0:
                buf = C.create_string_buffer(128)
                try:
                    pkg._check(lib.exaadmm_dist_unique_id(buf, 128), "exaadmm_dist_unique_id")
                    ident[0] = buf.raw
                except pkg.ExaAdmmError as e:
                    err = e
            if world > 1:
                dist.broadcast_object_list(ident, src=0)
            if ident[0] is None:
                return fail("exaadmm_dist_unique_id", err)
            # phase 3, collective: every rank is known to be ready
            try:
                pkg._check(lib.exaadmm_dist_connect(h, ident[0], 128), "exaadmm_dist_connect")
            except pkg.ExaAdmmError as e:
                err = e
            if not all_ok(err is None):
                return fail("exaadmm_dist_connect", err)
        self.native = True
        # phase 4 (optional): the mailboxes of the resident loop.  Every rank's 256-byte description of its mailbox goes to every rank;
        # a rank maps its peers' (same process: by address; other processes: IPC handle + peer access).  Only if EVERY rank could map
        # EVERY peer does the partition run the resident loop (what crosses GPUs then goes by direct stores over xGMI from inside the
        # launch); otherwise all keep two launches per iteration + the RCCL exchange.
        self.p2p = False
        if world > 1 and os.environ.get("EXAADMM_DIST_P2P", "1") != "0" and self.mod.get_option("dist_p2p"):
            err, blobs = None, [None] * world
            mine = C.create_string_buffer(256)
            try:
                pkg._check(lib.exaadmm_dist_p2p_export(h, mine, 256), "exaadmm_dist_p2p_export")
            except pkg.ExaAdmmError as e:
                err = e
            dist.all_gather_object(blobs, mine.raw if err is None else None)
            if err is None and all(b is not None for b in blobs):
                try:
                    pkg._check(lib.exaadmm_dist_p2p_import(h, b"".join(blobs), 256), "exaadmm_dist_p2p_import")
                except pkg.ExaAdmmError as e:
                    err = e
            else:
                err = err or pkg.ExaAdmmError("a peer has no mailbox")
            self.p2p = all_ok(err is None)
            if not self.p2p:
                if err is not None and rank == 0:
                    sys.stderr.write("exaadmm: rank %d: no mailboxes between the ranks (%s): two launches per iteration + RCCL exchange\n" % (rank, err))
                self.mod.set_option("dist_p2p", 0)
        else:
            self.mod.set_option("dist_p2p", 0)
        return True

    def run_inner_native(self, outer, max_inner, beta):
        done = C.c_int()
        res = np.zeros(4)
        self.pkg._check(self.lib.exaadmm_run_inner(self.mod.handle, int(outer), int(max_inner), float(beta), C.byref(done),
                                                   self.pkg._dp(res)), "exaadmm_run_inner")
        return done.value, res

    def run_fixed_native(self, iters, first_major_iter, beta):
        res = np.zeros(4)
        ms = C.c_double()
        self.pkg._check(self.lib.exaadmm_run_fixed(self.mod.handle, int(iters), int(first_major_iter), float(beta),
                                                   self.pkg._dp(res), C.byref(ms), None), "exaadmm_run_fixed")
        return res, ms.value

    def norm_z_native(self):
        v = C.c_double()
        self.pkg._check(self.lib.exaadmm_norm_z(self.mod.handle, C.byref(v)), "exaadmm_norm_z")
        return v.value

    def new_buffer(self, n):
        return self.torch.zeros(max(int(n), 1), dtype=self.torch.float64, device=self.dev)

    def upload(self, name, vec):
        self.mod.upload(name, vec)

    def download(self, name):
        self.torch.cuda.synchronize(self.dev)
        return self.mod.download(name)

    def set_check_every(self, k):
        self.env.params.check_every = int(k)
        self.mod.sync_params()

    def ctrl_begin(self, outer, max_inner, beta):
        self.pkg._check(self.lib.exaadmm_ctrl_begin(self.mod.handle, outer, max_inner, beta, self.part.nvar_global), "ctrl_begin")

    def ctrl_step(self):
        self.pkg._check(self.lib.exaadmm_ctrl_step(self.mod.handle, C.c_void_p(self.sums.data_ptr())), "ctrl_step")

    def ctrl_read(self):
        done, inner = C.c_int(), C.c_int()
        res = np.zeros(4)
        self.pkg._check(self.lib.exaadmm_ctrl_read(self.mod.handle, C.byref(done), C.byref(inner), self.pkg._dp(res)), "ctrl_read")
        return bool(done.value), inner.value, res

    def ctrl_end(self):
        self.lib.exaadmm_ctrl_end(self.mod.handle)

    def update_x(self, major_iter):
        self.pkg._check(self.lib.exaadmm_update_x(self.mod.handle, major_iter), "exaadmm_update_x")

    def update_xbar(self):
        self.pkg._check(self.lib.exaadmm_update_xbar(self.mod.handle), "exaadmm_update_xbar")

    def pack(self, slot, buf):
        self.pkg._check(self.lib.exaadmm_halo_pack(self.mod.handle, slot, C.c_void_p(buf.data_ptr())), "halo_pack")

    def unpack(self, slot, buf):
        self.pkg._check(self.lib.exaadmm_halo_unpack(self.mod.handle, slot, C.c_void_p(buf.data_ptr())), "halo_unpack")

    def update_zl_partial(self, beta):
        self.pkg._check(self.lib.exaadmm_update_zl_partial(self.mod.handle, beta, C.c_void_p(self.sums.data_ptr())), "update_zl_partial")
        return self.sums

    def sumsq_z_partial(self):
        self.pkg._check(self.lib.exaadmm_sumsq_z_partial(self.mod.handle, C.c_void_p(self.sums.data_ptr())), "sumsq_z_partial")
        return self.sums

    def update_lz(self, beta):
        self.pkg._check(self.lib.exaadmm_update_lz(self.mod.handle, beta), "exaadmm_update_lz")

    def objective_partial(self):
        v = C.c_double()
        self.torch.cuda.synchronize(self.dev)
        self.pkg._check(self.lib.exaadmm_objective(self.mod.handle, C.byref(v)), "exaadmm_objective")
        return v.value


# --------------------------------------------------------------------------- orchestration
class DistributedAcopf:
    """admm_two_level over P partitions.  `dist` is torch.distributed (initialised by the caller);
    `backend` implements the HipBackend interface."""

    def __init__(self, part, backend, dist, global_state, params, check_every=1):
        self.part, self.be, self.dist = part, backend, dist
        # check_every = 1: the reference's per-iteration break test (default).  k > 1 is an explicit
        # opt-in relaxation of the step-wise loop: the residual all-reduce (and the test) only happen every k-th inner
        # iteration, the loop may run up to k-1 iterations longer than the reference's.
        self.check_every = max(1, int(check_every))
        backend.set_check_every(self.check_every)
        self.par = params            # dict: initial_beta, inc_c, theta, outer_eps, outer_iterlim, inner_iterlim, MAX_MULTIPLIER
        for name in ("v_curr", "rho"):
            backend.upload(name, part.slice_global(global_state[name]))
        p = part
        self.buf_send_v = backend.new_buffer(sum(p.plan_send_v[2]))
        self.buf_recv_v = backend.new_buffer(sum(p.plan_recv_v[2]))
        self.info = {"outer": 0, "inner": 0, "cumul": 0, "status": "IterationLimit"}

    def _staged(self, t):
        """gloo cannot move device tensors: stage through the host (debug / no-RCCL fallback)"""
        return self.dist.get_backend() == "gloo" and getattr(t, "is_cuda", False)

    def _exchange(self, send, recv, send_counts, recv_counts):
        ns, nr = sum(send_counts), sum(recv_counts)
        if self.dist.get_world_size() == 1:
            return
        # every rank must enter the collective, also with empty lists
        s, r = send[:ns], recv[:nr]
        if self._staged(send):
            s_h = s.cpu()
            r_h = r.cpu()
            self.dist.all_to_all_single(r_h, s_h, output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts))
            r.copy_(r_h)
        else:
            self.dist.all_to_all_single(r, s, output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts))

    def _all_reduce(self, t):
        if self.dist.get_world_size() == 1:
            return
        if self._staged(t):
            t_h = t.cpu()
            self.dist.all_reduce(t_h)
            t.copy_(t_h)
        else:
            self.dist.all_reduce(t)

    def inner_iteration(self, major_iter, beta, max_inner=None):
        """step-wise twin of the in-library iteration: x-update of every local branch (natives and replicas), bus
        update, ONE exchange (the v entries of the cut branches' local ends), z / l / residual on every local element
        with the sums taken over the owned ones, all-reduce of the 4 sums"""
        be, p = self.be, self.part
        be.update_x(major_iter)
        be.update_xbar()
        be.pack(0, self.buf_send_v)
        self._exchange(self.buf_send_v, self.buf_recv_v, p.plan_send_v[2], p.plan_recv_v[2])
        be.unpack(1, self.buf_recv_v)
        sums = be.update_zl_partial(beta)
        if self.check_every == 1 or major_iter % self.check_every == 0 or major_iter == max_inner:
            self._all_reduce(sums)      # otherwise the (untested) local sums are only stored
        be.ctrl_step()

    def run_inner(self, outer, max_inner, beta, batch=8):
        """inner while-loop of admm_two_level.jl:34-63; loop control on the device, one host
        round trip per `batch` iterations; exact per-iteration break semantics."""
        be = self.be
        if getattr(be, "native", False):      # the whole loop runs inside the library (RCCL from C++)
            return be.run_inner_native(outer, max_inner, beta)
        be.ctrl_begin(outer, max_inner, beta)
        enq = 0
        done, inner, res = False, 0, np.zeros(4)
        while enq < max_inner and not done:
            nb = min(batch, max_inner - enq)
            for i in range(nb):
                self.inner_iteration(enq + i + 1, beta, max_inner)   # major_iter is re-derived on the device when armed
            enq += nb
            done, inner, res = be.ctrl_read()
        be.ctrl_end()
        return inner, res

    def norm_z(self):
        if getattr(self.be, "native", False):
            return self.be.norm_z_native()
        sums = self.be.sumsq_z_partial()
        self._all_reduce(sums)
        return math.sqrt(float(sums.cpu()[2]))

    def solve(self):
        """admm_two_level (algorithms/admm_two_level.jl:1-88), replicated scalar logic."""
        par, info, p = self.par, self.info, self.part
        sqrt_d = math.sqrt(p.nvar_global)
        OUTER_TOL = sqrt_d * par["outer_eps"]
        beta = par["initial_beta"]
        info.update(outer=0, inner=0, cumul=0, status="IterationLimit")
        t0 = time.perf_counter()
        while info["outer"] < par["outer_iterlim"]:
            info["outer"] += 1
            norm_z_prev = self.norm_z()
            inner, res = self.run_inner(info["outer"], par["inner_iterlim"], beta)
            info["inner"] = inner
            info["cumul"] += inner
            info["primres"], info["dualres"], info["norm_z_curr"], info["mismatch"] = (float(x) for x in res)
            if info["mismatch"] <= OUTER_TOL:
                info["status"] = "Solved"
                break
            self.be.update_lz(beta)
            if info["norm_z_curr"] > par["theta"] * norm_z_prev:
                beta = min(par["inc_c"] * beta, 1e24)
        info["time_overall"] = time.perf_counter() - t0
        info["beta"] = beta
        return info

    def gather_owned(self, name):
        """all ranks: global reference-layout vector assembled from every rank's owned entries"""
        import torch
        p = self.part
        loc = np.asarray(self.be.download(name))[:p.nvar_owned]
        gi = p.local_index_map()[:p.nvar_owned]
        out = np.zeros(p.nvar_global)
        out[gi] = loc
        t = torch.from_numpy(out)
        if self.dist.get_world_size() > 1:
            if self.dist.get_backend() == "nccl":
                t = t.cuda()
            self.dist.all_reduce(t)
            t = t.cpu()
        return t.numpy()


DEFAULT_PAR = dict(initial_beta=1e3, inc_c=6.0, theta=0.8, outer_eps=2e-4, outer_iterlim=20,
                   inner_iterlim=1000, MAX_MULTIPLIER=1e12)


# --------------------------------------------------------------------------- bench (N > 1)
def bench_main(args, rank, world, local_rank):
    """bench.py --gpus N [--case C] [--scaling weak|strong].
    weak (default): one banded synthetic grid of N areas of the case's size (N x ngen, N x nline, N x nbus) cut into N bus
      ranges, one per GPU; the cut branches are the tie lines between areas; value = N x (inner iterations/s of the
      N-area grid) = area-iterations/s.  `--case case_ACTIVSg70k` gives north_star's target configuration: N areas of
      70k buses.
    strong: ONE grid of the case's size sharded N ways (BASELINE.json configs 4 and 5 as written); value = inner
      iterations/s of that grid.
    The per-iteration sequence runs inside the library (csrc/dist_native.h, RCCL called from C++)."""
    import torch
    import torch.distributed as dist
    import exaadmm_loader
    pkg = exaadmm_loader.load_package()
    syn = exaadmm_loader.load_submodule("synthetic")
    import bench as B

    torch.cuda.set_device(local_rank)
    # tests of the supervisor (bench_supervisor.py): "hang:<attempt>" / "fail:<attempt>" -- this child, having touched the GPU, never
    # answers / exits non-zero in that attempt
    inject = os.environ.get("EXAADMM_BENCH_INJECT", "")
    if inject and inject.split(":")[1] == os.environ.get("EXAADMM_BENCH_ATTEMPT", "0"):
        torch.zeros(8, device="cuda").sum().item()
        if inject.startswith("hang"):
            while True:
                time.sleep(1.0)
        raise SystemExit(3)
    # python-gloo mode of the supervisor: no RCCL anywhere, device tensors staged through the host
    backend = os.environ.get("EXAADMM_BENCH_BACKEND", "nccl")
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    world = dist.get_world_size()
    red_dev = "cpu" if dist.get_backend() == "gloo" else "cuda"   # where the job's scalar reductions live
    leg = getattr(args, "leg", "main")
    light = (leg != "main")   # an extra leg of the line (north_star_scaling): bounded to about a minute, no CPU baseline
    params = B.CASE_PARAMS[args.case]
    g0, l0, b0 = syn.CASE_SIZES[args.case]
    sharded = (getattr(args, "scaling", "weak") == "strong")
    if sharded:
        grid = syn.sized_like(args.case, seed=0)
        workload = "synthetic-%s-sized sharded %d ways" % (args.case, world)
    else:
        grid = syn.sized_like(args.case, seed=0, areas=world)
        workload = "synthetic-%s-sized x %d areas" % (args.case, world)
    part = Partition(grid, world, rank)
    st = init_state(grid, params["rho_pq"], params["rho_va"])
    be = HipBackend(pkg, part, params["rho_pq"], params["rho_va"], params={"scale": params["scale"]}, device_index=local_rank)
    be.mod.sync_params()
    model = DistributedAcopf(part, be, dist, st, dict(DEFAULT_PAR))
    native = os.environ.get("EXAADMM_DIST_PYTHON_LOOP") is None and dist.get_backend() != "gloo"
    if native:
        native = bool(be.attach_native(dist))
    beta = DEFAULT_PAR["initial_beta"]
    exch0 = int(be.lib.exaadmm_dist_exchange_count())

    def run(n, first):
        if n <= 0:
            return
        if native:
            be.run_fixed_native(n, first, beta)
        else:
            for i in range(n):                           # step-wise twin: Python issues every step and collective
                model_inner_fixed(model, first + i, beta)

    def model_inner_fixed(m, major_iter, beta):
        bk, p = m.be, m.part
        bk.update_x(major_iter)
        bk.update_xbar()
        bk.pack(0, m.buf_send_v)
        m._exchange(m.buf_send_v, m.buf_recv_v, p.plan_send_v[2], p.plan_recv_v[2])
        bk.unpack(1, m.buf_recv_v)
        sums = bk.update_zl_partial(beta)
        m._all_reduce(sums)

    prep = max(0, getattr(args, "prep", 0))     # untimed preparation iterations that define the benchmark state
    # (the ranks enter their first launch together: a rank that waited a second for a peer still busy on its host would give the launch
    #  up -- capped waits -- and every rank would fall back to two launches per iteration for the rest of the job)
    torch.cuda.synchronize()
    dist.barrier()
    run(prep, 1)
    run(args.warmup, prep + 1)
    # timed region: exactly K steps between barrier + synchronize on both sides, MAX over the ranks.  Like the one-GPU line
    # the region is repeated, every time from the same ADMM state (every rank restores its own device-side snapshot --
    # the replicas of the cut branches stay consistent: they were when the snapshot was taken), until about a second of
    # GPU time has been spent, and the median region is reported.
    can_snapshot = native
    if can_snapshot:
        pkg._check(be.lib.exaadmm_state_save(be.mod.handle), "exaadmm_state_save")

    def region():
        if can_snapshot:
            pkg._check(be.lib.exaadmm_state_restore(be.mod.handle), "exaadmm_state_restore")
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.steps, prep + args.warmup + 1)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.cpu()[0])

    exch1 = int(be.lib.exaadmm_dist_exchange_count())
    regions = [region()]
    exch2 = int(be.lib.exaadmm_dist_exchange_count())
    want = getattr(args, "repeats", 0)
    gpu_seconds = min(2.0, getattr(args, "gpu_seconds", 1.0)) if light else getattr(args, "gpu_seconds", 1.0)
    repeats = 1 if (want == 1 or not can_snapshot) else (want if want > 1 else max(1, min(4000, int(round(gpu_seconds / max(regions[0], 1e-6))))))
    for _ in range(repeats - 1):       # (the same count on every rank: regions[0] is the all-reduced maximum)
        regions.append(region())
    dt = sorted(regions)[len(regions) // 2]
    if native:
        res, _ = be.run_fixed_native(1, prep + args.warmup + args.steps + 1, beta)     # norms of one more iteration
    else:
        res = torch.sqrt(be.sums).cpu().numpy()
    cuts = torch.tensor([part.n_cut, part.n_native], dtype=torch.float64, device=red_dev)
    dist.all_reduce(cuts)
    # what the in-library loop did: ranks of the communicator as RCCL counts them, RCCL version, duration of the sampled exchanges
    # (grouped send / receive + unpack kernel, HIP events on the exchange stream), batches replayed from a hipGraph / enqueued directly
    rccl = None
    exchange_us = None
    if native:
        n_, v_, gb_, db_, pr_ = C.c_int(), C.c_int(), C.c_int64(), C.c_int64(), C.c_int()
        us_ = C.c_double()
        pkg._check(be.lib.exaadmm_dist_stats(be.mod.handle, C.byref(n_), C.byref(v_), C.byref(us_), C.byref(gb_), C.byref(db_), C.byref(pr_)), "exaadmm_dist_stats")
        xs = torch.tensor([us_.value], dtype=torch.float64, device=red_dev)
        dist.all_reduce(xs, op=dist.ReduceOp.MAX)
        exchange_us = float(xs.cpu()[0]) or None
        rccl = {"nranks": n_.value, "version": v_.value, "path": be.lib.exaadmm_dist_rccl_path().decode(),
                "capture_probe": {1: "ok: batches of 12 iterations incl. their exchanges replayed from one hipGraph", 0: "refused: batches enqueued directly",
                                  -1: "not run (one rank)"}.get(pr_.value),
                "graph_batches_rank0": gb_.value, "direct_batches_rank0": db_.value}
    # N > 1: what ONE GPU does on one area with the loop the partitions run (two launches per iteration; the single-GPU bench
    # line uses the resident loop, one launch per batch): separates the cost of communication from the change of loop
    n1_same_loop = None
    cpu_base = None
    if rank == 0 and (world > 1 or getattr(args, "force_dist", False)):
        try:
            g1 = syn.sized_like(args.case, seed=0)
            env1 = pkg.AdmmEnv(args.case, params["rho_pq"], params["rho_va"], gpu_no=local_rank, verbose=0, grid=g1)
            m1 = pkg.ModelAcopf(env1)
            m1.set_option("resident", 0)
            env1.params.scale = params["scale"]
            m1.sync_params()
            m1.run_fixed(prep + args.warmup, first_major_iter=1, beta=beta)
            m1.state_save()
            ts = []
            for _ in range(10 if light else 30):
                m1.state_restore()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                m1.run_fixed(args.steps, first_major_iter=prep + args.warmup + 1, beta=beta)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            n1_same_loop = {"iterations_per_s": args.steps / sorted(ts)[len(ts) // 2], "loop_kind": m1.loop_kind(),
                            "what": "one case-sized area on one GPU, two launches per iteration (option resident = 0), same state, median of %d regions" % len(ts)}
            m1.close()
            if not getattr(args, "no_cpu_baseline", False) and not light:
                cpu_base = B.cpu_baseline(g1, params, budget_s=10.0, prep=min(prep, 100))
                cpu_base["sample"] += " -- ONE case-sized area (what one GPU of the job iterates on)"
        except Exception as e:   # never lose the line over the extras
            sys.stderr.write("bench: extras failed: %r\n" % (e,))
    # every rank pushes out whatever the runtime libraries buffered on C stdio (RCCL version banner) BEFORE
    # rank 0 prints, so that the JSON line is the last line of the job's stdout
    try:
        sys.stdout.flush()
        C.CDLL(None).fflush(None)
    except Exception:
        pass
    dist.barrier()
    if rank == 0:
        its = args.steps / dt
        G, Ln, Bn = int(grid["ngen"]), int(grid["nline"]), int(grid["nbus"])
        value = its if sharded else world * its
        out = {
            "metric": "ADMM inner iterations/sec", "value": value, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "value_definition": ("inner iterations/s of the one grid, sharded over the GPUs" if sharded else
                         "N x (inner iterations/s of the N-area grid): iterations/s per case-sized area, summed over the N areas "
                         "(equal to plain iterations/s at N = 1); the N-area grid itself iterates at grid_iterations_per_s"),
            "grid_iterations_per_s": its, "repeats": len(regions),
            "regions": {"first": args.steps / regions[0], "median": its, "min": args.steps / max(regions), "max": args.steps / min(regions),
                        "unit": "grid iterations/s", "gpu_seconds": sum(regions)},
            "config": {"workload": workload, "ngen": G, "nline": Ln, "nbus": Bn,
                       "rho_pq": params["rho_pq"], "rho_va": params["rho_va"], "scale": params["scale"],
                       "cut_branches": int(cuts.cpu()[0]), "cut_fraction": float(cuts.cpu()[0] / max(cuts.cpu()[1], 1.0)),
                       "mode": "fixed-count inner iterations, no early break",
                       "state": "ADMM state after %d untimed preparation iterations from init_solution!" % prep,
                       "loop": ("inside the library: resident loop, one launch per run, what crosses GPUs by direct stores into the peers' mailboxes (kernels.h ResidentPeers)"
                                if (native and getattr(be, "p2p", False) and be.mod.loop_kind() == 2) else
                                "inside the library, RCCL from C++ (dist_native.h)" if native else "step-wise twin: Python-driven steps + torch collectives"),
                       "loop_kind": be.mod.loop_kind(), "mailboxes_mapped": bool(getattr(be, "p2p", False)), "loop_downgrades": be.mod.loop_downgrades(),
                       "resident_form": be.mod.resident_form(),
                       # (the forms of the resident kernel that talk to PEERS have no first-use self check -- a collective one would
                       #  need the two-launch loop's communicator and buffer sets next to the resident loop's.  What stands in its place: mailbox
                       #  words validate themselves (payload + sequence number in ONE 8-byte store: a stale or torn word cannot be taken for this
                       #  iteration's), every wait is capped, a failed launch is reported to all ranks by the launch's last kernel and the call is
                       #  repeated with two launches per iteration; predicted figures to read this line against: profiles/r06_predicted_scaling.txt)
                       "peers_form_self_check": "none (self-validating mailbox words, capped waits, agreed fallback)" if (native and getattr(be, "p2p", False)) else None,
                       "exchanges_per_iteration_rank0": (exch2 - exch1) / float(max(args.steps, 1)) if native else None,
                       "overlap": "exchange + cut-branch solves on a second stream beside the interior branch solves of the next iteration"
                                  if be.mod.get_option("dist_overlap") else "off (option dist_overlap = 0): one stream",
                       "collectives": dist.get_backend(),
                       "parallelism": "bus/branch partition x%d (breadth-first bus ranges); cut branches solved on both owners; per "
                                      "iteration ONE grouped ncclSend/ncclRecv per peer (4 v entries per cut branch + 4 residual sums), "
                                      "sums added in rank order, no all-reduce" % world},
            "roofline": {"bound": "hbm", "achieved": B.BYTES_PER_ITER(G, Ln, Bn) * its / 1e9, "peak": world * B.HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": B.BYTES_PER_ITER(G, Ln, Bn) * its / 1e9 / (world * B.HBM_PEAK_GBS),
                         "traffic": None, "note": "whole-iteration algorithmic bytes of the grid / wall time"},
            "solver": {"primres": float(res[0])},
            "rccl": rccl, "exchange_us": exchange_us, "n1_same_loop": n1_same_loop,
        }
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
            out["speedup_vs_cpu_baseline_per_area"] = its / cpu_base["value"]
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    dist.barrier()
    if native:
        pkg._check(be.lib.exaadmm_dist_detach(be.mod.handle), "exaadmm_dist_detach")
    dist.destroy_process_group()
    try:
        sys.stdout.flush()
        C.CDLL(None).fflush(None)
    except Exception:
        pass
