"""Host-side mirror of the back-end's optimisation call surface.

  SlamGraphOptimizer.optimize(OptParams)  <- SlamGraph::optimize   slam_graph.hpp:457-462,
                                                                   slam_graph.cpp:312-355

The double-window bookkeeping that produces the window / active point set
(prepareForOptimization, slam_graph.cpp:288-310) is out of scope; this class
takes its output the way copyDataToG2o consumes it: poses, inverse-depth points, observation
edges, pose-pose constraints.

Multi-GPU: landmarks shard across ranks, each rank reduces its own landmarks into a
partial reduced camera system, ONE all-reduce (RCCL through torch.distributed; "nccl" backend on
ROCm) sums the packed system, every rank runs the identical 6Px6P Cholesky, back-substitutes its
own landmarks; two sums per LM trial (32 doubles: 16 partial slots each) are all-reduced for the accept/reject decision.
"""
import ctypes as C

import numpy as np
import torch

from . import capi
from .ctypes_types import BA_CONSTRAINT_DTYPE, BA_EDGE_DTYPE, BaParams, BaStats, Cam


class OptParams:
    """slam_graph.hpp:36-50.  huber_kernel_width is accepted and ignored, as in the reference
    (slam_graph-impl.cpp:86-90 never passes it to the kernel; delta stays g2o's default 1)."""

    def __init__(self, num_iters=2, use_robust_kernel=True, huber_kernel_width=3.0):
        self.num_iters, self.use_robust_kernel, self.huber_kernel_width = num_iters, use_robust_kernel, huber_kernel_width


def make_allreduce(stream, device, group=None):
    """svs_allreduce_fn backed by torch.distributed: sums `count` doubles in place on `stream`."""
    import torch.distributed as dist

    def _fn(d_buf, count, _user):
        try:
            # zero-copy view of the library's device buffer
            t = _as_tensor(d_buf, count, device)
            with torch.cuda.stream(stream):
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print("allreduce callback failed:", repr(e))
            return 1

    return capi.ALLREDUCE_FN(_fn)


class Communicator:
    """svs_comm: the library-owned RCCL communicator of the landmark-sharded back-end.  The unique id travels from rank 0 to the
    other ranks through torch.distributed (any backend; here only as the out-of-band channel a C++ host would replace by MPI or a
    TCP store) -- the collectives of svs_ba_optimize themselves are ncclAllReduce calls made by the library on its own stream."""

    def __init__(self, ctx, rank, world, device=None, group=None):
        import torch.distributed as dist
        self.ctx, self.rank, self.world = ctx, rank, world
        uid = np.zeros(128, np.uint8)
        if rank == 0:
            ctx.call("svs_comm_get_unique_id", uid.ctypes.data)
        if world > 1:
            t = torch.as_tensor(uid)
            if device is not None:
                t = t.to(device)
            dist.broadcast(t, src=0, group=group)
            uid = t.cpu().numpy().copy()
        self.h = C.c_void_p()
        ctx.check(ctx.lib.svs_comm_create(ctx.h, uid.ctypes.data, rank, world, C.byref(self.h)))
        ctx.children.add(self)

    @classmethod
    def p2p(cls, ctx, rank, world, capacity_doubles=1 << 16, group=None, device=None):
        """The one-shot transport (svs_comm_create_p2p / svs_comm_connect_p2p): peer-mapped mailboxes instead of RCCL.  The 64-byte IPC handles are
        all-gathered through torch.distributed -- CPU tensors (a gloo group) or, with `device`, tensors on that GPU (the nccl backend); a C++ host would use
        MPI or a TCP store.  Every rank learns whether ALL ranks could map their peers (a MIN all-reduce, so that nobody is left waiting in a barrier for a
        rank that raised): on failure every rank raises."""
        import torch.distributed as dist
        self = cls.__new__(cls)
        self.ctx, self.rank, self.world = ctx, rank, world
        self.h = C.c_void_p()
        mine = np.zeros(64, np.uint8)
        err = None
        try:
            ctx.check(ctx.lib.svs_comm_create_p2p(ctx.h, rank, world, capacity_doubles, C.byref(self.h), mine.ctypes.data))
            ctx.children.add(self)
        except Exception as e:              # e.g. hipIpcGetMemHandle refused: this rank still takes part in the two collectives below
            err = e
            self.h = None
            if world == 1:
                raise
        if world > 1:
            dev = torch.device("cpu") if device is None else device
            out = [torch.zeros(64, dtype=torch.uint8, device=dev) for _ in range(world)]
            dist.all_gather(out, torch.as_tensor(mine).to(dev), group=group)
            handles = np.ascontiguousarray(np.stack([t.cpu().numpy() for t in out]))
            if err is None:
                try:
                    ctx.check(ctx.lib.svs_comm_connect_p2p(self.h, handles.ctypes.data))
                except Exception as e:      # e.g. no peer access between two devices
                    err = e
            ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)      # also the barrier: every mailbox is mapped everywhere before the first push
            if int(ok.item()) == 0:
                self.close()
                raise err if err else RuntimeError("one-shot P2P transport: another rank could not map its peers' mailboxes")
        return self

    def transport(self):
        k, t = C.c_int32(), C.c_uint32()
        self.ctx.check(self.ctx.lib.svs_comm_transport(self.h, C.byref(k), C.byref(t)))
        return dict(kind="p2p" if k.value else "rccl", timeouts=t.value,
                    mailbox_memory={0: None, 1: "fine-grained", 2: "uncached", 3: "coarse-grained (hipMalloc)"}.get(k.value))

    def allreduce(self, d_ptr, count):
        self.ctx.check(self.ctx.lib.svs_comm_allreduce_f64(self.h, d_ptr, count))

    def stats(self):
        r, w, n, d = C.c_int32(), C.c_int32(), C.c_uint64(), C.c_uint64()
        self.ctx.lib.svs_comm_stats(self.h, C.byref(r), C.byref(w), C.byref(n), C.byref(d))
        return dict(rank=r.value, world=w.value, n_calls=n.value, n_doubles=d.value)

    def close(self):
        if self.h and self.ctx.h:
            self.ctx.lib.svs_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _CudaArray:
    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def _as_tensor(ptr, count, device):
    return torch.as_tensor(_CudaArray(ptr, count), device=torch.device("cuda", device))


class SlamGraphOptimizer:
    def __init__(self, ctx, stream=None):
        self.ctx, self.stream = ctx, stream
        self.h = C.c_void_p()
        ctx.check(ctx.lib.svs_ba_create(ctx.h, C.byref(self.h)))
        ctx.children.add(self)
        self.P = self.L = 0

    def copyDataToG2o(self, poses, psi, edges, cons, cam, prm=None, add_pose_terms=True):
        """copyDataToG2o + setupG2o (slam_graph.cpp:983-1032,1061-1080): upload the window."""
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 12)
        psi = np.ascontiguousarray(psi, np.float64).reshape(-1, 3)
        edges = np.ascontiguousarray(edges, BA_EDGE_DTYPE)
        cons = np.ascontiguousarray(cons if cons is not None else np.zeros(0, BA_CONSTRAINT_DTYPE), BA_CONSTRAINT_DTYPE)
        self.P, self.L = len(poses), len(psi)
        self.prm = prm or BaParams.reference_defaults()
        camc = cam if isinstance(cam, Cam) else Cam(cam["f"], cam["cx"], cam["cy"], cam["b"], cam["w"], cam["h"])
        self.ctx.check(self.ctx.lib.svs_ba_set_problem(
            self.h, self.P, poses.ctypes.data, self.L, psi.ctypes.data, len(edges), edges.ctypes.data, len(cons),
            cons.ctypes.data, C.byref(camc), C.byref(self.prm), int(add_pose_terms)))

    def windowUpdate(self, pose_ids, poses, point_ids, psi, anchor_pose_ids, new_obs, cons, cam, prm=None):
        """svs_ba_window_update: the persistent-window form of copyDataToG2o -- ids define the window, only the observations made since
        the last call are handed over (new_obs: BA_EDGE_DTYPE with ids in point / pose; cons: BA_CONSTRAINT_DTYPE with frame ids)."""
        pose_ids = np.ascontiguousarray(pose_ids, np.int32)
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 12)
        point_ids = np.ascontiguousarray(point_ids, np.int32)
        psi = np.ascontiguousarray(psi, np.float64).reshape(-1, 3)
        anchor_pose_ids = np.ascontiguousarray(anchor_pose_ids, np.int32)
        new_obs = np.ascontiguousarray(new_obs, BA_EDGE_DTYPE)
        cons = np.ascontiguousarray(cons if cons is not None else np.zeros(0, BA_CONSTRAINT_DTYPE), BA_CONSTRAINT_DTYPE)
        self.P, self.L = len(poses), len(psi)
        self.prm = prm or BaParams.reference_defaults()
        camc = cam if isinstance(cam, Cam) else Cam(cam["f"], cam["cx"], cam["cy"], cam["b"], cam["w"], cam["h"])
        self.ctx.check(self.ctx.lib.svs_ba_window_update(
            self.h, self.P, pose_ids.ctypes.data, poses.ctypes.data, self.L, point_ids.ctypes.data, psi.ctypes.data, anchor_pose_ids.ctypes.data,
            len(new_obs), new_obs.ctypes.data, len(cons), cons.ctypes.data, C.byref(camc), C.byref(self.prm)))

    def windowReset(self):
        self.ctx.check(self.ctx.lib.svs_ba_window_reset(self.h))

    def windowForgetKeyframes(self, pose_ids):
        """drop the stored observations of keyframes that will never be in a window again"""
        ids = np.ascontiguousarray(pose_ids, np.int32)
        self.ctx.check(self.ctx.lib.svs_ba_window_forget_keyframes(self.h, ids.ctypes.data, len(ids)))

    def reset_state(self, poses, psi):
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 12)
        psi = np.ascontiguousarray(psi, np.float64).reshape(-1, 3)
        self.ctx.check(self.ctx.lib.svs_ba_reset_state(self.h, poses.ctypes.data, psi.ctypes.data))

    def optimize(self, allreduce=None):
        """optimizer.optimize(num_iters) with the LM control flow of g2o (slam_graph.cpp:336-346)."""
        st = BaStats()
        cb = allreduce if allreduce is not None else None
        self.ctx.check(self.ctx.lib.svs_ba_optimize(self.h, C.cast(cb, C.c_void_p) if cb else None, None, C.byref(st)))
        return st

    def restoreDataFromG2o(self):
        """restoreDataFromG2o (slam_graph.cpp:1035-1058): poses [P,12], psi [L,3]."""
        poses = np.zeros((self.P, 12))
        psi = np.zeros((self.L, 3))
        self.ctx.check(self.ctx.lib.svs_ba_get_state(self.h, poses.ctypes.data, psi.ctypes.data))
        return poses, psi

    def reduced_system(self, lam):
        n = 6 * self.P
        H, b, chi2 = np.zeros((n, n)), np.zeros(n), np.zeros(1)
        self.ctx.check(self.ctx.lib.svs_ba_reduced_system(self.h, float(lam), H.ctypes.data, b.ctypes.data, chi2.ctypes.data))
        return H, b, float(chi2[0])

    def set_option(self, name, value):
        """experiment / test switches of this optimizer (svs_ba_set_option)"""
        self.ctx.check(self.ctx.lib.svs_ba_set_option(self.h, name.encode(), int(value)))

    def set_comm(self, comm):
        """attach a library-owned communicator: optimize() then all-reduces with ncclAllReduce on the ctx stream"""
        self.ctx.check(self.ctx.lib.svs_ba_set_comm(self.h, comm.h if comm is not None else None))

    SOLVE_KINDS = ("global-memory blocked Cholesky", "LDS-window pipeline", "fused register-resident elimination, one front",
                   "fused register-resident elimination, two fronts", "multi-workgroup blocked Cholesky (wide envelope)",
                   "multi-workgroup tile-resident blocked Cholesky (wide envelope)")

    def info(self):
        k, r, c, w = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        self.ctx.check(self.ctx.lib.svs_ba_info(self.h, C.byref(k), C.byref(r), C.byref(c), C.byref(w)))
        o, rn = C.c_int32(), C.c_int32()
        self.ctx.check(self.ctx.lib.svs_ba_order_info(self.h, C.byref(o), C.byref(rn), None))
        return dict(solve_kernel=self.SOLVE_KINDS[k.value], envelope_rows=r.value, pose_order="reverse Cuthill-McKee" if o.value else "caller's",
                    envelope_rows_callers_order=rn.value, wave_chunks=c.value, wide_landmarks=w.value)

    def pose_order(self):
        """solver row k is the caller's pose perm[k]"""
        perm = np.zeros(self.P, np.int32)
        self.ctx.check(self.ctx.lib.svs_ba_order_info(self.h, None, None, perm.ctypes.data))
        return perm

    def set_timing(self, on):
        """hipEvent brackets around the dominant kernels of every LM trial (profiling; ~4 us per event)."""
        self.ctx.check(self.ctx.lib.svs_ba_set_timing(self.h, int(bool(on))))

    def kernel_times(self):
        r, s, b, n = C.c_float(), C.c_float(), C.c_float(), C.c_int32()
        self.ctx.lib.svs_ba_kernel_times(self.h, C.byref(r), C.byref(s), C.byref(b), C.byref(n))
        return dict(reduce_ms=r.value, solve_ms=s.value, backsub_ms=b.value, n_trials=n.value)

    def graph_stats(self):
        """(optimizes replayed from a recorded HIP graph, recordings made) -- svs_ba_graph_stats"""
        a, b = C.c_int64(), C.c_int64()
        self.ctx.check(self.ctx.lib.svs_ba_graph_stats(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def close(self):
        if self.h and self.ctx.h:
            self.ctx.lib.svs_ba_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def optimize_batch(optimizers):
    """svs_ba_optimize_batch: the windows of `optimizers` (each on its own context) optimised concurrently; returns their BaStats."""
    n = len(optimizers)
    hs = (C.c_void_p * n)(*[o.h for o in optimizers])
    st = (BaStats * n)()
    rc = optimizers[0].ctx.lib.svs_ba_optimize_batch(hs, n, st)
    if rc:
        for o in optimizers:
            o.ctx.check(rc)
    return list(st)


def shard_problem(prob, rank, world, chunk=64):
    """Landmark shard of a BA window for `rank`: landmarks dealt in
    contiguous chunks of 64, round-robin; point ids stay global; constraints live on rank 0."""
    L = len(prob["psi"])
    owner = (np.arange(L) // chunk) % world
    mask = owner[prob["edges"]["point"]] == rank
    return dict(prob, edges=prob["edges"][mask], owner=owner, add_pose_terms=(rank == 0))


def merge_sharded_psi(psi_local, owner, rank, world, device=None, group=None):
    """Every rank only updates its own landmarks; combine them (sum of masked arrays)."""
    import torch.distributed as dist
    mine = np.where((owner == rank)[:, None], psi_local, 0.0)
    t = torch.as_tensor(mine)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()
