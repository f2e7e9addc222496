"""world_size-2 (and 4, and 8 = BASELINE configs[3]'s shard count) gloo test of the landmark-sharded BA path on CPU.

The HIP kernels cannot run here, so each rank builds its shard's partial reduced camera system with
the CPU oracle (the checker) and the test exercises what is specific to N>1: the product's
shard_problem() partition, pose terms counted on exactly one rank, lambda added once after the
all-reduce, and torch.distributed's SUM all-reduce of the packed system -- the same exchange the GPU
path performs through the svs_allreduce_fn hook (scavislam_amd/backend.py)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.backend import merge_sharded_psi, shard_problem
    from scavislam_amd.ctypes_types import BaParams, Cam
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prob = synth.ba_window(8, 600, seed=31)
    c = prob["cam"]
    cam = Cam(c["f"], c["cx"], c["cy"], c["b"], c["w"], c["h"])
    prm = BaParams.reference_defaults()
    lam = 50.0
    sh = shard_problem(prob, rank, world)
    cons = sh["cons"] if sh["add_pose_terms"] else sh["cons"][:0]
    H, b = O.ba_reduced_system(sh["poses"], sh["psi"], sh["edges"], cons, cam, prm, lam)
    n = H.shape[0]
    H -= lam * np.eye(n)                                  # partial system carries no damping; added once below
    chi = O.ba_chi2(sh["poses"], sh["psi"], sh["edges"], cons, cam, prm)
    buf = torch.as_tensor(np.concatenate([H.ravel(), b, [chi]]))
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    Hs = buf[:n * n].numpy().reshape(n, n) + lam * np.eye(n)
    bs = buf[n * n:n * n + n].numpy()
    x = np.linalg.solve(Hs, bs)                            # identical replicated solve on every rank
    # each rank "updates" its own landmarks; merged with one more all-reduce
    psi_local = sh["psi"].copy()
    psi_local[sh["owner"] == rank] += 1.0 + rank
    merged = merge_sharded_psi(psi_local, sh["owner"], rank, world)
    if rank == 0:
        Hf, bf = O.ba_reduced_system(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm, lam)
        chif = O.ba_chi2(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
        ok = (np.allclose(Hs, Hf, rtol=0, atol=1e-9 * np.abs(Hf).max()) and np.allclose(bs, bf, rtol=0, atol=1e-9 * np.abs(bf).max())
              and abs(float(buf[-1]) - chif) < 1e-9 * chif
              and np.allclose(x, np.linalg.solve(Hf, bf), rtol=0, atol=1e-9 * np.abs(x).max())
              and np.allclose(merged, prob["psi"] + (1.0 + sh["owner"])[:, None]))
        q.put(bool(ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_reduced_system_allreduce_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    [p.join(180) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) is True


# ---- GPU box: two PROCESSES share the one GPU, each builds its landmark shard with the HIP kernels, the partial reduced systems
# meet in a gloo all-reduce (RCCL needs one GPU per rank; the exchange semantics are the same: SUM of the packed system on every
# rank) -- the N > 1 optimize path end to end with HIP-built shards.
def _gpu_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import ctypes as C
    import torch
    import torch.distributed as dist
    from scavislam_amd import capi, synth
    from scavislam_amd.backend import SlamGraphOptimizer, _as_tensor, shard_problem
    from scavislam_amd.ctypes_types import BaParams, Cam
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx, stream = capi.torch_context(0)
    prob = synth.ba_window(12, 1500, seed=41)
    c = prob["cam"]
    cam = Cam(c["f"], c["cx"], c["cy"], c["b"], c["w"], c["h"])
    prm = BaParams.reference_defaults()
    sh = shard_problem(prob, rank, world)
    opt = SlamGraphOptimizer(ctx, stream)
    opt.copyDataToG2o(sh["poses"], sh["psi"], sh["edges"], sh["cons"], cam, prm, add_pose_terms=sh["add_pose_terms"])
    n_calls = [0]

    def allreduce(d_buf, count, _u):
        try:
            ctx.sync()
            t = _as_tensor(d_buf, count, 0)
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            t.copy_(h)
            torch.cuda.synchronize()
            n_calls[0] += 1
            return 0
        except Exception as e:
            print("allreduce failed", repr(e))
            return 1
    st = opt.optimize(capi.ALLREDUCE_FN(allreduce))
    poses, psi = opt.restoreDataFromG2o()
    mine = np.where((sh["owner"] == rank)[:, None], psi, 0.0)
    t = torch.as_tensor(mine)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if rank == 0:
        import oracle as O
        poses_ref, psi_ref, st_ref = O.ba_optimize(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
        upd = np.abs(poses_ref - prob["poses"]).max()
        err_p = np.abs(poses - poses_ref).max() / upd
        err_l = np.abs(t.numpy() - psi_ref).max() / np.abs(psi_ref - prob["psi"]).max()
        q.put((st.trials == st_ref.trials and st.accepted == st_ref.accepted, float(err_p), float(err_l), n_calls[0]))
    opt.close()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_optimize_two_processes_hip_shards():
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 90)
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    [p.join(300) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    same_traj, err_p, err_l, n_calls = q.get(timeout=5)
    assert same_traj and err_p < 1e-6 and err_l < 1e-6, (same_traj, err_p, err_l)
    assert n_calls == 1 + 2 * 2          # pattern once, then (system + scalars) per LM trial


# ---- the one-shot P2P transport (svs_comm_create_p2p / svs_comm_connect_p2p, comm.hip): `world` PROCESSES share the one GPU of the box, their mailboxes are
# mapped into each other with hipIpc (the same calls that map a peer GPU's memory over xGMI on a multi-GPU node), the exchange is the library's own push /
# reduce kernel pair.  (a) raw all-reduces of every message size of the back end, incl. messages longer than a mailbox slot and 40 back-to-back calls (slot
# parity), bit-equal to the sum in rank order; (b) the sharded optimize with svs_ba_set_comm on that transport vs the oracle.
def _p2p_worker(rank, world, port, q, one_device=True):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from scavislam_amd import capi, synth
    from scavislam_amd.backend import Communicator, SlamGraphOptimizer, shard_problem
    from scavislam_amd.ctypes_types import BaParams, Cam
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx, stream = capi.torch_context(0 if one_device else rank)
    if not one_device:
        torch.cuda.set_device(rank)
    comm = Communicator.p2p(ctx, rank, world, capacity_doubles=4096)
    assert comm.transport()["kind"] == "p2p"
    ok = True
    worst = 0
    with torch.cuda.stream(stream):
        for it, n in enumerate([32, 1, 300, 4096, 4097, 20000] + [777] * 40):
            vecs = [np.random.default_rng(1000 * it + r).standard_normal(n) * 10.0 ** np.random.default_rng(it).integers(-3, 4) for r in range(world)]
            t = torch.as_tensor(vecs[rank]).cuda()
            stream.synchronize()
            comm.allreduce(t.data_ptr(), n)
            ctx.sync()
            want = vecs[0].copy()
            for r in range(1, world):
                want = want + vecs[r]                   # rank order, like the reduce kernel
            got = t.cpu().numpy()
            ok = ok and np.array_equal(got, want)
            worst = max(worst, float(np.abs(got - want).max()))
    tr = comm.transport()
    # (b) sharded optimize over the same transport
    prob = synth.ba_window(12, 1500, seed=41)
    c = prob["cam"]
    cam = Cam(c["f"], c["cx"], c["cy"], c["b"], c["w"], c["h"])
    prm = BaParams.reference_defaults()
    sh = shard_problem(prob, rank, world)
    opt = SlamGraphOptimizer(ctx, stream)
    opt.set_comm(comm)
    opt.copyDataToG2o(sh["poses"], sh["psi"], sh["edges"], sh["cons"], cam, prm, add_pose_terms=sh["add_pose_terms"])
    n0 = comm.stats()["n_calls"]
    st = opt.optimize()
    poses, psi = opt.restoreDataFromG2o()
    n_calls = comm.stats()["n_calls"] - n0
    mine = np.where((sh["owner"] == rank)[:, None], psi, 0.0)
    t = torch.as_tensor(mine)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    allp = [torch.zeros_like(torch.as_tensor(poses)) for _ in range(world)]
    dist.all_gather(allp, torch.as_tensor(poses))
    if rank == 0:
        import oracle as O
        poses_ref, psi_ref, st_ref = O.ba_optimize(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
        upd = np.abs(poses_ref - prob["poses"]).max()
        err_p = np.abs(poses - poses_ref).max() / upd
        err_l = np.abs(t.numpy() - psi_ref).max() / np.abs(psi_ref - prob["psi"]).max()
        replicas_equal = all(torch.equal(allp[0], a) for a in allp)
        q.put(dict(raw_ok=bool(ok), raw_worst=worst, timeouts=tr["timeouts"], mailbox_memory=tr["mailbox_memory"], same_traj=bool(st.trials == st_ref.trials and st.accepted == st_ref.accepted),
                   err_p=float(err_p), err_l=float(err_l), n_calls=int(n_calls), trials=int(st.trials), replicas_equal=bool(replicas_equal)))
    dist.barrier()                                      # nobody tears its mailbox down while a peer may still push
    opt.close(); comm.close(); ctx.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 8])      # 8: eight mailboxes per rank, multi-piece messages (4097 / 20000 doubles against 4096-double slots)
def test_p2p_one_shot_transport_processes_on_one_gpu(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 90) + world
    procs = [ctx.Process(target=_p2p_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    [p.join(300) for p in procs]
    for p in procs:
        if p.is_alive():
            p.kill()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    r = q.get(timeout=5)
    print(f"P2P one-shot transport, {world} processes on one GPU: {r}")
    assert r["raw_ok"] and r["timeouts"] == 0, r
    assert r["same_traj"] and r["err_p"] < 1e-6 and r["err_l"] < 1e-6 and r["replicas_equal"], r
    assert r["n_calls"] == 1 + 2 * r["trials"], r
    assert r["mailbox_memory"] in ("fine-grained", "uncached"), r      # peers write the mailbox while its owner polls it: not coarse-grained memory


@pytest.mark.gpu
def test_p2p_one_shot_transport_across_devices():
    """The same exchange with one process per GPU: the peers' mailboxes are another device's memory (xGMI).  Needs two visible devices (the 1-GPU boxes skip it;
    on a multi-GPU node this is the validation of the transport across devices)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: the cross-device exchange cannot be exercised here")
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + (os.getpid() % 90)
    procs = [ctx.Process(target=_p2p_worker, args=(r, world, port, q, False)) for r in range(world)]
    [p.start() for p in procs]
    [p.join(300) for p in procs]
    for p in procs:
        if p.is_alive():
            p.kill()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    r = q.get(timeout=5)
    print(f"P2P one-shot transport, 2 processes on 2 GPUs: {r}")
    assert r["raw_ok"] and r["timeouts"] == 0 and r["same_traj"] and r["err_p"] < 1e-6 and r["err_l"] < 1e-6 and r["replicas_equal"], r


def test_bench_gpus_n_launches_n_ranks():
    """`python bench.py --gpus N` is the multi-GPU command the driver runs: with no launcher environment it must start N ranks itself (torch.distributed.run, one per
    GPU), and an N-GPU request must never come back as a one-rank line.  --dry-launch: the ranks meet over gloo on the CPU and rank 0 reports who answered."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-launch"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line == {"dry_launch": True, "n_gpus": 2, "requested": 2, "ranks_seen": [0, 1]}
    # a launcher that started the wrong number of ranks: refused with a reason, non-zero
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-launch"], env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                         capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "--gpus 2" in bad.stderr and not bad.stdout.strip()
