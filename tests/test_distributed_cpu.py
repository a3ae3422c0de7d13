"""The N>1 path on CPU: world_size-2 gloo processes.  Compute is out of reach without a GPU, so
the gradients come from the golden fixture G3 (per-shard reference gradients) and the test checks
the exchange itself: flat gradient arenas (every .grad a view), hook-driven bucket exchange + 1/G == the
fixture's shard mean over several steps, never-used parameters keep all-zero views, the clip with the 1/G
folded in == clip_grad_norm_ over averaged gradients, results are bitwise equal across ranks."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN


def _worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank))
        from cvc.distributed import GradReducer, init_from_env, shard_range
        r, w, _ = init_from_env("gloo")
        assert (r, w) == (rank, world)
        # the control plane's small all-reduce (bench.py's max over ranks, barriers' companion): host numbers over gloo
        from cvc.distributed import control_all_reduce
        assert control_all_reduce([float(rank), 10.0 - rank], "max") == [float(world - 1), 10.0]
        assert control_all_reduce([1.0], "sum") == [float(world)]
        z = np.load(os.path.join(GOLDEN, "g3_shards.npz"))
        names = sorted(k[len("mean.grad."):] for k in z.keys() if k.startswith("mean.grad.") and not k.endswith(".is_none"))
        dead = sorted(k[len("mean.grad."):-len(".is_none")] for k in z.keys() if k.startswith("mean.grad.") and k.endswith(".is_none"))
        params = []
        for n in names + dead:
            shape = z["full.grad." + n].shape if ("full.grad." + n) in z else (3,)
            params.append((n, torch.nn.Parameter(torch.zeros(shape))))
        for overlap in (False, True):
            red = GradReducer(params, bucket_mb=0.01, overlap=overlap)          # tiny buckets -> several messages
            assert len(red.buckets) > 2
            for n, p in params:                                                 # every .grad is a view into a flat arena
                assert p.grad is not None and p.grad.data_ptr() == red._views[id(p)].data_ptr()
            for step in range(3):                                               # step 0 learns the arrivals; 1, 2 launch from hooks
                red.zero_grad()
                for n, p in params:
                    if n in dead:
                        continue
                    g = torch.from_numpy(z["shard%d.grad.%s" % (rank, n)]).clone()
                    (p * g).sum().backward()                                      # accumulates in place, drives the hooks
                if step == 2 and overlap:
                    assert any(red._launched)                                     # buckets left from inside backward
                red.finalize()
                for n, p in params:
                    if n in dead:
                        # never-used parameters: found on the first step, .grad = None from then on (the reference's optimizers
                        # skip them: no weight decay, no Adam state); their arena slots stay zero
                        assert p.grad is None, n
                        assert float(red._views[id(p)].abs().max()) == 0.0, n
                    else:
                        assert p.grad.data_ptr() == red._views[id(p)].data_ptr(), n  # still the arena view
                        np.testing.assert_allclose(p.grad.numpy(), z["mean.grad." + n], rtol=1e-6, atol=1e-8)
                # bitwise equal across ranks
                flat = torch.cat([a for a in red.arenas])
                both = [torch.zeros_like(flat) for _ in range(world)]
                dist.all_gather(both, flat)
                assert torch.equal(both[0], both[1])
            # sums + clip with the 1/G folded in == clip_grad_norm_ over the averaged gradients (trainer.py:120-121)
            red.zero_grad()
            for n, p in params:
                if n not in dead:
                    (p * torch.from_numpy(z["shard%d.grad.%s" % (rank, n)])).sum().backward()
            red.finalize(average=False)
            total = red.clip_(0.1, summed=True)
            ref = [torch.from_numpy(z["mean.grad." + n]).clone() for n, p in params if n not in dead]
            ref_norm = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in ref]))
            np.testing.assert_allclose(float(total), float(ref_norm), rtol=1e-5)
            coef = min(1.0, 0.1 / (float(ref_norm) + 1e-6))
            for (n, p), g in zip([(n, p) for n, p in params if n not in dead], ref):
                np.testing.assert_allclose(p.grad.numpy(), (g * coef).numpy(), rtol=1e-5, atol=1e-9)
            red.remove_hooks()
            for n, p in params:
                p.grad = None
        # inference outputs of the clip shards are merged host-side, identically on every rank
        from cvc.distributed import gather_eval_outputs
        preds = {"v_%d" % rank: [{"sentence": "s%d" % rank, "segment": "0"}], "v_shared": [{"sentence": "r%d" % rank, "segment": str(rank)}]}
        grd = {"v_shared": {str(rank): {"clss": ["c%d" % rank]}}}
        all_p, all_g = gather_eval_outputs(preds, grd)
        assert set(all_p) == {"v_0", "v_1", "v_shared"}
        assert [x["sentence"] for x in all_p["v_shared"]] == ["r0", "r1"]      # rank order
        assert all_g == {"v_shared": {"0": {"clss": ["c0"]}, "1": {"clss": ["c1"]}}}
        # clip sharding covers the batch exactly once
        cover = [shard_range(7, r_, 2) for r_ in range(2)]
        assert cover[0].start == 0 and cover[0].stop == cover[1].start and cover[1].stop == 7
        # the data plane's rendezvous: when rank 0 cannot draw RCCL's unique id, EVERY rank gets the error (no rank is left waiting
        # for a broadcast that never comes)
        from cvc.comm import RcclComm

        def no_id():
            raise OSError("librccl is not here")
        real_id, RcclComm.unique_id = RcclComm.unique_id, staticmethod(no_id)
        try:
            with pytest.raises(RuntimeError, match="rank 0 could not draw the unique id.*librccl is not here"):
                RcclComm.from_process_group()
        finally:
            RcclComm.unique_id = real_id
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


def test_two_rank_gradient_exchange_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in res), res


def test_shard_batch_dict_and_tuple():
    from cvc.distributed import shard_batch
    b = {"a": torch.arange(10).view(5, 2), "ids": ["x"] * 5, "k": 3}
    s0, s1 = shard_batch(b, 0, 2), shard_batch(b, 1, 2)
    assert s0["a"].shape[0] == 3 and s1["a"].shape[0] == 2 and len(s0["ids"]) == 3 and s0["k"] == 3
    assert torch.equal(torch.cat([s0["a"], s1["a"]]), b["a"])


class _TwoUses(torch.autograd.Function):
    """A weight used twice per pass through cvc.functional's deferred-dW batcher: each backward call stashes its dW, the
    last outstanding use returns the sum -- exactly what the LSTM cells' T uses per step do."""

    @staticmethod
    def forward(ctx, w, x):
        from cvc import functional as F_
        ctx.key = ("test", w.data_ptr())
        ctx.save_for_backward(w, x)
        F_._BATCHER.note_use(ctx.key)
        return (w * x).sum().reshape(1)

    @staticmethod
    def backward(ctx, g):
        from cvc import functional as F_
        w, x = ctx.saved_tensors
        got = F_._BATCHER.add(ctx.key, (g * x,), (w,), lambda items: (sum(i[0] for i in items),))
        return (got[0] if got is not None else None), None


def test_late_flushed_weight_stays_live_and_in_the_arena():
    """a weight whose gradient arrives only through the end-of-backward flush (one of its uses got no
    gradient, so the use counter never reached zero) fires no post-accumulate hook.  It must not be marked dead on the first
    step: its gradient must stay in the arena view (exchanged, zeroed, seen by the optimizer) on every step."""
    from cvc.distributed import GradReducer
    w = torch.nn.Parameter(torch.ones(5))
    v = torch.nn.Parameter(torch.ones(3))          # an ordinary parameter (hook fires)
    dead = torch.nn.Parameter(torch.ones(2))       # never used
    red = GradReducer([("rest.w", w), ("rest.v", v), ("rest.dead", dead)])
    x1, x2 = torch.arange(5.0), torch.full((5,), 2.0)
    for step in range(3):
        red.zero_grad()
        a = _TwoUses.apply(w, x1)
        _unused = _TwoUses.apply(w, x2)            # second use: no gradient reaches it -> leftover -> late flush
        (a.sum() + (v * 3).sum()).backward()
        red.finalize()
        assert w.grad is not None and w.grad.data_ptr() == red._views[id(w)].data_ptr(), step
        np.testing.assert_allclose(w.grad.numpy(), x1.numpy())
        np.testing.assert_allclose(v.grad.numpy(), np.full(3, 3.0))
        assert dead.grad is None or float(dead.grad.abs().max()) == 0.0
        from cvc import functional as F_
        F_._BATCHER.uses.clear()
    assert id(w) not in red._dead and id(dead) in red._dead
    # a parameter dropped as dead that later receives a gradient is refused loudly
    red.zero_grad()
    with pytest.raises(RuntimeError, match="received no gradient on the first step"):
        (dead * 2).sum().backward()
    red.remove_hooks()


# ------------------------------------------------------------------ cvc_allreduce_grads at world 2, without GPUs
STUB_SRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stub_rccl", "stub_rccl.c")


def _build_stub(dirname) -> str:
    """tests/stub_rccl/stub_rccl.c -> <dirname>/librccl.so (host pointers over POSIX shared memory; test infrastructure)"""
    import subprocess
    out = os.path.join(str(dirname), "librccl.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", out, STUB_SRC, "-lpthread", "-lrt", "-ldl"])
    return out


def _stub_worker(rank, world, port, q):
    """One rank of the stub-transport test.  CVC_RCCL_LIB (set by the parent) names the stub: csrc/comm_rccl.hip opens that path
    instead of the "librccl.so" the process already holds (torch's bundled RCCL, loaded through libtorch_hip.so's RPATH)."""
    try:
        import ctypes as C
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        from cvc import hip
        from cvc.comm import RcclComm
        from cvc.distributed import GradReducer, init_from_env
        init_from_env("gloo")
        L = hip.lib()
        # ---- argument checks of the C-ABI itself: rank >= world, bad world, null pointers
        h = C.c_void_p()
        uid = (C.c_char * 128)()
        assert L.cvc_comm_unique_id(uid) == 0 and uid.raw.startswith(b"/cvc_stub_rccl_"), "the stub is not the library that was opened"
        for w_, r_ in ((2, 2), (2, 5), (2, -1), (0, 0)):
            assert L.cvc_comm_init(w_, r_, uid, C.byref(h)) == -1, (w_, r_)          # CVC_E_BADARG
        assert L.cvc_comm_init(2, 0, None, C.byref(h)) == -1 and L.cvc_comm_init(2, 0, uid, None) == -1
        # ---- the production rendezvous: rank 0 draws the id, gloo carries it, every rank joins, every rank agrees it worked
        comm = RcclComm.from_process_group(host_buffers=True)
        assert (comm.world, comm.rank) == (world, rank)
        assert comm.count_ranks() == world                         # count = 1: not divisible by 2 -> the all-reduce branch
        assert L.cvc_allreduce_grads(comm._h, None, 8, None) == -1 and L.cvc_allreduce_grads(comm._h, uid, 0, None) == -1
        # ---- in-place reduce-scatter + all-gather: the sum of shard r lands at grads + r * n, the gather fills the rest
        rng = np.random.default_rng(100)
        mine_all = [rng.standard_normal(4096).astype(np.float32) for _ in range(world)]       # every rank can compute every rank's input
        for count in (4096, 2, 130, 4095, 1, 3):                   # even -> RS + AG pair; odd -> one all-reduce
            bufs = [m[:count].copy() for m in mine_all]
            want = bufs[0].copy()
            for r_ in range(1, world):
                want = want + bufs[r_]                             # rank order, fp32: what the stub computes, bit for bit
            t = torch.from_numpy(bufs[rank].copy())
            comm.all_reduce_(t)
            assert np.array_equal(t.numpy(), want), (count, rank)
            both = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(both, t)
            assert torch.equal(both[0], both[1]), count            # bitwise equal across ranks
        # ---- the reducer's bookkeeping on this transport: arenas, buckets, hooks, sinks as on RCCL; CPU arenas, no streams
        z = np.load(os.path.join(GOLDEN, "g3_shards.npz"))
        names = sorted(k[len("mean.grad."):] for k in z.keys() if k.startswith("mean.grad.") and not k.endswith(".is_none"))
        params = [(n, torch.nn.Parameter(torch.zeros(z["full.grad." + n].shape))) for n in names]
        red = GradReducer(params, bucket_mb=0.01, comm=comm)
        assert red.backend == "rccl" and red.world == world and red.exchange and len(red.buckets) > 2
        assert all(a.numel() % (64 * world) == 0 for a in red.arenas)           # equal, aligned shards: the RS + AG branch
        for step in range(3):
            red.zero_grad()
            for n, p in params:
                (p * torch.from_numpy(z["shard%d.grad.%s" % (rank, n)]).clone()).sum().backward()
            if step == 2:
                assert any(red._launched)                                       # buckets left from inside backward
            red.finalize()
            for n, p in params:
                np.testing.assert_allclose(p.grad.numpy(), z["mean.grad." + n], rtol=1e-6, atol=1e-8)
            flat = torch.cat(list(red.arenas))
            both = [torch.zeros_like(flat) for _ in range(world)]
            dist.all_gather(both, flat)
            assert torch.equal(both[0], both[1])
        red.remove_hooks()
        dist.barrier()
        comm.destroy()
        with pytest.raises(RuntimeError, match="destroyed"):
            comm.all_reduce_(torch.zeros(4))
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.skipif(torch.cuda.is_available(), reason="host-buffer stand-in transport: CPU machines only (a GPU box runs the real RCCL tests)")
def test_allreduce_grads_cabi_world_2_over_a_stub_librccl(tmp_path, monkeypatch):
    """cvc_allreduce_grads (csrc/comm_rccl.hip) has only ever run with world == 1 on hardware: `rank * n == 0`, nothing on a link.
    Here two processes load the product's libcvc_hip.so, which dlopens a stand-in librccl.so built from tests/stub_rccl/ (found
    through CVC_RCCL_LIB: a bare "librccl.so" resolves to torch's bundled RCCL, already in the process), and run cvc_comm_init(2, r) + cvc_allreduce_grads on host
    buffers: the in-place reduce-scatter lands the sum of shard r at grads + r * n, the all-gather fills the rest, both ranks
    bitwise equal, odd counts take the all-reduce branch, rank >= world is CVC_E_BADARG; then GradReducer(comm=...) runs its
    bucket bookkeeping over the same transport against the G3 shard-mean fixture."""
    monkeypatch.setenv("CVC_RCCL_LIB", _build_stub(tmp_path))          # inherited by the two fresh interpreters below
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_stub_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in res), res
