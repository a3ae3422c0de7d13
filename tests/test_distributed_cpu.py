"""The N>1 path on CPU: world_size-2 gloo processes.  Compute is out of reach without a GPU, so
the gradients come from the golden fixture G3 (per-shard reference gradients) and the test checks
the exchange itself: bucketed all-reduce + 1/G == the fixture's shard mean, None-grad parameters are
skipped consistently, a rank-divergent None is healed, results are bitwise equal across ranks."""
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
            for step in range(2):                                                # step 0 agrees on active set; step 1 uses hooks
                for n, p in params:
                    p.grad = None
                for n, p in params:
                    if n in dead:
                        continue
                    g = torch.from_numpy(z["shard%d.grad.%s" % (rank, n)]).clone()
                    if overlap:
                        (p * g).sum().backward()                                  # drives the post-accumulate hooks
                    else:
                        p.grad = g
                red.finalize()
                for n, p in params:
                    if n in dead:
                        assert p.grad is None, n
                    else:
                        np.testing.assert_allclose(p.grad.numpy(), z["mean.grad." + n], rtol=1e-6, atol=1e-8)
                # bitwise equal across ranks
                flat = torch.cat([p.grad.reshape(-1) for n, p in params if n not in dead])
                both = [torch.zeros_like(flat) for _ in range(world)]
                dist.all_gather(both, flat)
                assert torch.equal(both[0], both[1])
            red.remove_hooks()
        # a gradient present on one rank only is healed with zeros on the other
        p1 = torch.nn.Parameter(torch.zeros(4))
        red = GradReducer([("only_rank0", p1)], overlap=False)
        p1.grad = torch.ones(4) if rank == 0 else None
        red.finalize()
        np.testing.assert_allclose(p1.grad.numpy(), np.full(4, 0.5))
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
