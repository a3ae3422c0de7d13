"""Is the captured end-to-end training step a linear chain?  Captures it (small sizes) with graph debug mode on, dumps the DOT file
of every captured graph and counts nodes / edges / nodes with more than one successor (forks)."""
import os
import re
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cyclical-visual-captioning_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ctypes as C
_Graph = torch.cuda.CUDAGraph


def _keep(*a, **k):            # every graph the trainer makes keeps its hipGraph_t (raw_cuda_graph)
    return _Graph(keep_graph=True)


torch.cuda.CUDAGraph = _keep
import bench
import bench_e2e
from cvc import synth
from cvc.distributed import GradReducer
from cvc.trainer import Trainer, build_optimizer

dev = torch.device("cuda:0")
d = synth.Dims(B=4, N=20, F=12, R=512, A=64, E=32, V=50, T=4, G=24, DET=6, K=3)
o, model, batch = bench_e2e.build_raw(d, dev, 3)
model.train()
red = GradReducer(model.named_parameters()) if "noreducer" not in sys.argv else None
tr = Trainer(o, None, model, build_optimizer(model, o), None, None, grad_reducer=red)
assert tr.graph_capable()
with tr.deferred_errors():
    b = tr._prepare(batch, True)
    for _ in range(4):
        tr.train_step_bucketed(b)
torch.cuda.synchronize()
print(tr.graph_stats)
hip = C.CDLL("libamdhip64.so")
KINDS = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "waitEvent", 7: "eventRecord"}
for i, (g, _s, _r) in enumerate(tr._graphs.values()):
    raw = C.c_void_p(g.raw_cuda_graph())
    n = C.c_size_t(0)
    assert hip.hipGraphGetNodes(raw, None, C.byref(n)) == 0
    nodes = (C.c_void_p * n.value)()
    assert hip.hipGraphGetNodes(raw, nodes, C.byref(n)) == 0
    ne = C.c_size_t(0)
    assert hip.hipGraphGetEdges(raw, None, None, C.byref(ne)) == 0
    fr, to = (C.c_void_p * ne.value)(), (C.c_void_p * ne.value)()
    assert hip.hipGraphGetEdges(raw, fr, to, C.byref(ne)) == 0
    succ, pred = {}, {}
    for a_, b_ in zip(fr, to):
        succ.setdefault(a_, set()).add(b_)
        pred.setdefault(b_, set()).add(a_)
    kinds = {}
    for nd in nodes:
        t = C.c_int(-1)
        hip.hipGraphNodeGetType(C.c_void_p(nd), C.byref(t))
        kinds[nd] = KINDS.get(t.value, str(t.value))
    from collections import Counter
    forks = [a_ for a_, s_ in succ.items() if len(s_) > 1]
    joins = [b_ for b_, p_ in pred.items() if len(p_) > 1]
    roots = [nd for nd in nodes if nd not in pred]
    print(f"graph {i}: {n.value} nodes {dict(Counter(kinds.values()))}, {ne.value} edges, {len(roots)} root(s), {len(forks)} fork node(s), {len(joins)} join node(s)")
    for a_ in forks[:8]:
        print("   fork at a", kinds[a_], "node ->", [kinds[x] for x in succ[a_]])
