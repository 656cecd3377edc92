import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from icet_amd import api
import icet_amd
from tests.param_sweep import draw_node_case, run_node_case
from oracle import pyoracle as po
rng = np.random.default_rng(63)
for c in range(7):
    kw, frames = draw_node_case(rng)
print(kw, [len(f) for f in frames])
ctx = icet_amd.Context(0)
for flags in (0, 64):
    g, o = api.Node(ctx, **dict(kw, flags=kw.get("flags", 0) | flags)), po.Node(**kw)
    for k, s in enumerate(frames):
        rg, ro = g.push(s), o.push(s)
        print(flags, k, "dev", rg["solved"], rg["diverged"], rg["n_kept"], np.array2string(rg["X"], precision=7), "| orc", ro["solved"], ro["diverged"], ro["n_kept"], np.array2string(ro["X"], precision=7))
    g.close(); o.close()
