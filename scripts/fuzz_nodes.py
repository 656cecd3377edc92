"""Random node settings and hostile frame sequences (tests/param_sweep.draw_node_case): the GPU node against the oracle's node (decisions and bookkeeping exactly) and a
device burst against frame-by-frame pushes (bits).  Usage (GPU box): python scripts/fuzz_nodes.py [cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from icet_amd import api
from tests.param_sweep import draw_node_case, run_node_case

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = api.Context(); nbad = 0
for c in range(cases):
    kw, frames = draw_node_case(rng)
    bad = run_node_case(ctx, kw, frames)
    nbad += 1 if bad else 0
    print("case %3d frames=%d rows=%s runlen=%d min_range=%.1f seed_x0=%d map=%d/%d thresh=%g/%g  %s" % (
        c, len(frames), [len(f) for f in frames], kw["runlen"], kw["min_range"], kw.get("seed_x0", 0), kw.get("map_capacity", 0), kw.get("map_downsample", 0),
        kw.get("trans_thresh", 0), kw.get("rot_thresh", 0), "ok" if not bad else "BAD " + "; ".join(bad[:4])), flush=True)
print("cases with complaints:", nbad)
