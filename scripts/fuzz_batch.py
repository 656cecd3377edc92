"""Randomised batches: ragged pairs (empty scans, tiny scans, strides, stretches of the real and synthetic scans), batch sizes on both sides of the library's
small-batch / throughput switch (32 pairs), random parameters, through icet_solve_batch (host pointers) and icet_solve_batch_device (device descriptors with
a leading dimension larger than the row count).  Every pair of every batch must carry the BITS of its own single solve.  Usage (GPU box):
python scripts/fuzz_batch.py [batches] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from icet_amd import api
from tests.param_sweep import pools as make_pools, run_batch


def main():
    batches = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    pools = make_pools()
    ctx = api.Context(); single = api.Context()
    bad = 0
    for bno in range(batches):
        desc, diffs = run_batch(ctx, single, rng, pools)
        bad += 1 if diffs else 0
        print("batch %2d %s  %s" % (bno, desc, "ok" if not diffs else "DIFF " + str(diffs[:6])), flush=True)
    print("batches with a pair that differs from its single solve:", bad)


if __name__ == "__main__":
    main()
