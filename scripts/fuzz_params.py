"""Randomised differential run: GPU (C ABI) against the CPU oracle over the PARAMETER space (tests/param_sweep.py).  The keyframe table must be the oracle's
bits; the solution is printed next to it (scripts/fuzz_diag.py looks into single cases).  Usage (GPU box): python scripts/fuzz_params.py [cases] [seed] [flags]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from icet_amd import api
from tests.param_sweep import draw_case, run_case, pools as make_pools


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    flags = len(sys.argv) > 3 and sys.argv[3] == "flags"
    rng = np.random.default_rng(seed)
    pools = make_pools()
    ctx = api.Context()
    bad = 0
    for c in range(cases):
        a, b, T, P, kw, runlen, x0 = draw_case(rng, pools, with_flags=flags)
        bits, d, r, ref, fits = run_case(ctx, a, b, T, P, kw, runlen, x0)
        ok = all(bits.values())
        bad += 0 if ok else 1
        tag = " flag=%d" % kw["_twin"][0] if "_twin" in kw else ""
        print("case %3d%s n1=%6d n2=%6d T=%3d P=%2d n=%3d thresh=%.2f buff=%.1f runlen=%d x0=%s fits=%4d keyframe_bits=%s dX_t=%.2e dX_r=%.2e pruned=%s%s" % (
            c, tag, a.shape[0], b.shape[0], T, P, kw["n"], kw["thresh"], kw["buff"], runlen, "0" if not x0.any() else "r", fits, "ok" if ok else "DIFF", d[:3].max(), d[3:].max(),
            ref["trace"]["pruned"][-1] if runlen else "-", "" if ok else "  " + str({k: v for k, v in bits.items() if not v})), flush=True)
    print("cases with differing keyframe bits:", bad)


if __name__ == "__main__":
    main()
