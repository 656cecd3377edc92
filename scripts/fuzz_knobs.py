"""Launch-shape and path-selection knobs (icet_set_option) drawn at random: a single solve and a 40-pair device batch must give the bits of the default
settings -- the sums are integers, the sort is a sort, the literal path decides what the fast path decides.  Usage (GPU box): python scripts/fuzz_knobs.py [draws] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from icet_amd import api
from tests.param_sweep import run_knob_draws


if __name__ == "__main__":
    draws = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    failed = run_knob_draws(draws, seed, log=lambda m: print(m, flush=True))
    print("draws with differing bits:", len(failed))
