#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/soak; mkdir -p $O
run() { name=$1; shift; timeout -k 10 280 "$@" > $O/$name.txt 2>&1; echo "$name: $(tail -1 $O/$name.txt)"; }
for s in 41 42 43 44 45; do run params_$s python scripts/fuzz_params.py 200 $s; done
for s in 46 47; do run params_flags_$s python scripts/fuzz_params.py 200 $s flags; done
for s in 41 42 43; do run adv_rt2_$s python scripts/fuzz_adversarial.py 150 $s rt2; done
for s in 41 42 43; do run batch_$s python scripts/fuzz_batch.py 60 $s; done
for s in 41 42 43; do run knobs_$s python scripts/fuzz_knobs.py 60 $s; done
for s in 41 42 43; do run nodes_$s python scripts/fuzz_nodes.py 60 $s; done
for s in 41 42; do run multi_$s python scripts/fuzz_multi.py 30 $s; done
