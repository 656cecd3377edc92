#!/bin/bash
# round 6: one last pass of every fuzz script on the final library, fresh seeds (61-63); summary lines to gpurun_out/r06_fuzz_final_pass.txt (run through gpurun)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/fuzz_r06; mkdir -p $O
F=gpurun_out/r06_fuzz_final_pass.txt
echo "# final pass of every fuzz script on round 6's final library, fresh seeds (61-63): summary lines" > $F
run() { name=$1; shift; timeout -k 10 170 "$@" > $O/$name.txt 2>&1; echo "$name: $(grep -v amdgpu.ids $O/$name.txt | tail -1)" | tee -a $F; }
run params_61 python scripts/fuzz_params.py 150 61
run params_62 python scripts/fuzz_params.py 150 62
run flags_63 python scripts/fuzz_params.py 120 63 flags
run adv_61 python scripts/fuzz_adversarial.py 120 61
run adv_rt2_62 python scripts/fuzz_adversarial.py 80 62 rt2
run batch_61 python scripts/fuzz_batch.py 40 61
run batch_62 python scripts/fuzz_batch.py 40 62
run knobs_61 python scripts/fuzz_knobs.py 40 61
run nodes_61 python scripts/fuzz_nodes.py 60 61
run nodes_62 python scripts/fuzz_nodes.py 60 62
run nodes_63 python scripts/fuzz_nodes.py 60 63
run grids_61 python scripts/fuzz_grids.py 40 61
run ties_61 python scripts/fuzz_ties.py 20 61
run multi_61 python scripts/fuzz_multi.py 20 61
run threads_61 python scripts/fuzz_threads.py 61
run members_61 python scripts/fuzz_members.py 60 61
