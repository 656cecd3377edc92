cd $GRAFT_REPO_ROOT
one() { python bench.py --workload $1 $2 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '$2', '$ICET_HIP_LIB', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  for a in "" "--set lds_rank=2" "--set lds_rank=0"; do one odometry "$a"; done
  ICET_HIP_LIB=$PWD/icet_amd/lib_exp1/libicet_hip.so one odometry ""
done
for a in "" "--set lds_rank=2" "--set lds_rank=0"; do one mapmaker "$a"; done
ICET_HIP_LIB=$PWD/icet_amd/lib_exp1/libicet_hip.so one mapmaker ""
