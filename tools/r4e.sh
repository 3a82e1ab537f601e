cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4e; mkdir -p $O
T=$PWD/fedmlp_amd/libfedmlp_hip_tune.so
export FEDMLP_HIP_LIB=$T
for v in "0 0" "0 1" "1 0" "1 1"; do set -- $v
  FM_PW_GEMM=$1 FM_PW_FULLM=$2 python tools/op_profile.py --streams 1 --steps 4 > $O/op1s_g$1_f$2.txt 2>/dev/null
done
FM_PW_GEMM=1 FM_PW_GEMM_NS=3 FM_PW_FULLM=1 python tools/op_profile.py --streams 1 --steps 4 > $O/op1s_g1ns3_f1.txt 2>/dev/null
for f in $O/op1s_*.txt; do echo "== $f"; head -1 $f; grep -E "^(exp_fwd|proj_fwd|exp_dgrad|proj_dgrad|conv_fwd|conv_dgrad)/" $f | tr '\n' ';'; echo; done
