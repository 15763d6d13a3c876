#!/bin/bash
# round 6, visit 10: fair shares for the CU's two level-4 workgroups (s_setprio turns by wave slot): product (per chunk) vs off / younger
# first / per stage, M S Sc; stamps of the product; parity of the irc tests.
tag=${1:-r6v10}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for cfg in M S Sc; do
  bash tools/gpu_variants.sh ${tag}_$cfg $cfg irc_prio0 irc_prio2 irc_prio3 > /dev/null 2>&1
  cat gpurun_out/variants_${tag}_$cfg.txt | grep -E "==|patch_irc" | cut -c1-160
done
HS_IR_MATH=auto HS_HIP_LIB=$R/hyperseg_amd/lib/libhyperseg_hip_stamps_irc.so timeout 120 python tools/ir_phase_times.py M > gpurun_out/irc_stamps_M_$tag.txt 2>&1
tail -13 gpurun_out/irc_stamps_M_$tag.txt | cut -c1-330
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -x -k "inverted_residual or split_ir or full_config or op_c" 2>&1 | tail -2
