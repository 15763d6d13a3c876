#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
tag=${1:-r2c}
timeout 300 python tools/diag_train_l3.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/diag_train_$tag.txt
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -k "full_config or op_d or inverted" 2>&1 | tail -3
HS_HIP_LIB=$R/hyperseg_amd/lib/libhyperseg_hip_stamps.so timeout 300 python tools/ir_phase_times.py M 2>&1 | grep -v amdgpu.ids > gpurun_out/phases_${tag}_M.txt; cat gpurun_out/phases_${tag}_M.txt
export TMPDIR=/tmp
for cfg in M L; do
  rm -rf /tmp/prof_$cfg
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$cfg -- python $R/tools/decoder_loop.py $cfg 20 > /tmp/prof_$cfg.log 2>&1 )
  grep -E "decoder (eager|graph)" /tmp/prof_$cfg.log
  f=$(find /tmp/prof_$cfg -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/kstats_${tag}_$cfg.csv && python tools/kstats.py "$f" "hs::" 40
done
