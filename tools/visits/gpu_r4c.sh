#!/bin/bash
# Round-4 visit c: level-4 kernel, same-box A/B against round 3's (variant irc_r3); banks pipelined on a side stream (HS_SIDE_STREAM).
tag=${1:-r4c}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
out=$R/gpurun_out/irc_$tag.txt; : > $out
timeout 400 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "${IRC_TESTS:-inverted_residual or split_ir or full_config or tiny_decoder or modes_agree or op_c or misaligned or reference_fixture}" 2>&1 | tail -6 | tee -a $out
prof() {   # $1 = label, $2 = config
  rm -rf /tmp/prof_$1_$2; cd /tmp
  HS_IR_MATH=auto timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1_$2 -- python $R/tools/decoder_loop.py $2 40 > /tmp/prof_$1_$2.log 2>&1
  cd $R
  f=$(find /tmp/prof_$1_$2 -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then cp $f gpurun_out/irc_${tag}_$1_$2_kernel_stats.csv; echo "== $1 $2" | tee -a $out; python tools/kstats.py $f hs:: 60 | head -${3:-3} | tee -a $out; else tail -3 /tmp/prof_$1_$2.log | tee -a $out; fi
}
for rep in 1 2; do
  prof product M 8
  HS_HIP_LIB=$R/hyperseg_amd/lib/libhyperseg_hip_irc_r3.so prof irc_r3 M 1
done
prof product S 2; HS_HIP_LIB=$R/hyperseg_amd/lib/libhyperseg_hip_irc_r3.so prof irc_r3 S 1
prof product Sc 2; HS_HIP_LIB=$R/hyperseg_amd/lib/libhyperseg_hip_irc_r3.so prof irc_r3 Sc 1
HS_IR_MATH=auto HS_HIP_LIB=$R/hyperseg_amd/lib/libhyperseg_hip_stamps_irc.so timeout 120 python tools/ir_phase_times.py M 2>&1 | grep -v amdgpu.ids | tee gpurun_out/irc_phase_cycles_$tag.txt | head -30
echo "== decoder loop, HS_SIDE_STREAM = 0 / 1 / 2 (eager and graph replay)" | tee -a $out
for rep in 1 2; do for m in 0 1 2; do
  echo "side=$m: $(HS_IR_MATH=auto HS_SIDE_STREAM=$m timeout 100 python tools/decoder_loop.py M 200 2>&1 | grep -v amdgpu | tr '\n' ' ')" | tee -a $out
done; done
echo "== whole frame (bench.py --no-extras), HS_SIDE_STREAM = 0 / 1 / 2" | tee -a $out
for rep in 1 2; do for m in 0 1 2; do
  HS_SIDE_STREAM=$m timeout 200 python bench.py --no-extras --steps 200 --warmup 20 --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('side=$m', d['value'], d['ms_per_step'], d['repeats']['ms_per_step'])" | tee -a $out
done; done
