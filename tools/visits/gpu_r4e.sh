#!/bin/bash
# Round-4 visit e: the heterogeneous launch (banks of later levels inside the k = 1 levels' launches) -- parity, decoder and frame A/B.
tag=${1:-r4e}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
out=$R/gpurun_out/cosched_$tag.txt; : > $out
timeout 500 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "${K_TESTS:-coscheduled or signal2weights or full_config or tiny_decoder or hyper_patch or meta_patch or model_m or benched or segment or pyramid or reference_fixture}" 2>&1 | tail -6 | tee -a $out
echo "== decoder loop (eager / graph replay), HS_COSCHEDULE_BANKS = 0 / 1" | tee -a $out
for rep in 1 2 3; do for m in 0 1; do
  echo "cosched=$m: $(HS_IR_MATH=auto HS_COSCHEDULE_BANKS=$m timeout 100 python tools/decoder_loop.py M 300 2>&1 | grep -v amdgpu | tr '\n' ' ')" | tee -a $out
done; done
for c in Sc; do for m in 0 1; do
  echo "$c cosched=$m: $(HS_IR_MATH=auto HS_COSCHEDULE_BANKS=$m timeout 100 python tools/decoder_loop.py $c 300 2>&1 | grep -v amdgpu | tr '\n' ' ')" | tee -a $out
done; done
for m in 0 1; do
  rm -rf /tmp/prof_co$m; cd /tmp
  HS_IR_MATH=auto HS_COSCHEDULE_BANKS=$m timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_co$m -- python $R/tools/decoder_loop.py M 40 > /tmp/prof_co$m.log 2>&1
  cd $R
  f=$(find /tmp/prof_co$m -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then cp $f gpurun_out/cosched_${tag}_$m_M_kernel_stats.csv; echo "== cosched=$m M" | tee -a $out; python tools/kstats.py $f hs:: 60 | head -10 | tee -a $out; fi
done
echo "== whole frame (bench.py --no-extras), HS_COSCHEDULE_BANKS = 0 / 1" | tee -a $out
for rep in 1 2; do for m in 0 1; do
  HS_COSCHEDULE_BANKS=$m timeout 200 python bench.py --no-extras --steps 200 --warmup 20 --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cosched=$m', d['value'], d['ms_per_step'], d['repeats']['ms_per_step'])" | tee -a $out
done; done
