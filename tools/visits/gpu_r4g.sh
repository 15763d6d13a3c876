#!/bin/bash
# Round-4 visit g: training-path changes (stage input in one launch, own final upsample, BN counter in-kernel): tests + step times + profile.
tag=${1:-r4k}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x tests/test_hip_training.py tests/test_checkpoint.py tests/test_distributed.py -k "not config5_full" > gpurun_out/pytest_$tag.log 2>&1; tail -4 gpurun_out/pytest_$tag.log
timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "meta_conv2d or coscheduled or config5 or train_step" > gpurun_out/pytest_${tag}_b.log 2>&1; tail -4 gpurun_out/pytest_${tag}_b.log
timeout 200 python tools/train_step_time.py 20 > gpurun_out/train_step_$tag.txt 2>&1; cat gpurun_out/train_step_$tag.txt | tail -4
for mode in fp32 bf16; do
  ( cd /tmp && rm -rf /tmp/prof_train_$mode && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train_$mode -- python $R/tools/train_step_time.py 20 $mode > /tmp/prof_train_$mode.log 2>&1
    f=$(find /tmp/prof_train_$mode -name '*kernel_stats.csv' | head -1)
    if [ -n "$f" ]; then cp "$f" $R/gpurun_out/train_kernel_stats_${mode}_$tag.csv; python - "$f" $mode <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
per = [(int(r['Calls']) / 22, float(r['AverageNs']) / 1e3, r['Name']) for r in rows if int(r['Calls']) >= 22]
stock = [p for p in per if 'hs::' not in p[2][:12]]
print(sys.argv[2], 'launches/step', round(sum(p[0] for p in per), 1), 'kernel us/step', round(sum(p[0] * p[1] for p in per), 1),
      '| stock launches/step', round(sum(p[0] for p in stock), 1), 'us', round(sum(p[0] * p[1] for p in stock), 1))
PY
    else echo "no stats $mode"; tail -5 /tmp/prof_train_$mode.log; fi )
done
