#!/bin/bash
# round 6, visit 3: CamVid HyperSeg-L (Lc) parity + bench line; timing-only bounds of the "4x fewer store instructions" idea for the
# level-4 kernel; stamps of the product kernel on this box; the default bench with its per-leg wall seconds.
tag=${1:-r6v3}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "full_config and Lc" > gpurun_out/pytest_$tag.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/pytest_$tag.log | cut -c1-300
grep -E "^(FAILED|ERROR)" gpurun_out/pytest_$tag.log | head -20 | cut -c1-300
bash tools/gpu_variants.sh $tag M irc_store_quarter irc_h1_quarter irc_both_quarter > /dev/null 2>&1
cat gpurun_out/variants_$tag.txt | grep -E "==|patch_irc" | cut -c1-160
HS_IR_MATH=auto HS_HIP_LIB=$R/hyperseg_amd/lib/libhyperseg_hip_stamps_irc.so timeout 120 python tools/ir_phase_times.py M > gpurun_out/irc_stamps_$tag.txt 2>&1
tail -32 gpurun_out/irc_stamps_$tag.txt | cut -c1-170
timeout 300 python bench.py --model lc --no-cpu-baseline --traffic off > gpurun_out/bench_lc_$tag.json 2> gpurun_out/bench_lc_$tag.err || tail -5 gpurun_out/bench_lc_$tag.err
python -c "
import json; d=json.load(open('gpurun_out/bench_lc_$tag.json')); print('lc', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d.get('parity')); print(d['config']['decoder_launches']); print([(l['kernel'], l['avg_us']) for l in d['decoder']['launches'] if l['in_decoder']]); print(d['legs_s'])"
( time timeout 600 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err ) 2>&1 | grep real
python -c "
import json; d=json.load(open('gpurun_out/bench_$tag.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['legs_s']); print(d['cpu_baseline'])"
