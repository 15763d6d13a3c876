#!/bin/bash
# round 6, visit 4: level-4 kernel with swapped pw3 operand roles (16-byte epilogue stores) + tap tables under the load round trip:
# parity in all math modes, same-box A/B against round 5's kernel (irc_r5), stamps; default bench wall time after the CPU-baseline fix.
tag=${1:-r6v4}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -x -k "inverted_residual or split_ir or full_config or op_c or tiny_decoder or benched" > gpurun_out/pytest_$tag.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/pytest_$tag.log | cut -c1-300
grep -E "^(FAILED|ERROR)" gpurun_out/pytest_$tag.log | head -20 | cut -c1-300
for cfg in M S Sc; do
  bash tools/gpu_variants.sh ${tag}_$cfg $cfg irc_r5 > /dev/null 2>&1
  cat gpurun_out/variants_${tag}_$cfg.txt | grep -E "==|patch_irc" | cut -c1-160
done
HS_IR_MATH=auto HS_HIP_LIB=$R/hyperseg_amd/lib/libhyperseg_hip_stamps_irc.so timeout 120 python tools/ir_phase_times.py M > gpurun_out/irc_stamps_$tag.txt 2>&1
tail -32 gpurun_out/irc_stamps_$tag.txt | cut -c1-170
( time timeout 600 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err ) 2>&1 | grep real
python -c "
import json; d=json.load(open('gpurun_out/bench_$tag.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['legs_s']); print(d['cpu_baseline'])"
