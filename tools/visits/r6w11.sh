#!/bin/bash
# round 6, visit w11: depthwise kernel without the (opt-in, unused) squeeze-excite tail compiled in: encoder tests, whole-frame A/B against the
# previous build (variant library enc_se_in); lean-kernel knobs re-swept per block with staging in; in-order dispatch list of the frame
#   gpurun --timeout 1500 -- 'bash tools/visits/r6w11.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp HS_BENCH_MBCONV_FIRST_TABLE_ONLY=1
out=$R/gpurun_out/dw_set_r6w11.txt; : > $out
timeout 600 python -m pytest tests/test_hip_encoder.py tests/test_model_boundary.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -5 | tee -a $out
for round in 1 2 3; do
  for v in enc_se_in product; do
    lib=$R/hyperseg_amd/lib/libhyperseg_hip_$v.so; [ $v = product ] && lib=$R/hyperseg_amd/lib/libhyperseg_hip.so
    HS_HIP_LIB=$lib timeout 200 python bench.py --model m --steps 300 --warmup 30 --no-cpu-baseline --no-extras 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$round m $v', d['value'], d['ms_per_step'])" | tee -a $out
  done
done
run() { echo "== $*" | tee -a $out; env "$@" timeout 120 python tools/bench_mbconv.py 2>&1 | tail -12 | grep -E "^ +[2-8] |sum" | cut -c1-75 | tee -a $out; }
run HS_MBX_MIN_WG=768
run HS_MBX_MIN_WG=512
run HS_MBX_MIN_WG=1024
run HS_MBX_MIN_WG=768 HS_MBX_OTH1=8
run HS_MBX_MIN_WG=768 HS_MBX_OTH2=4
cd /tmp && rm -rf /tmp/pf && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pf -- python $R/tools/prof_graph.py 20 fold_dw > /tmp/pf.log 2>&1
f=$(find /tmp/pf -name '*kernel_trace.csv' | head -1); cd $R
python tools/frame_sequence.py $f 20 > gpurun_out/frame_sequence_r6w11.txt 2>&1; tail -2 gpurun_out/frame_sequence_r6w11.txt
