#!/bin/bash
# Round-4 visit b: the restructured level-4 kernel -- parity tests in all math modes, per-kernel times (rocprofv3 stats) for M / S / Sc
# with the product build and the A/B variants, phase stamps.
tag=${1:-r4b}; shift; variants=${@:-irc_narrow_store}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
out=$R/gpurun_out/irc_$tag.txt; : > $out
timeout 400 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "${IRC_TESTS:-inverted_residual or split_ir or full_config or tiny_decoder or modes_agree or op_c or misaligned or bn_act or reference_fixture}" 2>&1 | tail -8 | tee -a $out
prof() {   # $1 = label, $2 = config, env from caller
  rm -rf /tmp/prof_$1_$2; cd /tmp
  HS_IR_MATH=auto timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1_$2 -- python $R/tools/decoder_loop.py $2 40 > /tmp/prof_$1_$2.log 2>&1
  cd $R
  f=$(find /tmp/prof_$1_$2 -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then cp $f gpurun_out/irc_${tag}_$1_$2_kernel_stats.csv; echo "== $1 $2" | tee -a $out; python tools/kstats.py $f hs:: 60 | tee -a $out; else tail -3 /tmp/prof_$1_$2.log | tee -a $out; fi
}
for c in M S Sc; do prof product $c; done
for v in $variants; do
  export HS_HIP_LIB=$R/hyperseg_amd/lib/libhyperseg_hip_$v.so
  prof $v M
  unset HS_HIP_LIB
done
HS_IR_MATH=auto HS_HIP_LIB=$R/hyperseg_amd/lib/libhyperseg_hip_stamps_irc.so timeout 120 python tools/ir_phase_times.py M 2>&1 | grep -v amdgpu.ids | tee gpurun_out/irc_phase_cycles_$tag.txt | head -40
