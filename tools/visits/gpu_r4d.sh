#!/bin/bash
# Round-4 visit d: the level-4 kernel on eight waves per region (variant irc_nw8) against the four-wave product build, same box.
tag=${1:-r4d}; var=${2:-irc_nw8}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
out=$R/gpurun_out/irc_$tag.txt; : > $out
HS_HIP_LIB=$R/hyperseg_amd/lib/libhyperseg_hip_$var.so timeout 400 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "${IRC_TESTS:-inverted_residual or split_ir or full_config or tiny_decoder or modes_agree or op_c or misaligned or reference_fixture}" 2>&1 | tail -6 | tee -a $out
prof() {   # $1 = label, $2 = config
  rm -rf /tmp/prof_$1_$2; cd /tmp
  HS_IR_MATH=auto timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1_$2 -- python $R/tools/decoder_loop.py $2 40 > /tmp/prof_$1_$2.log 2>&1
  cd $R
  f=$(find /tmp/prof_$1_$2 -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then cp $f gpurun_out/irc_${tag}_$1_$2_kernel_stats.csv; echo "== $1 $2" | tee -a $out; python tools/kstats.py $f hs:: 60 | head -${3:-1} | tee -a $out; else tail -3 /tmp/prof_$1_$2.log | tee -a $out; fi
}
for rep in 1 2; do for c in M S Sc; do
  prof product $c 1
  HS_HIP_LIB=$R/hyperseg_amd/lib/libhyperseg_hip_$var.so prof $var $c 1
done; done
echo "== decoder loop graph replay" | tee -a $out
for rep in 1 2; do
  echo "product: $(HS_IR_MATH=auto timeout 100 python tools/decoder_loop.py M 200 2>&1 | grep -v amdgpu | tr '\n' ' ')" | tee -a $out
  echo "$var: $(HS_HIP_LIB=$R/hyperseg_amd/lib/libhyperseg_hip_$var.so HS_IR_MATH=auto timeout 100 python tools/decoder_loop.py M 200 2>&1 | grep -v amdgpu | tr '\n' ' ')" | tee -a $out
done
