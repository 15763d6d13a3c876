#!/bin/bash
# round 5, visit 16: phase removal in the blocked signal2weights launch (no stores / no LDS-DMA fills / no matrix products): what is the 16 us made of?
# (the s2b_no* variant builds came from -DHS_S2B_DEV_* knobs that lived in hs_s2w_blocked.h at commit 47765e8 only)
tag=${1:-r5v16}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
out=$R/gpurun_out/s2w_phase_removal_$tag.txt; : > $out
for v in product s2b_nostore s2b_nofill s2b_nomfma; do
  lib=""; [ $v != product ] && lib=$R/hyperseg_amd/lib/libhyperseg_hip_$v.so
  for cfg in M L; do
    rm -rf /tmp/pv; cd /tmp
    HS_HIP_LIB=$lib timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -- python $R/tools/decoder_loop.py $cfg 30 > /tmp/pv.log 2>&1
    cd $R; echo "== $v ($cfg)" | tee -a $out
    f=$(find /tmp/pv -name '*kernel_stats.csv' | head -1)
    [ -n "$f" ] && python tools/kstats.py $f signal2weights 40 | cut -c1-150 | tee -a $out
  done
done
