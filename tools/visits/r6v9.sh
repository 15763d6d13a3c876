#!/bin/bash
# round 6, visit 9: where the level-4 launch's time goes beyond a workgroup's lifetime: starts / ends of the 512 workgroups on the chip-wide clock
tag=${1:-r6v9}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for cfg in M S Sc; do
HS_IR_MATH=auto HS_HIP_LIB=$R/hyperseg_amd/lib/libhyperseg_hip_stamps_irc.so timeout 120 python tools/ir_phase_times.py $cfg > gpurun_out/irc_stamps_${cfg}_$tag.txt 2>&1
tail -13 gpurun_out/irc_stamps_${cfg}_$tag.txt | cut -c1-260
done
