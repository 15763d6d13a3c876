#!/bin/bash
# SQ / LDS counters of the decoder kernels (separate rocprofv3 --pmc passes, eager decoder-only workload), averaged per
# kernel.  Usage on the GPU box:  bash tools/pmc_decoder.sh <tag>     -> gpurun_out/pmc_<tag>_*.txt
tag=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 60 rocprofv3 -L > $R/gpurun_out/pmc_${tag}_available.txt 2>&1
i=0
for group in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" \
             "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 120 rocprofv3 --pmc $group --kernel-trace --output-format csv -d /tmp/pmc_$i -- python $R/tools/prof_decoder.py ${PMC_CFG:-M} ${PMC_ITERS:-20} > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then
    python - "$f" > $R/gpurun_out/pmc_${tag}_$i.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0][:60]
    if not k.startswith('hs::') and 'hs::' not in k:
        continue
    a = agg[k][r['Counter_Name']]
    a[0] += 1; a[1] += float(r['Counter_Value'])
for k, cs in agg.items():
    print(k)
    for c, (n, v) in sorted(cs.items()):
        print(f'    {c:32s} {v / n:16.1f}   (mean of {n} dispatches)')
PY
    tail -40 $R/gpurun_out/pmc_${tag}_$i.txt
  else
    echo "pass $i failed"; tail -3 /tmp/pmc_$i.log | cut -c1-300
    tail -3 /tmp/pmc_$i.log > $R/gpurun_out/pmc_${tag}_$i.txt
  fi
done
