#!/bin/bash
# The scaling run on ONE 8-GPU MI355X node, exactly as the driver launches bench.py (one rank per GPU over RCCL / xGMI).
#   bash tools/scale_run.sh [model=m] [steps=200] [warmup=20] [collective=allgather|auto|ingraph|direct|gather] [gather=logits|masks]
# Writes gpurun_out/scale_<model>_<collective>_<gather>_N<n>.json (ONE JSON line each: `value` = whole-job frames/s) for N = 1 2 4 8 and
# prints value and value / (N x value_1).  --collective allgather is bench.py's default at N > 1; auto calibrates the two in-place all-gather forms
# (captured into the step's HIP graph | on RCCL's stream) and reports both figures under collective.calibration_ms_per_step.
# Per-link bytes per step and direction at N ranks, payload P bytes per rank (HyperSeg-M logits: 39.8 MB; masks: 0.5 MB):
#   allgather / ingraph : RCCL's choice; a single ring moves (N-1) x P through every link of the ring, its multi-ring / direct
#                         schedules on the fully connected xGMI mesh spread that to ~P per link
#   direct              : exactly P per link and direction (every shard crosses the one link between producer and consumer)
#   gather              : P per link into rank 0 only
# Each rank RECEIVES (N-1) x P per step in every all-to-all policy: 278.6 MB at N = 8 for HyperSeg-M = 357 GB/s at 1280 steps/s.
model=${1:-m}; steps=${2:-200}; warmup=${3:-20}; coll=${4:-allgather}; gather=${5:-logits}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
base=""
for n in 1 2 4 8; do
  out=gpurun_out/scale_${model}_${coll}_${gather}_N$n.json
  if [ "$n" = 1 ]; then
    timeout 900 python bench.py --gpus 1 --steps "$steps" --warmup "$warmup" --model "$model" --no-cpu-baseline --traffic off > "$out" 2> "${out%.json}.err"
  else
    timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $((29500 + n)) \
      bench.py --gpus "$n" --steps "$steps" --warmup "$warmup" --model "$model" --collective "$coll" --gather "$gather" > "$out" 2> "${out%.json}.err"
  fi
  rc=$?
  python - "$out" "$n" "$rc" "$base" <<'PY'
import json, sys
path, n, rc, base = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
try:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    c = d.get('collective') or {}
    eff = f"{d['value'] / (n * float(base)):.3f}" if base else '1.000'
    print(f"N={n} rc={rc} value={d['value']} frames/s ms/step={d['ms_per_step']} efficiency={eff} policy={c.get('policy')} "
          f"calibration={c.get('calibration_ms_per_step')} per_rank={d.get('per_rank_frames_per_s')}")
except Exception as e:
    print(f'N={n} rc={rc}: no JSON line ({e}); see {path[:-5]}.err')
PY
  [ "$n" = 1 ] && base=$(python -c "import json;print(json.loads(open('$out').read().strip().splitlines()[-1])['value'])" 2>/dev/null)
done
