#!/bin/bash
# Round 4: contexts per GPU against the number of hardware queues the HIP runtime uses (GPU_MAX_HW_QUEUES, default 4: with more than two contexts - a compute and a
# copy stream each - launches of different contexts share a queue and run one after the other).  Run on the GPU box: bash scripts/gpu_ctx_queues_r04.sh [sweep]
cd $GRAFT_REPO_ROOT
run() { # cfg ctx queues steps
  r=$(GPU_MAX_HW_QUEUES=$3 timeout 380 python bench.py --config $1 --contexts $2 --steps $4 --warmup 4 --cpu-sample 0 --seed-probes 0 --wave-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], (d.get('single_context') or {}).get('value'))" 2>/dev/null)
  echo "config $1 contexts $2 GPU_MAX_HW_QUEUES $3 -> $r"
}
if [ "${1:-first}" = first ]; then
run 1rep 2 4 12
run 1rep 4 8 16
run 1rep 4 16 16
run 1rep 6 16 18
run 1 2 4 20
run 1 3 8 24
run 1 4 8 24
else
run 1rep 3 8 15
run 1rep 5 16 20
run 2 2 4 12
run 2 3 8 12
run 2 4 8 16
run 4shape 2 4 8
run 4shape 3 8 9
run 4shape 4 8 12
run 1 2 8 20
fi
