#!/bin/bash
# round 6: the rejection filter in front of PathAligner (xm_bound.h) - tests, then configs[4] at reduced size with the filter off and on (one context)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r06_filter}
mkdir -p $O
cd $R
SCALE=${2:-0.05}
READS=${3:-400000}
timeout 900 python3 -m pytest tests/test_gpu_bound.py -x -q 2>&1 | tail -5 > $O/tests.log
cat $O/tests.log
for f in 1 0; do
  XM_BOUND_FILTER=$f timeout 1200 python3 bench.py --config 4 --big-scale $SCALE --reads $READS --contexts 1 --steps 1 --warmup 1 --seed-probes 0 --stream-batches 0 --single-context-steps 0 --cpu-sample 20000 2> $O/bench_config4_filter$f.err | tail -n 1 > $O/bench_config4_filter$f.json
  python3 - <<PY
import json
d = json.load(open("$O/bench_config4_filter$f.json"))
print("filter $f:", d["value"], d["unit"], "ms/step", d["ms_per_step"], "bit_identical", d.get("bit_identical"), "counters", json.dumps(d.get("counters"))[:900])
print("   cpu", d.get("cpu_baseline"))
PY
  tail -2 $O/bench_config4_filter$f.err
done
