#!/bin/bash
# round 3: gapped-pass launch shape on 1 kb queries with 1, 2 and 3 contexts (bench.py --config 4shape, 300 k queries per step)
cd $GRAFT_REPO_ROOT
for c in 1 2 3; do
  for s in "8 8" "4 8" "4 16" "8 6" "12 8"; do
    set -- $s
    line=$(XM_FULL_LPW=$1 XM_FULL_WAVES=$2 timeout 200 python bench.py --config 4shape --reads 300000 --steps 6 --contexts $c --cpu-sample 0 --seed-probes 0 --wave-steps 0 --single-context-steps 0 --stream-batches 0 2>/dev/null | tail -1)
    echo "contexts $c lpw $1 waves $2: $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["unit"], d["ms_per_step"], "ms/step identical", d.get("bit_identical"))')"
  done
done
