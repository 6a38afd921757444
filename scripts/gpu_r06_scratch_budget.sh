#!/bin/bash
# round 6 (verdict 5): the headline against the scratch a context may take - three contexts (the default) and two, 8 / 16 / 32 / 64 GiB each and unlimited
R=$GRAFT_REPO_ROOT
cd $R
Q="--cpu-sample 0 --seed-probes 0 --wave-steps 0 --single-context-steps 0 --stream-batches 0 --end-to-end-reads 0 --steps 12"
for c in 3 2; do for g in 8 16 32 64 0; do
  timeout 300 python3 bench.py $Q --contexts $c --scratch-gib $g 2>/dev/null | tail -n 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('contexts $c scratch limit $g GiB: %.2f M reads/s, %.1f ms per step, scratch each %s GiB, light %.1f ms gapped %.1f ms per step, launches per step %.1f' % (d['value'], d['ms_per_step'], d['contexts']['scratch_gib_each'], d['roofline']['kernel_ms_by_pass']['light_pass'], d['roofline']['kernel_ms_by_pass']['gapped_and_rerun_passes'], d['roofline']['launches_per_step']))"
done; done
