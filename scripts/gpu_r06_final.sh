#!/bin/bash
# round 6, final build: the judged measurements (scripts/gpu_profile_round_r06.sh all), the whole GPU tier, smoke(), the DPP check
R=$GRAFT_REPO_ROOT
cd $R
ulimit -c 0
O=$R/gpurun_out/final; mkdir -p $O
bash scripts/gpu_profile_round_r06.sh r06 all > $O/profile_round.log 2>&1; echo "profile round rc=$?"
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "gpu tier rc=$?"; tail -3 $O/gputests.log | cut -c1-300
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log | cut -c1-300
timeout 60 scripts/dpp_check/dpp_check > $O/dpp_check.log 2>&1; cat $O/dpp_check.log
cut -c1-600 $R/gpurun_out/r06/bench_line.json
