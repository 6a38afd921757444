#!/bin/bash
# the search kernel on its own: every PathAligner search of the bench workload through the memo and xm_wave_search_kernel (no inline searches)
cd $GRAFT_REPO_ROOT
export XM_WAVE=1 XM_WAVE_INLINE_SEARCH=0 XM_TRACE_PASSES=1
python3 scripts/gpu_wave_ticks.py se ${1:-1000000} 2>&1 | grep "search kernel\|Mticks\|tiers" | tail -14
