#!/bin/bash
# kernel logic under ASan + UBSan on the CPU, in the pass shapes of the host simulation
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/xm_asan
g++ -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fsanitize=address,undefined -fno-sanitize-recover=undefined -w -o /tmp/xm_asan/libxm_hostsim.so tests/hostsim/xm_hostsim.cpp
export XM_SANITIZED_LIB=/tmp/xm_asan/libxm_hostsim.so LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0
python scripts/cpu_sanitize.py                                             # light pass -> hand-over -> gapped pass (the product's default)
XMSIM_INLINE=1 python scripts/cpu_sanitize.py                              # plain inline run
XMSIM_NO_HANDOVER=1 python scripts/cpu_sanitize.py                         # light pass -> re-seeding gapped pass, two deferred rounds
XMSIM_NO_HANDOVER=1 XMSIM_DEFER_ROUNDS=100 python scripts/cpu_sanitize.py  # every search deferred
XMSIM_WAVE=1 python scripts/cpu_sanitize.py                                # the wave-per-read form (tiers, inline searches, memo rounds) in the host simulation
XM_BUILD_HYBRID_ON_HOST=1 XM_BUILD_SPLICE_MIN=2048 python scripts/cpu_sanitize.py  # index: plain rule + windows of multi blocks around ambiguous bases (the GPU build's composition), long runs of N split
