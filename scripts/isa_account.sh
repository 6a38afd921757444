#!/bin/bash
# Static ISA account of a device function of xm_capi.hip (default: the slot search, pathSearchLds): instructions by category, from llvm-objdump of the gfx950 code
# object, and what LLVM's uniformity analysis takes for divergent in it.  Runs on the CPU (hipcc cross-compiles).
# usage: scripts/isa_account.sh [function-name pattern] [source tree, default: this one] [extra hipcc flags]
set -e
PAT=${1:-pathSearchLds}
TREE=${2:-$(cd "$(dirname "$0")/.." && pwd)}
EXTRA=$3
T=$(mktemp -d)
LL=/opt/rocm/lib/llvm/bin
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function -DXM_BUILD_STAMP=\"x\" $EXTRA"
(cd $TREE/mapper_amd/csrc && /opt/rocm/bin/hipcc $FLAGS -mllvm -wwm-regalloc=basic --offload-device-only -c -o $T/dev.o xm_capi.hip 2>/dev/null)
$LL/clang-offload-bundler --unbundle --type=o --input=$T/dev.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co
$LL/llvm-objdump -d $T/dev.co > $T/dev.s
awk -v pat="$PAT" '/^[0-9a-f]+ </ { p = ($0 ~ pat) } p {print}' $T/dev.s > $T/fn.s
n() { grep -cE "^\s+($1)" $T/fn.s || true; }
echo "function matching '$PAT': $(grep -cE '^\s+[a-z]' $T/fn.s) instructions"
echo "  execution-mask bookkeeping (s_and_saveexec, s_or/andn2/xor/orn2/and _b64, s_cbranch_exec*, s_mov_b64): $(n 's_and_saveexec|s_or_b64|s_andn2|s_xor_b64|s_orn2|s_and_b64|s_cbranch_exec|s_mov_b64|s_andn2_saveexec|s_or_saveexec')"
echo "  scalar branches (s_cbranch_scc*, s_cbranch_vcc*, s_branch): $(n 's_cbranch_scc|s_cbranch_vcc|s_branch')"
echo "  SGPR spills to VGPR lanes (v_readlane, v_writelane): $(n 'v_readlane|v_writelane')"
echo "  v_readfirstlane: $(n 'v_readfirstlane')"
echo "  moves (v_mov_b32, v_mov_b64, s_mov_b32): $(n 'v_mov_b32|v_mov_b64|s_mov_b32')"
echo "  fp64 arithmetic and compares (v_*_f64): $(grep -cE '^\s+v_[a-z_]+_f64' $T/fn.s || true)"
echo "  integer SALU (s_add/sub/mul/lshl/lshr/ashr/and_b32/or_b32/min/max/cmp/cselect_b32): $(n 's_add_|s_addc|s_sub_|s_mul_|s_lshl_|s_lshr_|s_ashr_|s_and_b32|s_or_b32|s_min_|s_max_|s_cmp_|s_cselect_b32|s_bcnt|s_movk')"
echo "  integer VALU (v_add/sub/mul/lshl/and/or/cmp/cndmask, not f64): $(grep -E '^\s+v_(add|sub|mul|lshl|lshr|ashr|and|or|xor|cmp|cndmask|mad|bfe|bitop)' $T/fn.s | grep -vc f64 || true)"
echo "  LDS (ds_*): $(n 'ds_')   global_*: $(n 'global_')   flat_*: $(n 'flat_')   scratch_*: $(n 'scratch_')"
echo "  s_waitcnt: $(n 's_waitcnt')   s_nop: $(n 's_nop')"
(cd $TREE/mapper_amd/csrc && /opt/rocm/bin/hipcc $FLAGS --offload-device-only -emit-llvm -S -o $T/dev.ll xm_capi.hip 2>/dev/null)
$LL/opt -mtriple=amdgcn-amd-amdhsa -mcpu=gfx950 -passes='print<uniformity>' -disable-output $T/dev.ll 2> $T/uni.txt
awk -v pat="$PAT" '/UniformityInfo for function/ {p = ($0 ~ pat)} p' $T/uni.txt > $T/uni_fn.txt
echo "  LLVM uniformity analysis: $(grep -c 'DIVERGENT:' $T/uni_fn.txt) divergent values, $(grep -c 'DIVERGENT:   br' $T/uni_fn.txt) divergent branches"
if [ -n "$ISA_KEEP" ]; then cp $T/fn.s $ISA_KEEP.s; cp $T/uni_fn.txt $ISA_KEEP.uniformity.txt; fi
rm -rf $T
