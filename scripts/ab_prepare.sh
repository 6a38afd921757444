#!/bin/bash
# puts the kernel sources of a git revision (default HEAD) into .ab_head/ for scripts/gpu_ab.sh
REV=${1:-HEAD}
rm -rf .ab_head; mkdir -p .ab_head/csrc .ab_head/include
for f in $(git ls-tree --name-only $REV mapper_amd/csrc/); do git show $REV:$f > .ab_head/csrc/$(basename $f); done
for f in $(git ls-tree --name-only $REV include/); do git show $REV:$f > .ab_head/include/$(basename $f); done
ls .ab_head/csrc .ab_head/include
