#!/bin/bash
# Usage (build container): bash tools/build_tree.sh <git-ref>   -- checks <git-ref> out as a worktree under _tree/ (ignored by git through
# .git/info/exclude, shipped to the GPU box by gpurun) and builds its library there, so that tools/gpu_ab_tree.sh can run THAT tree's bench
# beside this one's on ONE box.  Frame rates differ by +-3 % from box to box: comparing with a previous round's NUMBERS hid a 4 % regression
# for half of round 6 (LAB_NOTES.md); compare with its tree.      Remove with: git worktree remove --force _tree
set -e
cd "$(dirname "$0")/.."
REF=${1:?git ref}
grep -qx "_tree/" .git/info/exclude 2>/dev/null || echo "_tree/" >> .git/info/exclude
[ -d _tree ] && git worktree remove --force _tree
git worktree add -f _tree "$REF" > /dev/null
(cd _tree && python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2)
git -C _tree log --oneline | head -1
