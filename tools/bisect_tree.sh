#!/bin/bash
# Build a past commit of this repository in a scratch worktree (tools/_build/wt_<rev>) with Trainer.capture's custom-loss refusal
# removed, for `bench.py --arch mae_b_16 --graph` under the ROCm debug agent (the round-3 replay fault).
REV=$1
ROOT=$(cd "$(dirname "$0")/.." && pwd)
WT=$ROOT/tools/_build/wt_$REV
rm -rf $WT; git -C $ROOT worktree prune
git -C $ROOT worktree add -f $WT $REV -q || exit 1
python - "$WT/noise_robust_vit_amd/train.py" <<'PY'
import re, sys
p = sys.argv[1]
s = open(p).read()
s2 = re.sub(r"        if self\.compute_loss is not None( and not _allow_custom_loss)?:\n(            #.*\n)*            raise RuntimeError\([^\n]*\n", "", s)
open(p, "w").write(s2)
print("refusal removed" if s2 != s else "no refusal in this revision")
PY
(cd $WT && python -m noise_robust_vit_amd.build 2>&1 | tail -1)
