#!/bin/bash
# Round 6: soak of the device chain with the CG loop's progress word (polling,
# sleeping between tests, other AHEADs): same seed, same start, run twice per
# variant -- every saved array bit for bit equal within a variant AND across
# variants (the host side must not change a sample).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
echo "Round 6 -- scripts/soak_chain.py under the variants of the CG loop's host side"
for env in "BBX_NOP=1" "BBX_CG_SLEEP=1" "BBX_CG_AHEAD=3" "BBX_CG_AHEAD=1 BBX_CG_SLEEP=1"; do
  echo "-- $env"
  env $env BBX_SOAK_SAVE=gpurun_out/soak_$(echo $env | tr ' =' '__').npz python3 scripts/soak_chain.py config2 1500 2>&1 | grep -v amdgpu.ids
  env $env BBX_SOAK_SAVE=gpurun_out/soak3_$(echo $env | tr ' =' '__').npz python3 scripts/soak_chain.py config3 400 2>&1 | grep -v amdgpu.ids
done
python3 - <<'PY'
import glob
import numpy as np
for pre in ("soak_", "soak3_"):
    files = sorted(glob.glob("gpurun_out/%s*.npz" % pre))
    base = np.load(files[0])
    for f in files[1:]:
        other = np.load(f)
        same = all(np.array_equal(base[k], other[k]) for k in base.files)
        print("%s == %s: %s" % (files[0].split("/")[-1], f.split("/")[-1],
                                "bit for bit" if same else "DIFFERENT"))
PY
rm -f gpurun_out/soak_*.npz gpurun_out/soak3_*.npz
