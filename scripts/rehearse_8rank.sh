#!/bin/bash
# Host-side rehearsal of BASELINE config 5 on the 1-GPU box: eight ranks (eight
# generators, eight layout builders side by side, rendezvous on 127.0.0.1, the
# 8-way gather over gloo) share the one device at config-3 size.  Run twice:
# with the builder's thread cap (cgroup quota / LOCAL_WORLD_SIZE) and with the
# old behaviour (64 threads per rank: BBX_BUILD_THREADS=0).  No scaling number
# comes out of this -- only "the set-up survives the CPU quota".
#   usage: scripts/rehearse_8rank.sh [out_dir] [tag]
out=${1:-gpurun_out}
tag=${2:-r06}
mkdir -p "$out"
{
  echo "nproc: $(nproc)   cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
  free -g | head -2
} | tee "$out/${tag}_8rank_host.txt"
avail_gb=$(awk '/MemAvailable/ {print int($2 / 1048576)}' /proc/meminfo)
if [ "$avail_gb" -lt 100 ]; then
  echo "only ${avail_gb} GB of host memory available: 8 ranks need ~70 GB; skipped" \
    | tee -a "$out/${tag}_8rank_host.txt"
  exit 0
fi
args="--gpus 8 --config config3 --steps 5 --warmup 2 --burnin 10 --cpu-baseline-iters 0 --repeat 1"
timeout 900 \
  python bench.py $args > "$out/${tag}_bench_8rank_shared_config3.json" \
  2> "$out/${tag}_bench_8rank_shared_config3.err"
echo "capped rc=$?" | tee -a "$out/${tag}_8rank_host.txt"
BBX_BUILD_THREADS=0 timeout 900 \
  python bench.py $args > "$out/${tag}_bench_8rank_shared_config3_uncapped.json" \
  2> "$out/${tag}_bench_8rank_shared_config3_uncapped.err"
echo "uncapped rc=$?" | tee -a "$out/${tag}_8rank_host.txt"
python - "$out" "$tag" <<'PY'
import json, sys
out = sys.argv[1]
tag = sys.argv[2]
for name in (tag + "_bench_8rank_shared_config3", tag + "_bench_8rank_shared_config3_uncapped"):
    try:
        line = json.loads(open("%s/%s.json" % (out, name)).read().strip().splitlines()[-1])
        print(name, "value", line["value"], "per_rank", line["config"]["per_rank"])
    except Exception as exc:
        print(name, "no line:", exc)
PY
