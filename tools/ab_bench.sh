#!/bin/bash
# bench.py of this tree against the round-2 tree kept under .ab/r2 (built there; git-ignored), alternating, in ONE
# gpurun call: devices differ by several percent, so step times are only comparable inside one call.
# usage: tools/ab_bench.sh OUT_PREFIX [bench args]
P=$1; shift
for rep in 1 2; do
  (cd .ab/r2 && timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1) > ${P}_r2_$rep.json
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 > ${P}_new_$rep.json
done
python - <<PY
import json, glob
for f in sorted(glob.glob("${P}_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"], 3), "ms  host", d.get("host_enqueue_ms_per_step"))
    except Exception as e:
        print(f, "failed", e)
PY
