#!/bin/bash
# bench.py with the shipped library against a diagnostic variant library (tools/build_variant.sh), alternating, in ONE
# gpurun call.   usage: tools/ab_lib.sh OUT_PREFIX VARIANT_NAME [bench args]
P=$1; V=$2; shift 2
for rep in 1 2; do
  SPARCH_HIP_LIB=$PWD/sparch_amd/libsparch_hip_$V.so timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 > ${P}_${V}_$rep.json
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 > ${P}_new_$rep.json
done
python - <<PY
import json, glob
for f in sorted(glob.glob("${P}_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = {n: v for n, v in d.get("kernels_ms_per_step", {}).items() if n.startswith("cell_")}
        print(f.split("/")[-1], round(d["ms_per_step"], 3), "ms", k)
    except Exception as e:
        print(f, "failed", e)
PY
