#!/bin/bash
# bench.py with the shipped library and with several diagnostic variant libraries, round-robin, in ONE gpurun call.
# usage: tools/ab_libs.sh "bench args" main variant1 variant2 ...
ARGS=$1; shift
for rep in 1 2; do
  for V in "$@"; do
    if [ "$V" = main ]; then F=$PWD/sparch_amd/libsparch_hip.so; else F=$PWD/sparch_amd/libsparch_hip_$V.so; fi
    SPARCH_HIP_LIB=$F timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 $ARGS 2>/dev/null | tail -1 > gpurun_out/abl_${V}_$rep.json
  done
done
python - "$@" <<'PY'
import json, sys
for V in sys.argv[1:]:
    for rep in (1, 2):
        try:
            d = json.loads(open(f"gpurun_out/abl_{V}_{rep}.json").read().strip().splitlines()[-1])
            k = d.get("kernels_ms_per_step", {})
            big = {n.split("[")[0] + "[" + n.split("[")[1][:18]: round(v, 3) for n, v in k.items() if v > 0.05}
            print(f"{V:6s} {d['ms_per_step']:.3f} ms  {big}")
        except Exception as e:
            print(V, rep, "failed", e)
PY
