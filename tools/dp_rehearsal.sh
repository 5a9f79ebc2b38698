#!/bin/bash
# Rehearsal of bench.py --gpus 2 on ONE GPU (both ranks on cuda:0, gloo rendezvous) under the three all-reduce
# policies; every rank's stderr is kept.  usage (through gpurun): bash tools/dp_rehearsal.sh [bench args]
set -u
export SPARCH_DIST_BACKEND=gloo SPARCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/dp_rehearsal; mkdir -p $O
port=29610
for pol in window deferred overlap; do
  port=$((port+1))
  SPARCH_DP_POLICY=$pol timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
     --master-port $port --redirects 2:3 --log-dir $O/logs_$pol bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline "$@" \
     > $O/bench_$pol.json 2> $O/bench_$pol.launcher.err
  echo "policy $pol rc=$?"
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$pol.json").read().strip().splitlines()[-1])
    print("  ", d["ms_per_step"], "ms/step", d["config"]["grad_allreduce"], d.get("degraded"))
except Exception as e:
    print("   no JSON line:", e)
PY
done
find $O -name "*.log" -size +0 | head -20
