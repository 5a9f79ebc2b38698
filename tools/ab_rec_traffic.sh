#!/bin/bash
# HBM write / fetch counters of the recurrent kernels alone (tools/rec_time.py, 8 launches) for several library builds,
# one rocprofv3 --pmc pass each (kernel trace only).  usage: tools/ab_rec_traffic.sh lib1 lib2 ...  ("main" = shipped)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $R
for L in "$@"; do
  if [ "$L" = main ]; then F=$R/sparch_amd/libsparch_hip.so; else F=$R/sparch_amd/libsparch_hip_$L.so; fi
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/rt_${L}_$C
    SPARCH_HIP_LIB=$F REC_TIME_ITERS=8 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/rt_${L}_$C -- python3 tools/rec_time.py > gpurun_out/rt_${L}_$C.log 2>&1 || exit 1
  done
done
python3 - "$@" <<'PY'
import csv, glob, sys, collections
for L in sys.argv[1:]:
    out = {}
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(f"gpurun_out/rt_{L}_{C}/*/*counter_collection.csv")[0]
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == C and ("rec_fwd" in r["Kernel_Name"] or "rec_bwd" in r["Kernel_Name"]):
                acc[r["Kernel_Name"].split("<")[0].split("::")[-1]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out.setdefault(k, {})[C] = sum(v) / len(v)
    for k, d in out.items():
        print(f"{L:6s} {k}: FETCH {d['FETCH_SIZE'] / 1024:.0f} MiB (x2 = {2 * d['FETCH_SIZE'] / 1024:.0f}), WRITE {d['WRITE_SIZE'] / 1024:.0f} MiB, corrected total {(2 * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024 / 1e6:.0f} MB")
PY
