#!/bin/bash
# A/B of diagnostic library builds on the recurrent cell kernels (tools/rec_time.py), all in ONE gpurun call.
# usage: tools/ab_rec.sh OUT.txt lib1 lib2 ...   (names as in sparch_amd/libsparch_hip_<name>.so; "main" = the shipped one)
OUT=$1; shift
: > $OUT
for rep in 1 2 3; do
for L in "$@"; do
  if [ "$L" = main ]; then F=sparch_amd/libsparch_hip.so; else F=sparch_amd/libsparch_hip_$L.so; fi
  echo -n "$L: " >> $OUT
  SPARCH_HIP_LIB=$F timeout -k 10 100 python tools/rec_time.py ${REC_ARGS:-} 2>&1 | tail -1 >> $OUT || exit 1
done
done
cat $OUT
