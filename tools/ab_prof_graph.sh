#!/bin/bash
# rocprofv3 kernel traces of bench.py --graph on with the shipped library and with a variant library, in one call.
# usage: tools/ab_prof_graph.sh VARIANT [bench args]   -> gpurun_out/pg_{new,VARIANT}
V=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cd $R
rm -rf gpurun_out/pg_new gpurun_out/pg_$V
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pg_new -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph on "$@" > gpurun_out/pg_new.log 2>&1 &&
SPARCH_HIP_LIB=$R/sparch_amd/libsparch_hip_$V.so rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pg_$V -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph on "$@" > gpurun_out/pg_$V.log 2>&1
echo rc=$?
