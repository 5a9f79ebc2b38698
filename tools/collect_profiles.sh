#!/bin/bash
# Collects the round's rocprofv3 evidence for bench.py on the GPU box (run through gpurun): kernel stats, the two HBM
# traffic passes and the MFMA-busy pass — each its own rocprofv3 run, PMC passes with kernel trace only.
# usage: bash tools/collect_profiles.sh TAG [bench args...]      -> gpurun_out/prof_TAG_{stats,fetch,write,mfma}
set -u
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_$TAG
rm -rf ${O}_stats ${O}_fetch ${O}_write ${O}_mfma
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --graph off "$@" > ${O}_stats.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d ${O}_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --graph off "$@" > ${O}_fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d ${O}_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --graph off "$@" > ${O}_write.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d ${O}_mfma -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --graph off "$@" > ${O}_mfma.log 2>&1
echo "rc=$?"; du -sh ${O}_* | tail -8
