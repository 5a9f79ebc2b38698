#!/bin/bash
# gpurun_out/prof_r3_* (tools/collect_round3.sh) -> the tracked summaries under profiles/ (run here, no GPU).
set -eu
cd "$(dirname "$0")/.."
export PROFILE_COMMIT=${PROFILE_COMMIT:-$(git rev-parse --short HEAD)}
S=tools/profile_summary.py; G=gpurun_out; P=profiles
stats() { python $S stats $G/prof_r3_$1_stats $P/r03_bench_kernel_stats_$1.md $P/r03_bench_kernel_stats_$1.csv "$2, commit $PROFILE_COMMIT" $3; }
pmc()   { python $S pmc $G/prof_r3_$1_fetch $G/prof_r3_$1_write $P/r03_pmc_traffic_$1.md $P/r03_pmc_traffic_$1.json "round 3: HBM traffic per launch, bench.py ($2), commit $PROFILE_COMMIT"; }
mfma()  { python $S mfma $G/prof_r3_$1_mfma $P/r03_pmc_mfma_$1.md "round 3: MFMA pipe utilisation, bench.py ($2), commit $PROFILE_COMMIT"; }
stats cfg3      "round 3: rocprofv3 --kernel-trace --stats of bench.py --steps 5 --warmup 2 (cfg3)" 7
stats cfg3_bf16 "round 3: rocprofv3 --kernel-trace --stats of bench.py --steps 5 --warmup 2 --compute-dtype bf16 (cfg3)" 7
stats cfg2      "round 3: rocprofv3 --kernel-trace --stats of bench.py --workload cfg2 --steps 5 --warmup 2, eager launches" 7
stats cfg4      "round 3: rocprofv3 --kernel-trace --stats of bench.py --workload cfg4 --steps 5 --warmup 2" 7
stats cfg5_bf16 "round 3: rocprofv3 --kernel-trace --stats of bench.py --workload cfg5 --compute-dtype bf16 --steps 5 --warmup 2" 7
stats cfg5      "round 3: rocprofv3 --kernel-trace --stats of bench.py --workload cfg5 --steps 3 --warmup 1 (fp32)" 4
for c in cfg3 cfg3_bf16 cfg2 cfg4 cfg5_bf16; do pmc $c "$c"; mfma $c "$c"; done
cp $G/r3_bench_lines.jsonl $P/r03_bench_lines.jsonl
cp $G/r3_rec_anatomy.txt $P/r03_rec_cycle_anatomy.txt
ls $P | grep -c r03
