set -u
bash tools/collect_profiles.sh r3_cfg4 --workload cfg4 | tail -1
bash tools/collect_profiles.sh r3_cfg5_bf16 --workload cfg5 --compute-dtype bf16 | tail -1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3_cfg5_stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --graph off --workload cfg5 > gpurun_out/prof_r3_cfg5_stats.log 2>&1
: > gpurun_out/r3_bench_lines.jsonl
for w in "cfg3" "cfg3 --compute-dtype bf16" "cfg2" "cfg4" "cfg4 --compute-dtype bf16" "cfg5 --steps 5 --warmup 2" "cfg5 --compute-dtype bf16 --steps 5 --warmup 2" "rnn" "mlp" "ligru" "gru"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --workload $w 2>/dev/null | tail -1 >> gpurun_out/r3_bench_lines.jsonl
done
SPARCH_HIP_LIB=sparch_amd/libsparch_hip_prof.so timeout -k 10 120 python tools/rec_prof.py > gpurun_out/r3_rec_anatomy.txt 2>&1
wc -l gpurun_out/r3_bench_lines.jsonl
