# Round-3 evidence, collected on the GPU box (run through gpurun, ~10 minutes in two calls):
#   part 1: bash tools/collect_round3.sh 1   -> cfg3 fp32 / bf16 and cfg2 passes (kernel stats, FETCH, WRITE, MFMA busy)
#   part 2: bash tools/collect_round3.sh 2   -> cfg4, cfg5 (bf16: all passes; fp32: kernel stats), the bench lines of every
#                                               workload from one call, the recurrent kernels' cycle anatomy (prof build)
# then here: bash tools/summarize_round3.sh  -> profiles/r03_*
set -u
if [ "${1:-1}" = 1 ]; then
  bash tools/collect_profiles.sh r3_cfg3 | tail -1
  bash tools/collect_profiles.sh r3_cfg3_bf16 --compute-dtype bf16 | tail -1
  bash tools/collect_profiles.sh r3_cfg2 --workload cfg2 | tail -1
else
  bash tools/collect_profiles.sh r3_cfg4 --workload cfg4 | tail -1
  bash tools/collect_profiles.sh r3_cfg5_bf16 --workload cfg5 --compute-dtype bf16 | tail -1
  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3_cfg5_stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --graph off --workload cfg5 > gpurun_out/prof_r3_cfg5_stats.log 2>&1
  : > gpurun_out/r3_bench_lines.jsonl
  for w in "cfg3" "cfg3 --compute-dtype bf16" "cfg2" "cfg4" "cfg4 --compute-dtype bf16" "cfg5 --steps 5 --warmup 2" "cfg5 --compute-dtype bf16 --steps 5 --warmup 2" "rnn" "mlp" "ligru" "gru"; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --workload $w 2> "gpurun_out/r3_bench_$(echo $w | tr -c 'a-z0-9\n' _).err" | tail -1 >> gpurun_out/r3_bench_lines.jsonl
  done
  SPARCH_HIP_LIB=sparch_amd/libsparch_hip_prof.so timeout -k 10 120 python tools/rec_prof.py > gpurun_out/r3_rec_anatomy.txt 2>&1
  wc -l gpurun_out/r3_bench_lines.jsonl
fi
