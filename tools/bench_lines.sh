: > gpurun_out/r3_bench_lines.jsonl
for w in "cfg3" "cfg3 --compute-dtype bf16" "cfg2" "cfg4" "cfg4 --compute-dtype bf16" "cfg5 --steps 5 --warmup 2" "cfg5 --compute-dtype bf16 --steps 5 --warmup 2" "rnn" "mlp" "ligru" "gru"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --workload $w 2> "gpurun_out/r3_bench_$(echo $w | tr -c 'a-z0-9\n' _).err" | tail -1 >> gpurun_out/r3_bench_lines.jsonl
done
wc -l gpurun_out/r3_bench_lines.jsonl
