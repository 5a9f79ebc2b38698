timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "recurrent or dyadic or golden or rnn or RNN or ann or bidirectional or long_seq" > gpurun_out/r2_t.log 2>&1; tail -3 gpurun_out/r2_t.log
SPARCH_HIP_LIB=sparch_amd/libsparch_hip_prof.so timeout -k 10 120 python tools/rec_prof.py > gpurun_out/r2_anat.txt 2>&1; tail -17 gpurun_out/r2_anat.txt
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/r2_b.json 2> gpurun_out/r2_b.err; python -c "
import json
d=json.loads(open('gpurun_out/r2_b.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], {k:v for k,v in d['kernels_ms_per_step'].items()})"
