cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_accumulate_image.py tests/test_gpu_repeat.py -x -q 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 --no-cpu --no-gibbs --no-c5 2>gpurun_out/r3/bench_q2.err | tail -1 > gpurun_out/r3/bench_q2.json; tail -3 gpurun_out/r3/bench_q2.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r3/bench_q2.json'))
print(d['ms_per_step'], [(k['kernel'],k['avg_ms'],k.get('executed_frac_of_sustained')) for k in d['roofline']['kernels']], d.get('parity',{}).get('max_rel_dG'), d.get('full_size_check',{}).get('pass'))
print('m1024', d.get('m1024',{}).get('ms_per_step'), [(k['kernel'],k['avg_ms']) for k in d.get('m1024',{}).get('roofline',{}).get('kernels',[])])
PY
