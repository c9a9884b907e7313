mkdir -p gpurun_out/r2o; O=gpurun_out/r2o
python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_sharded.py -m gpu -x -q 2>&1 | tail -8 | grep -v "^{" 
L=gaussian_splatting_amd
python scripts/kbench.py --workload D --full --tag coop > $O/kb_D.json 2>$O/err.txt; python scripts/kbench.py --workload B --full --tag coop > $O/kb_B.json 2>>$O/err.txt
python -c "
import json
for w in 'DB':
    d=json.load(open('$O/kb_%s.json'%w)); print(w, d['median_ms'])"
timeout 600 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --spinup-steps 10 --no-copy-bandwidth --train-loop 7000 > $O/bench_train_loop.json 2>>$O/err.txt; echo "rc=$?"
python - <<PY
import json
d=json.loads(open('$O/bench_train_loop.json').read().strip().splitlines()[-1])
t=d['train_loop']; print({k:v for k,v in t.items() if k not in ('n_gaussians_trace','note')})
PY
