mkdir -p gpurun_out/r2f; O=gpurun_out/r2f
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest.txt
tail -8 $O/pytest.txt
python bench.py --no-cpu-baseline > $O/bench_D.json 2>$O/err.txt
python bench.py --no-cpu-baseline --workload B > $O/bench_B.json 2>>$O/err.txt
python bench.py --no-cpu-baseline --workload C > $O/bench_C.json 2>>$O/err.txt
python scripts/render_stats.py --workload D --out $O/render_stats_D.json > /dev/null 2>>$O/err.txt
for w in D B C; do python -c "import json;d=json.load(open('$O/bench_$w.json'));print('$w',d['ms_per_step'],d['value'],d['roofline']['entry_ms_per_step'])"; done
