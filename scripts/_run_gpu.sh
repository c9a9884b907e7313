mkdir -p gpurun_out/r2k; O=gpurun_out/r2k
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
grep -v "^{" $O/pytest.txt | tail -12
