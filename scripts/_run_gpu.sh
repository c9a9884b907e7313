mkdir -p gpurun_out/r2m; O=gpurun_out/r2m
python -m pytest tests/test_gpu_densify.py tests/test_gpu_train_ops.py -m gpu -x -q 2>&1 | tail -30 > $O/pytest.txt
tail -30 $O/pytest.txt
