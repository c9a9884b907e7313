mkdir -p gpurun_out/r2p; O=gpurun_out/r2p
L=gaussian_splatting_amd
GSPLAT_HIP_LIB=$L/libgsplat_hip_base.so python scripts/kbench.py --workload D --save /tmp/ref_D.pt --tag base > $O/kb_base.json 2>$O/err.txt
python scripts/kbench.py --workload D --check /tmp/ref_D.pt --tag asm > $O/kb_asm.json 2>>$O/err.txt
GSPLAT_HIP_LIB=$L/libgsplat_hip_base.so python scripts/kbench.py --workload B --save /tmp/ref_B.pt --tag base > $O/kbB_base.json 2>>$O/err.txt
python scripts/kbench.py --workload B --check /tmp/ref_B.pt --tag asm > $O/kbB_asm.json 2>>$O/err.txt
cat $O/kb_asm.json $O/kbB_asm.json
python -m pytest tests/test_gpu_scale.py tests/test_gpu_fused.py -m gpu -x -q 2>&1 | tail -4 | grep -v "^{"
