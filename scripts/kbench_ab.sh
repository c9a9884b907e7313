#!/bin/bash
# Kernel-level A/B of the render kernels: the working tree's libgsplat_hip.so against a build of HEAD's render.hip, the two
# alternating on one box (scripts/kbench.py: median GPU ms per entry point, image / num_splats bit-equality and the slab's
# distance to the baseline's), at D (depth cut), on a band of D with the depth-segmented backward, and at B.  This is how
# the visit-loop changes of round 6 were measured (profiles/r06/no_zero_fill_ab.txt, fwd_done_from_acc_ab.txt).
#   here:        scripts/kbench_ab.sh build         (compiles HEAD's render.hip into gaussian_splatting_amd/libgsplat_hip_base.so)
#   on the box:  gpurun -- 'bash scripts/kbench_ab.sh run <outdir under gpurun_out> [pytest files...]'
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
C=$R/gaussian_splatting_amd/csrc
if [ "$1" = "build" ]; then
  cd $C
  make
  git show HEAD:gaussian_splatting_amd/csrc/render.hip > render_base_tmp.hip
  F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize -Wall -Wno-unused-function"
  /opt/rocm/bin/hipcc $F -c render_base_tmp.hip -o /tmp/render_base.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libgsplat_hip_base.so gs_error.o per_gaussian.o preprocess.o binning.o /tmp/render_base.o halo.o train_ops.o loss.o densify.o
  rm render_base_tmp.hip
  exit 0
fi
shift
O=$R/gpurun_out/${1:-kbench_ab}; shift || true
mkdir -p $O
cd $R
B=$R/gaussian_splatting_amd/libgsplat_hip_base.so
N=$R/gaussian_splatting_amd/libgsplat_hip.so
kb() { GSPLAT_HIP_LIB=$1 timeout 200 python scripts/kbench.py --reps 40 "${@:3}" > $O/$2.json 2>>$O/err.txt; }
kb $B k_base_1 --workload D --depth-cut 1 --tag base --save /tmp/ref.pt
for i in 1 2 3; do
  kb $N k_new_$i --workload D --depth-cut 1 --tag new --check /tmp/ref.pt
  kb $B k_base_$((i+1)) --workload D --depth-cut 1 --tag base --check /tmp/ref.pt
done
kb $B kb_base --workload D --rows 24 31 --segments 1 --tag base_band --save /tmp/refb.pt
kb $N kb_new --workload D --rows 24 31 --segments 1 --tag new_band --check /tmp/refb.pt
kb $B kB_base --workload B --tag base_B --save /tmp/refB.pt
kb $N kB_new --workload B --tag new_B --check /tmp/refB.pt
python - $O <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + '/k*.json')):
    try:
        d = json.loads(open(f).read().strip().split('\n')[-1])
        print(f.split('/')[-1], d['tag'], {k: v for k, v in d['median_ms'].items() if 'render' in k}, d.get('image_equal'),
              d.get('nsp_equal'), d.get('slab_rel_err_floor_1e-2'))
    except Exception as e:
        print(f, 'ERR', e)
PY
if [ $# -gt 0 ]; then
  timeout 1200 python -m pytest "$@" -x -q > $O/pytest.txt 2>&1 || true
  grep -E "passed|failed" $O/pytest.txt | tail -2
fi
