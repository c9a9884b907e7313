#!/bin/bash
# Collects PMC counters for the fused frame in separate rocprofv3 passes (counters only with
# --kernel-trace, as the pool requires).  usage: scripts/pmc_passes.sh <workload> <outdir>
WL=${1:-D}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/${2:-gpurun_out/pmc_$WL}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# (GSPLAT_DEPTH_CUT: 1 = the depth-cut kernels from the first frame on, so that a kernel name has one meaning in the run)
export GSPLAT_DEPTH_CUT=${GSPLAT_DEPTH_CUT:-1}
# PMC_CMD overrides the profiled command (e.g. one rank's band frame: PMC_CMD="python $R/scripts/host_timeline.py --world 8 --steps 3")
CMD=${PMC_CMD:-"python $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --also="}
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC -d $OUT -o pass$i -- $CMD > $OUT/pass$i.log 2>&1
  echo "pass $i ($PMC): rc=$?"
  python $R/scripts/rocpd_pmc.py $OUT/pass${i}_results.db --match gs:: > $OUT/pass$i.txt 2>&1
  rm -f $OUT/pass${i}_results.db
done
ls -la $OUT
