#!/bin/bash
# rocprofv3 kernel statistics of one kbench.py run: scripts/prof_kbench.sh <outdir> <lib suffix or ""> [kbench args...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/$1; SUF=$2; shift; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
GSPLAT_HIP_LIB=$R/gaussian_splatting_amd/libgsplat_hip$SUF.so rocprofv3 --kernel-trace --stats -d $OUT -o kb -- python $R/scripts/kbench.py "$@" > $OUT/kbench.log 2>&1
python $R/scripts/rocpd_stats.py $OUT/kb_results.db $OUT/kernel_stats.csv > $OUT/kernel_stats.txt 2>&1
rm -f $OUT/kb_results.db
tail -n 1 $OUT/kbench.log | cut -c1-300
grep -E "gs::" $OUT/kernel_stats.txt | head -14 | cut -c1-64,100-150
