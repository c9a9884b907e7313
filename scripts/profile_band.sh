#!/bin/bash
# rocprofv3 kernel trace of one rank's band frame (scripts/band_cost.py, owner mode); writes a per-kernel summary.
# usage: scripts/profile_band.sh <outdir> [band_cost args...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT -o band -- python $R/scripts/band_cost.py --grad-mode owner --steps 200 "$@" > $OUT/band.log 2>&1
python $R/scripts/rocpd_stats.py $OUT/band_results.db $OUT/kernel_stats.csv > $OUT/kernel_stats.txt 2>&1
rm -f $OUT/band_results.db
tail -n 3 $OUT/band.log
head -24 $OUT/kernel_stats.txt
