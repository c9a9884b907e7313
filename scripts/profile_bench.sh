#!/bin/bash
# rocprofv3 kernel trace of the bench command; writes a per-kernel summary CSV.
# usage: scripts/profile_bench.sh <outdir> [bench args...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT -o bench -- python $R/bench.py "$@" > $OUT/bench.log 2>&1
python $R/scripts/rocpd_stats.py $OUT/bench_results.db $OUT/kernel_stats.csv > $OUT/kernel_stats.txt 2>&1
rm -f $OUT/bench_results.db
tail -n 1 $OUT/bench.log
head -20 $OUT/kernel_stats.txt
