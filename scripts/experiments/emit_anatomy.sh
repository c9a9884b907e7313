#!/bin/bash
# What are the 86 us of k_bin_emit_buckets<1, 1024> (workload D)?  Timing-only variant builds of binning.hip (their tile
# lists are wrong), run through scripts/kbench.py --binning-only (nothing downstream of the sort runs: a first version of
# this script profiled bench.py and hung in the render on garbage lists) under rocprofv3, every step under a timeout.
#   base      the product
#   nostore   the key stores removed (walk, cursors and everything else kept)
#   plaincur  the cursor atomics replaced by plain LDS reads (keys land on top of each other; stores kept)
#   nowalk    no tile walk at all: cursor set-up + the trips' list / record loads + the per-Gaussian set-up
#   nosetup   cursor set-up only (no trips)
#   blk512    512 threads per bucket instead of 1024
R=$GRAFT_REPO_ROOT
F=$R/gaussian_splatting_amd/csrc/binning.hip
O=$R/gpurun_out/emit_anatomy; mkdir -p $O
cp $F /tmp/binning.orig
build_run() {
  (cd $R/gaussian_splatting_amd/csrc && timeout 300 make variant NAME=$1 VSRC=binning EXTRA= > $O/build_$1.log 2>&1) || { echo "build failed: $1"; cp /tmp/binning.orig $F; return; }
  cp /tmp/binning.orig $F
  echo "== $1"
  (cd /tmp && export TMPDIR=/tmp && GSPLAT_HIP_LIB=$R/gaussian_splatting_amd/libgsplat_hip_$1.so timeout 150 rocprofv3 --kernel-trace --stats -d $O/$1 -o k -- \
     python $R/scripts/kbench.py --workload D --binning-only --depth-cut 1 --reps 40 > $O/$1.log 2>&1)
  python $R/scripts/rocpd_stats.py $O/$1/k_results.db 2>/dev/null | grep -E "k_bin_emit_buckets<1|k_tile_sort_runs|k_bin_count_buckets|k_tile_sort<" | cut -c1-50,100-150
  rm -f $O/$1/k_results.db
}
edit() { python - "$1" <<'PY'
import sys
p='/root/repo/gaussian_splatting_amd/csrc/binning.hip'
import os
p=os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gaussian_splatting_amd/csrc/binning.hip'
s=open(p).read()
which=sys.argv[1]
store="                if (pos < cap) keys[pos] = k;\n            },\n            0, [&](int tile) { return s_cursor[tile - t0] >= 0; });"
cur="                const int pos = atomicAdd(&s_cursor[tile - t0], 1);\n                if (pos < cap) keys[pos] = k;\n            },\n            0, [&](int tile)"
walk="        wave_for_each_tile(\n            active, tw, ntx, key,\n            [&](int tile, uint64_t k) {\n                const int pos = atomicAdd(&s_cursor[tile - t0], 1);\n                if (pos < cap) keys[pos] = k;"
trips="    const int g0 = cs.boff2[sl], g1 = cs.boff2[sl + 1];\n    __syncthreads();\n    // (the next trip's record in flight"
for o in (store, cur, walk, trips): assert s.count(o)==1, o[:40]
if which=='nostore':
    s=s.replace(store, store.replace("if (pos < cap) keys[pos] = k;", "if (pos < -1) keys[pos] = k;"))
elif which=='plaincur':
    s=s.replace(cur, cur.replace("atomicAdd(&s_cursor[tile - t0], 1)", "s_cursor[tile - t0]"))
elif which=='nowalk':
    s=s.replace(walk, "        if (tw.w.sx == 12345 && key == 77) keys[0] = key;\n        if (cap < 0)\n"+walk)
elif which=='nosetup':
    s=s.replace(trips, trips.replace("g1 = cs.boff2[sl + 1];", "g1 = cap < 0 ? cs.boff2[sl + 1] : cs.boff2[sl];"))
elif which=='blk512':
    o="constexpr int GS_CUT_EMIT_BLOCK = 1024;"; assert s.count(o)==1
    s=s.replace(o,"constexpr int GS_CUT_EMIT_BLOCK = 512;")
open(p,'w').write(s)
PY
}
build_run base
for v in nostore plaincur nowalk nosetup blk512; do edit $v; build_run $v; done
cp /tmp/binning.orig $F
rm -f $R/gaussian_splatting_amd/libgsplat_hip_{base,nostore,plaincur,nowalk,nosetup,blk512}.so $R/gaussian_splatting_amd/csrc/binning_*.o
