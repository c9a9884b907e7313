#!/bin/bash
# sweep COOP_MIN on the GPU box (rebuilds in the snapshot)
for v in 8 12 16 24 32 48; do
  sed -i "s/^constexpr int COOP_MIN = [0-9]*;/constexpr int COOP_MIN = $v;/" gaussian_splatting_amd/csrc/binning.hip
  (cd gaussian_splatting_amd/csrc && make > /dev/null 2>&1)
  echo "== COOP_MIN $v"
  scripts/profile_bench.sh gpurun_out/r06p/coop_$v --no-cpu-baseline --also= 2>&1 | grep -E "k_bin_(count|emit)_buckets|ms_per_step" | cut -c1-40,100-150
  scripts/profile_band.sh gpurun_out/r06p/coopb_$v --world 8 2>&1 | grep -E "gs::k_bin_(count|emit)\(|ms/step" | cut -c1-60,100-150
done
