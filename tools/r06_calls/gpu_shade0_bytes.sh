#!/bin/bash
# round 6, VERDICT item 4: what k_shade_primary's 2.8 GB per 8-frame launch are made of — FETCH_SIZE and WRITE_SIZE (separate passes) with one stream switched off at a time
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_shade0; mkdir -p $O; cd $R
for v in default "halton_table=0" "primary_hint=0" "frame_bundle=0"; do
  tag=$(echo $v | tr '=' '_'); extra=""; [ "$v" = default ] || extra="--opt $v"
  echo "=== $v"
  BENCH_ARGS="$extra" bash tools/pmc_pass.sh r06_shade0/$tag "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" 2>&1 | grep -E "k_shade_primary|pass|failed"
done 2>&1 | tee $O/summary.txt
