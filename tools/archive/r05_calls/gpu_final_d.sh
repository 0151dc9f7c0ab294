#!/bin/bash
# round 5, final evidence on the final tree: the suite; the rocprofv3 evidence (tools/collect_profiles.sh r05); bench.py's own lines WITH that evidence in profiles/ (hash-gated)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05_final; mkdir -p $O; cd $R
timeout -k 10 800 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || { grep -E "Error|error|assert|FAILED" $O/pytest.log | head -20; exit 1; }
bash tools/archive/r05_calls/gpu_final_c.sh
