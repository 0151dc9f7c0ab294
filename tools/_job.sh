set -o pipefail
O=gpurun_out/r04j; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_instancing.py -q -m gpu -x > $O/pytest_tl.log 2>&1; rc=$?; tail -3 $O/pytest_tl.log; echo "pytest rc=$rc"
[ $rc -eq 0 ] || { grep -n "^FAILED\|^E " $O/pytest_tl.log | head -20; exit 1; }
bash tools/gpu_two_level_binned_ab.sh $O
