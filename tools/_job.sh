set -o pipefail
O=gpurun_out/r04f; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; rc=$?; tail -3 $O/pytest_gpu.log; echo "pytest rc=$rc"
[ $rc -eq 0 ] || { grep -n "^FAILED\|^E " $O/pytest_gpu.log | head -20; exit 1; }
grep -h "commit phases" $O/pytest_gpu.log | head -3
python tools/build_probe.py --reps 3 2>&1 | grep -v amdgpu.ids | tee $O/build_probe.txt
SKIP_TESTS=1 bash tools/gpu_variants_ab.sh early 2>&1 | tee $O/early_rays_ab.txt
