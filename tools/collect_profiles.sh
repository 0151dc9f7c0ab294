#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's numbers on the GPU box (run through gpurun):
#   kernel-trace + stats of the default bench command, then HBM traffic counters in separate --pmc passes
#   (FETCH_SIZE and WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --no-strict > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
echo "trace rc=$?"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --no-strict --steps 8 --warmup 2 > /dev/null 2> $OUT/pmc_fetch.err
echo "fetch rc=$?"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --no-cpu-baseline --no-strict --steps 8 --warmup 2 > /dev/null 2> $OUT/pmc_write.err
echo "write rc=$?"
timeout -k 10 200 rocprofv3 --pmc VALUBusy VALUUtilization --output-format csv -d $OUT/pmc_valu -- python3 $R/bench.py --no-cpu-baseline --no-strict --steps 8 --warmup 2 > /dev/null 2> $OUT/pmc_valu.err
echo "valu rc=$?"
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d $OUT/pmc_insts -- python3 $R/bench.py --no-cpu-baseline --no-strict --steps 8 --warmup 2 > /dev/null 2> $OUT/pmc_insts.err
echo "insts rc=$?"
cd $R && python3 tools/summarize_profiles.py $OUT $TAG
