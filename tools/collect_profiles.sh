#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's numbers on the GPU box (run through gpurun):  tools/collect_profiles.sh r02
#   1. kernel trace + stats of the DRIVER'S command shape (20 steps in passes of 7 + 7 + 6 frames; warm-up = one full 8-frame pass)
#   2. kernel trace + stats of the SERIALISED frame (one stream, one frame per pass): kernel times that are kernel times
#   3. counters in separate --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md "rocprofv3 PMC slots"),
#      on serialised 8-frame passes (the default pass size): steps and warm-up are multiples of the batch, so every dispatch carries eight full frames
# A step that fails or is killed ends the script (no further GPU step after a failed one).
#   4. the VALU instruction-rate table (tools/valu_rates.hip) and the on-chip calibration
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
COMMON="--no-cpu-baseline --no-strict --no-latency"
# the instruction-rate tool is built from source here (the binary is git-ignored); without it step 4 is skipped, the summary still runs
[ -x $R/tools/valu_rates ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $R/tools/valu_rates $R/tools/valu_rates.hip || echo "valu_rates did not build: step 4 will be skipped"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_driver -- python3 $R/bench.py --steps 20 --warmup 8 $COMMON > $OUT/bench_driver_under_rocprof.json 2> $OUT/trace_driver.err || { echo "trace driver failed"; tail -5 $OUT/trace_driver.err; exit 1; }; echo "trace driver ok"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_serial -- python3 $R/bench.py --steps 20 --warmup 4 $COMMON --opt frames_in_flight=1 --opt frame_batch=1 --opt tile_groups=1 > $OUT/bench_serial_under_rocprof.json 2> $OUT/trace_serial.err || { echo "trace serial failed"; tail -5 $OUT/trace_serial.err; exit 1; }; echo "trace serial ok"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_serial_pass -- python3 $R/bench.py --steps 32 --warmup 8 $COMMON --opt frames_in_flight=1 --opt frame_batch=8 --opt tile_groups=1 > $OUT/bench_serial_pass_under_rocprof.json 2> $OUT/trace_serial_pass.err || { echo "trace serial pass failed"; tail -5 $OUT/trace_serial_pass.err; exit 1; }; echo "trace serial pass ok"
PMC="--steps 16 --warmup 8 $COMMON --opt frames_in_flight=1 --opt frame_batch=8 --opt tile_groups=1"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "VALUBusy VALUUtilization" "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$i -- python3 $R/bench.py $PMC > /dev/null 2> $OUT/pmc_$i.err || { echo "pmc $i ($grp) failed"; tail -5 $OUT/pmc_$i.err; exit 1; }; echo "pmc $i ($grp) ok"
done
cd $R
if [ -x tools/valu_rates ]; then tools/valu_rates > $OUT/valu_rates.json 2> $OUT/valu_rates.err || { echo "valu_rates failed"; exit 1; }; fi
python3 tools/calibrate.py > $OUT/calibrate.json 2> $OUT/calibrate.err || { echo "calibrate failed"; exit 1; }
python3 tools/summarize_profiles.py $OUT $TAG
