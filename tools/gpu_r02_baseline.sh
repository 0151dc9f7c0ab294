#!/bin/bash
# Round-2 baseline measurements on the GPU box.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02a; mkdir -p $O
cd $R
python3 tools/calibrate.py > $O/calibrate.json 2> $O/calibrate.err; echo "calibrate rc=$?"; cat $O/calibrate.json
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver-style rc=$?"; cut -c1-400 $O/bench_driver.json
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strict --opt frames_in_flight=1 --opt frame_batch=1 > $O/bench_serial.json 2> $O/bench_serial.err; echo "serial rc=$?"; cut -c1-300 $O/bench_serial.json
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strict --opt frames_in_flight=3 --opt frame_batch=1 > $O/bench_fif3.json 2> $O/bench_fif3.err; echo "fif3 rc=$?"; cut -c1-300 $O/bench_fif3.json
python3 bench.py --steps 480 --warmup 48 --no-cpu-baseline > $O/bench_long.json 2> $O/bench_long.err; echo "long rc=$?"; cut -c1-300 $O/bench_long.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_serial -- python3 $R/bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-strict --opt frames_in_flight=1 --opt frame_batch=1 > $O/trace_serial.json 2> $O/trace_serial.err; echo "trace serial rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_serial_b4 -- python3 $R/bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-strict --opt frames_in_flight=1 --opt frame_batch=4 > $O/trace_serial_b4.json 2> $O/trace_serial_b4.err; echo "trace serial b4 rc=$?"
for d in trace_serial trace_serial_b4; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); echo "== $d"; head -8 $f | cut -c1-200; done
cd $R
tools/pmc_pass.sh r02a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" \
  "SQ_WAVES SQ_LEVEL_WAVES SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_THREAD_CYCLES_VALU" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
  "GRBM_GUI_ACTIVE GRBM_TA_BUSY" "TCP_GATE_EN1_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_READ_sum" "TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum TD_TC_STALL_sum" > $O/pmc.log 2>&1
echo "pmc rc=$?"; cat $O/pmc.log
