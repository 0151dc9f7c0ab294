#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02i; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver-style rc=$?"; python3 -c "
import json; d=json.loads(open('$O/bench_driver.json').read().strip().splitlines()[-1])
for k in ('value','ms_per_step','roofline','latency','valu_issue','strict_primary_plus_shadow'): print(k, json.dumps(d.get(k))[:900])"
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"; python3 -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
for k in ('value','ms_per_step','valu_issue','strict_primary_plus_shadow'): print(k, json.dumps(d.get(k))[:600])"
b() { python3 bench.py --steps ${STEPS:-20} --warmup ${WARM:-5} --no-cpu-baseline --no-strict --no-latency "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'])"; }
echo "c3: 4 bounces 64 frames"; STEPS=64 WARM=8 b --bounces 4
echo "irregular long"; STEPS=240 WARM=24 b --scene dragon_irregular
echo "dragon4 long"; STEPS=240 WARM=24 b --scene dragon4
echo "dragon4 two-level"; STEPS=64 WARM=8 b --scene dragon4 --sopt instancing=1
echo "garden 4k"; STEPS=96 WARM=12 b --scene garden --width 3840 --height 2160
echo "cornell 256"; STEPS=240 WARM=24 b --scene cornell --width 256 --height 256
echo "fif 3 batch 4 long"; STEPS=240 WARM=24 b --opt frames_in_flight=3
echo "fif 6 batch 4 long"; STEPS=240 WARM=24 b --opt frames_in_flight=6
echo "2-rank gloo tile"; python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --dist-backend gloo --steps 20 --warmup 5 2> $O/two.err | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'], d['scaling'], d['config']['shard'], d['config']['frame_batch'])"
