cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/builders.log; : > $L
b() { timeout -k 10 300 python3 bench.py --steps 240 --warmup 24 --no-cpu-baseline --no-latency --no-strict $1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('   value', d['value'], 'ms/step', d['ms_per_step'], 'build ms', c.get('bvh_build_ms'), 'sah', c.get('sah_cost'), 'nodes', c.get('bvh_nodes'), 'depth', c.get('wide_depth'))"; }
for rep in 1 2; do
for o in "" "--sopt ploc_radius=32" "--sopt ploc_radius=64" "--sopt ploc_radius=128" "--sopt builder=2" "--sopt builder=0"; do echo "[${o:-default: PLOC radius 16}]" >> $L; b "$o" >> $L; done; done
cat $L
