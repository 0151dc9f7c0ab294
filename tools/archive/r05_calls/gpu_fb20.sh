set -e
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/fb20.log; : > $L
for rep in 1 2 3; do
for o in "" "--opt equal_passes=0"; do
  echo "[${o:-default}] 20" >> $L
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-strict $o 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'], d['config'].get('passes_of_timed_draw'))" >> $L
done; done
cat $L
