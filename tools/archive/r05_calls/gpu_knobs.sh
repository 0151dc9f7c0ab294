cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/knobs.log; : > $L
b() { timeout -k 10 300 python3 bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-latency --no-strict $3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'])"; }
for rep in 1 2; do
for o in "" "--opt persist_chunk=128" "--opt persist_chunk=512" "--opt persist_chunk=1024" "--frames-in-flight 8" "--frames-in-flight 12" "--opt wave_slots=4096" "--opt wave_slots=2048" "--opt wave_slots=6144"; do
  echo "[${o:-default}] 240" >> $L; b 240 24 "$o" >> $L
done; done
cat $L
