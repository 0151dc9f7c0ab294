#!/bin/bash
# Whose writes does rocprofv3 count for a traversal launch?  WRITE_SIZE per dispatch of serialised 8-frame passes (the configuration of tools/collect_profiles.sh), for the library as it is and
# for a counters-only variant whose traversal kernels store NOTHING (tools/build_variant.sh skipboth "-DMRT_SKIP_HITS=1 -DMRT_SKIP_LIT=1" on a tree
# where the two emit lambdas of renderer.hip read `if (lit) { if (!MRT_SKIP_LIT) lit[...] = 1; }` and `if (!MRT_SKIP_HITS) qstore(&hits[j], ...)`, both macros defaulting to 0; the image is garbage).  DESIGN.md §6.82.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/write_attr; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in head skipboth; do
  if [ $v = head ]; then unset MRT_LIB_PATH; else export MRT_LIB_PATH=$R/metal-raytracing_amd/variants/libmrt_hip_$v.so; fi
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/$v -- python3 $R/bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-strict --no-latency --opt frames_in_flight=1 --opt frame_batch=8 > /dev/null 2> $O/$v.err || { echo "$v failed"; tail -3 $O/$v.err; exit 1; }
done
python3 - <<PY
import csv,glob
for v in ('head','skipboth'):
    f=glob.glob('$O/'+v+'/**/*counter_collection.csv',recursive=True)[0]
    rows=[r for r in csv.DictReader(open(f)) if r['Counter_Name']=='WRITE_SIZE']
    rows.sort(key=lambda r:int(r['Dispatch_Id']))
    seq=[(('trace' if 'k_trace_mixed' in r['Kernel_Name'] else 'shade' if 'k_shade' in r['Kernel_Name'] else 'accumulate' if 'k_accumulate' in r['Kernel_Name'] else 'other'), float(r['Counter_Value'])/1024) for r in rows]
    seq=[x for x in seq if x[0]!='other']
    last=seq[-7:]          # the last pass: shade(0) trace(0) shade(1) trace(1) shade(2) trace(2) accumulate
    print(f"{v:9s} last 8-frame pass, WRITE_SIZE in MB:", ", ".join(f"{k} {mb:.1f}" for k,mb in last), "| sum", round(sum(mb for k,mb in last),1))
PY
