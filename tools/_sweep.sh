python tools/bench_short.py --steps 96
for v in nopk pkw5; do echo $v; MRT_LIB_PATH=metal-raytracing_amd/variants/libmrt_hip_$v.so python tools/bench_short.py --steps 96; done
