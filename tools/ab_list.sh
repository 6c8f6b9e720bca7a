mkdir -p gpurun_out/r3b && cd /root/repo && timeout -k 10 300 python -u -m pytest tests/test_gpu_cull.py -x -q --timeout 200 --timeout-method thread > gpurun_out/r3b/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3b/tests.log; tail -4 gpurun_out/r3b/tests.log
export RT_BENCH_LEG=headline
for rep in 1 2; do
for cfg in "cull3:RT_BENCH_CULL=3" "listexact:RT_LIST_GRID=16906" "listfull:RT_LIST_GRID=0" ; do
  name=${cfg%%:*}; kv=${cfg#*:}
  env $kv timeout -k 10 120 python bench.py --steps 200 --no-cpu-baseline > gpurun_out/r3b/$name.$rep.json 2> gpurun_out/r3b/$name.$rep.err || echo "bench $name failed"
  python - <<PY
import json
d=json.load(open("gpurun_out/r3b/$name.$rep.json")); r=d["roofline"]
print("$name", "$rep", "ms/step", round(d["ms_per_step"]*1e3,2), "kernel_ms", round(r["kernel_ms"]*1e3,1), "single", round(r["single_frame_us"],1))
PY
done; done
