mkdir -p gpurun_out/r3g
for ws in 68719476736 137438953472 206158430208; do
  echo "workspace_bytes=$ws" >> gpurun_out/r3g/dragon4k_workspace.txt
  timeout -k 10 200 python bench.py --workload dragon4k --steps 2 --warmup 0 --no-cpu --in-flight 1 --ctx-option workspace_bytes=$ws 2>>gpurun_out/r3g/dragon4k.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], d['value'], r['launches_per_frame'], r['frame_device_ms_in_timed_region'])" >> gpurun_out/r3g/dragon4k_workspace.txt || exit 1
done
cat gpurun_out/r3g/dragon4k_workspace.txt
