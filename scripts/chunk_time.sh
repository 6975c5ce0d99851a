#!/bin/bash
# experiment: bench time for several chunk sizes (no profiler attached)
for cb in 1 2 3 4 8; do
  export ICP_HIP_CHUNK_BATCHES=$cb
  for wl in 64M 1M; do
  timeout -k 10 200 python3 bench.py --workload $wl --cpu-sample 0 > gpurun_out/ct_${wl}_$cb.json 2> gpurun_out/ct_${wl}_$cb.err || exit 1
  python3 - <<PY
import json
j=json.loads(open("gpurun_out/ct_${wl}_$cb.json").read().strip().splitlines()[-1])
print("$wl chunk_batches $cb: nn first %.3f last %.3f avg %.3f ms; ms/step %.4f" % (j["nn_ms_first_iter"], j["nn_ms_last_iter"], j["roofline"]["avg_launch_ms"], j["ms_per_step"]))
PY
  done
done
