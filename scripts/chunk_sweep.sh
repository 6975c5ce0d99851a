#!/bin/bash
# experiment: NN pass time and FETCH_SIZE of the 64M bench for several chunk sizes
export TMPDIR=/tmp
for cb in 1 2 4 8; do
  export ICP_HIP_CHUNK_BATCHES=$cb
  timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/cs_$cb -o pmc --output-format csv -- python3 bench.py --cpu-sample 0 > gpurun_out/cs_$cb.json 2> gpurun_out/cs_$cb.err || exit 1
  python3 - <<PY
import csv, json
v=[float(r["Counter_Value"])*2048 for r in csv.DictReader(open("gpurun_out/cs_$cb/pmc_counter_collection.csv")) if "nn_tile" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
j=json.loads(open("gpurun_out/cs_$cb.json").read().strip().splitlines()[-1])
print("chunk_batches $cb: fetch first %.2f last %.2f GB; nn first %.2f last %.2f ms; ms/step %.3f" % (v[3]/1e9, v[-1]/1e9, j["nn_ms_first_iter"], j["nn_ms_last_iter"], j["ms_per_step"]))
PY
done
