#!/bin/bash
# A/B of prebuilt library variants (variants/*.so) with bench.py on a custom cloud size, inside ONE gpurun call:
# usage: scripts/ab_points.sh <points> <rounds> [extra bench args]
pts=$1; rounds=$2; shift 2
cp icp_amd/lib/libicp_hip.so /tmp/libicp_hip.keep
for r in $(seq 1 $rounds); do
  for so in variants/*.so; do
    cp $so icp_amd/lib/libicp_hip.so
    timeout -k 10 300 python3 bench.py --workload custom --points $pts --cpu-sample 0 "$@" > gpurun_out/ab.json 2> gpurun_out/ab.err || { cp /tmp/libicp_hip.keep icp_amd/lib/libicp_hip.so; exit 1; }
    python3 -c "
import json,sys
j=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1])
print('%-9s %-28s step %.4f first %.4f last %.4f' % (sys.argv[2], sys.argv[1], j['ms_per_step'], j['nn_ms_first_iter'], j['nn_ms_last_iter']))" "$so" $pts
  done
done
cp /tmp/libicp_hip.keep icp_amd/lib/libicp_hip.so
