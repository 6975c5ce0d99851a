#!/bin/bash
# experiment helper (GPU box): bench lines for prebuilt library variants in variants/<name>.so
cp icp_amd/lib/libicp_hip.so /tmp/libicp_hip.keep
for so in variants/*.so; do
  cp $so icp_amd/lib/libicp_hip.so
  timeout -k 10 300 python3 bench.py --cpu-sample 0 > gpurun_out/try.json 2> gpurun_out/try.err || exit 1
  python3 -c "
import json; j=json.loads(open('gpurun_out/try.json').read().strip().splitlines()[-1]); print('$so', j['ms_per_step'], j['nn_ms_first_iter'], j['nn_ms_last_iter'], j['roofline']['avg_launch_ms'])"
done
cp /tmp/libicp_hip.keep icp_amd/lib/libicp_hip.so
