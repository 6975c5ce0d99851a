#!/bin/bash
# sweep of the workgroups per CU (ICP_HIP_TILE_WGS_PER_CU) with bench.py on the 64M workload, inside ONE gpurun call
for w in 7 6 5 4 3; do
ICP_HIP_TILE_WGS_PER_CU=$w timeout -k 10 300 python3 bench.py --workload 64M --cpu-sample 0 > gpurun_out/wgs.json 2> gpurun_out/wgs.err || exit 1
python3 -c "
import json,sys
j=json.loads(open('gpurun_out/wgs.json').read().strip().splitlines()[-1])
print('wgs $w step %.4f first %.4f last %.4f' % (j['ms_per_step'], j['nn_ms_first_iter'], j['nn_ms_last_iter']))"
done
