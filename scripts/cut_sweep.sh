#!/bin/bash
# sweep of the leaf-phase cut (ICP_HIP_TILE_CUT) with bench.py on the 64M workload, inside ONE gpurun call
for cut in 4 0 2 3 6 8 12 4; do
ICP_HIP_TILE_CUT=$cut timeout -k 10 300 python3 bench.py --workload 64M --cpu-sample 0 > gpurun_out/cut.json 2> gpurun_out/cut.err || exit 1
python3 -c "
import json,sys
j=json.loads(open('gpurun_out/cut.json').read().strip().splitlines()[-1])
print('cut $cut step %.4f first %.4f last %.4f' % (j['ms_per_step'], j['nn_ms_first_iter'], j['nn_ms_last_iter']))"
done
