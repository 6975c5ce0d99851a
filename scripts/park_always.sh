#!/bin/bash
# lots in every pass (ICP_HIP_PARK=always) against the rule (only after the second ordering pass), 64M bench, ONE gpurun call
for r in 1 2; do
for mode in rule always; do
if [ $mode = always ]; then export ICP_HIP_PARK=always; else unset ICP_HIP_PARK; fi
timeout -k 10 300 python3 bench.py --workload 64M --cpu-sample 0 > gpurun_out/pk.json 2> gpurun_out/pk.err || exit 1
python3 -c "
import json,sys
j=json.loads(open('gpurun_out/pk.json').read().strip().splitlines()[-1])
print('park $mode step %.4f first %.4f last %.4f' % (j['ms_per_step'], j['nn_ms_first_iter'], j['nn_ms_last_iter']))"
done; done
