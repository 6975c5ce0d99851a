#!/bin/bash
# A/B of environment settings with bench.py inside ONE gpurun call.
# usage: scripts/ab_env_loop.sh <workload> <rounds> ENV=VALUE [ENV=VALUE ...]   ("-" = no setting)
wl=$1; rounds=$2; shift 2
for r in $(seq 1 $rounds); do
  for e in "$@"; do
    [ "$e" = "-" ] && e="ICP_HIP_NOOP=1"
    env $e timeout -k 10 300 python3 bench.py --workload $wl --cpu-sample 0 > gpurun_out/ab.json 2> gpurun_out/ab.err || exit 1
    python3 -c "
import json,sys
j=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1])
print('%-6s %-22s step %.4f first %.4f last %.4f' % (sys.argv[2], sys.argv[1], j['ms_per_step'], j['nn_ms_first_iter'], j['nn_ms_last_iter']))" "$e" $wl
  done
done
