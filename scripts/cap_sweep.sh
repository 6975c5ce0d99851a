for cap in 256 128 64 512; do
timeout -k 10 300 python3 bench.py --workload 64M --capacity $cap --cpu-sample 0 > gpurun_out/cap_$cap.json 2> gpurun_out/cap_$cap.err || exit 1
python3 -c "
import json,sys
j=json.loads(open('gpurun_out/cap_$cap.json').read().strip().splitlines()[-1])
print('cap $cap step %.4f first %.4f last %.4f height %s setup %s' % (j['ms_per_step'], j['nn_ms_first_iter'], j['nn_ms_last_iter'], j['config']['tree_height'], j['setup_s']))"
done
