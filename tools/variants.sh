# Frame rate of the headline workload under the library's debugging switches (run through gpurun).
for v in "VOHIP_X=1" "VOHIP_NO_SPECULATION=1" "VOHIP_NO_WORKER=1" "VOHIP_NO_WORKER=1 VOHIP_NO_SPECULATION=1"; do
  env $v timeout -k 10 300 python -u bench.py --only-headline --no-cpu-baseline > gpurun_out/v.json 2>/dev/null || exit 1
  python3 -c "
import json
d=json.load(open('gpurun_out/v.json')); print('$v', '| fps %.0f ms %.4f enqueue %.0f wait %.0f' % (d['value'], d['ms_per_step'], d['config']['host_enqueue_us_per_frame'], d['config']['host_wait_us_per_frame']))"
done
