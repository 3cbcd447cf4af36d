# Frame rate of the headline workload under environment switches of the library (run through gpurun):
#   bash tools/variants_env.sh "A=1" "B=1 C=2" ...   ("X=0" for the plain run)
for v in "$@"; do
  for rep in 1 2; do
    env $v timeout -k 10 300 python -u bench.py --only-headline --no-cpu-baseline > gpurun_out/v.json 2>/dev/null || exit 1
    python3 -c "
import json
d=json.load(open('gpurun_out/v.json')); print('$v', '| fps %.0f ms %.4f enqueue %.0f wait %.0f' % (d['value'], d['ms_per_step'], d['config']['host_enqueue_us_per_frame'], d['config']['host_wait_us_per_frame']))"
  done
done
