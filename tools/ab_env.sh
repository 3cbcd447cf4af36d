# Headline frame rate under settings of one environment variable in turn (run through gpurun):
#   bash tools/ab_env.sh NAME value [value ...]      ("-" = unset)
name=$1; shift
for v in "$@"; do
  if [ "$v" = "-" ]; then unset $name; else export $name=$v; fi
  timeout -k 10 300 python -u bench.py --only-headline --no-cpu-baseline > gpurun_out/v.json 2>/dev/null || exit 1
  python3 -c "
import json
d=json.load(open('gpurun_out/v.json')); print('$name=$v', '| fps %.0f ms %.4f enqueue %.0f wait %.0f' % (d['value'], d['ms_per_step'], d['config']['host_enqueue_us_per_frame'], d['config']['host_wait_us_per_frame']))"
done
