# Frame rate of the headline workload and tracker launch times for library variants built with
# tools/build_variant.sh (run through gpurun): bash tools/variants.sh <tag> [<tag> ...]; "base" = libvohip.so
for tag in "$@"; do
  lib=visual-odometry-pipeline_amd/lib/libvohip_$tag.so
  [ "$tag" = base ] && lib=visual-odometry-pipeline_amd/lib/libvohip.so
  for n in 1 2000 8192; do
    VOHIP_LIB=$PWD/$lib timeout -k 10 100 python3 tools/tune_klt.py $n 2>/dev/null | grep "max_iter=20" | sed "s/^/$tag /" || exit 1
  done
  for rep in 1 2; do
    VOHIP_LIB=$PWD/$lib timeout -k 10 300 python -u bench.py --only-headline --no-cpu-baseline > gpurun_out/v.json 2>/dev/null || exit 1
    python3 -c "
import json
d=json.load(open('gpurun_out/v.json')); print('$tag', '| fps %.0f ms %.4f enqueue %.0f wait %.0f' % (d['value'], d['ms_per_step'], d['config']['host_enqueue_us_per_frame'], d['config']['host_wait_us_per_frame']))"
  done
done
