# Headline frame rate of library variants in turn (run through gpurun): bash tools/ab.sh <tag> [<tag> ...];
# "base" = lib/libvohip.so, any other tag = lib/libvohip_<tag>.so
for tag in "$@"; do
  lib=visual-odometry-pipeline_amd/lib/libvohip_$tag.so
  [ "$tag" = base ] && lib=visual-odometry-pipeline_amd/lib/libvohip.so
  VOHIP_LIB=$PWD/$lib timeout -k 10 300 python -u bench.py --only-headline --no-cpu-baseline > gpurun_out/v.json 2>/dev/null || exit 1
  python3 -c "
import json
d=json.load(open('gpurun_out/v.json')); print('$tag', '| fps %.0f ms %.4f enqueue %.0f wait %.0f' % (d['value'], d['ms_per_step'], d['config']['host_enqueue_us_per_frame'], d['config']['host_wait_us_per_frame']))"
done
