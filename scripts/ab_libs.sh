#!/bin/bash
# ab_libs.sh "name1 name2 ..." [rows] [reps]: bench.py once per library variant (vrod_amd/libvrod_NAME.so; "hip" = the
# product build), round-robin on the same box; prints ms per batch and the roofline fraction
set -e
names=$1; rows=${2:-0}; reps=${3:-2}
mkdir -p gpurun_out/ab
for rep in $(seq 1 $reps); do for v in $names; do
  VROD_HIP_LIB=$PWD/vrod_amd/libvrod_$v.so python bench.py --steps 40 --warmup 3 --rows $rows --no-cpu-baseline --no-hbm-probe --no-host-probe > gpurun_out/ab/$v.$rows.$rep.json 2> gpurun_out/ab/$v.$rows.$rep.err
  python -c "
import json,sys
d=json.loads(open('gpurun_out/ab/$v.$rows.$rep.json').read().strip().splitlines()[-1])
print('$v rows=$rows rep=$rep ms=%.4f frac=%.4f launches=%s fallbacks=%s' % (d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('launches_per_step'), d.get('certificate_fallbacks', d.get('fallbacks'))))"
done; done
