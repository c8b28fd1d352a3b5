#!/bin/bash
# env_sweep.sh VAR "v1 v2 ..." [rows] [reps]: bench.py once per value of one VROD_DEBUG_* switch, round-robin on one box
var=$1; vals=$2; rows=${3:-0}; reps=${4:-2}
mkdir -p gpurun_out/sweep
for rep in $(seq 1 $reps); do for v in $vals; do
  env $var=$v python bench.py --steps 40 --warmup 3 --rows $rows --no-cpu-baseline --no-hbm-probe --no-host-probe > gpurun_out/sweep/$var.$v.$rows.$rep.json 2> gpurun_out/sweep/$var.$v.$rows.$rep.err
  python -c "
import json
d=json.loads(open('gpurun_out/sweep/$var.$v.$rows.$rep.json').read().strip().splitlines()[-1])
print('$var=$v rows=$rows rep=$rep ms=%.4f frac=%.4f launches=%s fallback=%s band=%s' % (d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('launches_per_step'), d['exactness']['certificate_fallback_queries'], d['exactness']['resolved_by_band_pass']))"
done; done
