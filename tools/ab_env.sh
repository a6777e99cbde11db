#!/bin/bash
# Same-box A/B of an environment switch: bash tools/ab_env.sh VAR "v1 v2 .." [rounds]  (bench.py --steps ${AB_STEPS:-100} --windows ${AB_WINDOWS:-3})
var=$1; vals=$2; rounds=${3:-2}
for r in $(seq 1 $rounds); do for v in $vals; do
  env $var=$v python bench.py ${AB_EXTRA} --steps ${AB_STEPS:-100} --windows ${AB_WINDOWS:-3} --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
ks={k['kernel']:round(k['ms']*1e3,1) for k in j['roofline_kernels']}
print('$var=$v', 'two in flight', round(j['value']), 'serial', round(j['serial']['value']), 'stem', ks.get('stem'), 'lin1', ks.get('head.lin1'), 'gate', ks.get('gate_path (all binarised LUT launches)'))
"
done; done
