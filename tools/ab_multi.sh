#!/bin/bash
# Same-box A/B of several environment settings: bash tools/ab_multi.sh rounds "A=1 B=2" "A=0" ...
rounds=$1; shift
for r in $(seq 1 $rounds); do for cfg in "$@"; do
  env $cfg python bench.py ${AB_EXTRA} --steps ${AB_STEPS:-300} --windows ${AB_WINDOWS:-5} --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
ks={k['kernel']:round(k['ms']*1e3,1) for k in j['roofline_kernels']}
print('$cfg', '| two in flight', round(j['value']), 'serial', round(j['serial']['value']), 'stem', ks.get('stem'), 'lin1', ks.get('head.lin1'), 'mid', ks.get('head.bn_poly'), 'last', ks.get('gate_last'))
"
done; done
