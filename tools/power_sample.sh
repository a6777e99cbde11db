#!/bin/bash
# Runs on the GPU box: samples rocm-smi (power, clocks) while bench.py runs the TT-small forward back to back.
#   bash tools/power_sample.sh [bench args]   -> stdout
smi() { rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|GPU use" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';'; echo; }
echo "idle: $(smi)"
python bench.py --steps 150000 --windows 1 --no-cpu-baseline --no-extras "$@" > /tmp/power_bench.json 2>/dev/null &
pid=$!
for i in $(seq 1 200); do
  use=$(rocm-smi --showuse 2>/dev/null | grep -oE "GPU use \(%\): [0-9]+" | grep -oE "[0-9]+$")
  [ "${use:-0}" -ge 90 ] && break
  sleep 1
done
echo "busy after ${i}s"
for k in 1 2 3 4 5 6 7 8; do echo "t+$((2*k))s: $(smi)"; sleep 2; done
wait $pid
python -c "import json;j=json.loads(open('/tmp/power_bench.json').read().strip().splitlines()[-1]);print('value',j['value'],'serial',j.get('serial',{}).get('value'))"
