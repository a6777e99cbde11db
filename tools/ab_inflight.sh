for r in 1 2; do for n in 2 3 4; do
python bench.py --inflight $n --steps 300 --windows 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight $n', round(j['value']), 'serial', round(j['serial']['value']))"
done; done
