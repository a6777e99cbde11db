for g in 256 224 192 160 128; do
  TTNET_STEM_GRID=$g python bench.py --steps 100 --windows 3 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('grid', $g, 'value', j['value'], 'serial', j['serial']['value'], 'stem_ms', j['roofline']['ms'])
"
done
