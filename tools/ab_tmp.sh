for L in 32 34 28 43 22 32 34; do
  echo "L=$L"; python bench.py --no-cpu-baseline --steps 30 --segs-per-chunk $L 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])
"
done
