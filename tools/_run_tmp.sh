P='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(d["ms_per_step"], d["value"])'
for i in 1 2 3; do
  echo -n "mixed (wgx=fp16): "; python bench.py --train 3 --steps 10 --warmup 6 --no-cpu-baseline --precision mixed | python -c "$P"
  echo -n "mixed, wgx=layer: "; python bench.py --train 3 --steps 10 --warmup 6 --no-cpu-baseline --precision "proj=f16x3,score=f16x3,pv=fp16,conv=f16x3" | python -c "$P"
done
