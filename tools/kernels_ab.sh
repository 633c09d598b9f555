#!/bin/bash
# developer tool: the per-kernel table of bench.py (HIP events inside the timed region) with an environment switch on and off, same box
#   tools/kernels_ab.sh VAR [bench args]
var=$1; shift
for on in 1 0; do
  if [ $on = 1 ]; then export $var=1; else unset $var; fi
  rm -f /tmp/_kab.jsonl
  python bench.py --no-extras --no-scoring --no-cpu-baseline --sustain-seconds 0 --record-file /tmp/_kab.jsonl "$@" > /tmp/_kab.out 2>/dev/null
  python - <<PY
import json
d=json.loads(open('/tmp/_kab.out').read().strip().splitlines()[-1])
print("== $var=$on  value", round(d['value']), "ms/step", d['ms_per_step'])
for l in open('/tmp/_kab.jsonl'):
    r=json.loads(l)
    if r.get('record')=='kernels':
        ks=r['kernels']
        for k,v in sorted(ks.items(), key=lambda kv:-kv[1]['ms_per_step']):
            print(f"   {k:28s} {v['ms_per_step']*1e3:8.1f} us/step  {v['launches_per_step']:.0f} launches  avg {v['avg_ms']*1e3:7.1f} us")
PY
done
