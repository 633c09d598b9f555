#!/bin/bash
# developer tool: the per-kernel table of bench.py (HIP events inside the timed region) for two builds of the library, same box, A B A B:
#   A = tools/_ab/libsvhip_A.so (e.g. the previous commit: git stash; tools/build_variant.sh; mv tools/libsvhip_var.so tools/_ab/libsvhip_A.so; git stash pop)
#   B = the in-tree build.      tools/lib_ab.sh [name filter] [bench args]
filt=${1:-.}; shift
for rep in 1 2; do
for v in A B; do
  if [ $v = A ]; then export SVHIP_LIB_PATH=$PWD/tools/_ab/libsvhip_A.so; else unset SVHIP_LIB_PATH; fi
  rm -f /tmp/_lab.jsonl
  python bench.py --no-extras --no-scoring --no-cpu-baseline --sustain-seconds 0 --record-file /tmp/_lab.jsonl "$@" > /tmp/_lab.out 2>/dev/null
  V=$v FILT=$filt python - <<'PY'
import json, os, re
d = json.loads(open('/tmp/_lab.out').read().strip().splitlines()[-1])
print("==", os.environ["V"], " value", round(d['value']), "ms/step", d['ms_per_step'])
for l in open('/tmp/_lab.jsonl'):
    r = json.loads(l)
    if r.get('record') == 'kernels':
        for k, v in sorted(r['kernels'].items(), key=lambda kv: -kv[1]['ms_per_step']):
            if re.search(os.environ["FILT"], k):
                print(f"   {k:28s} {v['ms_per_step']*1e3:8.1f} us/step  {v['launches_per_step']:.0f} launches  avg {v['avg_ms']*1e3:7.1f} us")
PY
done
done
