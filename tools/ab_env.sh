#!/bin/bash
# same-box A/B of one environment switch: tools/ab_env.sh VAR N [bench args]  ->  N alternations of (VAR=1, VAR unset), sustained utt/s of each
var=$1; n=$2; shift 2
for i in $(seq 1 $n); do
  for on in 1 0; do
    if [ $on = 1 ]; then export $var=1; else unset $var; fi
    python bench.py --no-extras --no-scoring --no-cpu-baseline --sustain-seconds 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$on', round(d['value']), round(d['sustained']['value']), round(d['roofline']['avg_launch_ms']*1e3,1))"
  done
done
