import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d['value']), d['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in d['kernels'].items()})
