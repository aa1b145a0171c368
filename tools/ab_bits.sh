cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do LCR_ENUM_BITS=$v python bench.py --no-extras --no-cpu-baseline --no-traffic --steps 60 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('enum_bits=$v', 'ms/step %.3f'%d['ms_per_step'], 'value %.3e'%d['value'], 'api', {k:round(v,3) for k,v in d['stages']['api_ms'].items()}, 'isolated frac %.3f'%d['roofline']['isolated']['frac'])
"; done
