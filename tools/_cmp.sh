for impl in 1 3; do for w in "north" "north --shard 0/2" "north --shard 0/4" "north --shard 0/8" "cfg5 --shard 0/8" "cfg4"; do
python bench.py --workload $w --fir-impl $impl --no-cpu-baseline 2>&1 | tail -1 > /tmp/o.json; python -c "
import json; d=json.load(open('/tmp/o.json')); r=d['roofline']
print('impl $impl', '$w', 'step', round(d['ms_per_step'],4), 'fir', round(r['launch_ms']*1e3,1), round(r['frac'],3), 'alone', r.get('launch_ms_alone') and round(r['launch_ms_alone']*1e3,1), r.get('frac_alone') and round(r['frac_alone'],3))
"; done; done
