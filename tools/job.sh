out=gpurun_out/r5fin2
mkdir -p $out
timeout -k 10 600 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err || echo "default bench failed"
for w in c3 c2 c3t c2t c1; do timeout -k 10 200 python3 bench.py --workload $w --no-variants > $out/bench_$w.json 2> $out/bench_$w.err || echo "bench $w failed"; done
python3 -c "
import json
R='$out/'
d=json.loads(open(R+'bench_default.json').read().strip().split('\n')[-1]); r=d['roofline']
print('default', d['value'], d['ms_per_step'], r['kernel_ms_avg'], r['frac'], d['spinup_steps'], {k:(round(v['ms_per_step'],4), v.get('spinup_steps'), v.get('kernel_ms_avg')) for k,v in d['variants'].items()})
for w in ('c3','c2','c3t','c2t','c1'):
    d=json.loads(open(R+'bench_%s.json'%w).read().strip().split('\n')[-1]); r=d['roofline']
    print(w, d['value'], d['ms_per_step'], r['kernel_ms_avg'], r['frac'], d['spinup_steps'])
"
