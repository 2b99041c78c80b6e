out=gpurun_out/r5fin
mkdir -p $out
timeout -k 10 600 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err || echo "default bench failed"
for w in c3 c2 c3t c2t; do timeout -k 10 200 python3 bench.py --workload $w --no-variants > $out/bench_$w.json 2> $out/bench_$w.err || echo "bench $w failed"; done
tail -c 600 $out/bench_default.json
