out=gpurun_out/r5fin2
mkdir -p $out
timeout -k 10 600 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err || echo "default bench failed"
for w in c3 c2 c3t c2t; do timeout -k 10 200 python3 bench.py --workload $w --no-variants > $out/bench_$w.json 2> $out/bench_$w.err || echo "bench $w failed"; done
timeout -k 10 100 python3 tools/latency.py > $out/latency_single_evaluation.txt 2>&1
tail -c 300 $out/bench_default.json
