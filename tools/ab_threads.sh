R=/root/repo
for rep in 1 2; do
  for t in 768 1024 512; do
    line=$(ET_KMEANS_FILTER_THREADS=$t timeout 300 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1)
    echo "threads=$t $(echo "$line" | python -c 'import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; st=j["stages"]; print("step_ms", j["ms_per_step"], "lloyd_us/iter", round(1e3*r["avg_launch_ms"]/r["lloyd_iterations_per_launch"],2), "lloyd_ms", st["kmeans_lloyd"]["ms"])')"
  done
done
