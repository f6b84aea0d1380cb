python -m pytest tests -m gpu -x -q -k "metrics or evaluate or wrapper or g6" 2>&1 | tail -3
for d in 2 3; do echo "STAGES=$d"; ET_METRICS_STAGES=$d python tools/ab_metrics.py 2>&1 | grep -E "round 1|max"; done
