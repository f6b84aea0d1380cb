python -m pytest tests -m gpu -x -q -k "metrics or evaluate or wrapper or g6" 2>&1 | tail -3
python tools/ab_metrics.py 2>&1 | grep -E "round|max"
