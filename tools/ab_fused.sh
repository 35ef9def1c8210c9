for rep in 1 2 3; do for wl in study replica; do for f in "--fused on" "--fused off"; do
python bench.py --steps 3 --warmup 1 --cpu-rays 0 --no-split-legs --no-train-leg --precision bf16x3 --workload $wl $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep$rep $wl $f', round(d['value']), 'rays/s  field ms', round(d['roofline']['avg_launch_ms'],3))"
done; done; done
