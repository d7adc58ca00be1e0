timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-160
timeout -k 10 500 python bench.py --workload 10m --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1200
