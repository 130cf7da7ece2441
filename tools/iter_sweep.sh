#!/bin/bash
# study: K1 kernel time on C2 vs an iteration cap (per-iteration cost at full and at tail occupancy)
for k in 1 4 8 12 16 24 100; do
  echo -n "max_iterations=$k "
  python bench.py --steps 50 --warmup 5 --no-cpu-baseline --max-iterations $k 2>/dev/null | tail -1 | \
    python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("kernel_ms %.4f" % d["roofline"]["kernel_ms"], d["solver"])'
done
