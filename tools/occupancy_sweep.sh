#!/bin/bash
# study: K1 kernel time vs number of resident waves (batch) at a fixed iteration count
for b in 256 512 1024 2048 4096 8192 16384; do
  echo -n "batch=$b "
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --max-iterations ${1:-8} --batch $b 2>/dev/null | tail -1 | \
    python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("kernel_ms %.4f" % d["roofline"]["kernel_ms"], d["solver"])'
done
