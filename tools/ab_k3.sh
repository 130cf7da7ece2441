for lib in neo_mpc_planner2_amd/libneo_mpc_prev.so neo_mpc_planner2_amd/libneo_mpc.so; do
 for cfg in "4096 200" "64 1000" "1024 500" "1 1000"; do
  for ch in default; do
   NEO_MPC_LIB=$lib python tools/bench_ingest.py $cfg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $lib) $cfg', '%.4f ms %.0f GB/s'%(d['ms'],d['achieved_GBps']))"
  done
 done
done
for ch in 1 2 8; do NEO_MPC_INGEST_CHUNKS=$ch python tools/bench_ingest.py 4096 200 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new grid-chunks=$ch 4096x200', '%.4f ms %.0f GB/s'%(d['ms'],d['achieved_GBps']))"; done
python -m pytest tests -m gpu -q -x -k "pool or edges or objective" 2>&1 | tail -2
