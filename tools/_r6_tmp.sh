O=gpurun_out/check_r06f; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -80 > $O/gpu_tests.log
cat $O/gpu_tests.log
