mkdir -p gpurun_out/r03c
timeout 900 python -m pytest tests -m gpu -q -k "mirror or reconfigure or fuzz" 2>&1 | tail -15 > gpurun_out/r03c/tests.log
timeout 300 python tools/diag_mirror.py --steps 8 --count 256 --show 1 > gpurun_out/r03c/diag8.log 2>&1
timeout 300 python tools/diag_mirror.py --steps 3 --method 3 --count 512 --show 1 > gpurun_out/r03c/diag3.log 2>&1
cat gpurun_out/r03c/tests.log; grep -v amdgpu gpurun_out/r03c/diag8.log | cut -c1-400; grep -v amdgpu gpurun_out/r03c/diag3.log | cut -c1-300
