mkdir -p gpurun_out
timeout 600 python scripts/_diag_hw.py > gpurun_out/r3_hw.txt 2>&1; cat gpurun_out/r3_hw.txt
