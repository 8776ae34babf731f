mkdir -p gpurun_out
timeout 600 python scripts/cfg5_time.py > gpurun_out/r3_cfg5_c.txt 2>&1; cat gpurun_out/r3_cfg5_c.txt
