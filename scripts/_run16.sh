mkdir -p gpurun_out
timeout 600 python scripts/cfg5_time.py > gpurun_out/r3_cfg5_d.txt 2>&1; cat gpurun_out/r3_cfg5_d.txt
timeout 600 python scripts/profile_phases.py 3 4096 > gpurun_out/r3_phases_4096.txt 2>&1; cat gpurun_out/r3_phases_4096.txt
timeout 600 python scripts/profile_phases.py 3 256 > gpurun_out/r3_phases_256.txt 2>&1; cat gpurun_out/r3_phases_256.txt
