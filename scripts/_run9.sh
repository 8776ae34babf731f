mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "matches_oracle or time_sliced or moving or fit_surround or wire or launch_shape or random_layouts" > gpurun_out/r3_tests9.txt 2>&1
tail -4 gpurun_out/r3_tests9.txt
timeout 600 python scripts/cfg5_time.py > gpurun_out/r3_cfg5_a.txt 2>&1
timeout 600 python scripts/profile_phases.py 5 1024 >> gpurun_out/r3_cfg5_a.txt 2>&1
cat gpurun_out/r3_cfg5_a.txt
