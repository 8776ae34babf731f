mkdir -p gpurun_out
ORDER=ref timeout 600 python scripts/profile_phases.py 3 32 > gpurun_out/r3_refprof_a.txt 2>&1
ORDER=ref timeout 600 python scripts/profile_phases.py 1 32 >> gpurun_out/r3_refprof_a.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gputests2.txt 2>&1
cat gpurun_out/r3_refprof_a.txt; tail -15 gpurun_out/r3_gputests2.txt
