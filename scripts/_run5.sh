mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q --durations=8 ) > gpurun_out/r3_gputests5.txt 2>&1
tail -25 gpurun_out/r3_gputests5.txt
