mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r3_smoke.txt 2>&1; tail -3 gpurun_out/r3_smoke.txt
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 ) > gpurun_out/r3_gputests17.txt 2>&1; tail -14 gpurun_out/r3_gputests17.txt
timeout 300 python scripts/pipeline_time.py > gpurun_out/r3_pipeline.txt 2>&1; tail -12 gpurun_out/r3_pipeline.txt
