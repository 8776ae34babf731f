mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r3_build.txt 2>&1
( time timeout 1500 python bench.py ) > gpurun_out/r3_bench_a.json 2> gpurun_out/r3_bench_a.err
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 ) > gpurun_out/r3_gputests4.txt 2>&1
tail -5 gpurun_out/r3_bench_a.err; tail -25 gpurun_out/r3_gputests4.txt
