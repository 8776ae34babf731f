mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r3_build.txt 2>&1
timeout 600 python scripts/ref_order_diag.py 1 3 > gpurun_out/r3_diag1.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gputests1.txt 2>&1
tail -30 gpurun_out/r3_diag1.txt; tail -15 gpurun_out/r3_gputests1.txt
