#!/bin/bash
# GPU box, round 6, the final tree: the whole GPU suite, the fuzzes, then scripts/r06_profiles.sh (profiles + the bench line)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.txt 2>&1; tail -3 gpurun_out/r06_smoke.txt
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r06_gpu_suite.txt 2>&1; tail -4 gpurun_out/r06_gpu_suite.txt
timeout 900 python scripts/fuzz_quad.py 400 1000 > gpurun_out/r06_fuzz_quad.txt 2>&1; tail -2 gpurun_out/r06_fuzz_quad.txt
timeout 900 python scripts/fuzz_quadm.py 300 2000 > gpurun_out/r06_fuzz_quadm.txt 2>&1; tail -2 gpurun_out/r06_fuzz_quadm.txt
timeout 1200 python scripts/fuzz_reference_order.py 200 3000 > gpurun_out/r06_fuzz_reference_order.txt 2>&1; tail -2 gpurun_out/r06_fuzz_reference_order.txt
timeout 900 python scripts/fuzz_steps.py 300 5000 > gpurun_out/r06_fuzz_steps.txt 2>&1; tail -2 gpurun_out/r06_fuzz_steps.txt
bash scripts/r06_profiles.sh
