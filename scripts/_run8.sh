mkdir -p gpurun_out
bash scripts/collect_profiles.sh r03 > gpurun_out/r3_collect.txt 2>&1
tail -12 gpurun_out/r3_collect.txt
