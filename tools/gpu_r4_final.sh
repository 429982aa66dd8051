#!/bin/bash
# round-4 evidence in ONE box: PMC passes first (so that the bench line quotes traffic measured on these very sources), then tools/gpu_final.sh
cd $GRAFT_REPO_ROOT
bash tools/gpu_pmc_bench.sh > gpurun_out/pmc_run.log 2>&1; tail -n 30 gpurun_out/pmc_run.log
if [ -s gpurun_out/pmc_traffic.json ]; then cp gpurun_out/pmc_traffic.json profiles/r04_pmc_traffic.json; fi
bash tools/gpu_final.sh
bash tools/gpu_soak.sh > /dev/null 2>&1; cat gpurun_out/soak.txt | grep -v "^    " 
