#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/pf /tmp/pw
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -o p -- python bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-events > /dev/null 2>gpurun_out/pmcf.err; echo "fetch pass $?"
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -o p -- python bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-events > /dev/null 2>gpurun_out/pmcw.err; echo "write pass $?"
python tools/pmc_summary.py /tmp/pf/p_counter_collection.csv /tmp/pw/p_counter_collection.csv gpurun_out/pmc_traffic.json
