#!/bin/bash
# round 4, call A: full GPU suite (new: pretext branch, variant identity, device guard) + a baseline bench line on the same box
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > gpurun_out/r4a_pytest.log 2>&1; echo "pytest exit $?"; tail -n 30 gpurun_out/r4a_pytest.log
cat gpurun_out/nt_variant_identity.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4a_bench.json 2>gpurun_out/r4a_bench.err; echo "bench exit $?"; cut -c1-400 gpurun_out/r4a_bench.json
