#!/bin/bash
# round-end evidence: full GPU suite log, smoke, default bench line, configs 4/5 bench lines, rocprofv3 kernel-trace summary of the bench command
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
{ echo "# box: $(rocm-smi --showproductname 2>/dev/null | grep -m1 'Card Model' | sed 's/.*: *//') $(nproc) host threads, $(date -u +%FT%TZ), git $(cat .git_head 2>/dev/null)"; } > gpurun_out/pytest_gpu_full.log
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=10 >> gpurun_out/pytest_gpu_full.log 2>&1; echo "pytest exit $?"; tail -n 4 gpurun_out/pytest_gpu_full.log
timeout 300 python __graft_entry__.py > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -n 1 gpurun_out/smoke.log
timeout 600 python bench.py > gpurun_out/bench_default.json 2>gpurun_out/bench_default.err; echo "bench exit $?"; cat gpurun_out/bench_default.json | cut -c1-1500
timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --s2s_prob 0.75 > gpurun_out/bench_cc.json 2>/dev/null; cut -c1-200 gpurun_out/bench_cc.json
timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --tasks vqa2 --s2s_prob 0 > gpurun_out/bench_vqa.json 2>/dev/null; cut -c1-200 gpurun_out/bench_vqa.json
rm -rf /tmp/prof_f; VLP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_bench_line.json 2>gpurun_out/prof_f.err; echo "rocprof exit $?"
python tools/prof_summary.py $(find /tmp/prof_f -name "*kernel_trace.csv" | head -1) 0.5 > gpurun_out/prof_summary.txt 2>&1; head -n 14 gpurun_out/prof_summary.txt
cp $(find /tmp/prof_f -name "*kernel_stats.csv" | head -1) gpurun_out/prof_kernel_stats.csv 2>/dev/null
# copy block for profiles/ (run in the container after the call):
#   cp gpurun_out/pytest_gpu_full.log profiles/rNN_pytest_gpu_full.log; cp gpurun_out/bench_default.json profiles/rNN_bench.json
#   cp gpurun_out/parity_fullsize.json profiles/rNN_parity_report.json; cp gpurun_out/parity_report.json profiles/rNN_parity_report_fixtures.json
#   cp gpurun_out/parity_fullsize_grads_*.json profiles/ (as rNN_parity_fullsize_grads_*.json); prof_summary.txt, prof_kernel_stats.csv, prof_bench_line.json, bench_cc.json, bench_vqa.json
