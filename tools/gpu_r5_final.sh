#!/bin/bash
# round-5 evidence on ONE box: full GPU suite log (product library), the investigation variants against the lab library, smoke, default bench
# line (dense `value` + config.varlen), CC / VQA shapes, rocprofv3 kernel-trace summaries of the dense and of the padding-free step
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
{ echo "# box: $(rocm-smi --showproductname 2>/dev/null | grep -m1 'Card Model' | sed 's/.*: *//') $(nproc) host threads, $(date -u +%FT%TZ), git $(cat .git_head 2>/dev/null)"; } > gpurun_out/pytest_gpu_full.log
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=10 >> gpurun_out/pytest_gpu_full.log 2>&1; echo "pytest exit $?"; tail -n 4 gpurun_out/pytest_gpu_full.log
if [ -f vlp_amd/libvlp_hip_lab.so ]; then
  VLP_HIP_LIB=$PWD/vlp_amd/libvlp_hip_lab.so timeout 900 python -m pytest tests/test_00_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "phased or stream_k or attention_fwd_bwd or epilogues or asymmetric" > gpurun_out/pytest_gpu_lab_variants.log 2>&1; echo "lab pytest exit $?"; tail -n 2 gpurun_out/pytest_gpu_lab_variants.log
fi
timeout 300 python __graft_entry__.py > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -n 1 gpurun_out/smoke.log
timeout 600 python bench.py > gpurun_out/bench_default.json 2>gpurun_out/bench_default.err; echo "bench exit $?"; cat gpurun_out/bench_default.json | cut -c1-2500
timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --s2s_prob 0.75 > gpurun_out/bench_cc.json 2>/dev/null; cut -c1-200 gpurun_out/bench_cc.json
timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --tasks vqa2 --s2s_prob 0 > gpurun_out/bench_vqa.json 2>/dev/null; cut -c1-200 gpurun_out/bench_vqa.json
for mode in dense varlen; do
  V=0; [ $mode = varlen ] && V=1
  rm -rf /tmp/prof_$mode
  VLP_VARLEN=$V VLP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-varlen > gpurun_out/prof_bench_line_$mode.json 2>gpurun_out/prof_$mode.err; echo "rocprof $mode exit $?"
  python tools/prof_summary.py $(find /tmp/prof_$mode -name "*kernel_trace.csv" | head -1) 0.45 > gpurun_out/prof_summary_$mode.txt 2>&1; head -n 16 gpurun_out/prof_summary_$mode.txt
  cp $(find /tmp/prof_$mode -name "*kernel_stats.csv" | head -1) gpurun_out/prof_kernel_stats_$mode.csv 2>/dev/null
done
